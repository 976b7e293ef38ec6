"""GPU parity of the SDXL UNet (BASELINE.json configs[0]).

* tests/golden/tiny_sdxl.npz: a narrow UNet (32/64/128 channels, same wiring) assembled from the reference's own blocks and
  run through the reference's forward on CPU, bf16 / fp8 / int8;
* tests/golden/sdxl_full.npz: SDXL-base at full width (2.6 B synthetic parameters, generated bit-identically on the GPU)
  through the reference's own core and loaders, 256x256 image, CFG batch 2, bf16 / fp8.
"""
import os

import pytest
import torch

from oracle import sdxl as OX
from oracle.make_golden import GOLD, load

pytestmark = pytest.mark.gpu
BF16, FP8 = torch.bfloat16, torch.float8_e4m3fn


def to_cuda(inp):
    return dict(sample=inp["sample"].cuda(), encoder_hidden_states=inp["encoder_hidden_states"].cuda(),
                added_cond_kwargs={k: v.cuda() for k, v in inp["added_cond_kwargs"].items()})


def tiny_model(quant, seed, channels_last=True):
    from fastdm_amd.model_entry import create_model
    cfg = OX.TINY
    return create_model("sdxl", ckpt_path=OX.synth_state_dict(cfg, seed=seed), dtype=BF16, quant_type=quant, kernel_backend="hip",
                        block_out_channels=cfg.block_out_channels, transformer_layers=cfg.transformer_layers, mid_layers=cfg.mid_layers,
                        channels_last=channels_last).eval()


@pytest.mark.parametrize("channels_last", [True, False])
@pytest.mark.parametrize("tag,qt", [("bf16", None), ("fp8", FP8), ("int8", torch.int8)])
def test_tiny_sdxl_against_reference_golden(tag, qt, channels_last):
    g = load("tiny_sdxl.npz")
    cfg = OX.TINY
    model = tiny_model(qt, int(g["seed"]), channels_last)
    assert model.config.in_channels == 4 and model.config.addition_time_embed_dim == 256 and model.add_embedding is model.core_model.add_embedding
    inp = to_cuda(OX.synth_inputs(cfg, int(g["batch"]), int(g["height"]), int(g["width"]), int(g["n_txt"]), seed=int(g["seed"])))
    cpu_inp = OX.synth_inputs(cfg, int(g["batch"]), int(g["height"]), int(g["width"]), int(g["n_txt"]), seed=int(g["seed"]))
    oracle = OX.SDXLCore(OX.synth_state_dict(cfg, seed=int(g["seed"])), cfg, qt)
    for ti, t in enumerate((981.0, 341.0)):
        out = model(inp["sample"], torch.tensor(t).cuda(), inp["encoder_hidden_states"], inp["added_cond_kwargs"])
        assert isinstance(out, list) and out[0].shape == (2, 4, 16, 24) and out[0].is_contiguous()
        got = out[0].float().cpu()
        ref = g[f"out_{tag}_t{ti}"].float()
        orc = oracle.forward(cpu_inp["sample"], torch.tensor(t), cpu_inp["encoder_hidden_states"], cpu_inp["added_cond_kwargs"])[0].float()
        d, do = (got - ref).abs(), (got - orc).abs()
        print(f"tiny SDXL {tag} cl={channels_last} t{ti}: vs reference {d.max().item():.4f}/{d.mean().item():.5f}  vs oracle "
              f"{do.max().item():.4f}/{do.mean().item():.5f}")
        # oracle-vs-reference on this fixture (torch's CPU flash SDPA keeps P in bf16; everything else is bit-exact):
        # bf16 0.042 / 0.0076, int8 0.063 / 0.011, fp8 0.20 / 0.032 on |out| ~ 0.44
        lim = {"bf16": (1.2e-1, 2e-2), "fp8": (3.5e-1, 5e-2), "int8": (1.4e-1, 2.4e-2)}[tag]      # GPU adds MIOpen conv / group-norm rounding (measured up to 0.063 / 0.010 bf16)
        assert torch.isfinite(got).all() and d.max() <= lim[0] and d.mean() <= lim[1], (tag, ti, d.max().item(), d.mean().item())
        assert do.max() <= lim[0] and do.mean() <= lim[1], (tag, ti, do.max().item(), do.mean().item())


@pytest.mark.parametrize("tag,qt", [("bf16", None), ("fp8", FP8)])
def test_full_width_sdxl_against_reference_core(tag, qt):
    from fastdm_amd.model_entry import create_model
    from fastdm_amd.synthetic import SyntheticSDXLStateDict, sdxl_step_inputs
    g = load("sdxl_full.npz")
    seed = int(g["seed"])
    model = create_model("sdxl", ckpt_path=SyntheticSDXLStateDict(seed=seed), dtype=BF16, quant_type=qt, kernel_backend="hip").eval()
    inp = sdxl_step_inputs(int(g["batch"]), int(g["height"]), int(g["width"]), int(g["n_txt"]), seed=seed)
    # the GPU generator and the numpy twin the reference consumed agree bit for bit
    want = OX.synth_inputs(OX.SDXLConfig(), int(g["batch"]), int(g["height"]), int(g["width"]), int(g["n_txt"]), seed=seed)
    assert torch.equal(inp["sample"].cpu(), want["sample"]) and torch.equal(inp["encoder_hidden_states"].cpu(), want["encoder_hidden_states"])
    assert torch.equal(inp["added_cond_kwargs"]["text_embeds"].cpu(), want["added_cond_kwargs"]["text_embeds"])
    out = model(inp["sample"], torch.tensor(601.0).cuda(), inp["encoder_hidden_states"], inp["added_cond_kwargs"])[0]
    ref = g[f"out_{tag}"]
    assert out.shape == ref.shape == (2, 4, 32, 32)
    d = (out.float().cpu() - ref.float()).abs()
    print(f"SDXL full width {tag}: max|d| {d.max().item():.4f} mean|d| {d.mean().item():.5f} on mean|ref| {ref.float().abs().mean().item():.3f}")
    lim = {"bf16": (1.2e-1, 1.5e-2), "fp8": (4e-1, 5e-2)}[tag]
    assert torch.isfinite(out.float()).all() and d.max() <= lim[0] and d.mean() <= lim[1], (tag, d.max().item(), d.mean().item())


@pytest.mark.parametrize("tag,qt", [("bf16", None), ("fp8", FP8)])
def test_baseline_config1_sdxl_512x512_four_steps_against_reference(tag, qt):
    """BASELINE.json configs[0] on the HIP path: SDXL-base-1.0 UNet at full width, 512 x 512 (latent 64 x 64), 4 denoise steps, CFG as one
    batch-2 forward.  The fixture (`python -m oracle.make_golden c1`) holds the same 4-step loop through the reference's own core on CPU:
    every step's UNet output and latents.  Two checks per step: the UNet output given the REFERENCE's latents of the previous step (no
    error carried over: the per-forward deviation, limits as the 256 x 256 full-width test), and the free-running trajectory's latents."""
    from fastdm_amd.model_entry import create_model
    from fastdm_amd.pipeline import SyntheticSDXLPipeline
    from fastdm_amd.synthetic import SyntheticSDXLStateDict, sdxl_step_inputs
    path = "sdxl_c1_512x512_4steps.npz"
    if not os.path.exists(os.path.join(GOLD, path)):
        pytest.skip(f"{path} not generated")
    g = load(path)
    if f"lat_{tag}_0" not in g:
        pytest.skip(f"{path} has no {tag} trajectory")
    seed, steps, guidance, hw = int(g["seed"]), int(g["steps"]), float(g["guidance"]), int(g["height"])
    model = create_model("sdxl", ckpt_path=SyntheticSDXLStateDict(seed=seed), dtype=BF16, quant_type=qt, kernel_backend="hip").eval()
    pipe = SyntheticSDXLPipeline(model, seed=seed)
    ts, sig = pipe.schedule(steps)
    assert [int(t) for t in ts] == [int(t) for t in g["timesteps"]] and torch.equal(torch.tensor(sig), g["sigmas"].float())
    inp = sdxl_step_inputs(2, hw, hw, int(g["n_txt"]), seed=seed)
    lat = (inp["sample"][0:1].float() * float(sig[0])).to(BF16)
    lines = []
    # per-forward limits: the two-implementation control (sdxl_c1_control.npz, see SDXL_C1_RATIO_LIMIT below) x the measured ratio + 10 %
    # -- mean 0.0091 / 0.043 at step 0 instead of the 0.015 / 0.05 guessed in round 4; the free-running bound
    # below propagates the same number
    c = load("sdxl_c1_control.npz") if os.path.exists(os.path.join(GOLD, "sdxl_c1_control.npz")) else None
    for i in range(steps):
        if c is not None:
            ce = (c[f"ceps_{tag}_{i}"].float() - g[f"eps_{tag}_{i}"].float()).abs()
            lim_eps = (2.0 * ce.max().item(), SDXL_C1_RATIO_LIMIT[tag][i] * ce.mean().item())
        else:
            lim_eps = {"bf16": (1.5e-1, 1.5e-2), "fp8": (5e-1, 5e-2)}[tag]
        # (a) one forward from the reference's own latents
        prev = (inp["sample"][0:1].float() * float(sig[0])).to(BF16) if i == 0 else g[f"lat_{tag}_{i - 1}"].cuda()
        x = (prev.float() / ((float(sig[i]) ** 2 + 1.0) ** 0.5)).to(BF16)
        eps = model(x.expand(2, -1, -1, -1).contiguous(), torch.tensor(float(ts[i])).cuda(), inp["encoder_hidden_states"], inp["added_cond_kwargs"])[0]
        de = (eps.float().cpu() - g[f"eps_{tag}_{i}"].float()).abs()
        # (b) the free-running loop
        lat = pipe.step(lat, inp, ts[i], float(sig[i]), float(sig[i + 1]), guidance)
        dl = (lat.float().cpu() - g[f"lat_{tag}_{i}"].float()).abs()
        ref_mag = g[f"lat_{tag}_{i}"].float().abs().mean().item()
        lines.append(f"SDXL 512x512 {tag} step {i}: UNet output max|d| {de.max().item():.4f} mean {de.mean().item():.5f} (mean|eps| {g[f'eps_{tag}_{i}'].float().abs().mean().item():.3f})"
                     f" | latents max|d| {dl.max().item():.4f} mean {dl.mean().item():.5f} (mean|lat| {ref_mag:.3f})")
        assert torch.isfinite(eps.float()).all() and de.max() <= lim_eps[0] and de.mean() <= lim_eps[1], lines[-1]
        # latents: x += d_sigma * (eps_u + g (eps_c - eps_u)): a per-forward deviation e reaches the latents as ~ sqrt((g - 1)^2 + g^2) |d_sigma| e
        # (first order, g = 5: 6.4 x; sigma runs 14.6 -> 0 over the 4 steps), so the free-running limit is that propagation of the
        # per-forward limit over the sigma already travelled -- a gross-error bound, the per-forward check above is the sharp one
        travelled = float(sig[0] - sig[i + 1])
        lim_traj = max(lim_traj, lim_eps[1]) if i else lim_eps[1]        # the largest per-forward limit so far (step 0's: errors shrink with sigma)
        assert dl.mean() <= 6.4 * lim_traj * travelled, lines[-1] + f" (limit {6.4 * lim_traj * travelled:.3f})"
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/sdxl_c1_parity_{tag}.txt", "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))

# measured (gpurun_out/sdxl_c1_control_<tag>.txt) + 10 %: mean|HIP - ref| / mean|control - ref| per step, where control = the reference's own
# classes with an fp32-softmax attention (and, fp8, a float64 W8A8 accumulate): the distance between two legitimate implementations.
# Round 6 (the default model is channels-last with HIP convolutions): bf16 1.023 1.007 1.025 1.010, fp8 1.075 1.011 1.003 1.022 -- ON the envelope.
# Round 5's NCHW / MIOpen path measured 1.15 1.30 1.45 1.48 (bf16); tools/diag_sdxl_isolate.py showed that fp32 convolutions alone take it to
# 1.02 1.01 1.03 1.01: MIOpen's bf16 convolutions (every one of them, conv_in and the 1 x 1 shortcuts included) were the whole excess
# (profiles/r06g_sdxl_isolate.txt, r06q_sdxl_isolate.txt).  The NCHW model keeps the round-5 limits (SDXL_C1_RATIO_LIMIT_NCHW).
SDXL_C1_RATIO_LIMIT = {"bf16": (1.13, 1.11, 1.13, 1.12), "fp8": (1.19, 1.12, 1.11, 1.13)}
SDXL_C1_RATIO_LIMIT_NCHW = {"bf16": (1.27, 1.43, 1.59, 1.63), "fp8": (1.18, 1.17, 1.23, 1.30)}


@pytest.mark.parametrize("channels_last", [True, False])
@pytest.mark.parametrize("tag,qt", [("bf16", None), ("fp8", FP8)])
def test_baseline_config1_sdxl_against_the_two_implementation_control(tag, qt, channels_last):
    """BASELINE.json configs[0] with a yardstick.  `python -m oracle.make_golden c1_control` ran the reference's core
    once more per step from the reference's own latents with an fp32-softmax attention (fp8: plus a float64 accumulate) -- `ceps_*`.  The HIP
    path, fed the same latents, must stay within SDXL_C1_RATIO_LIMIT x mean|control - ref| of the reference at every step, and the seven UNet
    stage outputs of the step-0 forward (down 0-2, mid, up 0-2) are compared the same way so that a deviation can be LOCATED: a block that
    sticks out shows as a ratio jump between two consecutive stages."""
    from fastdm_amd.model_entry import create_model
    from fastdm_amd.pipeline import SyntheticSDXLPipeline
    from fastdm_amd.synthetic import SyntheticSDXLStateDict, sdxl_step_inputs
    for path in ("sdxl_c1_512x512_4steps.npz", "sdxl_c1_control.npz"):
        if not os.path.exists(os.path.join(GOLD, path)):
            pytest.skip(f"{path} not generated")
    g, c = load("sdxl_c1_512x512_4steps.npz"), load("sdxl_c1_control.npz")
    seed, steps, hw = int(g["seed"]), int(g["steps"]), int(g["height"])
    assert int(c["seed"]) == seed and int(c["steps"]) == steps
    model = create_model("sdxl", ckpt_path=SyntheticSDXLStateDict(seed=seed), dtype=BF16, quant_type=qt, kernel_backend="hip", channels_last=channels_last).eval()
    pipe = SyntheticSDXLPipeline(model, seed=seed)
    ts, sig = pipe.schedule(steps)
    inp = sdxl_step_inputs(2, hw, hw, int(g["n_txt"]), seed=seed)
    sub = tuple(int(x) for x in c["tap_sub"])
    lines, ratios = [], []
    for i in range(steps):
        prev = (inp["sample"][0:1].float() * float(sig[0])).to(BF16) if i == 0 else g[f"lat_{tag}_{i - 1}"].cuda()
        x = (prev.float() / ((float(sig[i]) ** 2 + 1.0) ** 0.5)).to(BF16)
        taps = {}
        restore = OX.sdxl_block_taps(model.core_model, taps, sub) if i == 0 else (lambda: None)
        eps = model(x.expand(2, -1, -1, -1).contiguous(), torch.tensor(float(ts[i])).cuda(), inp["encoder_hidden_states"], inp["added_cond_kwargs"])[0]
        restore()
        ref = g[f"eps_{tag}_{i}"].float()
        d_hip = (eps.float().cpu() - ref).abs().mean().item()
        d_ctl = (c[f"ceps_{tag}_{i}"].float() - ref).abs().mean().item()
        ratios.append(d_hip / d_ctl)
        lines.append(f"SDXL 512x512 {tag} step {i}: mean|HIP - ref| {d_hip:.5f}  mean|control - ref| {d_ctl:.5f}  ratio {ratios[-1]:.3f}  (mean|eps| {ref.abs().mean().item():.3f})")
        if i == 0:
            for name in OX.SDXL_TAPS:
                r = c[f"tap_{tag}_ref_{name}"].float()
                th, tc = (taps[name].float() - r).abs().mean().item(), (c[f"tap_{tag}_ctl_{name}"].float() - r).abs().mean().item()
                lines.append(f"    step 0 stage {name:5s}: mean|HIP - ref| {th:.5f}  mean|control - ref| {tc:.5f}  ratio {(f'{th / tc:.3f}' if tc > 0 else 'n/a (this stage has no attention / W8A8 linear: control = ref)')}  (mean|ref| {r.abs().mean().item():.3f})")
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/sdxl_c1_control_{tag}{'' if channels_last else '_nchw'}.txt", "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))
    for i, r in enumerate(ratios):
        assert r <= (SDXL_C1_RATIO_LIMIT if channels_last else SDXL_C1_RATIO_LIMIT_NCHW)[tag][i], lines


def test_sdxl_blocks_keep_reference_signatures_and_errors():
    from fastdm_amd.layer.unetblock import Attention_SDXL, BasicTransformerBlock, FeedForward_SDXL
    from fastdm_amd.model.sdxl import SDXLUNetModelCore
    with pytest.raises(NotImplementedError):
        Attention_SDXL(128, 2048, has_ipadapter=True)
    with pytest.raises(NotImplementedError):
        SDXLUNetModelCore(is_ip_adapter=True)
    model = tiny_model(None, 41)
    core = model.core_model
    # block-level forwards with the reference's argument lists: [B*S, C] tokens in and out
    blk = core.down_blocks[1].attentions[0].transformer_blocks[0]
    assert isinstance(blk, BasicTransformerBlock) and isinstance(blk.ff, FeedForward_SDXL)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2 * 24, 64, generator=g).to(BF16).cuda()
    enc = torch.rand(2, 11, 2048, generator=g).to(BF16).cuda()
    y = blk.forward(x, enc, 2)
    y_parts = x
    import torch.nn.functional as F
    ln = lambda t, i: F.layer_norm(t, [64], getattr(blk, f"norm{i}_gamma"), getattr(blk, f"norm{i}_beta"), eps=1e-5)
    y_parts = blk.attn1.forward(ln(y_parts, 1), batch=2) + y_parts
    y_parts = blk.attn2.forward(ln(y_parts, 2), enc, batch=2) + y_parts
    y_parts = blk.ff.forward(ln(y_parts, 3)) + y_parts
    d = (y.float() - y_parts.float()).abs()
    assert y.shape == x.shape and d.max() <= 2 ** -5 and (d > 0).float().mean() < 0.05, (d.max().item(), (d > 0).float().mean().item())


def test_hip_graph_replay_equals_eager_forward():
    """fastdm_amd.graph.GraphedForward: the UNet forward captured into a HIP graph (library kernels + MIOpen) and replayed on
    new latents / timesteps gives the eager forward's bits."""
    from fastdm_amd.graph import GraphedForward
    cfg = OX.TINY
    model = tiny_model(FP8, 41)
    inp = to_cuda(OX.synth_inputs(cfg, 2, 16, 24, 11, seed=41))
    unet = lambda lat, t: model(lat, t, inp["encoder_hidden_states"], inp["added_cond_kwargs"])[0]
    graphed = GraphedForward(unet, inp["sample"], torch.zeros((), device="cuda"))
    g = torch.Generator().manual_seed(5)
    for t in (981.0, 341.0, 20.0):
        lat = torch.randn(inp["sample"].shape, generator=g).to(BF16).cuda()
        tt = torch.tensor(t, device="cuda")
        want = unet(lat, tt).clone()
        got = graphed(lat, tt)
        # bit-identical in every run so far; the bound only allows for MIOpen answering a shape with another solver (see sdxl.py)
        assert torch.equal(got, want) or (got.float() - want.float()).abs().max() <= 0.12, t


def test_engine_sdxl_and_sd35_synthetic_pipelines(tmp_path):
    """FastDMEngine(architecture='sdxl' | 'sd35') without a checkpoint: CFG as one batch-2 forward per step on synthetic weights;
    the SDXL engine with use_graph replays the UNet from a HIP graph and lands on the eager trajectory's latents."""
    import json
    from fastdm_amd.model_entry import FastDMEngine
    cfg = OX.TINY
    kw = dict(block_out_channels=cfg.block_out_channels, transformer_layers=cfg.transformer_layers, mid_layers=cfg.mid_layers)
    eager = FastDMEngine("", architecture="sdxl", use_fp8=True, kernel_backend="hip", **kw)
    lat = eager.generate(steps=4, guidance_scale=5.0, gen_seed=3, gen_width=192, gen_height=128, max_seq_len=11)
    assert lat.shape == (1, 4, 16, 24) and torch.isfinite(lat.float()).all()
    lat = eager.generate(steps=4, guidance_scale=5.0, gen_seed=3, gen_width=192, gen_height=128, max_seq_len=11)     # MIOpen settled (see sdxl.py)
    graphed = FastDMEngine("", architecture="sdxl", use_fp8=True, kernel_backend="hip", use_graph=True, **kw)
    lat_g = graphed.generate(steps=4, guidance_scale=5.0, gen_seed=3, gen_width=192, gen_height=128, max_seq_len=11)
    assert (lat_g.float() - lat.float()).abs().max() <= 0.25 * lat.float().abs().max()
    cc = tmp_path / "tea.json"
    cc.write_text(json.dumps({"cache_algorithm": "teacache", "enable_caching": True, "threshold": 0.25, "coefficients": [1.0, 0.0]}))
    with pytest.raises(ValueError, match="no step cache"):
        FastDMEngine("", architecture="sdxl", cache_config=str(cc), **kw)

    from oracle import sd35 as OS
    c3 = OS.TINY
    arch = dict(sample_size=c3.sample_size, patch_size=c3.patch_size, in_channels=c3.in_channels, out_channels=c3.out_channels,
                num_layers=c3.num_layers, attention_head_dim=c3.attention_head_dim, num_attention_heads=c3.num_attention_heads,
                joint_attention_dim=c3.joint_attention_dim, caption_projection_dim=c3.caption_projection_dim,
                pooled_projection_dim=c3.pooled_projection_dim, pos_embed_max_size=c3.pos_embed_max_size,
                dual_attention_layers=c3.dual_attention_layers)
    eng = FastDMEngine("", architecture="sd35", use_fp8=True, kernel_backend="hip", cache_config=str(cc), **arch)
    lat = eng.generate(steps=6, guidance_scale=4.5, gen_seed=5, gen_width=96, gen_height=64, max_seq_len=9)
    assert lat.shape == (1, 16, 8, 12) and torch.isfinite(lat.float()).all()
    calc = [t[3] for t in eng.cache.trace]
    assert len(calc) == 6 and calc[0] and not all(calc), calc        # TeaCache driven by the pipeline's step index skipped something


def test_controlnet_residual_inputs_on_the_hip_path():
    """FLUX (both index modes), SD3.5 and the SDXL UNet accept the residual tensors of a ControlNet the caller ran
    (flux.py:454-480, sd35.py:401-404, sdxl.py:405-440); outputs against the reference (tests/golden/controlnet_residual_inputs.npz)."""
    from fastdm_amd.model_entry import create_model
    from oracle import flux as OF, sd35 as OS
    from oracle.make_golden import controlnet_residuals as cn
    g = load("controlnet_residual_inputs.npz")
    cu = lambda ts: [t.cuda() for t in ts]

    def close(got, ref, mx, mean, what):
        d = (got.float().cpu() - ref.float()).abs()
        assert d.max() <= mx and d.mean() <= mean, (what, d.max().item(), d.mean().item())

    cfg = OF.TINY
    flux = create_model("flux", ckpt_path=OF.synth_state_dict(cfg, seed=7), dtype=BF16, quant_type=None, kernel_backend="hip",
                        in_channels=cfg.in_channels, out_channels=cfg.out_channels, num_layers=cfg.num_layers,
                        num_single_layers=cfg.num_single_layers, attention_head_dim=cfg.attention_head_dim,
                        num_attention_heads=cfg.num_attention_heads, joint_attention_dim=cfg.joint_attention_dim,
                        pooled_projection_dim=cfg.pooled_projection_dim, axes_dims_rope=cfg.axes_dims_rope).eval()
    inp = {k: v.cuda() for k, v in OF.synth_inputs(cfg, 1, 4, 6, 8, seed=9).items()}
    D = cfg.attention_head_dim * cfg.num_attention_heads
    dbl, sgl = cn("flux.double", [(1, 24, D)] * 2), cn("flux.single", [(1, 24, D)])
    for tag, rep in (("flux", False), ("flux_repeat", True)):
        out = flux(timestep=torch.tensor([0.5]).cuda(), guidance=torch.tensor([3.5]).cuda(), return_dict=False,
                   controlnet_block_samples=cu(dbl[:2 if rep else 1]), controlnet_single_block_samples=cu(sgl), controlnet_blocks_repeat=rep,
                   **inp)[0]
        close(out, g[tag], 6e-2, 8e-3, tag)
    c3 = OS.TINY
    sd3 = create_model("sd35", ckpt_path=OS.synth_state_dict(c3, seed=31), dtype=BF16, quant_type=None, kernel_backend="hip",
                       sample_size=c3.sample_size, patch_size=c3.patch_size, in_channels=c3.in_channels, out_channels=c3.out_channels,
                       num_layers=c3.num_layers, attention_head_dim=c3.attention_head_dim, num_attention_heads=c3.num_attention_heads,
                       joint_attention_dim=c3.joint_attention_dim, caption_projection_dim=c3.caption_projection_dim,
                       pooled_projection_dim=c3.pooled_projection_dim, pos_embed_max_size=c3.pos_embed_max_size,
                       dual_attention_layers=c3.dual_attention_layers).eval()
    i3 = {k: v.cuda() for k, v in OS.synth_inputs(c3, 2, 8, 12, 9, seed=31).items()}
    out = sd3(timestep=torch.tensor([500.0, 500.0]).cuda(), return_dict=False,
              block_controlnet_hidden_states=cu(cn("sd3", [(2, 24, c3.inner_dim)])), **i3)[0]
    close(out, g["sd3"], 6e-2, 8e-3, "sd3")
    cx = OX.TINY
    unet = tiny_model(None, 41)
    ix = to_cuda(OX.synth_inputs(cx, 2, 16, 24, 11, seed=41))
    c0, c1, c2 = cx.block_out_channels
    shapes = [(2, c0, 16, 24)] * 3 + [(2, c0, 8, 12)] + [(2, c1, 8, 12)] * 2 + [(2, c1, 4, 6)] + [(2, c2, 4, 6)] * 2
    out = unet(ix["sample"], torch.tensor(500.0).cuda(), ix["encoder_hidden_states"], ix["added_cond_kwargs"],
               down_block_additional_residuals=cu(cn("sdxl.down", shapes)), mid_block_additional_residual=cn("sdxl.mid", [(2, c2, 4, 6)])[0].cuda())[0]
    close(out, g["sdxl"], 1e-1, 1.6e-2, "sdxl")
