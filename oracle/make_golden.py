"""Generate golden vectors by RUNNING THE REFERENCE ITSELF (build container only).

TEST INFRASTRUCTURE ONLY.  Usage (from the repo root, in the container that has
/root/reference):

    python -m oracle.make_golden small        # ops / layers / tiny-FLUX / TeaCache  (~1 min)
    python -m oracle.make_golden c3 [--blocks 19 38]   # full-width FLUX @ 1024x2048 (tens of minutes)

The reference is imported from /root/reference with two stub modules
(`fastdm.kernel.cuda`, `fastdm.kernel.triton`: their real versions need the CUDA
extension / triton kernels) and KERNEL_BACKEND=torch, QLinear forced to the CPU
(SURVEY.md §8c).  `fp8_matmul` has no CPU implementation upstream
(torch._scaled_mm row-wise is CUDA-only), so an extra backend is registered for that one
op: an fp32 matmul of the fp8 operands (every fp8 x fp8 product is exact in fp32; the same
`a.float() @ b.float()` idiom the reference's int8 torch op uses, torch/matrixmul.py:67) yields
the accumulator, to which the row/column scales and bias are applied in fp32 exactly as
csrc/gemm/hopper_w8a8_fp8.cu:95-142 does.  (CPU `torch._scaled_mm` with unit per-tensor scales
gives the same numbers to ~6e-5 but runs at ~25 MFLOP/s, unusable at full width.)  Nothing here travels to the GPU box except
the vectors it writes under tests/golden/.
"""
import argparse
import json
import os
import sys
import time
import types

import numpy as np
import torch

REF = "/root/reference"
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


# ------------------------------------------------------------------ reference import
def import_reference():
    os.environ["KERNEL_BACKEND"] = "torch"
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for m in ("fastdm.kernel.cuda", "fastdm.kernel.triton"):
        sys.modules.setdefault(m, types.ModuleType(m))
    import fastdm.kernel  # noqa: F401
    from fastdm.kernel.registry import kernel_registry
    from fastdm.layer import qlinear as ref_qlinear

    reg = kernel_registry._registry
    if "cpu_fp8" not in reg["fp8_matmul"]:
        def fp8_matmul_cpu(a, b, scale_a, scale_b, out_dtype, bias=None):
            outs = []
            bf = b.float()
            for s in range(0, a.shape[0], 2048):
                acc = a[s:s + 2048].float() @ bf
                r = acc * scale_a[s:s + 2048] * scale_b.reshape(1, -1)
                if bias is not None:
                    r = r + bias.float()
                outs.append(r.to(out_dtype))
            return torch.cat(outs, 0)
        reg["fp8_matmul"]["cpu_fp8"] = fp8_matmul_cpu
    # gelu_and_mul is force-dispatched to triton (operators_set.py:54): alias the torch impl.
    reg["gelu_and_mul"].setdefault("triton", reg["gelu_and_mul"]["torch"])

    if not getattr(ref_qlinear.QLinear, "_cpu_patched", False):
        orig_init = ref_qlinear.QLinear.__init__

        def cpu_init(self, in_features, out_features, bias=True, data_type=torch.bfloat16, device_type="cpu"):
            orig_init(self, in_features, out_features, bias=bias, data_type=data_type, device_type="cpu")
        ref_qlinear.QLinear.__init__ = cpu_init
        ref_qlinear.QLinear._cpu_patched = True
    return kernel_registry


def use_backend(name):
    """'torch' for everything; 'cpu_fp8' makes fp8_matmul use the CPU restatement (other ops
    fall back to torch, registry.py:31)."""
    os.environ["KERNEL_BACKEND"] = name


# ------------------------------------------------------------------ npz packing
def pack(d):
    out = {}
    for k, v in d.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().contiguous()
            if v.dtype == torch.bfloat16:
                out[k + "__bf16"] = v.view(torch.int16).numpy().view(np.uint16)
            elif v.dtype == torch.float8_e4m3fn:
                out[k + "__fp8"] = v.view(torch.uint8).numpy()
            else:
                out[k] = v.numpy()
        else:
            out[k] = np.asarray(v)
    return out


def unpack(npz):
    out = {}
    for k in npz.files:
        a = npz[k]
        if k.endswith("__bf16"):
            out[k[:-6]] = torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16)
        elif k.endswith("__fp8"):
            out[k[:-5]] = torch.from_numpy(a.copy()).view(torch.float8_e4m3fn)
        else:
            out[k] = torch.from_numpy(a.copy()) if a.shape != () else a.item()
    return out


def load(name):
    return unpack(np.load(os.path.join(GOLD, name)))


def save(name, d):
    os.makedirs(GOLD, exist_ok=True)
    path = os.path.join(GOLD, name)
    np.savez_compressed(path, **pack(d))
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB")


def provenance():
    """Where a long-running fixture came from (a fixture nobody can re-derive is not evidence):
    stored as meta_* strings next to the tensors."""
    import platform
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    def git(*a):
        try:
            return subprocess.run(["git", "-C", root, *a], capture_output=True, text=True, timeout=20).stdout.strip()
        except Exception:
            return "?"
    cpu = "?"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    dirty = git("status", "--porcelain", "--", "oracle", "fastdm_amd/pipeline.py")
    return {"torch": torch.__version__, "numpy": np.__version__, "threads": torch.get_num_threads(), "cpu": cpu, "machine": platform.machine(),
            "git_head": git("rev-parse", "HEAD"), "git_dirty_oracle": "clean" if dirty == "" else dirty.replace("\n", "; ")}


def randn_bf16(gen, *shape, scale=1.0):
    return (torch.randn(*shape, generator=gen) * scale).to(torch.bfloat16)


# ------------------------------------------------------------------ small fixtures
def gen_ops():
    import_reference()
    use_backend("torch")
    from fastdm.kernel import operators_set as R
    g = torch.Generator().manual_seed(1234)
    d = {}
    # rms_norm: 4-D FLUX q/k-norm layout, 3-D Wan layout, odd row count (tests/test_rmsnorm.py:5-14 shapes, reduced)
    for tag, shape in (("rms4d", (1, 37, 4, 128)), ("rms3d", (2, 9, 320)), ("rms64", (3, 5, 2, 64))):
        x = randn_bf16(g, *shape, scale=3.0)
        w = (1 + 0.1 * torch.randn(shape[-1], generator=g)).to(torch.bfloat16)
        d[f"{tag}_x"], d[f"{tag}_w"] = x, w
        d[f"{tag}_out"] = R.rms_norm(x, w, 1e-6)
    # rope GPT-J (tests/test_rope.py:5-7 layout, reduced) and NeoX
    for tag, (B, S, H, hd, neox) in (("ropej", (2, 33, 4, 64, False)), ("ropej128", (1, 70, 3, 128, False)),
                                      ("ropen", (1, 17, 2, 64, True))):
        q = randn_bf16(g, B, S, H * hd)
        k = randn_bf16(g, B, S, H * hd)
        ang = torch.rand(S, hd // 2, generator=g) * 6.28
        cache = torch.cat((ang.cos(), ang.sin()), dim=-1).to(torch.bfloat16)
        d[f"{tag}_q"], d[f"{tag}_k"], d[f"{tag}_cache"] = q.clone(), k.clone(), cache
        R.rotary_pos_embedding(q, k, hd, cache, is_neox=neox)
        d[f"{tag}_qout"], d[f"{tag}_kout"] = q, k
    # per-token quant incl. constant / all-negative / zero / huge rows (tests/test_quant.py shapes, reduced)
    x = randn_bf16(g, 19, 256, scale=2.0)
    x[3] = 0.75
    x[5] = -x[5].abs() - 0.5
    x[7] = 0.0
    x[9] *= 1000.0
    x[11, 17] = 300.0
    d["quant_x"] = x
    q8, s8 = R.quantize_to_fp8(x)
    d["quant_fp8_q"], d["quant_fp8_s"] = q8, s8
    xi = x.clone()
    xi[7, 0] = 1.0  # an all-zero row is 0/0 in the reference's int8 path; keep it finite
    xi[3, 1] = 0.5  # constant rows give scale 0 in the asymmetric path
    d["quant_xi"] = xi
    qi, si, zi = R.quantize_to_int8(xi, symmetric=False)
    d["quant_i8a_q"], d["quant_i8a_s"], d["quant_i8a_zp"] = qi, si, zi
    qs, ss, _ = R.quantize_to_int8(xi, symmetric=True)
    d["quant_i8s_q"], d["quant_i8s_s"] = qs, ss
    # W8A8 GEMMs (tests/test_matmul.py:48-71,104-123 recipe at reduced shapes)
    for tag, (M, K, N) in (("mm14", (14, 256, 192)), ("mm130", (130, 384, 320))):
        a = torch.randint(-128, 128, (M, K), generator=g).to(torch.int8)
        b = torch.randint(-128, 128, (K, N), generator=g).to(torch.int8).t().contiguous().t()
        sa = torch.randn(M, 1, generator=g)
        sb = torch.randn(N, 1, generator=g)
        adj = torch.randint(-128, 127, (1, N), generator=g).to(torch.int32)
        azp = torch.randint(-128, 127, (M, 1), generator=g).to(torch.int32)
        bias = randn_bf16(g, N)
        d.update({f"{tag}_i8_a": a, f"{tag}_i8_b": b, f"{tag}_sa": sa, f"{tag}_sb": sb, f"{tag}_adj": adj,
                  f"{tag}_azp": azp, f"{tag}_bias": bias})
        d[f"{tag}_i8_out_bias"] = R.int8_matmul(a, b, sa, sb, torch.bfloat16, adj, azp, bias)
        d[f"{tag}_i8_out"] = R.int8_matmul(a, b, sa, sb, torch.bfloat16, adj, azp, None)
        fa = torch.randn(M, K, generator=g).to(torch.float8_e4m3fn)
        fb = torch.randn(K, N, generator=g).to(torch.float8_e4m3fn).t().contiguous().t()
        d[f"{tag}_f8_a"], d[f"{tag}_f8_b"] = fa, fb
        use_backend("cpu_fp8")
        d[f"{tag}_f8_out_bias"] = R.fp8_matmul(fa, fb, sa, sb, torch.bfloat16, bias)
        d[f"{tag}_f8_out"] = R.fp8_matmul(fa, fb, sa, sb, torch.bfloat16, None)
        use_backend("torch")
    # attention (tests/test_attention.py:7-21 families: hd 128 self, hd 64 self, hd 64 cross Sq != Skv)
    torch.manual_seed(0)
    for tag, (B, Sq, Sk, H, hd) in (("att128", (1, 200, 200, 3, 128)), ("att64", (2, 130, 130, 2, 64)),
                                    ("attx64", (2, 96, 77, 2, 64))):
        q = randn_bf16(g, B, Sq, H * hd)
        k = randn_bf16(g, B, Sk, H * hd)
        v = randn_bf16(g, B, Sk, H * hd)
        d[f"{tag}_q"], d[f"{tag}_k"], d[f"{tag}_v"] = q, k, v
        d[f"{tag}_out"] = R.scaled_dot_product_attention(q, k, v, H, H, hd, scale=hd ** -0.5)
    # gelu_and_mul (tests/test_gelu_and_mul.py:5-8, reduced)
    x = randn_bf16(g, 21, 2 * 160, scale=2.0)
    d["gm_x"] = x
    d["gm_out"] = R.gelu_and_mul(x)
    save("ops.npz", d)


def gen_layers():
    import_reference()
    use_backend("torch")
    from fastdm.utils.quantization import fp8_quantization, int8_quantization
    from fastdm.layer.embeddings import FluxPosEmbed, get_timestep_embedding
    from fastdm.layer.qlinear import QLinear
    g = torch.Generator().manual_seed(4321)
    d = {}
    # weight quantisation exactly as QLinear.weight_loading_and_quant does it (qlinear.py:18-54)
    w_list = [randn_bf16(g, 96, 128, scale=0.1) for _ in range(3)]      # three [out,in] weights fused along N
    b_list = [randn_bf16(g, 96, scale=0.02) for _ in range(3)]
    for i in range(3):
        d[f"w{i}"], d[f"b{i}"] = w_list[i], b_list[i]
    for tag, qt in (("fp8", torch.float8_e4m3fn), ("int8", torch.int8)):
        ql = QLinear(128, 288)
        ql.weight_loading_and_quant([w.transpose(0, 1) for w in w_list], list(b_list), quant_type=qt)
        d[f"ql_{tag}_w_nk"] = ql.weight.t().contiguous()        # stored [K,N] col-major == [N,K] row-major
        d[f"ql_{tag}_scale"] = ql.weight_quant_scale
        d[f"ql_{tag}_bias"] = ql.bias
        if qt == torch.int8:
            d["ql_int8_colsum"] = ql.weight_asym_sumcol
        x = randn_bf16(g, 2, 11, 128)
        d[f"ql_{tag}_x"] = x
        if qt == torch.float8_e4m3fn:
            use_backend("cpu_fp8")
        d[f"ql_{tag}_y"] = ql.forward(x)
        use_backend("torch")
    # RoPE table for a small grid + text tokens, FLUX axes (embeddings.py:527-549, flux.py:425-428)
    ids = torch.zeros(5 + 6 * 7, 3)
    ids[5:, 1] = torch.arange(6).repeat_interleave(7).float()
    ids[5:, 2] = torch.arange(7).repeat(6).float()
    cos, sin = FluxPosEmbed(theta=10000, axes_dim=[16, 56, 56]).forward(ids)
    d["rope_ids"] = ids
    d["rope_cos"], d["rope_sin"] = cos, sin
    d["rope_cache"] = torch.cat((cos[:, 0::2], sin[:, 1::2]), dim=-1).to(torch.bfloat16)
    # timestep embedding on the bf16-rounded timestep*1000 (flux.py:399-401)
    ts = torch.tensor([1.0, 0.75, 0.0361, 3.5]).to(torch.bfloat16) * 1000
    d["ts_in"] = ts
    d["ts_emb"] = get_timestep_embedding(ts, 256, flip_sin_to_cos=True, downscale_freq_shift=0)
    save("layers.npz", d)


def ref_flux_core(cfg, quant, sd, cache=None):
    from fastdm.model.flux import FluxTransformer2DModelCore
    core = FluxTransformer2DModelCore(
        in_channels=cfg.in_channels, out_channels=cfg.out_channels, num_layers=cfg.num_layers,
        num_single_layers=cfg.num_single_layers, attention_head_dim=cfg.attention_head_dim,
        num_attention_heads=cfg.num_attention_heads, joint_attention_dim=cfg.joint_attention_dim,
        pooled_projection_dim=cfg.pooled_projection_dim, guidance_embeds=True,
        axes_dims_rope=cfg.axes_dims_rope, data_type=torch.bfloat16, quant_dtype=quant, cache=cache)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        core.weight_loading(dict(sd), device_type="cpu")
    return core


def gen_tiny_flux():
    import_reference()
    from oracle import flux as OF
    cfg = OF.TINY
    sd = OF.synth_state_dict(cfg, seed=7)
    inp = OF.synth_inputs(cfg, batch=2, img_h=4, img_w=6, n_txt=8, seed=7)
    d = {"seed": 7, "batch": 2, "img_h": 4, "img_w": 6, "n_txt": 8}
    for tag, qt in (("bf16", None), ("int8", torch.int8), ("fp8", torch.float8_e4m3fn)):
        use_backend("cpu_fp8" if qt == torch.float8_e4m3fn else "torch")
        core = ref_flux_core(cfg, qt, sd)
        for ti, t in enumerate((1.0, 0.75)):
            out = core.forward(timestep=torch.tensor([t, t]), guidance=torch.tensor([3.5, 3.5]), **inp)[0]
            d[f"out_{tag}_t{ti}"] = out
    use_backend("torch")
    save("tiny_flux.npz", d)


def gen_teacache():
    """TeaCache decision + tensors on the tiny core over a 12-step synthetic trajectory
    (xcaching.py:187-276 driven exactly as model_entry.py:635-637 does, via a step callback)."""
    import_reference()
    use_backend("torch")
    from fastdm.caching.xcaching import AutoCache
    from oracle import flux as OF
    cfg = OF.TINY
    sd = OF.synth_state_dict(cfg, seed=7)
    inp = OF.synth_inputs(cfg, batch=1, img_h=4, img_w=6, n_txt=8, seed=9)
    coeffs = json.load(open(os.path.join(REF, "examples/xcaching/configs/teacache_flux.json")))
    # the shipped threshold never skips on this tiny random net; a loose one exercises both branches
    state = {"step": 0}
    trace = []
    for thr_tag, thr in (("shipped", coeffs["threshold"]), ("loose", 2.0)):
        cfgd = dict(coeffs)
        cfgd["threshold"] = thr
        cache = AutoCache.from_dict(cfgd)
        cache.config.current_steps_callback = lambda: state["step"]
        core = ref_flux_core(cfg, None, sd, cache=cache)
        lat = inp["hidden_states"].clone()
        outs, decisions = [], []
        sig = torch.linspace(1.0, 1.0 / 12, 12)
        import contextlib
        import io
        for i in range(12):
            state["step"] = i
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                out = core.forward(hidden_states=lat, encoder_hidden_states=inp["encoder_hidden_states"],
                                   pooled_projections=inp["pooled_projections"], timestep=sig[i:i + 1],
                                   img_ids=inp["img_ids"], txt_ids=inp["txt_ids"], guidance=torch.tensor([3.5]))[0]
            decisions.append(0 if "Skipping" in buf.getvalue() else 1)
            outs.append(out)
            dt = (sig[i + 1] - sig[i]) if i + 1 < 12 else -sig[i]
            lat = (lat.float() + dt * out.float()).to(torch.bfloat16)     # flow-match Euler step
        trace.append((thr_tag, thr, decisions, torch.stack(outs), lat))
    d = {"seed_w": 7, "seed_in": 9, "coefficients": np.array(coeffs["coefficients"])}
    for thr_tag, thr, decisions, outs, lat in trace:
        d[f"{thr_tag}_threshold"] = thr
        d[f"{thr_tag}_should_calc"] = np.array(decisions, dtype=np.int8)
        d[f"{thr_tag}_outs"] = outs
        d[f"{thr_tag}_final"] = lat
    # decision rule on its own: polynomial rescale of a rel-L1 sequence (np.poly1d, xcaching.py:212-216)
    rels = np.array([0.0, 0.02, 0.05, 0.11, 0.23, 0.4, 0.31, 0.07, 0.01, 0.6], dtype=np.float64)
    d["poly_in"] = rels
    d["poly_out"] = np.poly1d(coeffs["coefficients"])(rels)
    save("teacache.npz", d)


# ------------------------------------------------------------------ Qwen-Image
def ref_qwen_core(cfg, quant, sd, cache=None):
    from fastdm.model.qwenimage import QwenImageTransformer2DModelCore
    core = QwenImageTransformer2DModelCore(
        patch_size=cfg.patch_size, in_channels=cfg.in_channels, out_channels=cfg.out_channels, num_layers=cfg.num_layers,
        attention_head_dim=cfg.attention_head_dim, num_attention_heads=cfg.num_attention_heads,
        joint_attention_dim=cfg.joint_attention_dim, axes_dims_rope=cfg.axes_dims_rope, data_type=torch.bfloat16,
        quant_dtype=quant, cache=cache)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        core.weight_loading(dict(sd), device_type="cpu")
    return core


def gen_tiny_qwen():
    """Tiny Qwen-Image core (3 blocks) through the reference in bf16 / int8 / fp8 at two timesteps, the rope table
    and timestep embedding it builds, and a TeaCache trace with the negative-prompt cache (two forwards per step,
    xcaching.py:66-76) over a 10-step trajectory."""
    import_reference()
    from fastdm.caching.xcaching import AutoCache
    from oracle import qwen as OQ
    cfg = OQ.TINY
    sd = OQ.synth_state_dict(cfg, seed=11)
    inp = OQ.synth_inputs(cfg, batch=2, img_h=6, img_w=4, n_txt=7, seed=11)
    d = {"seed": 11, "batch": 2, "img_h": 6, "img_w": 4, "n_txt": 7}
    for tag, qt in (("bf16", None), ("int8", torch.int8), ("fp8", torch.float8_e4m3fn)):
        use_backend("cpu_fp8" if qt == torch.float8_e4m3fn else "torch")
        core = ref_qwen_core(cfg, qt, sd)
        for ti, t in enumerate((1.0, 0.62)):
            out = core.forward(timestep=torch.tensor([t, t]), **inp)[0]
            d[f"out_{tag}_t{ti}"] = out
    use_backend("torch")
    core = ref_qwen_core(cfg, None, sd)
    img_f, txt_f = core.pos_embed.forward(inp["img_shapes"], inp["txt_seq_lens"], device=torch.device("cpu"))
    d["rope"] = torch.cat([torch.cat([txt_f.real, txt_f.imag], -1), torch.cat([img_f.real, img_f.imag], -1)], 0).to(torch.bfloat16)
    d["temb_t062"] = core.time_text_embed.forward(torch.tensor([0.62]).to(torch.bfloat16), torch.zeros(1, 1, dtype=torch.bfloat16))
    # TeaCache with negative cache: cond and uncond forwards alternate within a step
    coeffs = json.load(open(os.path.join(REF, "examples/xcaching/configs/teacache_qwenimage.json")))
    inp1 = OQ.synth_inputs(cfg, batch=1, img_h=6, img_w=4, n_txt=7, seed=12)
    neg = OQ.synth_inputs(cfg, batch=1, img_h=6, img_w=4, n_txt=7, seed=13)["encoder_hidden_states"]
    state = {"step": 0}
    cfgd = dict(coeffs)
    cfgd["negtive_cache"] = True
    cfgd["threshold"] = 0.25         # between one and two per-step increments on this tiny random net: both branches are taken
    cache = AutoCache.from_dict(cfgd)
    cache.config.current_steps_callback = lambda: state["step"]
    core = ref_qwen_core(cfg, torch.int8, sd, cache=cache)
    lat = inp1["hidden_states"].clone()
    sig = torch.linspace(1.0, 0.1, 10)
    decisions, outs = [], []
    import contextlib
    import io
    for i in range(10):
        state["step"] = i
        row = []
        for enc in (inp1["encoder_hidden_states"], neg):
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                out = core.forward(hidden_states=lat, encoder_hidden_states=enc, encoder_hidden_states_mask=inp1["encoder_hidden_states_mask"],
                                   timestep=sig[i:i + 1], img_shapes=inp1["img_shapes"], txt_seq_lens=inp1["txt_seq_lens"])[0]
            row.append(out)
            decisions.append(0 if "Skipping" in buf.getvalue() else 1)
        outs.append(torch.stack(row))
        comb = row[1].float() + 4.0 * (row[0].float() - row[1].float())          # true-CFG combine, scale 4
        dt = (sig[i + 1] - sig[i]) if i + 1 < 10 else -sig[i]
        lat = (lat.float() + dt * comb[..., : cfg.in_channels]).to(torch.bfloat16)
    d["tc_threshold"] = 0.25
    d["tc_coefficients"] = np.array(coeffs["coefficients"])
    d["tc_negtive_coefficients"] = np.array(coeffs["negtive_coefficients"])
    d["tc_should_calc"] = np.array(decisions, dtype=np.int8)
    d["tc_outs"] = torch.stack(outs)
    d["tc_final"] = lat
    save("tiny_qwen.npz", d)



def gen_qwen_full(n_blocks=4, img_h=83, img_w=83, n_txt=128, seed=0):
    """Qwen-Image at FULL WIDTH (D = 3072, 24 x 128 heads, 1328 x 1328 -> 6889 image + 128 text tokens) through the
    reference's own core at reduced depth, int8 W8A8 (BASELINE.json configs[3]) and bf16; the whole output is kept."""
    import_reference()
    from oracle import qwen as OQ
    cfg = OQ.QwenConfig(num_layers=n_blocks)
    t0 = time.time()
    sd = OQ.synth_state_dict(cfg, seed=seed)
    print(f"synthetic weights: {time.time() - t0:.0f}s", flush=True)
    inp = OQ.synth_inputs(cfg, batch=1, img_h=img_h, img_w=img_w, n_txt=n_txt, seed=seed)
    d = {"seed": seed, "n_blocks": n_blocks, "img_h": img_h, "img_w": img_w, "n_txt": n_txt, "timestep": 0.7}
    for tag, qt in (("int8", torch.int8), ("bf16", None)):
        use_backend("torch")
        core = ref_qwen_core(cfg, qt, sd)
        t0 = time.time()
        out = core.forward(timestep=torch.tensor([0.7]), **inp)[0]
        print(f"{tag}: forward {time.time() - t0:.0f}s, mean|out| {out.float().abs().mean():.3f}", flush=True)
        d[f"out_{tag}"] = out
        del core
    save(f"qwen_full_{n_blocks}blocks.npz", d)


# ------------------------------------------------------------------ Wan2.x
def ref_wan_core(cfg, quant, sd, cache=None):
    from fastdm.model.wan import WanTransformer3DModelCore
    core = WanTransformer3DModelCore(
        patch_size=cfg.patch_size, num_attention_heads=cfg.num_attention_heads, attention_head_dim=cfg.attention_head_dim,
        in_channels=cfg.in_channels, out_channels=cfg.out_channels, text_dim=cfg.text_dim, freq_dim=cfg.freq_dim, ffn_dim=cfg.ffn_dim,
        num_layers=cfg.num_layers, cross_attn_norm=True, eps=cfg.eps, rope_max_seq_len=cfg.rope_max_seq_len,
        image_dim=cfg.image_dim, added_kv_proj_dim=cfg.added_kv_proj_dim,
        data_type=torch.bfloat16, quant_dtype=quant, cache=cache, sparse_attn=None)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        core.weight_loading(dict(sd), device_type="cpu")
    return core


def gen_wan_full(n_blocks=2, frames=5, height=32, width=48, n_txt=512, seed=0):
    """Wan2.2-T2V-A14B at FULL WIDTH (D = 5120, 40 x 128 heads, ffn 13824, text 4096 -> 512 tokens) through the reference's own
    core at reduced depth, fp8 W8A8 (BASELINE.json configs[4]) and bf16, on a small video (5 latent frames of 32 x 48 -> 1920 tokens)."""
    import_reference()
    from oracle import wan as OW
    cfg = OW.WanConfig(num_layers=n_blocks)
    t0 = time.time()
    sd = OW.synth_state_dict(cfg, seed=seed)
    print(f"synthetic weights: {sum(v.numel() for v in sd.values()) / 1e6:.0f} M parameters in {time.time() - t0:.0f}s", flush=True)
    inp = OW.synth_inputs(cfg, batch=1, frames=frames, height=height, width=width, n_txt=n_txt, seed=seed)
    d = {"seed": seed, "n_blocks": n_blocks, "frames": frames, "height": height, "width": width, "n_txt": n_txt, "timestep": 700.0}
    for tag, qt in (("fp8", torch.float8_e4m3fn), ("bf16", None)):
        use_backend("cpu_fp8" if tag == "fp8" else "torch")
        core = ref_wan_core(cfg, qt, sd)
        t0 = time.time()
        out = core.forward(hidden_states=inp["hidden_states"], timestep=torch.tensor([700.0]), encoder_hidden_states=inp["encoder_hidden_states"],
                           return_dict=False)
        out = out[0] if isinstance(out, (tuple, list)) else out
        print(f"{tag}: forward {time.time() - t0:.0f}s, mean|out| {out.float().abs().mean():.3f}", flush=True)
        d[f"out_{tag}"] = out
        del core
    use_backend("torch")
    save(f"wan_full_{n_blocks}blocks.npz", d)


def gen_tiny_wan():
    """Tiny Wan core (2 blocks, 2 x 64 heads) through the reference in bf16 / fp8 (dense attention: the reference has no
    CPU sparse kernel), plus radial block-sparse masks from the reference's generator on three geometries."""
    import_reference()
    from fastdm.sparse.xsparse import SparseAttn, sparge_mask_convert
    from oracle import wan as OW
    cfg = OW.TINY
    sd = OW.synth_state_dict(cfg, seed=21)
    inp = OW.synth_inputs(cfg, batch=2, frames=3, height=8, width=8, n_txt=9, seed=21)
    d = {"seed": 21, "batch": 2, "frames": 3, "height": 8, "width": 8, "n_txt": 9}
    for tag, qt in (("bf16", None), ("fp8", torch.float8_e4m3fn), ("int8", torch.int8)):
        use_backend("cpu_fp8" if qt == torch.float8_e4m3fn else "torch")
        core = ref_wan_core(cfg, qt, sd)
        for ti, t in enumerate((999.0, 500.0)):
            out = core.forward(hidden_states=inp["hidden_states"], timestep=torch.tensor([t, t]),
                               encoder_hidden_states=inp["encoder_hidden_states"], return_dict=False)
            out = out[0] if isinstance(out, (tuple, list)) else out
            d[f"out_{tag}_t{ti}"] = out
    use_backend("torch")
    core = ref_wan_core(cfg, None, sd)
    cos, sin = core.rope.forward(inp["hidden_states"])
    d["rope"] = torch.cat((cos.squeeze()[:, 0::2], sin.squeeze()[:, 1::2]), dim=-1).to(torch.bfloat16)
    save("tiny_wan.npz", d)
    m = {}
    import contextlib
    import io
    for gi, (frames, tpf, bs, decay) in enumerate(((6, 320, 64, 0.3), (4, 480, 64, 0.5), (9, 256, 128, 0.3))):
        sa = SparseAttn.from_dict({"sparse_algorithm": "radial", "model_type": "wan", "block_size": bs, "enable_sparse": True,
                                   "decay_factor": decay, "dense_layers": 1, "dense_steps": 2})
        sa.post_init(video_token_num=frames * tpf, num_frame=frames)
        with contextlib.redirect_stdout(io.StringIO()):
            lm = sa.gen_log_mask_shrinked(frames * tpf, torch.device("cpu"))
        m[f"g{gi}_geom"] = np.array([frames, tpf, bs], dtype=np.int64)
        m[f"g{gi}_decay"] = decay
        m[f"g{gi}_log_mask"] = lm.to(torch.uint8)
        m[f"g{gi}_kernel_mask"] = sparge_mask_convert(lm, block_size=bs, arch="sm80").to(torch.uint8)
    save("wan_radial_mask.npz", m)


def gen_tiny_wan_i2v():
    """Wan2.1-I2V wiring (image_dim / added_kv_proj_dim: CLIP image tokens through the image embedder, prepended to the 512
    text tokens, attended through add_k_proj / add_v_proj in every cross-attention) on a tiny core through the reference."""
    import_reference()
    from oracle import wan as OW
    cfg = OW.TINY_I2V
    sd = OW.synth_state_dict(cfg, seed=27)
    inp = OW.synth_inputs(cfg, batch=2, frames=3, height=8, width=8, n_txt=OW.TEXT_LEN_I2V, seed=27)
    d = {"seed": 27, "batch": 2, "frames": 3, "height": 8, "width": 8}
    for tag, qt in (("bf16", None), ("fp8", torch.float8_e4m3fn), ("int8", torch.int8)):
        use_backend("cpu_fp8" if qt == torch.float8_e4m3fn else "torch")
        core = ref_wan_core(cfg, qt, sd)
        out = core.forward(hidden_states=inp["hidden_states"], timestep=torch.tensor([700.0, 700.0]),
                           encoder_hidden_states=inp["encoder_hidden_states"], encoder_hidden_states_image=inp["encoder_hidden_states_image"],
                           return_dict=False)
        d[f"out_{tag}"] = out[0] if isinstance(out, (tuple, list)) else out
    use_backend("torch")
    save("tiny_wan_i2v.npz", d)


def gen_tiny_wan_ti2v():
    """Wan2.2-TI2V: a [B, S] timestep tensor (one timestep per token: the conditioning frame's tokens sit at t = 0), i.e.
    per-token AdaLN modulation in every block and in the output norm (wan.py:77-88, 319-329, 355-359), on the tiny core
    through the reference."""
    import_reference()
    from oracle import wan as OW
    cfg = OW.TINY
    sd = OW.synth_state_dict(cfg, seed=33)
    inp = OW.synth_inputs(cfg, batch=2, frames=3, height=8, width=8, n_txt=9, seed=33)
    S = 3 * 4 * 4
    ts = torch.full((2, S), 812.0)
    ts[:, :16] = 0.0                      # first latent frame = the clean conditioning image
    ts[1, 16:] = 377.0
    d = {"seed": 33, "batch": 2, "frames": 3, "height": 8, "width": 8, "n_txt": 9, "timestep": ts}
    for tag, qt in (("bf16", None), ("fp8", torch.float8_e4m3fn), ("int8", torch.int8)):
        use_backend("cpu_fp8" if qt == torch.float8_e4m3fn else "torch")
        core = ref_wan_core(cfg, qt, sd)
        out = core.forward(hidden_states=inp["hidden_states"], timestep=ts, encoder_hidden_states=inp["encoder_hidden_states"], return_dict=False)
        d[f"out_{tag}"] = out[0] if isinstance(out, (tuple, list)) else out
    use_backend("torch")
    save("tiny_wan_ti2v.npz", d)


def controlnet_residuals(tag, shapes, seed=0, scale=0.25):
    """Stand-ins for the tensors a ControlNet would hand over: N(0, scale^2) from the shared counter-based generator."""
    from oracle import synth
    from oracle.flux import bf16_from_bits
    out = []
    for i, shp in enumerate(shapes):
        bits = synth.normal_bf16_bits(synth.key_seed(f"controlnet.{tag}.{i}", seed), int(np.prod(shp)), scale, 0.0)
        out.append(bf16_from_bits(bits, shp))
    return out


def gen_controlnet_residual_inputs():
    """The cores' ControlNet residual INPUTS (flux.py:454-480, sd35.py:401-404, sdxl.py:405-440) through the reference: FLUX with
    one residual per double block and one for both single blocks, in both index modes; SD3.5 with one residual shared by the
    first two blocks; the SDXL UNet with nine down residuals + the mid residual."""
    import_reference()
    use_backend("torch")
    from oracle import flux as OF, sd35 as OS, sdxl as OX
    d = {}
    cfg = OF.TINY
    core = ref_flux_core(cfg, None, OF.synth_state_dict(cfg, seed=7))
    inp = OF.synth_inputs(cfg, batch=1, img_h=4, img_w=6, n_txt=8, seed=9)
    D = cfg.attention_head_dim * cfg.num_attention_heads
    dbl = controlnet_residuals("flux.double", [(1, 24, D)] * 2)
    sgl = controlnet_residuals("flux.single", [(1, 24, D)])
    for tag, rep in (("flux", False), ("flux_repeat", True)):
        d[tag] = core.forward(hidden_states=inp["hidden_states"], encoder_hidden_states=inp["encoder_hidden_states"],
                              pooled_projections=inp["pooled_projections"], timestep=torch.tensor([0.5]), img_ids=inp["img_ids"],
                              txt_ids=inp["txt_ids"], guidance=torch.tensor([3.5]), controlnet_block_samples=[t.clone() for t in dbl[:1 if not rep else 2]],
                              controlnet_single_block_samples=[t.clone() for t in sgl], controlnet_blocks_repeat=rep)[0]
    c3 = OS.TINY
    core3 = ref_sd3_core(c3, None, OS.synth_state_dict(c3, seed=31))
    i3 = OS.synth_inputs(c3, batch=2, height=8, width=12, n_txt=9, seed=31)
    d["sd3"] = core3.forward(timestep=torch.tensor([500.0, 500.0]), return_dict=False,
                             block_controlnet_hidden_states=controlnet_residuals("sd3", [(2, 24, c3.inner_dim)]), **i3)[0]
    cx = OX.TINY
    corex = ref_sdxl_core(cx, None, OX.synth_state_dict(cx, seed=41))
    ix = OX.synth_inputs(cx, batch=2, height=16, width=24, n_txt=11, seed=41)
    c0, c1, c2 = cx.block_out_channels
    down_shapes = [(2, c0, 16, 24)] * 3 + [(2, c0, 8, 12)] + [(2, c1, 8, 12)] * 2 + [(2, c1, 4, 6)] + [(2, c2, 4, 6)] * 2
    down = controlnet_residuals("sdxl.down", down_shapes)
    mid = controlnet_residuals("sdxl.mid", [(2, c2, 4, 6)])[0]
    d["sdxl"] = corex.forward(ix["sample"], torch.tensor(500.0), ix["encoder_hidden_states"], ix["added_cond_kwargs"],
                              down_block_additional_residuals=[t.clone() for t in down], mid_block_additional_residual=mid.clone())[0]
    save("controlnet_residual_inputs.npz", d)


def gen_fbcache():
    """FBCache decisions + final latents on the tiny FLUX core (bf16) and the tiny Qwen-Image core (int8) over 12-step
    trajectories, driven through the reference's AutoCache exactly like gen_teacache."""
    import_reference()
    use_backend("torch")
    from fastdm.caching.xcaching import AutoCache
    from oracle import flux as OF, qwen as OQ
    import contextlib
    import io
    d = {"threshold": 0.41, "warmup_steps": 2}     # threshold with the widest decision margins on these trajectories
    state = {"step": 0}
    sig = torch.linspace(1.0, 1.0 / 12, 12)
    # ---- FLUX
    cfg = OF.TINY
    sd = OF.synth_state_dict(cfg, seed=7)
    inp = OF.synth_inputs(cfg, batch=1, img_h=4, img_w=6, n_txt=8, seed=9)
    cache = AutoCache.from_dict({"cache_algorithm": "fbcache", "enable_caching": True, "threshold": 0.41, "warmup_steps": 2})
    cache.config.current_steps_callback = lambda: state["step"]
    core = ref_flux_core(cfg, None, sd, cache=cache)
    lat, dec = inp["hidden_states"].clone(), []
    for i in range(12):
        state["step"] = i
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            out = core.forward(hidden_states=lat, encoder_hidden_states=inp["encoder_hidden_states"], pooled_projections=inp["pooled_projections"],
                               timestep=sig[i:i + 1], img_ids=inp["img_ids"], txt_ids=inp["txt_ids"], guidance=torch.tensor([3.5]))[0]
        dec.append(0 if "Skipping" in buf.getvalue() else 1)
        dt = (sig[i + 1] - sig[i]) if i + 1 < 12 else -sig[i]
        lat = (lat.float() + dt * out.float()).to(torch.bfloat16)
    d["flux_should_calc"], d["flux_final"] = np.array(dec, dtype=np.int8), lat
    # ---- Qwen-Image (single forward per step, int8)
    qcfg = OQ.TINY
    qsd = OQ.synth_state_dict(qcfg, seed=11)
    qin = OQ.synth_inputs(qcfg, batch=1, img_h=6, img_w=4, n_txt=7, seed=12)
    cache = AutoCache.from_dict({"cache_algorithm": "fbcache", "enable_caching": True, "threshold": 0.41, "warmup_steps": 2})
    cache.config.current_steps_callback = lambda: state["step"]
    core = ref_qwen_core(qcfg, torch.int8, qsd, cache=cache)
    lat, dec = qin["hidden_states"].clone(), []
    for i in range(12):
        state["step"] = i
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            out = core.forward(hidden_states=lat, encoder_hidden_states=qin["encoder_hidden_states"],
                               encoder_hidden_states_mask=qin["encoder_hidden_states_mask"], timestep=sig[i:i + 1],
                               img_shapes=qin["img_shapes"], txt_seq_lens=qin["txt_seq_lens"])[0]
        dec.append(0 if "Skipping" in buf.getvalue() else 1)
        dt = (sig[i + 1] - sig[i]) if i + 1 < 12 else -sig[i]
        lat = (lat.float() + dt * out.float()[..., : qcfg.in_channels]).to(torch.bfloat16)
    d["qwen_should_calc"], d["qwen_final"] = np.array(dec, dtype=np.int8), lat
    print("flux decisions", d["flux_should_calc"].tolist(), "qwen decisions", d["qwen_should_calc"].tolist())
    save("fbcache.npz", d)


# ------------------------------------------------------------------ SD3.5
def ref_sd3_core(cfg, quant, sd, cache=None):
    from fastdm.model.sd35 import SD3TransformerModelCore
    core = SD3TransformerModelCore(
        sample_size=cfg.sample_size, patch_size=cfg.patch_size, in_channels=cfg.in_channels, num_layers=cfg.num_layers,
        attention_head_dim=cfg.attention_head_dim, num_attention_heads=cfg.num_attention_heads, joint_attention_dim=cfg.joint_attention_dim,
        caption_projection_dim=cfg.caption_projection_dim, pooled_projection_dim=cfg.pooled_projection_dim, out_channels=cfg.out_channels,
        pos_embed_max_size=cfg.pos_embed_max_size, dual_attention_layers=cfg.dual_attention_layers, qk_norm="rms_norm",
        data_type=torch.bfloat16, quant_dtype=quant, cache=cache)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        core.weight_loading(dict(sd), device_type="cpu")
    return core


def gen_tiny_sd3():
    """Tiny SD3.5 MMDiT (3 blocks: one with dual attention, the last context_pre_only) through the reference, bf16 / fp8 / int8."""
    import_reference()
    from oracle import sd35 as OS
    cfg = OS.TINY
    sd = OS.synth_state_dict(cfg, seed=31)
    inp = OS.synth_inputs(cfg, batch=2, height=8, width=12, n_txt=9, seed=31)
    d = {"seed": 31, "batch": 2, "height": 8, "width": 12, "n_txt": 9}
    for tag, qt in (("bf16", None), ("fp8", torch.float8_e4m3fn), ("int8", torch.int8)):
        use_backend("cpu_fp8" if qt == torch.float8_e4m3fn else "torch")
        core = ref_sd3_core(cfg, qt, sd)
        for ti, t in enumerate((999.0, 400.0)):
            out = core.forward(timestep=torch.tensor([t, t]), return_dict=False, **inp)
            d[f"out_{tag}_t{ti}"] = out[0] if isinstance(out, (tuple, list)) else out
    use_backend("torch")
    save("tiny_sd3.npz", d)


CACHE_SD3_WAN = {   # thresholds with the widest decision margins on these 12-step trajectories (searched with the oracle)
    "sd3_teacache": {"cache_algorithm": "teacache", "enable_caching": True, "threshold": 0.25, "coefficients": [1.0, 0.0]},
    "sd3_fbcache": {"cache_algorithm": "fbcache", "enable_caching": True, "threshold": 0.14, "warmup_steps": 2},
    "sd3_dicache": {"cache_algorithm": "dicache", "enable_caching": True, "threshold": 0.22, "probe_depth": 1, "ret_ratio": 0.2,
                    "rel_l1_distance_algo": "delta_y"},
    "wan_fbcache": {"cache_algorithm": "fbcache", "enable_caching": True, "threshold": 0.14, "warmup_steps": 2},
}


def gen_cache_sd3_wan():
    """Step caches on the SD3.5 and Wan cores through the reference (caching/xcaching.py sd35 and wan branches): decisions and
    final latents of 12-step trajectories.  TeaCache uses a linear rescale polynomial here: the published SD3.5 coefficients are
    fitted to distances of 0.01-0.05 and evaluate to ~0 on the tiny random model's 0.1 (it would skip every step)."""
    import contextlib
    import io
    import_reference()
    from fastdm.caching.xcaching import AutoCache
    from oracle import sd35 as OS, wan as OW
    state = {"step": 0}
    sig = torch.linspace(1.0, 1.0 / 12, 12)
    d = {}

    def make(name):
        c = AutoCache.from_dict(dict(CACHE_SD3_WAN[name]))
        c.config.current_steps_callback = lambda: state["step"]
        c.config.total_steps_callback = lambda: 12
        return c

    def run(core_forward, lat):
        dec = []
        for i in range(12):
            state["step"] = i
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                out = core_forward(lat, sig[i:i + 1] * 1000)
            dec.append(0 if "Skipping" in buf.getvalue() else 1)
            dt = (sig[i + 1] - sig[i]) if i + 1 < 12 else -sig[i]
            lat = (lat.float() + dt * out.float()).to(torch.bfloat16)
        return np.array(dec, dtype=np.int8), lat

    use_backend("torch")
    cfg = OS.TINY
    sd = OS.synth_state_dict(cfg, seed=31)
    inp = OS.synth_inputs(cfg, batch=1, height=8, width=12, n_txt=9, seed=33)
    for name in ("sd3_teacache", "sd3_fbcache", "sd3_dicache"):
        core = ref_sd3_core(cfg, None, sd, cache=make(name))
        fwd = lambda lat, t: core.forward(hidden_states=lat, encoder_hidden_states=inp["encoder_hidden_states"],
                                          pooled_projections=inp["pooled_projections"], timestep=t, return_dict=False)[0]
        d[f"{name}_should_calc"], d[f"{name}_final"] = run(fwd, inp["hidden_states"].clone())
        print(name, d[f"{name}_should_calc"].tolist())
    use_backend("cpu_fp8")
    wcfg = OW.TINY
    wsd = OW.synth_state_dict(wcfg, seed=21)
    winp = OW.synth_inputs(wcfg, batch=1, frames=3, height=8, width=8, n_txt=9, seed=23)
    core = ref_wan_core(wcfg, torch.float8_e4m3fn, wsd, cache=make("wan_fbcache"))

    def wfwd(lat, t):
        out = core.forward(hidden_states=lat, timestep=t, encoder_hidden_states=winp["encoder_hidden_states"], return_dict=False)
        return out[0] if isinstance(out, (tuple, list)) else out
    d["wan_fbcache_should_calc"], d["wan_fbcache_final"] = run(wfwd, winp["hidden_states"].clone())
    print("wan_fbcache", d["wan_fbcache_should_calc"].tolist())
    use_backend("torch")
    save("cache_sd3_wan.npz", d)


# ------------------------------------------------------------------ SDXL UNet (BASELINE configs[0])
def ref_sdxl_core(cfg, quant, sd):
    """The reference's SDXLUNetModelCore.  Its constructor and loaders hard-code SDXL-base's widths (sdxl.py:53-113,
    178-373) while its blocks take them as arguments, so for a narrow config the core object is assembled here from the
    reference's own block classes and filled through the reference's `init_weight`; `forward` is the reference's."""
    import contextlib
    import io
    from collections import namedtuple
    from oracle import sdxl as OX
    from fastdm.model.sdxl import SDXLUNetModelCore
    from fastdm.model.basemodel import BaseModelCore
    from fastdm.layer.embeddings import Timesteps, TimestepEmbedding
    from fastdm.layer import unetblock as U
    bf = torch.bfloat16
    if cfg == OX.SDXLConfig():
        core = SDXLUNetModelCore(data_type=bf, quant_dtype=quant)
        with contextlib.redirect_stdout(io.StringIO()):
            core.weight_loading(dict(sd), device_type="cpu")
        return core
    c0, c1, c2 = cfg.block_out_channels
    l1, l2 = cfg.transformer_layers
    core = object.__new__(SDXLUNetModelCore)
    BaseModelCore.__init__(core, type="Unet")
    core.config = namedtuple("config", "is_ip_adapter")
    core.config.is_ip_adapter = False
    core.time_proj = Timesteps(cfg.time_proj_dim, True, 0)
    core.time_embedding = TimestepEmbedding(in_features=cfg.time_proj_dim, out_features=cfg.time_embed_dim, data_type=bf)
    core.add_time_proj = Timesteps(cfg.addition_time_embed_dim, True, 0)
    core.add_embedding = TimestepEmbedding(in_features=cfg.text_embed_dim + 6 * cfg.addition_time_embed_dim, out_features=cfg.time_embed_dim,
                                           data_type=bf)
    core.down_blocks = [U.DownBlock2D(in_channels=c0, out_channels=c0, data_type=bf),
                        U.CrossAttnDownBlock2D(in_channels=c0, out_channels=c1, n_layers=l1, data_type=bf),
                        U.CrossAttnDownBlock2D(in_channels=c1, out_channels=c2, n_layers=l2, has_downsamplers=False, data_type=bf)]
    core.mid_block = U.UNetMidBlock2DCrossAttn(c2, data_type=bf)
    core.mid_block.attentions = [U.Transformer2DModel(c2, c2, n_layers=cfg.mid_layers, data_type=bf)]
    core.up_blocks = [U.CrossAttnUpBlock2D(in_channels=c1, out_channels=c2, prev_output_channel=c2, n_layers=l2, data_type=bf),
                      U.CrossAttnUpBlock2D(in_channels=c0, out_channels=c1, prev_output_channel=c2, n_layers=l1, data_type=bf),
                      U.UpBlock2D(in_channels=c0, out_channels=c0, prev_output_channel=c1, data_type=bf)]
    core.origin_tensor_dict, core.device, core.dtype = dict(sd), "cpu", bf
    core.unmatched_tensors = list(sd.keys())
    ld = core.init_weight

    def resnet(r, n):
        r.norm1_gamma, r.norm1_beta = ld([f"{n}.norm1.weight"]), ld([f"{n}.norm1.bias"])
        r.conv1_weight, r.conv1_bias = ld([f"{n}.conv1.weight"]), ld([f"{n}.conv1.bias"])
        ld([f"{n}.time_emb_proj"], r.time_emb_proj)
        r.norm2_gamma, r.norm2_beta = ld([f"{n}.norm2.weight"]), ld([f"{n}.norm2.bias"])
        r.conv2_weight, r.conv2_bias = ld([f"{n}.conv2.weight"]), ld([f"{n}.conv2.bias"])
        if r.conv_shortcut:
            r.convshortcut_weight, r.convshortcut_bias = ld([f"{n}.conv_shortcut.weight"]), ld([f"{n}.conv_shortcut.bias"])

    def transformer(t, n):
        t.norm_gamma, t.norm_beta = ld([f"{n}.norm.weight"]), ld([f"{n}.norm.bias"])
        ld([f"{n}.proj_in"], t.proj_in, quant)
        ld([f"{n}.proj_out"], t.proj_out, quant)
        for j, b in enumerate(t.transformer_blocks):
            k = f"{n}.transformer_blocks.{j}"
            for i in (1, 2, 3):
                setattr(b, f"norm{i}_gamma", ld([f"{k}.norm{i}.weight"]))
                setattr(b, f"norm{i}_beta", ld([f"{k}.norm{i}.bias"]))
            ld([f"{k}.attn1.to_q", f"{k}.attn1.to_k", f"{k}.attn1.to_v"], b.attn1.qkv_proj, quant)
            ld([f"{k}.attn1.to_out.0"], b.attn1.out_proj, quant)
            ld([f"{k}.attn2.to_q"], b.attn2.q_proj, quant)
            ld([f"{k}.attn2.to_k", f"{k}.attn2.to_v"], b.attn2.kv_proj, quant)
            ld([f"{k}.attn2.to_out.0"], b.attn2.out_proj, quant)
            ld([f"{k}.ff.net.0.proj"], b.ff.proj1, quant)
            ld([f"{k}.ff.net.2"], b.ff.proj2, quant)

    with contextlib.redirect_stdout(io.StringIO()):
        core.conv_in_weight, core.conv_in_bias = ld(["conv_in.weight"]), ld(["conv_in.bias"])
        for emb, n in ((core.time_embedding, "time_embedding"), (core.add_embedding, "add_embedding")):
            ld([f"{n}.linear_1"], emb.linear1, quant)
            ld([f"{n}.linear_2"], emb.linear2, quant)
        for m, blk in enumerate(core.down_blocks):
            for i, r in enumerate(blk.resnets):
                resnet(r, f"down_blocks.{m}.resnets.{i}")
            for i, t in enumerate(getattr(blk, "attentions", [])):
                transformer(t, f"down_blocks.{m}.attentions.{i}")
            if m < 2:
                blk.downsample_conv_weight = ld([f"down_blocks.{m}.downsamplers.0.conv.weight"])
                blk.downsample_conv_bias = ld([f"down_blocks.{m}.downsamplers.0.conv.bias"])
        for i, r in enumerate(core.mid_block.resnets):
            resnet(r, f"mid_block.resnets.{i}")
        transformer(core.mid_block.attentions[0], "mid_block.attentions.0")
        for m, blk in enumerate(core.up_blocks):
            for i, r in enumerate(blk.resnets):
                resnet(r, f"up_blocks.{m}.resnets.{i}")
            for i, t in enumerate(getattr(blk, "attentions", [])):
                transformer(t, f"up_blocks.{m}.attentions.{i}")
            if m < 2:
                blk.upsample_conv_weight = ld([f"up_blocks.{m}.upsamplers.0.conv.weight"])
                blk.upsample_conv_bias = ld([f"up_blocks.{m}.upsamplers.0.conv.bias"])
        core.conv_norm_out_gemma, core.conv_norm_out_beta = ld(["conv_norm_out.weight"]), ld(["conv_norm_out.bias"])
        core.conv_out_weight, core.conv_out_bias = ld(["conv_out.weight"]), ld(["conv_out.bias"])
    assert not core.unmatched_tensors, core.unmatched_tensors[:5]
    return core


def gen_tiny_sdxl():
    """A narrow SDXL UNet (32/64/128 channels, 1 and 2 transformer layers, same wiring) through the reference's blocks and
    forward, bf16 / fp8 / int8, CFG-style batch 2, two timesteps."""
    import_reference()
    from oracle import sdxl as OX
    cfg = OX.TINY
    sd = OX.synth_state_dict(cfg, seed=41)
    inp = OX.synth_inputs(cfg, batch=2, height=16, width=24, n_txt=11, seed=41)
    d = {"seed": 41, "batch": 2, "height": 16, "width": 24, "n_txt": 11}
    for tag, qt in (("bf16", None), ("fp8", torch.float8_e4m3fn), ("int8", torch.int8)):
        use_backend("cpu_fp8" if qt == torch.float8_e4m3fn else "torch")
        core = ref_sdxl_core(cfg, qt, sd)
        for ti, t in enumerate((981.0, 341.0)):
            d[f"out_{tag}_t{ti}"] = core.forward(inp["sample"], torch.tensor(t), inp["encoder_hidden_states"], inp["added_cond_kwargs"])[0]
    use_backend("torch")
    save("tiny_sdxl.npz", d)


def gen_sdxl_full(quants=("fp8", "bf16")):
    """SDXL-base-1.0 at full width (2.6 B synthetic parameters) through the reference's own core and loaders, 256x256 image
    (latent 32x32: 1024 / 256 tokens at the two attention levels), CFG batch 2."""
    import_reference()
    from oracle import sdxl as OX
    cfg = OX.SDXLConfig()
    t0 = time.time()
    sd = OX.synth_state_dict(cfg, seed=43)
    print(f"synthetic SDXL state_dict: {sum(v.numel() for v in sd.values()) / 1e9:.2f} B parameters in {time.time() - t0:.0f}s", flush=True)
    inp = OX.synth_inputs(cfg, batch=2, height=32, width=32, n_txt=77, seed=43)
    d = {"seed": 43, "batch": 2, "height": 32, "width": 32, "n_txt": 77}
    for tag in quants:
        qt = {"bf16": None, "fp8": torch.float8_e4m3fn, "int8": torch.int8}[tag]
        use_backend("cpu_fp8" if tag == "fp8" else "torch")
        t0 = time.time()
        core = ref_sdxl_core(cfg, qt, sd)
        out = core.forward(inp["sample"], torch.tensor(601.0), inp["encoder_hidden_states"], inp["added_cond_kwargs"])[0]
        print(f"reference SDXL {tag}: load+forward {time.time() - t0:.0f}s, mean|out| {out.float().abs().mean():.4f}", flush=True)
        d[f"out_{tag}"] = out
        del core
    use_backend("torch")
    save("sdxl_full.npz", d)


def gen_sdxl_c1(quants=("bf16", "fp8")):
    """BASELINE.json configs[0]: SDXL-base-1.0 UNet, 512 x 512 (latent 64 x 64: 4096 / 1024 tokens at the two attention levels), 4 denoise
    steps, classifier-free guidance as one batch-2 UNet forward -- at full width (2.6 B synthetic parameters) through the reference's own
    core and loaders (fastdm/model/sdxl.py; examples/profiling/sdxl_profiling.py:17 feeds the same tensors).  The loop is the one of
    fastdm_amd.pipeline.SyntheticSDXLPipeline (EulerDiscrete on the scaled-linear schedule, epsilon prediction): every step's UNet output
    (sub-sampled) and latents, and the final latents."""
    import_reference()
    from oracle import sdxl as OX
    cfg = OX.SDXLConfig()
    seed, steps, guidance, hw = 47, 4, 5.0, 64
    sd = OX.synth_state_dict(cfg, seed=seed)
    inp = OX.synth_inputs(cfg, batch=2, height=hw, width=hw, n_txt=77, seed=seed)
    betas = np.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=np.float64) ** 2
    alphas = np.cumprod(1.0 - betas)
    sig_all = ((1.0 - alphas) / alphas) ** 0.5
    ts = np.linspace(999, 0, steps).round().astype(np.int64)
    sig = np.concatenate([sig_all[ts], [0.0]]).astype(np.float32)
    d = {"seed": seed, "batch": 2, "height": hw, "width": hw, "n_txt": 77, "steps": steps, "guidance": guidance, "timesteps": torch.tensor(ts), "sigmas": torch.tensor(sig)}
    d.update(provenance())
    for tag in quants:
        qt = {"bf16": None, "fp8": torch.float8_e4m3fn}[tag]
        use_backend("cpu_fp8" if tag == "fp8" else "torch")
        core = ref_sdxl_core(cfg, qt, sd)
        lat = (inp["sample"][0:1].float() * float(sig[0])).to(torch.bfloat16)
        for i in range(steps):
            t0 = time.time()
            x = (lat.float() / ((float(sig[i]) ** 2 + 1.0) ** 0.5)).to(lat.dtype)
            eps = core.forward(x.expand(2, -1, -1, -1).contiguous(), torch.tensor(float(ts[i])), inp["encoder_hidden_states"], inp["added_cond_kwargs"])[0].float()
            noise = eps[0:1] + guidance * (eps[1:2] - eps[0:1])
            lat = (lat.float() + float(sig[i + 1] - sig[i]) * noise).to(lat.dtype)
            d[f"eps_{tag}_{i}"] = eps.to(torch.bfloat16)
            d[f"lat_{tag}_{i}"] = lat.clone()
            print(f"reference SDXL c1 {tag} step {i}: {time.time() - t0:.0f}s, mean|eps| {eps.abs().mean():.4f} mean|lat| {lat.float().abs().mean():.4f}", flush=True)
        del core
    use_backend("torch")
    save("sdxl_c1_512x512_4steps.npz", d)


def gen_sdxl_c1_control(quants=("bf16", "fp8")):
    """CONTROL for BASELINE.json configs[0] (the FLUX `c3_depth` recipe applied to the SDXL UNet): how far apart are two
    LEGITIMATE implementations of the same network?  For every step of the committed 4-step trajectory (sdxl_c1_512x512_4steps.npz) the
    reference's own core is run once more from the REFERENCE's latents of the previous step, with
      * attention = fp32 scores / fp32 softmax (oracle.ops.scaled_dot_product_attention) instead of the bf16 SDPA, and
      * fp8: the W8A8 products accumulated in float64 instead of the fp32 chunks of `cpu_fp8` (another summation order / width),
    giving `ceps_<tag>_<i>`; mean|control - ref| per step is the yardstick the GPU test holds mean|HIP - ref| to.  Step 0 also records
    the seven stage outputs (down 0-2, mid, up 0-2; sub-sampled) of the reference run (`tap_<tag>_ref_*`, re-run here and asserted equal to
    the committed eps) and of the control (`tap_<tag>_ctl_*`), so a deviation can be located."""
    reg = import_reference()
    from oracle import sdxl as OX
    from oracle import ops as OO
    from oracle.sdxl import sdxl_block_taps
    g = load("sdxl_c1_512x512_4steps.npz")
    cfg = OX.SDXLConfig()
    seed, steps, hw = int(g["seed"]), int(g["steps"]), int(g["height"])
    sig, ts = g["sigmas"].float().numpy(), g["timesteps"].numpy()
    sd = OX.synth_state_dict(cfg, seed=seed)
    inp = OX.synth_inputs(cfg, batch=2, height=hw, width=hw, n_txt=int(g["n_txt"]), seed=seed)
    d = {"seed": seed, "steps": steps, "height": hw, "tap_sub": torch.tensor([4, 2, 2])}
    d.update({f"meta_prov_{k}": str(v) for k, v in provenance().items()})
    d["meta_control"] = "attention: fp32 scores + fp32 softmax; fp8_matmul: float64 accumulate (bf16 run: attention only)"

    def attn_fp32(query, key, value, num_q_heads, num_kv_heads, head_dim, is_causal=False, scale=None):
        return OO.scaled_dot_product_attention(query, key, value, num_q_heads, num_kv_heads, head_dim, is_causal=is_causal, scale=scale)

    def fp8_matmul_f64(a, b, scale_a, scale_b, out_dtype, bias=None):
        outs = []
        bd = b.to(torch.float64)
        for s0 in range(0, a.shape[0], 1024):
            acc = a[s0:s0 + 1024].to(torch.float64) @ bd
            r = acc.float() * scale_a[s0:s0 + 1024] * scale_b.reshape(1, -1)
            if bias is not None:
                r = r + bias.float()
            outs.append(r.to(out_dtype))
        return torch.cat(outs, 0)

    def forward(core, i):
        prev = (inp["sample"][0:1].float() * float(sig[0])).to(torch.bfloat16) if i == 0 else g[f"lat_{tag}_{i - 1}"]
        x = (prev.float() / ((float(sig[i]) ** 2 + 1.0) ** 0.5)).to(torch.bfloat16)
        return core.forward(x.expand(2, -1, -1, -1).contiguous(), torch.tensor(float(ts[i])), inp["encoder_hidden_states"], inp["added_cond_kwargs"])[0].float()

    for tag in quants:
        qt = {"bf16": None, "fp8": torch.float8_e4m3fn}[tag]
        use_backend("cpu_fp8" if tag == "fp8" else "torch")
        core = ref_sdxl_core(cfg, qt, sd)
        # reference, step 0, with taps: must reproduce the committed eps bit for bit (the fixture and this run are the same computation)
        taps = {}
        restore = sdxl_block_taps(core, taps)
        t0 = time.time()
        eps = forward(core, 0)
        restore()
        assert torch.equal(eps.to(torch.bfloat16), g[f"eps_{tag}_0"]), f"re-run of the reference step 0 ({tag}) differs from the committed fixture"
        print(f"reference SDXL c1 {tag} step 0 re-run with taps: {time.time() - t0:.0f}s (bit-identical to the fixture)", flush=True)
        for k, v in taps.items():
            d[f"tap_{tag}_ref_{k}"] = v
        # control
        saved_attn = dict(reg._registry["sdpa"])
        saved_mm = dict(reg._registry["fp8_matmul"])
        for b in list(reg._registry["sdpa"]):
            reg._registry["sdpa"][b] = attn_fp32
        if tag == "fp8":
            reg._registry["fp8_matmul"]["cpu_fp8"] = fp8_matmul_f64
        try:
            for i in range(steps):
                taps = {}
                restore = sdxl_block_taps(core, taps) if i == 0 else (lambda: None)
                t0 = time.time()
                ceps = forward(core, i)
                restore()
                ref = g[f"eps_{tag}_{i}"].float()
                dd = (ceps.to(torch.bfloat16).float() - ref).abs()
                print(f"control SDXL c1 {tag} step {i}: {time.time() - t0:.0f}s, |control - ref| max {dd.max():.4f} mean {dd.mean():.5f} on mean|eps| {ref.abs().mean():.3f}", flush=True)
                d[f"ceps_{tag}_{i}"] = ceps.to(torch.bfloat16)
                for k, v in taps.items():
                    d[f"tap_{tag}_ctl_{k}"] = v
        finally:
            reg._registry["sdpa"].update(saved_attn)
            reg._registry["fp8_matmul"].update(saved_mm)
        del core
    use_backend("torch")
    save("sdxl_c1_control.npz", d)



def gen_sd3_full(seed=0):
    """SD3.5-large WIDTH (38 x 64 heads, D = 2432, text 4096, pooled 2048) at three blocks -- a dual-attention block (as SD3.5-medium
    has), a plain one and the context_pre_only last one -- through the reference's own core, fp8 and bf16, 512 x 512 (1024 image +
    154 text tokens), CFG batch 2."""
    import_reference()
    from oracle import sd35 as OS
    cfg = OS.SD3Config(sample_size=128, num_layers=3, attention_head_dim=64, num_attention_heads=38, caption_projection_dim=2432,
                       pos_embed_max_size=192, dual_attention_layers=(0,))
    sd = OS.synth_state_dict(cfg, seed=seed)
    print(f"synthetic weights: {sum(v.numel() for v in sd.values()) / 1e6:.0f} M parameters", flush=True)
    inp = OS.synth_inputs(cfg, batch=2, height=64, width=64, n_txt=154, seed=seed)
    d = {"seed": seed, "batch": 2, "height": 64, "width": 64, "n_txt": 154}
    for tag, qt in (("fp8", torch.float8_e4m3fn), ("bf16", None)):
        use_backend("cpu_fp8" if tag == "fp8" else "torch")
        core = ref_sd3_core(cfg, qt, sd)
        out = core.forward(timestep=torch.tensor([600.0, 600.0]), return_dict=False, **inp)
        d[f"out_{tag}"] = out[0] if isinstance(out, (tuple, list)) else out
        print(tag, "mean|out|", d[f"out_{tag}"].float().abs().mean().item(), flush=True)
    use_backend("torch")
    save("sd3_full_3blocks.npz", d)


def gen_dicache():
    """DiCache (probe depth 1, delta_y) decisions + final latents on the tiny FLUX (bf16) and Qwen-Image (int8) cores over
    12-step trajectories through the reference's AutoCache."""
    import_reference()
    use_backend("torch")
    from fastdm.caching.xcaching import AutoCache
    from oracle import flux as OF, qwen as OQ
    import contextlib
    import io
    d = {"threshold": 0.52, "probe_depth": 1, "ret_ratio": 0.2, "total_steps": 12}
    state = {"step": 0}
    sig = torch.linspace(1.0, 1.0 / 12, 12)

    def make_cache():
        c = AutoCache.from_dict({"cache_algorithm": "dicache", "enable_caching": True, "threshold": 0.52, "probe_depth": 1, "ret_ratio": 0.2,
                                 "rel_l1_distance_algo": "delta_y"})
        c.config.current_steps_callback = lambda: state["step"]
        c.config.total_steps_callback = lambda: 12
        return c

    cfg = OF.TINY
    sd = OF.synth_state_dict(cfg, seed=7)
    inp = OF.synth_inputs(cfg, batch=1, img_h=4, img_w=6, n_txt=8, seed=9)
    core = ref_flux_core(cfg, None, sd, cache=make_cache())
    lat, dec = inp["hidden_states"].clone(), []
    for i in range(12):
        state["step"] = i
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            out = core.forward(hidden_states=lat, encoder_hidden_states=inp["encoder_hidden_states"], pooled_projections=inp["pooled_projections"],
                               timestep=sig[i:i + 1], img_ids=inp["img_ids"], txt_ids=inp["txt_ids"], guidance=torch.tensor([3.5]))[0]
        dec.append(0 if "Skipping" in buf.getvalue() else 1)
        dt = (sig[i + 1] - sig[i]) if i + 1 < 12 else -sig[i]
        lat = (lat.float() + dt * out.float()).to(torch.bfloat16)
    d["flux_should_calc"], d["flux_final"] = np.array(dec, dtype=np.int8), lat
    qcfg = OQ.TINY
    qsd = OQ.synth_state_dict(qcfg, seed=11)
    qin = OQ.synth_inputs(qcfg, batch=1, img_h=6, img_w=4, n_txt=7, seed=12)
    core = ref_qwen_core(qcfg, torch.int8, qsd, cache=make_cache())
    lat, dec = qin["hidden_states"].clone(), []
    for i in range(12):
        state["step"] = i
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            out = core.forward(hidden_states=lat, encoder_hidden_states=qin["encoder_hidden_states"],
                               encoder_hidden_states_mask=qin["encoder_hidden_states_mask"], timestep=sig[i:i + 1],
                               img_shapes=qin["img_shapes"], txt_seq_lens=qin["txt_seq_lens"])[0]
        dec.append(0 if "Skipping" in buf.getvalue() else 1)
        dt = (sig[i + 1] - sig[i]) if i + 1 < 12 else -sig[i]
        lat = (lat.float() + dt * out.float()[..., : qcfg.in_channels]).to(torch.bfloat16)
    d["qwen_should_calc"], d["qwen_final"] = np.array(dec, dtype=np.int8), lat
    print("flux decisions", d["flux_should_calc"].tolist(), "qwen decisions", d["qwen_should_calc"].tolist())
    save("dicache.npz", d)


# ------------------------------------------------------------------ full-width C3 fixture
def gen_c3(n_double, n_single, quant_tags, img_h=64, img_w=128, n_txt=512, seed=0, sub=64, attention="ref", rec_double=None,
           rec_single=None, name=None):
    """Full-width FLUX.1-dev at 1024x2048 (S = 8192 + 512) through the reference's own block
    classes, streamed one block at a time so RSS stays small.  The block objects,
    their weight loading/quantisation and forward are the reference's; only the loop over blocks
    (flux.py:445-485) is spelled out here so that each block's weights can be freed."""
    reg = import_reference()
    from oracle import flux as OF
    from oracle import ops as OO
    from fastdm.model.flux import FluxTransformer2DModelCore, FluxTransformerBlock, FluxSingleTransformerBlock
    import contextlib
    import io
    cfg = OF.FluxConfig(num_layers=n_double, num_single_layers=n_single)
    inp = OF.synth_inputs(cfg, 1, img_h, img_w, n_txt, seed)
    t0 = time.time()
    rec_double = (0, n_double - 1) if rec_double is None else tuple(rec_double)
    rec_single = (0, n_single - 1) if rec_single is None else tuple(rec_single)
    attn_key, saved_attn = "sdpa", None          # fastdm/kernel/torch/attention.py:7 registers the op as 'sdpa'
    if attention == "fp32":                      # the A/B control of gen_c3_traj: fp32-score / fp32-softmax attention
        saved_attn = dict(reg._registry[attn_key])

        def attn_fp32(query, key, value, num_q_heads, num_kv_heads, head_dim, is_causal=False, scale=None):
            return OO.scaled_dot_product_attention(query, key, value, num_q_heads, num_kv_heads, head_dim, is_causal=is_causal, scale=scale)
        for b in list(reg._registry[attn_key]):
            reg._registry[attn_key][b] = attn_fp32

    def specs_with_prefix(prefix):
        full = OF.FluxConfig()
        return [(n, s, k) for n, s, k in OF.param_specs(full) if n.startswith(prefix)]

    def load_part(core, names_map, loader):
        sd = {}
        for name, shape, kind in names_map:
            sd[name[1]] = OF.synth_tensor(name[0], shape, kind, seed)
        core.origin_tensor_dict = sd
        core.unmatched_tensors = list(sd.keys())
        core.device, core.dtype = "cpu", torch.bfloat16
        with contextlib.redirect_stdout(io.StringIO()):
            loader()
        assert len(core.unmatched_tensors) == 0, core.unmatched_tensors
        core.origin_tensor_dict = {}

    results = {}
    for tag in quant_tags:
        qt = {"bf16": None, "fp8": torch.float8_e4m3fn, "int8": torch.int8}[tag]
        use_backend("cpu_fp8" if tag == "fp8" else "torch")
        core = FluxTransformer2DModelCore(num_layers=1, num_single_layers=1, quant_dtype=qt)
        # pre/post parts (never quantised)
        pre = [((n, n), s, k) for n, s, k in OF.param_specs(OF.FluxConfig())
               if n.startswith(("time_text_embed", "context_embedder", "x_embedder"))]
        post = [((n, n), s, k) for n, s, k in OF.param_specs(OF.FluxConfig()) if n.startswith(("norm_out", "proj_out"))]
        load_part(core, pre, core._pre_part_loading)
        load_part(core, post, core._post_part_loading)
        d = {}
        with torch.no_grad():
            hidden = core.x_embedder.forward(inp["hidden_states"])
            ts = torch.tensor([1.0]).to(hidden.dtype) * 1000
            gd = torch.tensor([3.5]).to(hidden.dtype) * 1000
            temb = core.time_text_embed.forward(ts, gd, inp["pooled_projections"])
            encoder = core.context_embedder.forward(inp["encoder_hidden_states"])
            ids = torch.cat((inp["txt_ids"], inp["img_ids"]), dim=0)
            rc, rs = core.pos_embed.forward(ids)
            rope = torch.cat((rc[:, 0::2], rs[:, 1::2]), dim=-1).to(hidden.dtype)
            d["temb"] = temb
            d["hidden_in_sub"] = hidden[:, ::sub].clone()
            d["encoder_in_sub"] = encoder[:, ::sub].clone()
            for i in range(n_double):
                names = [((n, n.replace(f"transformer_blocks.{i}.", "transformer_blocks.0.")), s, k)
                         for n, s, k in specs_with_prefix(f"transformer_blocks.{i}.")]
                # fresh block object: QLinear.weight_loading_and_quant casts to the *previous* weight dtype
                core.transformer_blocks = [FluxTransformerBlock(dim=cfg.inner_dim, num_attention_heads=cfg.num_attention_heads,
                                                                attention_head_dim=cfg.attention_head_dim, data_type=torch.bfloat16)]
                load_part(core, names, core._FluxTransformer2DModelCore__transformer_block_loading)
                encoder, hidden = core.transformer_blocks[0].forward(
                    hidden_states=hidden, encoder_hidden_states=encoder, temb=temb, image_rotary_emb=rope)
                print(f"[{tag}] double {i} done  t={time.time() - t0:.0f}s  |h|={hidden.float().abs().mean():.4f}", flush=True)
                if i in rec_double:
                    d[f"hidden_d{i}_sub"] = hidden[:, ::sub].clone()
                    d[f"encoder_d{i}_sub"] = encoder[:, ::sub].clone()
            x = torch.cat([encoder, hidden], dim=1)
            for i in range(n_single):
                names = [((n, n.replace(f"single_transformer_blocks.{i}.", "single_transformer_blocks.0.")), s, k)
                         for n, s, k in specs_with_prefix(f"single_transformer_blocks.{i}.")]
                core.single_transformer_blocks = [FluxSingleTransformerBlock(dim=cfg.inner_dim, num_attention_heads=cfg.num_attention_heads,
                                                                             attention_head_dim=cfg.attention_head_dim, data_type=torch.bfloat16)]
                load_part(core, names, core._FluxTransformer2DModelCore__single_trans_block_loading)
                x = core.single_transformer_blocks[0].forward(hidden_states=x, temb=temb, image_rotary_emb=rope)
                print(f"[{tag}] single {i} done  t={time.time() - t0:.0f}s  |x|={x.float().abs().mean():.4f}", flush=True)
                if i in rec_single:
                    d[f"x_s{i}_sub"] = x[:, ::sub].clone()
            hidden = x[:, n_txt:]
            hidden = core.norm_out.forward(hidden, temb)
            out = core.proj_out.forward(hidden)
            if name is None or "depth" not in name:
                d["out"] = out
            else:
                d["out_sub"] = out[:, ::sub].clone()
        results[tag] = d
        meta = {"seed": seed, "img_h": img_h, "img_w": img_w, "n_txt": n_txt, "sub": sub, "n_double": n_double,
                "n_single": n_single, "timestep": 1.0, "guidance": 3.5}
        d.update({f"meta_{k}": v for k, v in meta.items()})
        if name is None:
            save(f"flux_c3_{tag}_{n_double}x{n_single}.npz", d)
        else:
            d.update({f"meta_prov_{k}": str(v) for k, v in provenance().items()})
            d["meta_attention"] = attention
            save(name.format(tag=tag, attention=attention), d)
    if saved_attn is not None:
        reg._registry[attn_key].update(saved_attn)
    use_backend("torch")
    return results


def gen_c3_depth(quant_tags, attentions, n_double=8, n_single=4, sub=128):
    """DEPTH CURVE (VERDICT r2 'Next round' 1d): after 57 random-weight blocks any 1-ulp difference has saturated at the
    network's chaos floor, so a full-depth comparison cannot see a systematic error.  This fixture is a full-WIDTH C3 forward of
    8 double + 4 single blocks through the reference's own block classes with sub-sampled intermediates after 1, 2, 4, 8 double
    and 1, 2, 4 single blocks -- with the reference's attention ('ref') and with the fp32-softmax control ('fp32').  At these
    depths |control - ref| is still far below saturation, and the HIP path is held to 1.25 x of it at every depth."""
    for att in attentions:
        gen_c3(n_double, n_single, quant_tags, sub=sub, attention=att, rec_double=(0, 1, 3, 7), rec_single=(0, 1, 3),
               name="flux_c3_depth_{tag}_{attention}_%dx%d.npz" % (n_double, n_single))

def gen_c3_traj(n_double, n_single, tag, n_steps=4, attention="ref", img_h=64, img_w=128, n_txt=512, seed=0, sub=8):
    """A multi-step TRAJECTORY of the full-width, full-depth C3 denoise loop through the reference's own block classes
    (north_star: tolerance on FINAL latents): the first `n_steps` steps of the 25-step flow-match schedule with the Euler
    update of pipeline.py / bench.py, `timestep = sigma` in bf16, guidance 3.5.  Unlike gen_c3 the (quantised) blocks stay
    resident across the steps (12 GB fp8 / 24 GB bf16).

    attention='ref'  : the reference torch backend as is (torch SDPA, bf16, CPU flash kernel)
    attention='fp32' : the SAME reference model with its attention operator replaced by the oracle's fp32-score /
                       fp32-softmax restatement (oracle/ops.py) -- the A/B control: two legitimate implementations of the
                       reference's attention contract; their distance after n_steps bounds what a third (the HIP kernel)
                       may differ by."""
    reg = import_reference()
    from oracle import flux as OF
    from oracle import ops as OO
    from fastdm.model.flux import FluxTransformer2DModelCore, FluxTransformerBlock, FluxSingleTransformerBlock
    import contextlib
    import io
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from fastdm_amd.pipeline import flow_match_sigmas
    cfg = OF.FluxConfig(num_layers=n_double, num_single_layers=n_single)
    inp = OF.synth_inputs(cfg, 1, img_h, img_w, n_txt, seed)
    sig = flow_match_sigmas(25, img_h * img_w)
    t0 = time.time()
    qt = {"bf16": None, "fp8": torch.float8_e4m3fn, "int8": torch.int8}[tag]
    use_backend("cpu_fp8" if tag == "fp8" else "torch")
    attn_key, saved_attn = "sdpa", None          # fastdm/kernel/torch/attention.py:7 registers the op as 'sdpa'
    if attention == "fp32":
        saved_attn = dict(reg._registry[attn_key])

        def attn_fp32(query, key, value, num_q_heads, num_kv_heads, head_dim, is_causal=False, scale=None):
            return OO.scaled_dot_product_attention(query, key, value, num_q_heads, num_kv_heads, head_dim, is_causal=is_causal, scale=scale)
        for b in list(reg._registry[attn_key]):
            reg._registry[attn_key][b] = attn_fp32

    def specs_with_prefix(prefix):
        return [(n, s_, k) for n, s_, k in OF.param_specs(OF.FluxConfig()) if n.startswith(prefix)]

    def load_part(core, names_map, loader):
        sd = {}
        for name, shape, kind in names_map:
            sd[name[1]] = OF.synth_tensor(name[0], shape, kind, seed)
        core.origin_tensor_dict = sd
        core.unmatched_tensors = list(sd.keys())
        core.device, core.dtype = "cpu", torch.bfloat16
        with contextlib.redirect_stdout(io.StringIO()):
            loader()
        assert len(core.unmatched_tensors) == 0, core.unmatched_tensors
        core.origin_tensor_dict = {}

    core = FluxTransformer2DModelCore(num_layers=1, num_single_layers=1, quant_dtype=qt)
    pre = [((n, n), s_, k) for n, s_, k in OF.param_specs(OF.FluxConfig()) if n.startswith(("time_text_embed", "context_embedder", "x_embedder"))]
    post = [((n, n), s_, k) for n, s_, k in OF.param_specs(OF.FluxConfig()) if n.startswith(("norm_out", "proj_out"))]
    load_part(core, pre, core._pre_part_loading)
    load_part(core, post, core._post_part_loading)
    doubles, singles = [], []
    for i in range(n_double):
        names = [((n, n.replace(f"transformer_blocks.{i}.", "transformer_blocks.0.")), s_, k) for n, s_, k in specs_with_prefix(f"transformer_blocks.{i}.")]
        core.transformer_blocks = [FluxTransformerBlock(dim=cfg.inner_dim, num_attention_heads=cfg.num_attention_heads,
                                                        attention_head_dim=cfg.attention_head_dim, data_type=torch.bfloat16)]
        load_part(core, names, core._FluxTransformer2DModelCore__transformer_block_loading)
        doubles.append(core.transformer_blocks[0])
    for i in range(n_single):
        names = [((n, n.replace(f"single_transformer_blocks.{i}.", "single_transformer_blocks.0.")), s_, k)
                 for n, s_, k in specs_with_prefix(f"single_transformer_blocks.{i}.")]
        core.single_transformer_blocks = [FluxSingleTransformerBlock(dim=cfg.inner_dim, num_attention_heads=cfg.num_attention_heads,
                                                                     attention_head_dim=cfg.attention_head_dim, data_type=torch.bfloat16)]
        load_part(core, names, core._FluxTransformer2DModelCore__single_trans_block_loading)
        singles.append(core.single_transformer_blocks[0])
    print(f"[{tag}/{attention}] {n_double}+{n_single} blocks resident  t={time.time() - t0:.0f}s", flush=True)

    d = {}
    lat = inp["hidden_states"].clone()
    with torch.no_grad():
        ids = torch.cat((inp["txt_ids"], inp["img_ids"]), dim=0)
        rc, rs = core.pos_embed.forward(ids)
        rope = torch.cat((rc[:, 0::2], rs[:, 1::2]), dim=-1).to(torch.bfloat16)
        for step in range(n_steps):
            hidden = core.x_embedder.forward(lat)
            ts = torch.tensor([float(sig[step])]).to(hidden.dtype) * 1000
            gd = torch.tensor([3.5]).to(hidden.dtype) * 1000
            temb = core.time_text_embed.forward(ts, gd, inp["pooled_projections"])
            encoder = core.context_embedder.forward(inp["encoder_hidden_states"])
            for blk in doubles:
                encoder, hidden = blk.forward(hidden_states=hidden, encoder_hidden_states=encoder, temb=temb, image_rotary_emb=rope)
            x = torch.cat([encoder, hidden], dim=1)
            for blk in singles:
                x = blk.forward(hidden_states=x, temb=temb, image_rotary_emb=rope)
            out = core.proj_out.forward(core.norm_out.forward(x[:, n_txt:], temb))
            lat = (lat.float() + float(sig[step + 1] - sig[step]) * out.float()).to(torch.bfloat16)
            last = step == n_steps - 1
            d[f"out_s{step}"] = out if last else out[:, ::sub].clone()
            d[f"lat_s{step}"] = lat if last else lat[:, ::sub].clone()
            print(f"[{tag}/{attention}] step {step} sigma {sig[step]:.4f} done  t={time.time() - t0:.0f}s  |out|={out.float().abs().mean():.4f} "
                  f"|lat|={lat.float().abs().mean():.4f}", flush=True)
            c3_path = os.path.join(GOLD, f"flux_c3_{tag}_{n_double}x{n_single}.npz")
            if step == 0 and attention == "ref" and os.path.exists(c3_path):
                # the single-forward fixture is the same model, inputs and t = 1.0: step 0 must reproduce it bit for bit
                same = torch.equal(load(os.path.basename(c3_path))["out"], out)
                print(f"[{tag}/{attention}] step-0 output == flux_c3_{tag}_{n_double}x{n_single}.npz: {same}", flush=True)
                assert same, "trajectory step 0 does not reproduce the single-forward fixture: environment or tree differs"
    meta = {"seed": seed, "img_h": img_h, "img_w": img_w, "n_txt": n_txt, "sub": sub, "n_double": n_double, "n_single": n_single,
            "n_steps": n_steps, "guidance": 3.5, "schedule_steps": 25}
    d.update({f"meta_{k}": v for k, v in meta.items()})
    d.update({f"meta_prov_{k}": str(v) for k, v in provenance().items()})
    d["sigmas"] = torch.from_numpy(sig[: n_steps + 1].copy())
    save(f"flux_c3_traj_{tag}_{attention}_{n_double}x{n_single}_{n_steps}steps.npz", d)
    if saved_attn is not None:
        reg._registry[attn_key].update(saved_attn)
    use_backend("torch")
    return d

def gen_c3_equal_inputs(rows=128, seed=0):
    """is the fp8 path's extra noise after ONE block implementation noise of the quantised chain, or
    something else?  Feed EQUAL inputs to the chain: `rows` full-width rows of the C3 block-0 image stream (x_embedder output of
    the synthetic latents) and temb go through the reference's own block-0 pieces -- AdaLayerNormZero, per-token fp8 quantisation,
    the q columns of the fused qkv QLinear, and the feed-forward (GELU-tanh GEMM, re-quantisation, down GEMM) -- and every intermediate is stored, so the
    HIP chain can be held to: identical fp8 codes and scales, GEMM outputs within one bf16 ulp."""
    reg = import_reference()
    from oracle import flux as OF
    from fastdm.model.flux import FluxTransformer2DModelCore, FluxTransformerBlock
    from fastdm.kernel.operators_set import quantize_to_fp8
    import contextlib
    import io
    cfg = OF.FluxConfig(num_layers=1, num_single_layers=1)
    inp = OF.synth_inputs(cfg, 1, 64, 128, 512, seed)
    use_backend("cpu_fp8")
    core = FluxTransformer2DModelCore(num_layers=1, num_single_layers=1, quant_dtype=torch.float8_e4m3fn)

    def load_part(names_map, loader):
        sd = {}
        for name, shape, kind in names_map:
            sd[name[1]] = OF.synth_tensor(name[0], shape, kind, seed)
        core.origin_tensor_dict = sd
        core.unmatched_tensors = list(sd.keys())
        core.device, core.dtype = "cpu", torch.bfloat16
        with contextlib.redirect_stdout(io.StringIO()):
            loader()
        core.origin_tensor_dict = {}
    full = list(OF.param_specs(OF.FluxConfig()))
    load_part([((n, n), s_, k) for n, s_, k in full if n.startswith(("time_text_embed", "context_embedder", "x_embedder"))], core._pre_part_loading)
    core.transformer_blocks = [FluxTransformerBlock(dim=cfg.inner_dim, num_attention_heads=cfg.num_attention_heads,
                                                    attention_head_dim=cfg.attention_head_dim, data_type=torch.bfloat16)]
    load_part([((n, n), s_, k) for n, s_, k in full if n.startswith("transformer_blocks.0.")], core._FluxTransformer2DModelCore__transformer_block_loading)
    blk = core.transformer_blocks[0]
    with torch.no_grad():
        hidden = core.x_embedder.forward(inp["hidden_states"][:, :rows])
        ts = torch.tensor([1.0]).to(hidden.dtype) * 1000
        gd = torch.tensor([3.5]).to(hidden.dtype) * 1000
        temb = core.time_text_embed.forward(ts, gd, inp["pooled_projections"])
        norm_hidden, gate_msa, shift_mlp, scale_mlp, gate_mlp = blk.norm1.forward(hidden, emb=temb)
        codes, scales = quantize_to_fp8(norm_hidden.reshape(-1, norm_hidden.shape[-1]))
        q_out = blk.attn.qkv.forward(norm_hidden)[:, :, : cfg.inner_dim].contiguous()      # the q third of the fused qkv projection
        ff_out = blk.ff.forward(norm_hidden)
    d = {"hidden": hidden, "temb": temb, "norm_hidden": norm_hidden, "codes": codes, "scales": scales.reshape(-1).float(), "q_out": q_out,
         "ff_out": ff_out, "gate_msa": gate_msa, "meta_rows": rows, "meta_seed": seed}
    save("flux_c3_equal_inputs.npz", d)
    use_backend("torch")
    return d

def gen_c3_equal_single(rows=128, seed=0):
    """nothing pinned a SINGLE-stream block's fused chain on equal inputs.  `rows` full-width rows through the
    reference's own FluxSingleTransformerBlock pieces (fastdm/model/flux.py:52-76): AdaLayerNormZeroSingle, proj_mlp (erf-GELU
    afterwards), the attention module on those rows, the per-token fp8 quantisation of cat([attn | mlp]) (15360 wide) and
    proj_out (K = 15360) with gate and residual.  Every intermediate is stored, so each HIP stage can be fed the reference's tensors."""
    import_reference()
    from oracle import flux as OF, layers as OL
    from fastdm.model.flux import FluxTransformer2DModelCore, FluxSingleTransformerBlock
    from fastdm.kernel.operators_set import quantize_to_fp8
    import contextlib
    import io
    import torch.nn.functional as F
    cfg = OF.FluxConfig(num_layers=1, num_single_layers=1)
    inp = OF.synth_inputs(cfg, 1, 64, 128, 512, seed)
    use_backend("cpu_fp8")
    core = FluxTransformer2DModelCore(num_layers=1, num_single_layers=1, quant_dtype=torch.float8_e4m3fn)

    def load_part(names, loader):
        sd = {n: OF.synth_tensor(n, shape, kind, seed) for n, shape, kind in names}
        core.origin_tensor_dict = sd
        core.unmatched_tensors = list(sd.keys())
        core.device, core.dtype = "cpu", torch.bfloat16
        with contextlib.redirect_stdout(io.StringIO()):
            loader()
        core.origin_tensor_dict = {}
    full = list(OF.param_specs(OF.FluxConfig()))
    load_part([x for x in full if x[0].startswith(("time_text_embed", "context_embedder", "x_embedder"))], core._pre_part_loading)
    core.single_transformer_blocks = [FluxSingleTransformerBlock(dim=cfg.inner_dim, num_attention_heads=cfg.num_attention_heads,
                                                                 attention_head_dim=cfg.attention_head_dim, data_type=torch.bfloat16)]
    load_part([x for x in full if x[0].startswith("single_transformer_blocks.0.")], core._FluxTransformer2DModelCore__single_trans_block_loading)
    blk = core.single_transformer_blocks[0]
    with torch.no_grad():
        hidden = core.x_embedder.forward(inp["hidden_states"][:, :rows])
        ts = torch.tensor([1.0]).to(hidden.dtype) * 1000
        gd = torch.tensor([3.5]).to(hidden.dtype) * 1000
        temb = core.time_text_embed.forward(ts, gd, inp["pooled_projections"])
        ids = torch.zeros(rows, 3)
        ids[:, 1], ids[:, 2] = torch.arange(rows) // 16, torch.arange(rows) % 16
        rope = OL.flux_rope_cache(torch.zeros(0, 3), ids, cfg.axes_dims_rope)
        norm_hidden, gate = blk.norm.forward(hidden, emb=temb)
        codes_in, scales_in = quantize_to_fp8(norm_hidden.reshape(-1, norm_hidden.shape[-1]))
        mlp = F.gelu(blk.proj_mlp.forward(norm_hidden))
        attn = blk.attn.forward(hidden_states=norm_hidden, image_rotary_emb=rope)
        cat = torch.cat([attn, mlp], dim=2)
        codes_cat, scales_cat = quantize_to_fp8(cat.reshape(-1, cat.shape[-1]))
        proj = blk.proj_out.forward(cat)
        out = hidden + gate.unsqueeze(1) * proj
    d = {"hidden": hidden, "temb": temb, "norm_hidden": norm_hidden, "gate": gate, "codes_in": codes_in, "scales_in": scales_in.reshape(-1).float(),
         "mlp": mlp, "attn": attn, "codes_cat": codes_cat, "scales_cat": scales_cat.reshape(-1).float(), "proj": proj, "out": out,
         "rope": rope, "meta_rows": rows, "meta_seed": seed}
    d.update(provenance())
    save("flux_c3_equal_single.npz", d)
    use_backend("torch")
    return d


def gen_teacache_at_size(model="flux", n_steps=25):
    """TeaCache at the SIZE the metric is quoted on (VERDICT r2 missing item 2): full width, reduced depth (FLUX 1024 x 2048, S = 8704,
    2 double + 2 single blocks, fp8; Qwen-Image 1328 x 1328, 2 blocks, int8, negative cache), the full 25-step flow-match schedule through the
    reference's own TeaCache.apply_cache (xcaching.py:187-276) with the SHIPPED coefficients and threshold
    (examples/xcaching/configs/teacache_{flux,qwenimage}.json).  Stored per forward: the decision, the rel-L1 distance of the modulated
    input and the accumulated rescaled distance; the final latents.  The two random-weight blocks move the modulated input much less per
    step than a trained 57-block model does, so a second trace with the threshold scaled to the trace's own increments exercises both branches."""
    import_reference()
    import contextlib
    import io
    from fastdm.caching.xcaching import AutoCache
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from fastdm_amd.pipeline import flow_match_sigmas
    t0 = time.time()
    if model == "flux":
        from oracle import flux as OF
        cfg = OF.FluxConfig(num_layers=2, num_single_layers=2)
        sd = OF.synth_state_dict(cfg, seed=0)
        inp = OF.synth_inputs(cfg, 1, 64, 128, 512, 0)
        coeffs = json.load(open(os.path.join(REF, "examples/xcaching/configs/teacache_flux.json")))
        sig = flow_match_sigmas(n_steps, 64 * 128)
        quant, qname = torch.float8_e4m3fn, "fp8"
        use_backend("cpu_fp8")
        conds = [inp["encoder_hidden_states"]]
    else:
        from oracle import qwen as OQ
        cfg = OQ.QwenConfig(num_layers=2)
        sd = OQ.synth_state_dict(cfg, seed=0)
        inp = OQ.synth_inputs(cfg, batch=1, img_h=83, img_w=83, n_txt=128, seed=0)
        neg = OQ.synth_inputs(cfg, batch=1, img_h=83, img_w=83, n_txt=128, seed=1)["encoder_hidden_states"]
        coeffs = json.load(open(os.path.join(REF, "examples/xcaching/configs/teacache_qwenimage.json")))
        coeffs["negtive_cache"] = True
        sig = flow_match_sigmas(n_steps, 83 * 83)
        quant, qname = torch.int8, "int8"
        use_backend("torch")
        conds = [inp["encoder_hidden_states"], neg]
    print(f"[{model}] synthetic weights {time.time() - t0:.0f}s", flush=True)
    d = {"meta_model": model, "meta_quant": qname, "meta_n_steps": n_steps, "sigmas": torch.from_numpy(np.asarray(sig, dtype=np.float64).copy()),
         "coefficients": np.array(coeffs["coefficients"])}
    if "negtive_coefficients" in coeffs and coeffs.get("negtive_coefficients") is not None:
        d["negtive_coefficients"] = np.array(coeffs["negtive_coefficients"])

    def run(tag, thr):
        state = {"step": 0}
        cfgd = dict(coeffs)
        cfgd["threshold"] = thr
        cache = AutoCache.from_dict(cfgd)
        cache.config.current_steps_callback = lambda: state["step"]
        core = ref_flux_core(cfg, quant, sd, cache=cache) if model == "flux" else ref_qwen_core(cfg, quant, sd, cache=cache)
        lat = inp["hidden_states"].clone()
        dec, rel, acc = [], [], []
        for i in range(n_steps):
            state["step"] = i
            row = []
            for ci, enc in enumerate(conds):
                # TeaCache keys: "positive" on the first forward of a step, "negative" on the second (xcaching.py:66-76)
                prev = {k: v.clone() for k, v in cache.previous_modulated_input_dict.items() if v is not None}
                buf = io.StringIO()
                with contextlib.redirect_stdout(buf):
                    if model == "flux":
                        out = core.forward(hidden_states=lat, encoder_hidden_states=enc, pooled_projections=inp["pooled_projections"],
                                           timestep=torch.tensor([float(sig[i])]), img_ids=inp["img_ids"], txt_ids=inp["txt_ids"],
                                           guidance=torch.tensor([3.5]))[0]
                    else:
                        out = core.forward(hidden_states=lat, encoder_hidden_states=enc, encoder_hidden_states_mask=inp["encoder_hidden_states_mask"],
                                           timestep=torch.tensor([float(sig[i])]), img_shapes=inp["img_shapes"], txt_seq_lens=inp["txt_seq_lens"])[0]
                key = "negative" if (len(conds) > 1 and ci == 1) else "positive"     # get_cache_key alternates (xcaching.py:66-76)
                cur = cache.previous_modulated_input_dict[key]
                r = float(((cur - prev[key]).abs().mean() / prev[key].abs().mean()).cpu().item()) if key in prev else 0.0
                rel.append(r)
                acc.append(float(cache.accumulated_rel_l1_distance_dict[key]))
                dec.append(0 if "Skipping" in buf.getvalue() else 1)
                row.append(out)
            if model == "flux":
                comb = row[0].float()
            else:
                comb = (row[1].float() + 4.0 * (row[0].float() - row[1].float()))[..., : cfg.in_channels]
            lat = (lat.float() + float(sig[i + 1] - sig[i]) * comb).to(torch.bfloat16)
            print(f"[{model}/{tag}] step {i} t={time.time() - t0:.0f}s decisions {dec[-len(conds):]} rel {rel[-len(conds):]} acc {acc[-len(conds):]}", flush=True)
        d[f"{tag}_threshold"] = thr
        d[f"{tag}_should_calc"] = np.array(dec, dtype=np.int8)
        d[f"{tag}_rel"] = np.array(rel, dtype=np.float64)
        d[f"{tag}_acc"] = np.array(acc, dtype=np.float64)
        d[f"{tag}_final"] = lat
        return rel

    rel = run("shipped", float(coeffs["threshold"]))
    # second trace: a threshold of ~2.5 typical per-forward increments of THIS trajectory, so that skips and recomputes alternate
    polys = {0: np.poly1d(coeffs["coefficients"])}
    if len(conds) > 1:
        polys[1] = np.poly1d(coeffs["negtive_coefficients"])
    incs = [float(polys[j % len(conds)](r)) for j, r in enumerate(rel) if j >= len(conds)]
    thr2 = float(np.round(2.5 * float(np.median(np.abs(incs))), 6))
    run("scaled", thr2)
    d.update({f"meta_prov_{k}": str(v) for k, v in provenance().items()})
    save(f"teacache_at_size_{model}.npz", d)
    use_backend("torch")


def gen_radial_full():
    """BASELINE.json configs[4] at its real size: Wan2.2-T2V-A14B 768x1280x81f = 21 latent frames x 48 x 80 tokens (S = 80 640),
    the reference's own radial-attention config (examples/sparse/radial_attn_wan.json: block 64, decay 0.3).  The reference's mask
    generator (fastdm/sparse/xsparse.py:71-183) runs on CPU in about a minute; the 1260 x 1260 block mask and the kernel-format
    mask are stored bit-packed."""
    import_reference()
    import contextlib
    import io
    import json
    from fastdm.sparse.xsparse import SparseAttn, sparge_mask_convert
    cfg = json.load(open(os.path.join(REF, "examples", "sparse", "radial_attn_wan.json")))
    frames, tpf = 21, 48 * 80
    sa = SparseAttn.from_dict(cfg)
    sa.post_init(video_token_num=frames * tpf, num_frame=frames)
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        lm = sa.gen_log_mask_shrinked(frames * tpf, torch.device("cpu"))
    km = sparge_mask_convert(lm, block_size=int(cfg["block_size"]), arch="sm80")
    print(f"radial mask {tuple(lm.shape)} -> kernel mask {tuple(km.shape)} in {time.time() - t0:.0f}s; density {lm.float().mean():.4f}")
    d = {"geom": np.array([frames, tpf, int(cfg["block_size"])], dtype=np.int64), "decay": float(cfg["decay_factor"]),
         "log_mask_shape": np.array(lm.shape, dtype=np.int64), "kernel_mask_shape": np.array(km.shape, dtype=np.int64),
         "log_mask_bits": np.packbits(lm.to(torch.uint8).numpy().reshape(-1)),
         "kernel_mask_bits": np.packbits(km.to(torch.uint8).numpy().reshape(-1))}
    os.makedirs(GOLD, exist_ok=True)
    path = os.path.join(GOLD, "wan_radial_mask_a14b_768x1280x81f.npz")
    np.savez_compressed(path, **d)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["small", "ops", "layers", "tiny", "teacache", "qwen", "qwen_full", "wan", "fbcache", "dicache", "sd3", "sd3_full", "wan_i2v", "wan_full", "controlnet", "sd3_wan_cache", "sdxl", "sdxl_full", "c1", "c1_control", "c3", "c2", "c3_traj", "c3_depth", "teacache_flux", "teacache_qwen", "radial_full", "c3_equal", "c3_equal_single", "wan_ti2v"])
    ap.add_argument("--blocks", type=int, nargs=2, default=[19, 38])
    ap.add_argument("--quant", nargs="+", default=["fp8", "bf16"])
    ap.add_argument("--steps", type=int, default=4, help="c3_traj: denoise steps of the 25-step schedule")
    ap.add_argument("--attention", nargs="+", default=["ref"], help="c3_traj: 'ref' (reference torch backend) and / or 'fp32' (A/B control)")
    args = ap.parse_args()
    torch.set_grad_enabled(False)
    if args.what in ("small", "ops"):
        gen_ops()
    if args.what in ("small", "layers"):
        gen_layers()
    if args.what in ("small", "tiny"):
        gen_tiny_flux()
    if args.what in ("small", "teacache"):
        gen_teacache()
    if args.what in ("small", "qwen"):
        gen_tiny_qwen()
    if args.what in ("small", "wan"):
        gen_tiny_wan()
    if args.what in ("small", "fbcache"):
        gen_fbcache()
    if args.what in ("small", "dicache"):
        gen_dicache()
    if args.what in ("small", "sd3"):
        gen_tiny_sd3()
    if args.what in ("small", "controlnet"):
        gen_controlnet_residual_inputs()
    if args.what in ("small", "wan_i2v"):
        gen_tiny_wan_i2v()
    if args.what in ("small", "wan_ti2v"):
        gen_tiny_wan_ti2v()
    if args.what in ("small", "sd3_wan_cache"):
        gen_cache_sd3_wan()
    if args.what in ("small", "sdxl"):
        gen_tiny_sdxl()
    if args.what == "sdxl_full":
        gen_sdxl_full(args.quant)
    if args.what == "c1":
        gen_sdxl_c1()
    if args.what == "c1_control":
        gen_sdxl_c1_control()
    if args.what == "qwen_full":
        gen_qwen_full()
    if args.what == "wan_full":
        gen_wan_full()
    if args.what == "sd3_full":
        gen_sd3_full()
    if args.what == "c3":
        gen_c3(args.blocks[0], args.blocks[1], args.quant)
    if args.what == "c2":        # BASELINE.json configs[1]: FLUX bf16 1024 x 1024 (S = 4096 + 512), full depth
        gen_c3(19, 38, ["bf16"], img_h=64, img_w=64, name="flux_c2_{tag}_19x38.npz")
    if args.what == "c3_traj":
        for tag in args.quant:
            for att in args.attention:
                gen_c3_traj(args.blocks[0], args.blocks[1], tag, n_steps=args.steps, attention=att)
    if args.what == "teacache_flux":
        gen_teacache_at_size("flux")
    if args.what == "teacache_qwen":
        gen_teacache_at_size("qwen")
    if args.what == "c3_depth":
        gen_c3_depth(args.quant, args.attention)
    if args.what == "radial_full":
        gen_radial_full()
    if args.what == "c3_equal":
        gen_c3_equal_inputs()
    if args.what == "c3_equal_single":
        gen_c3_equal_single()


if __name__ == "__main__":
    main()
