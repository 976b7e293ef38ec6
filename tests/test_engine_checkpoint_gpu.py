"""FastDMEngine's CHECKPOINT branch (fastdm_amd/model_entry.py `_init_checkpoint_pipeline`; reference /root/reference/fastdm/
model_entry.py:586-690, 700-793) driven end to end without diffusers: a TEST DOUBLE of the pipeline classes the engine imports
(`DiffusionPipeline`, `WanPipeline`, `WanImageToVideoPipeline`, `AutoencoderKLWan`) is installed as the `diffusers` module.  The
double is not a copy of diffusers: it is the minimal object the engine touches -- `from_pretrained(dir)` reading
`<dir>/transformer[_2]/diffusion_pytorch_model.safetensors` (+ config.json) written by the test from synthetic weights, `.transformer`,
`.transformer_2`, `.scheduler.step_index / .timesteps`, `.to()`, and a `__call__` that runs a short Euler loop calling the denoiser
with exactly the keyword arguments of SURVEY.md §8b B1 inside `cache_context(...)`.

What is asserted: the denoisers are swapped for our wrappers built from the pipeline's own state_dict (both Wan experts, each with
its own step cache and its own config.json), the cache / radial-attention callbacks read the pipeline's scheduler, `generate()`
forwards the reference's kwargs (num_frames, sizes rounded to the sparse block, guidance vs true-CFG), and the images / frames
equal an independent loop over `create_model(...)` on the same weights bit for bit."""
import json
import os
import sys
import types
from contextlib import nullcontext

import pytest
import torch

pytestmark = pytest.mark.gpu
BF16, FP8 = torch.bfloat16, torch.float8_e4m3fn
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------------------------------------ the double
class _Stock(torch.nn.Module):
    """what `pipe.transformer` is before the swap: holds the checkpoint's tensors, knows nothing else"""

    def __init__(self, path):
        super().__init__()
        from safetensors import safe_open
        self._sd = {}
        with safe_open(os.path.join(path, "diffusion_pytorch_model.safetensors"), framework="pt") as f:
            for k in f.keys():
                self._sd[k] = f.get_tensor(k)
        cfg = os.path.join(path, "config.json")
        self.config = types.SimpleNamespace(**(json.load(open(cfg)) if os.path.isfile(cfg) else {}))

    def state_dict(self):
        return dict(self._sd)

    def forward(self, *a, **k):
        raise AssertionError("the stock denoiser must have been replaced before the pipeline is called")


class _Scheduler:
    def __init__(self):
        self.step_index, self.timesteps = None, torch.zeros(0)

    def set_timesteps(self, n):
        self.timesteps = torch.linspace(1000.0, 0.0, n + 1)[:-1]
        self.sigmas = torch.linspace(1.0, 0.0, n + 1)
        self.step_index = 0


class _Out:
    def __init__(self, images=None, frames=None):
        self.images, self.frames = images, frames


class _PipeBase:
    calls = []                      # every __call__'s kwargs, for the assertions
    vae_scale_factor_spatial = 8

    def __init__(self, path, **kw):
        self.path, self.load_kwargs = path, kw
        self.transformer = _Stock(os.path.join(path, "transformer"))
        t2 = os.path.join(path, "transformer_2")
        self.transformer_2 = _Stock(t2) if os.path.isdir(t2) else None
        self.scheduler = _Scheduler()
        self.vae = kw.get("vae")
        self.device_moves = []
        self.denoiser_kwargs = []   # kwargs of every denoiser call

    @classmethod
    def from_pretrained(cls, path, **kw):
        return cls(path, **kw)

    def to(self, dev):
        self.device_moves.append(dev)
        return self


class DiffusionPipeline(_PipeBase):
    """FLUX-shaped double: packed latents [1, S, 64], guidance-distilled (one forward per step)."""

    def __call__(self, prompt=None, negative_prompt=None, num_inference_steps=4, generator=None, max_sequence_length=16, width=64, height=64,
                 guidance_scale=3.5, **extra):
        type(self).calls.append(dict(prompt=prompt, negative_prompt=negative_prompt, num_inference_steps=num_inference_steps, width=width, height=height,
                                     max_sequence_length=max_sequence_length, guidance_scale=guidance_scale, **extra))
        from oracle import flux as OF
        inp = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in OF.synth_inputs(OF.TINY, 1, height // 16, width // 16, max_sequence_length, seed=7).items()}
        lat = inp["hidden_states"].clone()
        self.scheduler.set_timesteps(num_inference_steps)
        g = torch.full((1,), guidance_scale, device="cuda", dtype=torch.float32)
        for i in range(num_inference_steps):
            self.scheduler.step_index = i
            t = torch.full((1,), float(self.scheduler.sigmas[i]), device="cuda", dtype=lat.dtype)
            kw = dict(hidden_states=lat, timestep=t, guidance=g, pooled_projections=inp["pooled_projections"],
                      encoder_hidden_states=inp["encoder_hidden_states"], txt_ids=inp["txt_ids"], img_ids=inp["img_ids"],
                      joint_attention_kwargs=None, return_dict=False)
            self.denoiser_kwargs.append(sorted(kw))
            with self.transformer.cache_context("cond"):
                v = self.transformer(**kw)[0]
            lat = (lat.float() + float(self.scheduler.sigmas[i + 1] - self.scheduler.sigmas[i]) * v.float()).to(lat.dtype)
        return _Out(images=[lat])


class WanPipeline(_PipeBase):
    """Wan2.2-A14B-shaped double: two experts split at a boundary step, CFG = two forwards per step."""

    def __call__(self, prompt=None, negative_prompt=None, num_inference_steps=4, generator=None, max_sequence_length=16, width=64, height=64,
                 guidance_scale=5.0, num_frames=5, **extra):
        type(self).calls.append(dict(prompt=prompt, negative_prompt=negative_prompt, num_inference_steps=num_inference_steps, width=width, height=height,
                                     max_sequence_length=max_sequence_length, guidance_scale=guidance_scale, num_frames=num_frames, **extra))
        from oracle import wan as OW
        inp = {k: (v.cuda() if torch.is_tensor(v) else v)
               for k, v in OW.synth_inputs(OW.TINY, 1, (num_frames - 1) // 4 + 1, height // 8, width // 8, max_sequence_length, seed=5).items()}
        neg = torch.roll(inp["encoder_hidden_states"], 1, dims=1)
        lat = inp["hidden_states"].clone()
        self.scheduler.set_timesteps(num_inference_steps)
        boundary = num_inference_steps // 2
        self.expert_of_step = []
        for i in range(num_inference_steps):
            self.scheduler.step_index = i
            model = self.transformer if (i < boundary or self.transformer_2 is None) else self.transformer_2
            self.expert_of_step.append(1 if model is self.transformer else 2)
            t = torch.full((1,), float(self.scheduler.timesteps[i]), device="cuda", dtype=torch.float32)
            outs = []
            for name, enc in (("cond", inp["encoder_hidden_states"]), ("uncond", neg)):
                kw = dict(hidden_states=lat, timestep=t, encoder_hidden_states=enc, attention_kwargs=None, return_dict=False)
                self.denoiser_kwargs.append(sorted(kw))
                with model.cache_context(name):
                    outs.append(model(**kw)[0].float())
            v = outs[1] + guidance_scale * (outs[0] - outs[1])
            lat = (lat.float() + float(self.scheduler.sigmas[i + 1] - self.scheduler.sigmas[i]) * v).to(lat.dtype)
        return _Out(frames=[lat])


class WanImageToVideoPipeline(WanPipeline):
    pass


class AutoencoderKLWan:
    loaded = []

    @classmethod
    def from_pretrained(cls, path, subfolder=None, torch_dtype=None):
        cls.loaded.append((path, subfolder, torch_dtype))
        return cls()


@pytest.fixture
def fake_diffusers(monkeypatch):
    mod = types.ModuleType("diffusers")
    for c in (DiffusionPipeline, WanPipeline, WanImageToVideoPipeline, AutoencoderKLWan):
        setattr(mod, c.__name__, c)
    monkeypatch.setitem(sys.modules, "diffusers", mod)
    _PipeBase.calls = []
    AutoencoderKLWan.loaded = []
    return mod


def _write_denoiser(dirpath, sd, config):
    from safetensors.torch import save_file
    os.makedirs(dirpath)
    save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(dirpath, "diffusion_pytorch_model.safetensors"))
    with open(os.path.join(dirpath, "config.json"), "w") as f:
        json.dump(config, f)


# ------------------------------------------------------------------------------------------------ FLUX
def test_engine_swaps_a_flux_checkpoint_pipeline_and_matches_the_direct_loop(fake_diffusers, tmp_path):
    from fastdm_amd.model_entry import FastDMEngine, FluxTransformerWrapper, create_model
    from oracle import flux as OF
    cfg = OF.TINY
    sd = OF.synth_state_dict(cfg, seed=11)
    arch = dict(in_channels=cfg.in_channels, out_channels=cfg.out_channels, num_layers=cfg.num_layers, num_single_layers=cfg.num_single_layers,
                attention_head_dim=cfg.attention_head_dim, num_attention_heads=cfg.num_attention_heads, joint_attention_dim=cfg.joint_attention_dim,
                pooled_projection_dim=cfg.pooled_projection_dim, axes_dims_rope=list(cfg.axes_dims_rope))
    _write_denoiser(str(tmp_path / "flux" / "transformer"), sd, arch)
    # the wrapper takes its architecture from keyword arguments (as the reference's does, defaulting to FLUX.1-dev): a tiny
    # checkpoint needs them, so the tiny shape is registered as the 'flux' wrapper's defaults for this test
    from fastdm_amd.model_entry import ModelWrapperFactory
    tiny = lambda **kw: FluxTransformerWrapper(**{**arch, "axes_dims_rope": cfg.axes_dims_rope, **kw})
    old = ModelWrapperFactory._WRAPPER_MAPPING["flux"]
    ModelWrapperFactory._WRAPPER_MAPPING["flux"] = tiny
    try:
        eng = FastDMEngine(str(tmp_path / "flux"), architecture="flux", use_fp8=True, use_diffusers=False,
                           cache_config=os.path.join(ROOT, "examples", "xcaching", "configs", "teacache_flux.json"))
        assert isinstance(eng.pipe.transformer, FluxTransformerWrapper) and eng.pipe.device_moves == ["cuda"]
        assert eng.pipe.load_kwargs == {"torch_dtype": BF16, "use_safetensors": True}
        img = eng.generate(prompt="p", negative_prompt=None, steps=3, guidance_scale=2.5, gen_seed=1, gen_width=96, gen_height=64, max_seq_len=8)
        call = _PipeBase.calls[-1]
        assert call["num_inference_steps"] == 3 and (call["width"], call["height"]) == (96, 64) and call["max_sequence_length"] == 8
        assert call["guidance_scale"] == 2.5 and "true_cfg_scale" not in call and "num_frames" not in call
        # the denoiser is called with exactly these keywords
        assert eng.pipe.denoiser_kwargs[0] == sorted(["hidden_states", "timestep", "guidance", "pooled_projections", "encoder_hidden_states", "txt_ids",
                                                       "img_ids", "joint_attention_kwargs", "return_dict"])
        # the step-cache callbacks read the pipeline's scheduler
        eng.pipe.scheduler.step_index = 2
        assert eng.cache.config.current_steps_callback() == 2 and eng.cache.config.total_steps_callback() == 3
        # same weights, same inputs, our wrapper called directly with a TeaCache of its own (so both loops take the same decisions)
        from fastdm_amd.caching import AutoCache
        cache2 = AutoCache.from_json(os.path.join(ROOT, "examples", "xcaching", "configs", "teacache_flux.json"))
        pipe2 = DiffusionPipeline.__new__(DiffusionPipeline)
        pipe2.scheduler, pipe2.denoiser_kwargs = _Scheduler(), []
        cache2.config.current_steps_callback = lambda: pipe2.scheduler.step_index
        cache2.config.total_steps_callback = lambda: pipe2.scheduler.timesteps.shape[0]
        pipe2.transformer = tiny(ckpt_path=sd, dtype=BF16, quant_type=FP8, kernel_backend="hip", cache=cache2).eval()
        want = pipe2(num_inference_steps=3, guidance_scale=2.5, width=96, height=64, max_sequence_length=8).images[0]
        assert torch.equal(img, want)
        # use_diffusers=True leaves the stock modules alone (model_entry.py:631-632)
        stock = FastDMEngine(str(tmp_path / "flux"), architecture="flux", use_fp8=True, use_diffusers=True)
        assert isinstance(stock.pipe.transformer, _Stock)
    finally:
        ModelWrapperFactory._WRAPPER_MAPPING["flux"] = old


# ------------------------------------------------------------------------------------------------ Wan2.2: two experts
@pytest.mark.parametrize("arch_name", ["wan", "wan-i2v"])
def test_engine_swaps_both_wan_experts_with_their_own_cache_and_config(fake_diffusers, tmp_path, arch_name):
    from fastdm_amd.model_entry import FastDMEngine, WanTransformer3DWrapper, create_model
    from oracle import wan as OW
    cfg = OW.TINY
    conf = dict(patch_size=list(cfg.patch_size), num_attention_heads=cfg.num_attention_heads, attention_head_dim=cfg.attention_head_dim,
                in_channels=cfg.in_channels, out_channels=cfg.out_channels, text_dim=cfg.text_dim, freq_dim=cfg.freq_dim, ffn_dim=cfg.ffn_dim,
                num_layers=cfg.num_layers, qk_norm="rms_norm_across_heads", eps=cfg.eps, rope_max_seq_len=cfg.rope_max_seq_len)
    sd1, sd2 = OW.synth_state_dict(cfg, seed=21), OW.synth_state_dict(cfg, seed=22)
    root = tmp_path / "wan"
    _write_denoiser(str(root / "transformer"), sd1, conf)
    _write_denoiser(str(root / "transformer_2"), sd2, conf)
    sparse_json = tmp_path / "radial.json"
    sparse_json.write_text(json.dumps({"sparse_algorithm": "radial", "backend": "sparse_sageattn", "block_size": 64, "decay_factor": 0.3, "dense_layers": 1,
                                       "dense_steps": 1, "enable_sparse": True, "model_type": "wan"}))
    eng = FastDMEngine(str(root), architecture=arch_name, use_fp8=True, use_diffusers=False, task="t2v",
                       cache_config=os.path.join(ROOT, "examples", "xcaching", "configs", "fbcache_wan.json"), sparse_attn_config=str(sparse_json))
    p = eng.pipe
    assert isinstance(p, WanImageToVideoPipeline if arch_name == "wan-i2v" else WanPipeline)
    if arch_name == "wan":       # the Wan VAE is loaded separately, in fp32 (model_entry.py:608-618)
        assert AutoencoderKLWan.loaded == [(str(root), "vae", torch.float32)] and isinstance(p.vae, AutoencoderKLWan)
    assert isinstance(p.transformer, WanTransformer3DWrapper) and isinstance(p.transformer_2, WanTransformer3DWrapper)
    # each expert got its own config.json and its own step cache; both share the sparse-attention object
    assert len(p.transformer.core_model.blocks) == cfg.num_layers == len(p.transformer_2.core_model.blocks)
    assert p.transformer.core_model.cache is eng.cache and p.transformer_2.core_model.cache is eng.cache_2 and eng.cache is not eng.cache_2
    assert p.transformer.core_model.sparse_attn is eng.sparse_attn is p.transformer_2.core_model.sparse_attn
    frames = eng.generate(prompt="p", negative_prompt="n", num_frames=5, steps=4, guidance_scale=4.0, gen_seed=3, gen_width=100, gen_height=70, max_seq_len=8)
    call = _PipeBase.calls[-1]
    # sizes are rounded UP to the sparse block (64), num_frames is forwarded, guidance_scale (not true_cfg_scale) is used
    assert (call["width"], call["height"], call["num_frames"], call["guidance_scale"]) == (128, 128, 5, 4.0) and "true_cfg_scale" not in call
    assert p.expert_of_step == [1, 1, 2, 2]
    assert p.denoiser_kwargs[0] == sorted(["hidden_states", "timestep", "encoder_hidden_states", "attention_kwargs", "return_dict"])
    for c in (eng.cache, eng.cache_2):
        p.scheduler.step_index = 3
        assert c.config.current_steps_callback() == 3 and c.config.total_steps_callback() == 4
    p.scheduler.step_index = 1
    assert eng.sparse_attn.config.current_steps_callback() == 1
    # independent loop: the same two weight sets through create_model with fresh caches / sparse state
    from fastdm_amd.caching import AutoCache
    from fastdm_amd.sparse import SparseAttn
    sp = SparseAttn.from_json(str(sparse_json))
    caches = [AutoCache.from_json(os.path.join(ROOT, "examples", "xcaching", "configs", "fbcache_wan.json")) for _ in range(2)]
    pipe2 = WanPipeline.__new__(WanPipeline)
    pipe2.scheduler, pipe2.denoiser_kwargs = _Scheduler(), []
    for c in caches:
        c.config.current_steps_callback = lambda: pipe2.scheduler.step_index
        c.config.total_steps_callback = lambda: pipe2.scheduler.timesteps.shape[0]
    sp.config.current_steps_callback = lambda: pipe2.scheduler.step_index
    pipe2.transformer = create_model("wan", ckpt_path=sd1, dtype=BF16, quant_type=FP8, kernel_backend="hip", model_config=conf, cache=caches[0], sparse_attn=sp).eval()
    pipe2.transformer_2 = create_model("wan", ckpt_path=sd2, dtype=BF16, quant_type=FP8, kernel_backend="hip", model_config=conf, cache=caches[1], sparse_attn=sp).eval()
    want = pipe2(num_inference_steps=4, guidance_scale=4.0, width=128, height=128, num_frames=5, max_sequence_length=8).frames[0]
    assert torch.isfinite(frames.float()).all() and torch.equal(frames, want)


def test_engine_rejects_what_it_cannot_do(fake_diffusers, tmp_path):
    from fastdm_amd.model_entry import FastDMEngine
    with pytest.raises(ValueError, match="only supported for Wan"):
        FastDMEngine("", architecture="flux", sparse_attn_config="x.json")
    with pytest.raises(NotImplementedError, match="bfloat16"):
        FastDMEngine("", architecture="flux", data_type="float16")


def test_engine_synthetic_wan_i2v_feeds_latents_and_condition():
    """architecture='wan-i2v' without a checkpoint: synthetic weights of the Wan2.2-I2V wiring (36 = 16 latent + 20 condition input
    channels, 16 out) and the image-to-video loop -- every forward sees cat([latents, condition], dim=1), the latents stay 16-channel."""
    from fastdm_amd.model_entry import FastDMEngine
    eng = FastDMEngine("", architecture="wan-i2v", use_fp8=True, num_attention_heads=2, attention_head_dim=64, text_dim=96, ffn_dim=256, num_layers=2)
    seen = []
    orig = eng.pipe.transformer.forward

    def spy(hidden_states, **kw):
        seen.append(tuple(hidden_states.shape))
        return orig(hidden_states, **kw)
    eng.pipe.transformer.forward = spy
    out = eng.generate(num_frames=5, steps=2, guidance_scale=4.0, gen_seed=1, gen_width=64, gen_height=64, max_seq_len=8)
    assert tuple(out.shape) == (1, 16, 2, 8, 8) and torch.isfinite(out.float()).all()
    assert seen == [(1, 36, 2, 8, 8)] * 4              # 2 steps x (cond, uncond)
