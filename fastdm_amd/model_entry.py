"""Drop-in wrappers and engine (mirror of fastdm/model_entry.py:30-146,463-502,529-793).

`create_model('flux', ckpt_path=<state_dict | path>, dtype=, quant_type=, kernel_backend=, cache=)`
returns an `nn.Module` exposing what a diffusers pipeline touches on `pipe.transformer`: `.config`
(in_channels, out_channels, guidance_embeds, dtype), `.dtype`, `.device`, a no-op `.to()`, `.eval()`,
`cache_context(name)` and `forward(hidden_states, **kw) -> (tensor,)` under no_grad.
`FastDMEngine` wires it into a diffusers pipeline when diffusers is importable; without diffusers
(or with model_path='') it drives the built-in synthetic denoise loop instead.
"""
import json
import os
import types
from contextlib import contextmanager
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from fastdm_amd.caching.xcaching import AutoCache
from fastdm_amd.kernel.utils import set_global_backend
from fastdm_amd.model.flux import FluxTransformer2DModelCore
from fastdm_amd.model.qwenimage import QwenImageTransformer2DModelCore
from fastdm_amd.model.sd35 import SD3TransformerModelCore
from fastdm_amd.model.sdxl import SDXLUNetModelCore
from fastdm_amd.model.wan import WanTransformer3DModelCore


class ConfigMixin:
    def _create_diffusers_config(self, **config_params):
        return types.SimpleNamespace(**config_params)


class BaseModelWrapper(nn.Module, ConfigMixin):
    """model_entry.py:39-93."""

    def __init__(self, kernel_backend: str = "hip", config_path: Optional[str] = None):
        super().__init__()
        set_global_backend(kernel_backend)       # raises ValueError for anything but the HIP backend (alias 'cuda')
        self.model_config_dict = self._load_config(config_path)
        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.core_model = None

    def _load_config(self, config_path: Optional[str]) -> Optional[Dict]:
        if not config_path or not os.path.isfile(config_path):
            return None
        with open(config_path, "r", encoding="utf-8") as f:
            return json.load(f)

    @contextmanager
    def cache_context(self, name: str):
        """diffusers >= 0.35 enters `transformer.cache_context("cond")` around each call; no-op like the reference."""
        yield

    def _load_weights(self, ckpt_path, dtype, quantized_cache=None):
        """Checkpoint (path / directory / state_dict) -> GPU layers.  `quantized_cache`: a safetensors file of the layers as
        the kernels consume them -- written on the first load, read INSTEAD of the checkpoint afterwards (basemodel.py)."""
        have_ckpt = isinstance(ckpt_path, dict) or bool(ckpt_path and os.path.exists(ckpt_path))
        have_cache = bool(quantized_cache and os.path.isfile(quantized_cache))
        if have_cache:      # the checkpoint, when also given, is only fingerprinted and compared with the cache's record
            self.core_model.weight_loading(None, data_type=dtype, device_type="cuda", quantized_cache=quantized_cache,
                                           verify_against=ckpt_path if have_ckpt else None)
        elif have_ckpt:
            self.core_model.weight_loading(ckpt_path, data_type=dtype, device_type="cuda", quantized_cache=quantized_cache)
        elif ckpt_path:
            print(f"Warning: Checkpoint path {ckpt_path} does not exist")

    def to(self, *args, **kwargs):
        """Weights already live on the GPU in their final (quantised) form: `pipe.to('cuda')` must not touch them."""
        return self

    def forward(self, *args, **kwargs):
        if self.core_model is None:
            raise NotImplementedError("Core model not initialized")
        return self.core_model.forward(*args, **kwargs)


class FluxTransformerWrapper(BaseModelWrapper):
    """model_entry.py:96-146."""

    def __init__(self, ckpt_path=None, in_channels: int = 64, out_channels: int = 64, dtype: torch.dtype = torch.bfloat16,
                 quant_type: Optional[torch.dtype] = None, kernel_backend: str = "hip", need_resolve_oom=False,
                 cache: AutoCache = None, **kwargs):
        super().__init__(kernel_backend)
        quantized_cache = kwargs.pop("quantized_cache", None)
        guidance_embeds = kwargs.pop("guidance_embeds", True)
        self.config = self._create_diffusers_config(in_channels=in_channels, out_channels=out_channels,
                                                    guidance_embeds=guidance_embeds, dtype=dtype)
        self.need_resolve_oom = need_resolve_oom
        self.dtype = dtype
        self.core_model = FluxTransformer2DModelCore(in_channels=in_channels, out_channels=out_channels, guidance_embeds=guidance_embeds,
                                                     data_type=dtype, quant_dtype=quant_type, cache=cache, oom_ressolve=need_resolve_oom,
                                                     **kwargs)
        self._load_weights(ckpt_path, dtype, quantized_cache)

    @torch.no_grad()
    def forward(self, hidden_states: torch.Tensor, **kwargs):
        return self.core_model.forward(hidden_states, **kwargs)


class QwenTransformerWrapper(BaseModelWrapper):
    """model_entry.py:148-189.  NB the reference's defaults (out_channels=64) are what its pipelines pass; the
    Qwen-Image checkpoint itself has out_channels=16 with patch_size 2 (proj_out -> 64)."""

    def __init__(self, ckpt_path=None, in_channels: int = 64, out_channels: int = 64, dtype: torch.dtype = torch.bfloat16,
                 quant_type: Optional[torch.dtype] = None, kernel_backend: str = "hip", need_resolve_oom=False,
                 cache: AutoCache = None, **kwargs):
        super().__init__(kernel_backend)
        quantized_cache = kwargs.pop("quantized_cache", None)
        self.config = self._create_diffusers_config(in_channels=in_channels, out_channels=out_channels, guidance_embeds=False, dtype=dtype)
        self.dtype = dtype
        self.need_resolve_oom = need_resolve_oom
        self.core_model = QwenImageTransformer2DModelCore(in_channels=in_channels, out_channels=out_channels, guidance_embeds=False,
                                                          data_type=dtype, quant_dtype=quant_type, cache=cache,
                                                          oom_ressolve=need_resolve_oom, **kwargs)
        self._load_weights(ckpt_path, dtype, quantized_cache)

    @torch.no_grad()
    def forward(self, hidden_states: torch.Tensor, **kwargs):
        return self.core_model.forward(hidden_states, **kwargs)


class SD35TransformerWrapper(BaseModelWrapper):
    """model_entry.py:192-240."""

    def __init__(self, ckpt_path=None, in_channels: int = 16, out_channels: int = 16, dtype: torch.dtype = torch.bfloat16,
                 quant_type: Optional[torch.dtype] = None, kernel_backend: str = "hip", cache: AutoCache = None, need_resolve_oom=False,
                 **kwargs):
        super().__init__(kernel_backend)
        quantized_cache = kwargs.pop("quantized_cache", None)
        self.dtype = dtype
        self.need_resolve_oom = need_resolve_oom        # accepted as in the reference (model_entry.py:206); no tiled VAE is needed here
        self.core_model = SD3TransformerModelCore(in_channels=in_channels, out_channels=out_channels, data_type=dtype, quant_dtype=quant_type,
                                                  cache=cache, **kwargs)
        c = self.core_model.config
        self.config = self._create_diffusers_config(in_channels=in_channels, out_channels=out_channels, dtype=dtype, patch_size=c.patch_size,
                                                    sample_size=c.sample_size, joint_attention_dim=c.joint_attention_dim,
                                                    pooled_projection_dim=c.pooled_projection_dim)
        self._load_weights(ckpt_path, dtype, quantized_cache)

    @torch.no_grad()
    def forward(self, hidden_states: torch.Tensor, **kwargs):
        return self.core_model.forward(hidden_states, **kwargs)


class SDXLUNetModelWrapper(BaseModelWrapper):
    """model_entry.py:240-291 (IP-adapter arguments are accepted and rejected: that branch is not built)."""

    def __init__(self, ckpt_path=None, in_channels=4, dtype=torch.bfloat16, is_ipadapter=False, ipadapter_scale=0.6,
                 diffu_ipadapter_encoder_hid_proj=None, quant_type=None, kernel_backend: str = "hip", **kwargs):
        super().__init__(kernel_backend)
        quantized_cache = kwargs.pop("quantized_cache", None)
        self.config = self._create_diffusers_config(in_channels=in_channels, addition_time_embed_dim=256, sample_size=128,
                                                    time_cond_proj_dim=None, ip_adapter=is_ipadapter, dtype=dtype)
        self.dtype = dtype
        self.core_model = SDXLUNetModelCore(in_channels=in_channels, is_ip_adapter=is_ipadapter, ip_adapter_scale=ipadapter_scale,
                                            diffu_ipadapter_encoder_hid_proj=diffu_ipadapter_encoder_hid_proj, data_type=dtype,
                                            quant_dtype=quant_type, **kwargs)
        self.add_embedding = self.core_model.add_embedding
        self.encoder_hid_proj = self.core_model.encoder_hid_proj
        self._load_weights(ckpt_path, dtype, quantized_cache)

    @torch.no_grad()
    def forward(self, sample, timesteps, encoder_hidden_states, added_cond_kwargs, **kwargs):
        return self.core_model.forward(sample, timesteps, encoder_hidden_states, added_cond_kwargs, **kwargs)


_WAN_A14B_CONFIG = dict(patch_size=(1, 2, 2), num_attention_heads=40, attention_head_dim=128, in_channels=16, out_channels=16, text_dim=4096,
                        freq_dim=256, ffn_dim=13824, num_layers=40, qk_norm="rms_norm_across_heads", eps=1e-6, rope_max_seq_len=1024,
                        image_dim=None)


class WanTransformer3DWrapper(BaseModelWrapper):
    """model_entry.py:410-462.  The architecture comes from the checkpoint's config.json (`config_json`), or from
    `model_config` (a dict with the same keys), defaulting to Wan2.2-T2V-A14B."""

    def __init__(self, ckpt_path=None, in_channels: int = 16, out_channels: int = 16, dtype: torch.dtype = torch.bfloat16,
                 quant_type: Optional[torch.dtype] = None, kernel_backend: str = "hip", config_json=None, cache: AutoCache = None,
                 model_config: Optional[Dict] = None, **kwargs):
        super().__init__(kernel_backend, config_path=config_json)
        quantized_cache = kwargs.pop("quantized_cache", None)
        cfg = dict(_WAN_A14B_CONFIG)
        cfg.update(self.model_config_dict or {})
        cfg.update(model_config or {})
        self.model_config_dict = cfg
        self.config = self._create_diffusers_config(in_channels=cfg["in_channels"], out_channels=cfg["out_channels"], dtype=dtype,
                                                    image_dim=cfg.get("image_dim"), patch_size=cfg["patch_size"])
        self.dtype = dtype
        self.core_model = WanTransformer3DModelCore(
            patch_size=tuple(cfg["patch_size"]), num_attention_heads=cfg["num_attention_heads"], attention_head_dim=cfg["attention_head_dim"],
            in_channels=cfg["in_channels"], out_channels=cfg["out_channels"], text_dim=cfg["text_dim"], freq_dim=cfg["freq_dim"],
            ffn_dim=cfg["ffn_dim"], num_layers=cfg["num_layers"], qk_norm=cfg["qk_norm"], eps=cfg["eps"],
            rope_max_seq_len=cfg["rope_max_seq_len"], image_dim=cfg.get("image_dim"), added_kv_proj_dim=cfg.get("added_kv_proj_dim"),
            pos_embed_seq_len=cfg.get("pos_embed_seq_len"), data_type=dtype, quant_dtype=quant_type, cache=cache,
            sparse_attn=kwargs.get("sparse_attn", None))
        self._load_weights(ckpt_path, dtype, quantized_cache)

    @torch.no_grad()
    def forward(self, hidden_states: torch.Tensor, **kwargs):
        return self.core_model.forward(hidden_states, **kwargs)


class ModelWrapperFactory:
    """model_entry.py:463-497."""
    _WRAPPER_MAPPING = {"flux": FluxTransformerWrapper, "qwen": QwenTransformerWrapper, "wan": WanTransformer3DWrapper,
                        "sd35": SD35TransformerWrapper, "sdxl": SDXLUNetModelWrapper}
    _NOT_BUILT = ("sdxl-controlnet", "flux-controlnet")

    @classmethod
    def create_wrapper(cls, model_type: str, **kwargs) -> BaseModelWrapper:
        if model_type in cls._WRAPPER_MAPPING:
            return cls._WRAPPER_MAPPING[model_type](**kwargs)
        if model_type in cls._NOT_BUILT:
            raise NotImplementedError(f"model type '{model_type}' is on the roadmap (SURVEY.md §8f) but not built yet")
        raise ValueError(f"Unsupported model type: {model_type}. Supported types: {list(cls._WRAPPER_MAPPING)}")

    @classmethod
    def get_available_models(cls) -> List[str]:
        return list(cls._WRAPPER_MAPPING)

    @classmethod
    def register_wrapper(cls, model_type: str, wrapper_class: type):
        if not issubclass(wrapper_class, BaseModelWrapper):
            raise TypeError("Wrapper class must inherit from BaseModelWrapper")
        cls._WRAPPER_MAPPING[model_type] = wrapper_class


def create_model(model_type: str, **kwargs) -> BaseModelWrapper:
    return ModelWrapperFactory.create_wrapper(model_type, **kwargs)


def list_available_models() -> List[str]:
    return ModelWrapperFactory.get_available_models()


class FastDMEngine:
    """model_entry.py:529-834.

    With diffusers + a checkpoint directory (`_init_checkpoint_pipeline`): the architecture's pipeline class is loaded from `model_path`
    and its denoiser -- `unet` (sdxl), `transformer` (flux / sd35 / qwen), `transformer` + `transformer_2` with one step cache each
    (wan, wan-i2v: Wan2.2-A14B's two experts) -- is replaced by ours built from the pipeline's own state_dict; `generate()` calls the
    pipeline with the reference's kwargs.  diffusers is not in this image: tests/test_engine_checkpoint_gpu.py drives this branch with
    a test double of the pipeline classes fed from synthetic safetensors directories.
    Without (model_path == '' or no such directory): synthetic weights + the `Synthetic*Pipeline` of the architecture, whose
    `generate()` returns the final packed latents instead of a decoded image."""

    def __init__(self, model_path, architecture="flux", device=0, data_type="bfloat16", use_fp8=False, use_int8=False,
                 kernel_backend="hip", cache_config=None, oom_resolve=False, use_diffusers=True, task="t2i",
                 sparse_attn_config=None, synthetic_seed=0, **model_kwargs):
        if architecture not in ("flux", "qwen", "wan", "wan-i2v", "sd35", "sdxl"):
            raise NotImplementedError(f"architecture '{architecture}' is not built yet (SURVEY.md §8f)")
        if data_type != "bfloat16":
            raise NotImplementedError("bfloat16 activations only: the HIP kernels of this path are written for bf16 (the reference's float16 "
                                      "column is its RTX-8000 one, /root/reference/fastdm/model_entry.py:572)")
        if sparse_attn_config and architecture not in ("wan", "wan-i2v"):
            raise ValueError("Sparse attention is only supported for Wan models")
        self.architecture, self.device, self.task = architecture, device, task
        self.oom_resolve, self.use_diffusers = oom_resolve, use_diffusers
        torch.cuda.set_device(device)
        self.dtype = torch.bfloat16
        self.quant_type = torch.float8_e4m3fn if use_fp8 else (torch.int8 if use_int8 else None)
        self.cache = AutoCache.from_json(cache_config) if cache_config else None
        # Wan2.2-A14B is two experts (transformer: high noise, transformer_2: low noise) with a step cache each (model_entry.py:586-590)
        self.cache_2 = AutoCache.from_json(cache_config) if cache_config and architecture in ("wan", "wan-i2v") else None
        self.sparse_attn = None
        self.pipe = None
        have_ckpt = bool(model_path) and os.path.exists(model_path)
        if have_ckpt:
            self._init_checkpoint_pipeline(model_path, kernel_backend, sparse_attn_config)
        elif architecture in ("wan", "wan-i2v"):
            if architecture == "wan-i2v":      # Wan2.2-I2V-A14B: the T2V wiring with 16 latent + 20 condition input channels
                model_kwargs.setdefault("in_channels", 36)
                model_kwargs.setdefault("out_channels", 16)
            from fastdm_amd.pipeline import SyntheticWanPipeline
            from fastdm_amd.sparse import SparseAttn
            from fastdm_amd.synthetic import SyntheticWanStateDict
            self.sparse_attn = SparseAttn.from_json(sparse_attn_config) if sparse_attn_config else None
            sd = SyntheticWanStateDict(seed=synthetic_seed, **model_kwargs)
            transformer = create_model("wan", ckpt_path=sd, dtype=self.dtype, quant_type=self.quant_type, kernel_backend=kernel_backend,
                                       cache=self.cache, model_config=model_kwargs, sparse_attn=self.sparse_attn).eval()
            self.pipe = SyntheticWanPipeline(transformer, seed=synthetic_seed)
            if self.sparse_attn:
                self.sparse_attn.config.current_steps_callback = lambda: self.pipe.scheduler_step_index
        elif architecture == "sd35":
            from fastdm_amd.pipeline import SyntheticSD3Pipeline
            from fastdm_amd.synthetic import SyntheticSD3StateDict
            arch = dict(num_layers=38, num_attention_heads=38, attention_head_dim=64, caption_projection_dim=2432, pos_embed_max_size=192,
                        dual_attention_layers=())                                     # SD3.5-large unless overridden
            arch.update(model_kwargs)
            transformer = create_model("sd35", ckpt_path=SyntheticSD3StateDict(seed=synthetic_seed, **arch), dtype=self.dtype,
                                       quant_type=self.quant_type, kernel_backend=kernel_backend, cache=self.cache, **arch).eval()
            self.pipe = SyntheticSD3Pipeline(transformer, seed=synthetic_seed)
            if self.cache:
                self.cache.config.current_steps_callback = lambda: self.pipe.scheduler_step_index
                self.cache.config.total_steps_callback = lambda: self.pipe.num_steps
        elif architecture == "sdxl":
            from fastdm_amd.pipeline import SyntheticSDXLPipeline
            from fastdm_amd.synthetic import SyntheticSDXLStateDict
            if self.cache:
                raise ValueError("the SDXL UNet has no step cache (the reference has none either, README perf table)")
            use_graph = bool(model_kwargs.pop("use_graph", False))
            unet = create_model("sdxl", ckpt_path=SyntheticSDXLStateDict(seed=synthetic_seed, **model_kwargs), dtype=self.dtype,
                                quant_type=self.quant_type, kernel_backend=kernel_backend, **model_kwargs).eval()
            self.pipe = SyntheticSDXLPipeline(unet, seed=synthetic_seed, use_graph=use_graph)
        elif architecture == "qwen":
            from fastdm_amd.pipeline import SyntheticQwenPipeline
            from fastdm_amd.synthetic import SyntheticQwenStateDict
            batch_cfg = model_kwargs.pop("batch_cfg", None)      # cond + uncond as one batch-2 forward (pipeline.py); None: whenever no step cache is set
            batch_cfg = (self.cache is None) if batch_cfg is None else bool(batch_cfg)
            if batch_cfg and self.cache is not None:
                raise ValueError("batch_cfg runs both prompts in one forward; the step caches key on the cond / uncond alternation")
            keys = ("patch_size", "num_layers", "attention_head_dim", "num_attention_heads", "joint_attention_dim", "in_channels", "out_channels")
            sd = SyntheticQwenStateDict(seed=synthetic_seed, **{k: v for k, v in model_kwargs.items() if k in keys})
            model_kwargs.setdefault("out_channels", 16)      # the checkpoint's value (proj_out = patch^2 * 16 = 64 = in_channels)
            transformer = create_model("qwen", ckpt_path=sd, dtype=self.dtype, quant_type=self.quant_type, kernel_backend=kernel_backend,
                                       cache=self.cache, **model_kwargs).eval()
            self.pipe = SyntheticQwenPipeline(transformer, seed=synthetic_seed, batch_cfg=batch_cfg)
            if self.cache:
                self.cache.config.current_steps_callback = lambda: self.pipe.scheduler_step_index
                self.cache.config.total_steps_callback = lambda: self.pipe.num_steps
        else:
            from fastdm_amd.pipeline import SyntheticFluxPipeline
            from fastdm_amd.synthetic import SyntheticFluxStateDict
            sd = SyntheticFluxStateDict(seed=synthetic_seed, **{k: v for k, v in model_kwargs.items() if k in
                                                                ("num_layers", "num_single_layers", "attention_head_dim", "num_attention_heads",
                                                                 "joint_attention_dim", "pooled_projection_dim", "in_channels", "out_channels")})
            transformer = create_model("flux", ckpt_path=sd, dtype=self.dtype, quant_type=self.quant_type, kernel_backend=kernel_backend,
                                       cache=self.cache, **model_kwargs).eval()
            self.pipe = SyntheticFluxPipeline(transformer, seed=synthetic_seed)
            if self.cache:
                self.cache.config.current_steps_callback = lambda: self.pipe.scheduler_step_index
                self.cache.config.total_steps_callback = lambda: self.pipe.num_steps

    def _init_checkpoint_pipeline(self, model_path, kernel_backend, sparse_attn_config):
        """model_entry.py:606-700: load the diffusers pipeline of `model_path`, then (unless use_diffusers asks for the stock modules)
        swap its denoiser(s) for ours built from the pipeline's own state_dict, wire the step-cache / radial-attention callbacks to the
        pipeline's scheduler, and move the pipeline to the GPU."""
        try:
            import diffusers
        except ImportError as e:
            raise ImportError("a checkpoint path was given but diffusers is not installed; use model_path='' for the synthetic pipeline") from e
        arch = self.architecture
        if sparse_attn_config:
            from fastdm_amd.sparse import SparseAttn
            self.sparse_attn = SparseAttn.from_json(sparse_attn_config)
        if arch == "wan":          # model_entry.py:608-618: the Wan VAE is loaded in fp32
            vae = diffusers.AutoencoderKLWan.from_pretrained(model_path, subfolder="vae", torch_dtype=torch.float32)
            self.pipe = diffusers.WanPipeline.from_pretrained(model_path, vae=vae, torch_dtype=torch.bfloat16)
        elif arch == "wan-i2v":
            self.pipe = diffusers.WanImageToVideoPipeline.from_pretrained(model_path, torch_dtype=torch.bfloat16, use_safetensors=True)
        else:
            self.pipe = diffusers.DiffusionPipeline.from_pretrained(model_path, torch_dtype=self.dtype, use_safetensors=True)
        if self.use_diffusers:      # the stock pipeline, untouched (model_entry.py:631-632)
            self.pipe.to("cuda")
            return
        self.model_path = model_path
        for cache in (self.cache, self.cache_2):
            if cache is not None:
                cache.config.current_steps_callback = lambda: self.pipe.scheduler.step_index
                cache.config.total_steps_callback = lambda: self.pipe.scheduler.timesteps.shape[0]
        if self.sparse_attn is not None and self.sparse_attn.config.sparse_algorithm == "radial":
            self.sparse_attn.config.current_steps_callback = lambda: self.pipe.scheduler.step_index
        common = dict(dtype=self.dtype, quant_type=self.quant_type, kernel_backend=kernel_backend)
        if arch == "sdxl":          # the UNet is the swapped module and has no step cache
            self.pipe.unet = create_model("sdxl", ckpt_path=self.pipe.unet.state_dict(), **common).eval()
        elif arch in ("flux", "sd35", "qwen"):
            self.pipe.transformer = create_model(arch, ckpt_path=self.pipe.transformer.state_dict(), cache=self.cache,
                                                 need_resolve_oom=self.oom_resolve, **common).eval()
            # (the reference's oom_resolve also swaps tiled VAE decoders in, fastdm/utils/{qwen,flux}_vae.py: with 288 GB of HBM the
            # stock decoders fit, so the flag is accepted and changes nothing here)
        else:                       # wan / wan-i2v: one or two experts, each with its config.json and its own cache
            self.pipe.transformer = create_model("wan", ckpt_path=self.pipe.transformer.state_dict(), cache=self.cache, sparse_attn=self.sparse_attn,
                                                 config_json=f"{model_path}/transformer/config.json", **common).eval()
            if getattr(self.pipe, "transformer_2", None) is not None:
                self.pipe.transformer_2 = create_model("wan", ckpt_path=self.pipe.transformer_2.state_dict(), cache=self.cache_2,
                                                       sparse_attn=self.sparse_attn, config_json=f"{model_path}/transformer_2/config.json",
                                                       **common).eval()
        self.pipe.to("cuda")
        torch.cuda.empty_cache()

    def i2v_image_processor(self, input_image, width=832, height=480):
        """model_entry.py:796-810: resize the source image to the requested AREA at its own aspect ratio, on the VAE x patch grid."""
        import numpy as np
        if isinstance(input_image, str):
            if not os.path.isfile(input_image):
                raise FileNotFoundError(f"Input image file does not exist: {input_image}")
            from diffusers.utils import load_image
            input_image = load_image(input_image)
        aspect = input_image.height / input_image.width
        mod = self.pipe.vae_scale_factor_spatial * self.pipe.transformer.config.patch_size[1]
        h = round(np.sqrt(height * width * aspect)) // mod * mod
        w = round(np.sqrt(height * width / aspect)) // mod * mod
        return input_image.resize((w, h)), h, w

    def i2i_image_processor(self, input_image, width=832, height=480):
        """model_entry.py:812-834: one image (path or already loaded) or a list of them; the generation takes their largest size."""
        def load(x):
            if isinstance(x, str) and os.path.isfile(x):
                from diffusers.utils import load_image
                return load_image(x)
            return x
        if isinstance(input_image, list):
            images = [load(x) for x in input_image]
            return images, max(i.height for i in images), max(i.width for i in images)
        image = load(input_image)
        return image, image.height, image.width

    def generate(self, prompt=None, negative_prompt=None, src_image=None, num_frames=None, steps=50, guidance_scale=3.5,
                 true_cfg_scale=None, gen_seed=42, gen_width=512, gen_height=512, max_seq_len=512):
        from fastdm_amd.pipeline import (SyntheticFluxPipeline, SyntheticQwenPipeline, SyntheticSD3Pipeline, SyntheticSDXLPipeline,
                                         SyntheticWanPipeline)
        if isinstance(self.pipe, (SyntheticSD3Pipeline, SyntheticSDXLPipeline)):
            return self.pipe(num_inference_steps=steps, guidance_scale=guidance_scale, width=gen_width, height=gen_height,
                             max_sequence_length=max_seq_len, seed=gen_seed)
        if isinstance(self.pipe, SyntheticWanPipeline):
            return self.pipe(num_inference_steps=steps, guidance_scale=guidance_scale, width=gen_width, height=gen_height,
                             num_frames=num_frames or 81, max_sequence_length=max_seq_len, seed=gen_seed)
        if isinstance(self.pipe, SyntheticQwenPipeline):
            return self.pipe(num_inference_steps=steps, true_cfg_scale=true_cfg_scale, width=gen_width, height=gen_height,
                             max_sequence_length=max_seq_len, seed=gen_seed)
        if isinstance(self.pipe, SyntheticFluxPipeline):
            return self.pipe(num_inference_steps=steps, guidance_scale=guidance_scale, width=gen_width, height=gen_height,
                             max_sequence_length=max_seq_len, seed=gen_seed)
        # checkpoint pipelines: the reference's kwargs (model_entry.py:748-793)
        gen = torch.Generator().manual_seed(gen_seed)
        kwargs = {"prompt": prompt, "negative_prompt": negative_prompt, "num_inference_steps": steps, "generator": gen,
                  "max_sequence_length": max_seq_len}
        if src_image is not None and self.task == "i2v":
            kwargs["image"], gen_height, gen_width = self.i2v_image_processor(src_image, width=gen_width, height=gen_height)
        elif src_image is not None and self.task == "i2i":
            kwargs["image"], gen_height, gen_width = self.i2i_image_processor(src_image, width=gen_width, height=gen_height)
        if self.sparse_attn is not None:      # sizes rounded up to the sparse block
            bs = self.sparse_attn.config.block_size
            gen_height, gen_width = (gen_height + bs - 1) // bs * bs, (gen_width + bs - 1) // bs * bs
        kwargs["width"], kwargs["height"] = gen_width, gen_height
        if num_frames is not None:
            kwargs["num_frames"] = num_frames
        if true_cfg_scale is not None:        # Qwen-Image
            kwargs["true_cfg_scale"] = true_cfg_scale
        else:
            kwargs["guidance_scale"] = guidance_scale
        with torch.inference_mode():
            output = self.pipe(**kwargs)
        return output.frames[0] if num_frames is not None else output.images[0]
