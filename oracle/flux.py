"""FLUX transformer oracle (torch-CPU restatement of fastdm/model/flux.py).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Citations relative to /root/reference.
"""
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import layers as L
from . import synth


@dataclass
class FluxConfig:
    """Defaults = FLUX.1-dev (fastdm/model/flux.py:204-221)."""
    in_channels: int = 64
    out_channels: int = 64
    num_layers: int = 19
    num_single_layers: int = 38
    attention_head_dim: int = 128
    num_attention_heads: int = 24
    joint_attention_dim: int = 4096
    pooled_projection_dim: int = 768
    axes_dims_rope: Tuple[int, ...] = (16, 56, 56)
    mlp_ratio: float = 4.0

    @property
    def inner_dim(self):
        return self.attention_head_dim * self.num_attention_heads


# head_dim 64 (the smallest the HIP attention kernel supports), everything else shrunk
TINY = FluxConfig(in_channels=16, out_channels=16, num_layers=2, num_single_layers=2, attention_head_dim=64,
                  num_attention_heads=2, joint_attention_dim=64, pooled_projection_dim=48, axes_dims_rope=(16, 24, 24))


# ------------------------------------------------------------------ synthetic weights
def param_specs(cfg: FluxConfig):
    """Yield (diffusers key, shape, kind) for every tensor FluxTransformer2DModelCore consumes
    (fastdm/model/flux.py:274-332).  kind in {'w', 'w_mod', 'b', 'rms'} selects the distribution."""
    D, mlp = cfg.inner_dim, int(cfg.inner_dim * cfg.mlp_ratio)
    hd = cfg.attention_head_dim

    def lin(name, k, n, kind="w"):
        yield f"{name}.weight", (n, k), kind
        yield f"{name}.bias", (n,), "b"

    for p in ("timestep_embedder", "guidance_embedder"):
        yield from lin(f"time_text_embed.{p}.linear_1", 256, D)
        yield from lin(f"time_text_embed.{p}.linear_2", D, D)
    yield from lin("time_text_embed.text_embedder.linear_1", cfg.pooled_projection_dim, D)
    yield from lin("time_text_embed.text_embedder.linear_2", D, D)
    yield from lin("context_embedder", cfg.joint_attention_dim, D)
    yield from lin("x_embedder", cfg.in_channels, D)
    for i in range(cfg.num_layers):
        p = f"transformer_blocks.{i}"
        yield from lin(f"{p}.norm1.linear", D, 6 * D, "w_mod")
        yield from lin(f"{p}.norm1_context.linear", D, 6 * D, "w_mod")
        for n in ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_out.0", "to_add_out"):
            yield from lin(f"{p}.attn.{n}", D, D)
        for n in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
            yield f"{p}.attn.{n}.weight", (hd,), "rms"
        for ff in ("ff", "ff_context"):
            yield from lin(f"{p}.{ff}.net.0.proj", D, mlp)
            yield from lin(f"{p}.{ff}.net.2", mlp, D)
    for i in range(cfg.num_single_layers):
        p = f"single_transformer_blocks.{i}"
        yield from lin(f"{p}.norm.linear", D, 3 * D, "w_mod")
        yield from lin(f"{p}.proj_mlp", D, mlp)
        yield from lin(f"{p}.proj_out", D + mlp, D)
        for n in ("to_q", "to_k", "to_v"):
            yield from lin(f"{p}.attn.{n}", D, D)
        for n in ("norm_q", "norm_k"):
            yield f"{p}.attn.{n}.weight", (hd,), "rms"
    yield from lin("norm_out.linear", D, 2 * D, "w_mod")
    yield from lin("proj_out", D, cfg.out_channels)


def synth_dist(kind: str, shape):
    """(std, offset) of the synthetic distribution: Linear W ~ N(0, 1/fan_in); modulation
    linears and all biases N(0, 0.02^2); RMS-norm weights 1 + 0.1 N(0,1)."""
    if kind == "w":
        return (1.0 / shape[1]) ** 0.5, 0.0
    if kind in ("w_mod", "b"):
        return 0.02, 0.0
    if kind == "rms":
        return 0.1, 1.0
    raise ValueError(kind)


def bf16_from_bits(bits: np.ndarray, shape) -> torch.Tensor:
    return torch.from_numpy(bits.view(np.int16).copy()).view(torch.bfloat16).reshape(shape)


def synth_tensor(name: str, shape, kind: str, seed: int = 0) -> torch.Tensor:
    std, off = synth_dist(kind, shape)
    n = int(np.prod(shape))
    return bf16_from_bits(synth.normal_bf16_bits(synth.key_seed(name, seed), n, std, off), shape)


def synth_state_dict(cfg: FluxConfig, seed: int = 0, only_prefix: Optional[str] = None) -> Dict[str, torch.Tensor]:
    sd = {}
    for name, shape, kind in param_specs(cfg):
        if only_prefix is not None and not name.startswith(only_prefix):
            continue
        sd[name] = synth_tensor(name, shape, kind, seed)
    return sd


def synth_inputs(cfg: FluxConfig, batch: int, img_h: int, img_w: int, n_txt: int, seed: int = 0):
    """Synthetic step inputs, mirroring examples/profiling/flux_profiling.py:16-23 (U[0,1) bf16),
    with the real pipelines' img_ids grid (fastdm/comfyui_entry.py:302-309).  img_h x img_w is the
    packed-latent grid (pixels / 16)."""
    n_img = img_h * img_w

    def uni(tag, shape):
        bits = synth.uniform_bf16_bits(synth.key_seed(f"input.{tag}", seed), int(np.prod(shape)))
        return bf16_from_bits(bits, shape)

    ids = torch.zeros(n_img, 3)
    ids[:, 1] = torch.arange(img_h).repeat_interleave(img_w).float()
    ids[:, 2] = torch.arange(img_w).repeat(img_h).float()
    return dict(
        hidden_states=uni("latents", (batch, n_img, cfg.in_channels)),
        encoder_hidden_states=uni("prompt_embeds", (batch, n_txt, cfg.joint_attention_dim)),
        pooled_projections=uni("pooled", (batch, cfg.pooled_projection_dim)),
        img_ids=ids,
        txt_ids=torch.zeros(n_txt, 3),
    )


# ------------------------------------------------------------------ blocks (a13)
class FluxTransformerBlock:
    """fastdm/model/flux.py:78-178."""

    def __init__(self, sd, i, cfg: FluxConfig, quant):
        p = f"transformer_blocks.{i}"
        self.norm1 = L.AdaLayerNormZero(sd, f"{p}.norm1.linear")
        self.norm1_context = L.AdaLayerNormZero(sd, f"{p}.norm1_context.linear")
        self.attn = L.FluxAttention(sd, f"{p}.attn", cfg.num_attention_heads, cfg.attention_head_dim, quant, joint=True)
        self.ff = L.FeedForwardGeluTanh(sd, f"{p}.ff", quant)
        self.ff_context = L.FeedForwardGeluTanh(sd, f"{p}.ff_context", quant)

    def __call__(self, hidden, encoder, temb, rope):
        nh, gate_msa, shift_mlp, scale_mlp, gate_mlp = self.norm1(hidden, temb)
        ne, c_gate_msa, c_shift_mlp, c_scale_mlp, c_gate_mlp = self.norm1_context(encoder, temb)
        attn, c_attn = self.attn(nh, ne, rope)
        hidden = hidden + gate_msa.unsqueeze(1) * attn
        hidden = hidden + gate_mlp.unsqueeze(1) * self.ff(L.modulate(hidden, scale_mlp, shift_mlp))
        encoder = encoder + c_gate_msa.unsqueeze(1) * c_attn
        encoder = encoder + c_gate_mlp.unsqueeze(1) * self.ff_context(L.modulate(encoder, c_scale_mlp, c_shift_mlp))
        return encoder, hidden


class FluxSingleTransformerBlock:
    """fastdm/model/flux.py:17-76: erf-GELU MLP, pre_only attention, cat -> proj_out."""

    def __init__(self, sd, i, cfg: FluxConfig, quant):
        p = f"single_transformer_blocks.{i}"
        self.norm = L.AdaLayerNormZeroSingle(sd, f"{p}.norm.linear")
        self.proj_mlp = L.linear_from(sd, [f"{p}.proj_mlp"], quant)
        self.proj_out = L.linear_from(sd, [f"{p}.proj_out"], quant)
        self.attn = L.FluxAttention(sd, f"{p}.attn", cfg.num_attention_heads, cfg.attention_head_dim, quant, joint=False)

    def __call__(self, hidden, temb, rope):
        nh, gate = self.norm(hidden, temb)
        mlp = F.gelu(self.proj_mlp(nh))
        attn = self.attn(nh, None, rope)
        out = self.proj_out(torch.cat([attn, mlp], dim=2))
        return hidden + gate.unsqueeze(1) * out


class FluxCore:
    """fastdm/model/flux.py:180-494 (FluxTransformer2DModelCore), no controlnet, no OOM mode."""

    def __init__(self, sd, cfg: FluxConfig = FluxConfig(), quant=torch.float8_e4m3fn, cache=None):
        self.cfg, self.quant, self.cache = cfg, quant, cache
        self.time_text_embed = L.TimeTextEmbed(sd)
        self.context_embedder = L.linear_from(sd, ["context_embedder"])
        self.x_embedder = L.linear_from(sd, ["x_embedder"])
        self.transformer_blocks = [FluxTransformerBlock(sd, i, cfg, quant) for i in range(cfg.num_layers)]
        self.single_transformer_blocks = [FluxSingleTransformerBlock(sd, i, cfg, quant) for i in range(cfg.num_single_layers)]
        self.norm_out = L.AdaLayerNormContinuous(sd, "norm_out.linear")
        self.proj_out = L.linear_from(sd, ["proj_out"])

    def embed(self, hidden_states, encoder_hidden_states, pooled_projections, timestep, img_ids, txt_ids, guidance):
        """flux.py:397-428.  timestep/guidance are cast to the activation dtype and multiplied
        by 1000 *in that dtype* (bf16: 0.75 -> 752)."""
        hidden = self.x_embedder(hidden_states)
        ts = timestep.to(hidden.dtype) * 1000
        gd = guidance.to(hidden.dtype) * 1000
        temb = self.time_text_embed(ts, gd, pooled_projections)
        encoder = self.context_embedder(encoder_hidden_states)
        if txt_ids.ndim == 3:
            txt_ids = txt_ids[0]
        if img_ids.ndim == 3:
            img_ids = img_ids[0]
        rope = L.flux_rope_cache(txt_ids, img_ids, self.cfg.axes_dims_rope, hidden.dtype)
        return hidden, encoder, temb, rope

    def run_blocks(self, hidden, encoder, temb, rope, cn_double=None, cn_single=None, cn_repeat=False):
        """flux.py:445-485, with the ControlNet residual inputs (flux.py:454-480)."""
        import math
        nd, ns = len(self.transformer_blocks), len(self.single_transformer_blocks)
        for i, blk in enumerate(self.transformer_blocks):
            encoder, hidden = blk(hidden, encoder, temb, rope)
            if cn_double is not None:
                hidden = hidden + cn_double[i % len(cn_double) if cn_repeat else i // int(math.ceil(nd / len(cn_double)))]
        n_txt = encoder.shape[1]
        x = torch.cat([encoder, hidden], dim=1)
        for i, blk in enumerate(self.single_transformer_blocks):
            x = blk(x, temb, rope)
            if cn_single is not None:
                x[:, n_txt:] = x[:, n_txt:] + cn_single[i // int(math.ceil(ns / len(cn_single)))]
        return x[:, n_txt:]

    def forward(self, hidden_states, encoder_hidden_states=None, pooled_projections=None, timestep=None,
                img_ids=None, txt_ids=None, guidance=None, controlnet_block_samples=None, controlnet_single_block_samples=None,
                controlnet_blocks_repeat=False, **_):
        hidden, encoder, temb, rope = self.embed(hidden_states, encoder_hidden_states, pooled_projections,
                                                 timestep, img_ids, txt_ids, guidance)
        if self.cache is not None and self.cache.config.enable_caching:
            hidden = self.cache.apply_cache(self, hidden, encoder, temb, rope)
        else:
            hidden = self.run_blocks(hidden, encoder, temb, rope, controlnet_block_samples, controlnet_single_block_samples,
                                     controlnet_blocks_repeat)
        hidden = self.norm_out(hidden, temb)
        return (self.proj_out(hidden),)
