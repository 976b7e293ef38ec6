"""Qwen-Image transformer oracle (torch-CPU restatement of fastdm/model/qwenimage.py and the Qwen pieces of
fastdm/layer/embeddings.py / transformer.py).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Citations relative to /root/reference.
"""
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import math

import numpy as np
import torch
import torch.nn.functional as F

from . import layers as L
from . import ops
from .flux import bf16_from_bits, synth_dist
from . import synth


@dataclass
class QwenConfig:
    """Defaults = Qwen-Image (fastdm/model/qwenimage.py:144-158)."""
    patch_size: int = 2
    in_channels: int = 64
    out_channels: int = 16
    num_layers: int = 60
    attention_head_dim: int = 128
    num_attention_heads: int = 24
    joint_attention_dim: int = 3584
    axes_dims_rope: Tuple[int, ...] = (16, 56, 56)
    mlp_ratio: float = 4.0

    @property
    def inner_dim(self):
        return self.attention_head_dim * self.num_attention_heads

    @property
    def proj_out_dim(self):
        return self.patch_size * self.patch_size * self.out_channels


TINY = QwenConfig(in_channels=16, out_channels=4, num_layers=3, attention_head_dim=64, num_attention_heads=2,
                  joint_attention_dim=96, axes_dims_rope=(16, 24, 24))


def param_specs(cfg: QwenConfig):
    """(diffusers key, shape, kind) for every tensor QwenImageTransformer2DModelCore loads (qwenimage.py:201-243)."""
    D, mlp = cfg.inner_dim, int(cfg.inner_dim * cfg.mlp_ratio)
    hd = cfg.attention_head_dim

    def lin(name, k, n, kind="w"):
        yield f"{name}.weight", (n, k), kind
        yield f"{name}.bias", (n,), "b"

    yield from lin("time_text_embed.timestep_embedder.linear_1", 256, D)
    yield from lin("time_text_embed.timestep_embedder.linear_2", D, D)
    yield "txt_norm.weight", (cfg.joint_attention_dim,), "rms"
    yield from lin("img_in", cfg.in_channels, D)
    yield from lin("txt_in", cfg.joint_attention_dim, D)
    for i in range(cfg.num_layers):
        p = f"transformer_blocks.{i}"
        yield from lin(f"{p}.img_mod.1", D, 6 * D, "w_mod")
        yield from lin(f"{p}.txt_mod.1", D, 6 * D, "w_mod")
        for n in ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_out.0", "to_add_out"):
            yield from lin(f"{p}.attn.{n}", D, D)
        for n in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
            yield f"{p}.attn.{n}.weight", (hd,), "rms"
        for ff in ("img_mlp", "txt_mlp"):
            yield from lin(f"{p}.{ff}.net.0.proj", D, mlp)
            yield from lin(f"{p}.{ff}.net.2", mlp, D)
    yield from lin("norm_out.linear", D, 2 * D, "w_mod")
    yield from lin("proj_out", D, cfg.proj_out_dim)


def synth_tensor(name: str, shape, kind: str, seed: int = 0) -> torch.Tensor:
    std, off = synth_dist(kind, shape)
    n = int(np.prod(shape))
    return bf16_from_bits(synth.normal_bf16_bits(synth.key_seed("qwen." + name, seed), n, std, off), shape)


def synth_state_dict(cfg: QwenConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    return {name: synth_tensor(name, shape, kind, seed) for name, shape, kind in param_specs(cfg)}


def synth_inputs(cfg: QwenConfig, batch: int, img_h: int, img_w: int, n_txt: int, seed: int = 0):
    """Mirrors examples/profiling/qwenimg_profiling.py:31-37: U[0,1) bf16 latents / prompt embeddings, an all-ones
    text mask, img_shapes = [(1, h, w)] per sample, txt_seq_lens = [n_txt]."""
    def uni(tag, shape):
        bits = synth.uniform_bf16_bits(synth.key_seed(f"qwen.input.{tag}", seed), int(np.prod(shape)))
        return bf16_from_bits(bits, shape)

    return dict(
        hidden_states=uni("latents", (batch, img_h * img_w, cfg.in_channels)),
        encoder_hidden_states=uni("prompt_embeds", (batch, n_txt, cfg.joint_attention_dim)),
        encoder_hidden_states_mask=torch.ones(batch, n_txt, dtype=torch.int64),
        img_shapes=[[(1, img_h, img_w)]] * batch,
        txt_seq_lens=[n_txt] * batch,
    )


# ------------------------------------------------------------------ embeddings
def qwen_timestep_embedding(t: torch.Tensor, dim: int = 256) -> torch.Tensor:
    """Timesteps(num_channels=256, flip_sin_to_cos=True, downscale_freq_shift=0, scale=1000)
    (fastdm/layer/embeddings.py:18-69,377-393,751): fp32, `(t * freq) * 1000`, output [cos | sin]."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    arg = 1000 * (t[:, None].float() * freqs[None, :])
    return torch.cat([torch.cos(arg), torch.sin(arg)], dim=-1)


def _rope_params(index: torch.Tensor, dim: int, theta: float = 10000.0) -> torch.Tensor:
    """embeddings.py:791-799: complex unit vectors exp(i * index * theta^(-2j/dim))."""
    freqs = torch.outer(index, 1.0 / torch.pow(theta, torch.arange(0, dim, 2).to(torch.float32).div(dim)))
    return torch.polar(torch.ones_like(freqs), freqs)


def qwen_rope_cache(img_shape, n_txt: int, axes_dim=(16, 56, 56), theta: float = 10000.0, dtype=torch.bfloat16) -> torch.Tensor:
    """QwenEmbedRope(scale_rope=True).forward (embeddings.py:762-857) followed by the merge of
    qwenimage.py:308-314: rows = [text ; image], columns = [real(hd/2) | imag(hd/2)], cast to `dtype`.
    img_shape = (frame, height, width) of ONE sample (the reference takes video_fhw[0])."""
    if isinstance(img_shape, list):
        img_shape = img_shape[0]
    frame, height, width = img_shape
    pos_index = torch.arange(4096)
    neg_index = torch.arange(4096).flip(0) * -1 - 1
    pos = [_rope_params(pos_index, d, theta) for d in axes_dim]
    neg = [_rope_params(neg_index, d, theta) for d in axes_dim]
    f_frame = pos[0][0:frame].view(frame, 1, 1, -1).expand(frame, height, width, -1)
    f_h = torch.cat([neg[1][-(height - height // 2):], pos[1][: height // 2]], dim=0).view(1, height, 1, -1).expand(frame, height, width, -1)
    f_w = torch.cat([neg[2][-(width - width // 2):], pos[2][: width // 2]], dim=0).view(1, 1, width, -1).expand(frame, height, width, -1)
    vid = torch.cat([f_frame, f_h, f_w], dim=-1).reshape(frame * height * width, -1)
    max_vid_index = max(height // 2, width // 2)
    txt = torch.cat(pos, dim=1)[max_vid_index: max_vid_index + n_txt]
    img_c = torch.cat([vid.real, vid.imag], dim=-1).to(dtype)
    txt_c = torch.cat([txt.real, txt.imag], dim=-1).to(dtype)
    return torch.cat([txt_c, img_c], dim=0).contiguous()


# ------------------------------------------------------------------ block / core
class QwenImageTransformerBlock:
    """fastdm/model/qwenimage.py:17-124.  The modulation projections are never quantised on a large-memory card
    (qwenimage.py:165-166,221-222)."""

    def __init__(self, sd, i, cfg: QwenConfig, quant):
        p = f"transformer_blocks.{i}"
        self.img_mod = L.linear_from(sd, [f"{p}.img_mod.1"])
        self.txt_mod = L.linear_from(sd, [f"{p}.txt_mod.1"])
        self.attn = L.FluxAttention(sd, f"{p}.attn", cfg.num_attention_heads, cfg.attention_head_dim, quant, joint=True)
        self.img_mlp = L.FeedForwardGeluTanh(sd, f"{p}.img_mlp", quant)
        self.txt_mlp = L.FeedForwardGeluTanh(sd, f"{p}.txt_mlp", quant)

    @staticmethod
    def _modulate(x, mod):
        shift, scale, gate = mod.chunk(3, dim=-1)
        return x * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1), gate.unsqueeze(1)

    def modulated_inputs(self, hidden, encoder, temb):
        img_mod1, img_mod2 = self.img_mod(F.silu(temb)).chunk(2, dim=-1)
        txt_mod1, txt_mod2 = self.txt_mod(F.silu(temb)).chunk(2, dim=-1)
        img_m, img_g1 = self._modulate(L.layer_norm_noaffine(hidden), img_mod1)
        txt_m, txt_g1 = self._modulate(L.layer_norm_noaffine(encoder), txt_mod1)
        return img_m, img_g1, img_mod2, txt_m, txt_g1, txt_mod2

    def __call__(self, hidden, encoder, temb, rope):
        img_m, img_g1, img_mod2, txt_m, txt_g1, txt_mod2 = self.modulated_inputs(hidden, encoder, temb)
        img_attn, txt_attn = self.attn(img_m, txt_m, rope)       # forward_qwen == the joint FLUX attention (transformer.py:319-391)
        hidden = hidden + img_g1 * img_attn
        encoder = encoder + txt_g1 * txt_attn
        img_m2, img_g2 = self._modulate(L.layer_norm_noaffine(hidden), img_mod2)
        hidden = hidden + img_g2 * self.img_mlp(img_m2)
        txt_m2, txt_g2 = self._modulate(L.layer_norm_noaffine(encoder), txt_mod2)
        encoder = encoder + txt_g2 * self.txt_mlp(txt_m2)
        return encoder, hidden


class QwenCore:
    """fastdm/model/qwenimage.py:126-343 (QwenImageTransformer2DModelCore), no OOM mode."""

    def __init__(self, sd, cfg: QwenConfig = QwenConfig(), quant=torch.int8, cache=None):
        self.cfg, self.quant, self.cache = cfg, quant, cache
        self.t1 = L.linear_from(sd, ["time_text_embed.timestep_embedder.linear_1"])
        self.t2 = L.linear_from(sd, ["time_text_embed.timestep_embedder.linear_2"])
        self.txt_norm_weight = sd["txt_norm.weight"]
        self.img_in = L.linear_from(sd, ["img_in"])
        self.txt_in = L.linear_from(sd, ["txt_in"])
        self.transformer_blocks = [QwenImageTransformerBlock(sd, i, cfg, quant) for i in range(cfg.num_layers)]
        self.norm_out = L.AdaLayerNormContinuous(sd, "norm_out.linear")
        self.proj_out = L.linear_from(sd, ["proj_out"])

    def embed(self, hidden_states, encoder_hidden_states, timestep, img_shapes, txt_seq_lens):
        """qwenimage.py:289-314."""
        hidden = self.img_in(hidden_states)
        timestep = timestep.to(hidden.dtype)
        encoder = ops.rms_norm(encoder_hidden_states, self.txt_norm_weight, 1e-6)
        encoder = self.txt_in(encoder)
        temb = self.t2(F.silu(self.t1(qwen_timestep_embedding(timestep).to(hidden.dtype))))
        rope = qwen_rope_cache(img_shapes[0] if isinstance(img_shapes[0], list) else img_shapes, max(txt_seq_lens),
                               self.cfg.axes_dims_rope, dtype=hidden.dtype)
        return hidden, encoder, temb, rope

    def run_blocks(self, hidden, encoder, temb, rope):
        for blk in self.transformer_blocks:
            encoder, hidden = blk(hidden, encoder, temb, rope)
        return hidden

    def forward(self, hidden_states, encoder_hidden_states=None, encoder_hidden_states_mask=None, timestep=None,
                img_shapes=None, txt_seq_lens=None, **_):
        hidden, encoder, temb, rope = self.embed(hidden_states, encoder_hidden_states, timestep, img_shapes, txt_seq_lens)
        if self.cache is not None and self.cache.config.enable_caching:
            hidden = self.cache.apply_cache_qwen(self, hidden, encoder, temb, rope)
        else:
            hidden = self.run_blocks(hidden, encoder, temb, rope)
        hidden = self.norm_out(hidden, temb)
        return (self.proj_out(hidden),)
