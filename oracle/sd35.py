"""Stable Diffusion 3 / 3.5 MMDiT oracle (torch-CPU restatement of fastdm/model/sd35.py, PatchEmbed and
CombinedTimestepTextProjEmbeddings of fastdm/layer/embeddings.py, SD35AdaLayerNormZeroX of fastdm/layer/normalization.py).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Citations relative to /root/reference.
"""
from dataclasses import dataclass
from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import layers as L
from . import ops
from . import synth
from .flux import bf16_from_bits, synth_dist


@dataclass
class SD3Config:
    """Defaults = SD3.5-medium-like (fastdm/model/sd35.py:203-221); SD3.5-large: 38 layers x 38 heads, no dual attention."""
    sample_size: int = 128
    patch_size: int = 2
    in_channels: int = 16
    num_layers: int = 24
    attention_head_dim: int = 64
    num_attention_heads: int = 24
    joint_attention_dim: int = 4096
    caption_projection_dim: int = 1536
    pooled_projection_dim: int = 2048
    out_channels: int = 16
    pos_embed_max_size: int = 384
    dual_attention_layers: Tuple[int, ...] = tuple(range(13))
    mlp_ratio: float = 4.0

    @property
    def inner_dim(self):
        return self.attention_head_dim * self.num_attention_heads


# caption_projection_dim must equal inner_dim (the context embedder feeds the joint attention directly)
TINY = SD3Config(sample_size=16, num_layers=3, attention_head_dim=64, num_attention_heads=2, joint_attention_dim=80,
                 caption_projection_dim=128, pooled_projection_dim=48, pos_embed_max_size=12, dual_attention_layers=(0,))


def param_specs(cfg: SD3Config):
    """(diffusers key, shape, kind) for every tensor SD3TransformerModelCore loads (sd35.py:286-333)."""
    D, mlp, hd, p = cfg.inner_dim, int(cfg.inner_dim * cfg.mlp_ratio), cfg.attention_head_dim, cfg.patch_size

    def lin(name, k, n, kind="w"):
        yield f"{name}.weight", (n, k), kind
        yield f"{name}.bias", (n,), "b"

    yield "pos_embed.proj.weight", (D, cfg.in_channels, p, p), "conv"
    yield "pos_embed.proj.bias", (D,), "b"
    yield "pos_embed.pos_embed", (1, cfg.pos_embed_max_size ** 2, D), "b"
    yield from lin("time_text_embed.timestep_embedder.linear_1", 256, D)
    yield from lin("time_text_embed.timestep_embedder.linear_2", D, D)
    yield from lin("time_text_embed.text_embedder.linear_1", cfg.pooled_projection_dim, D)
    yield from lin("time_text_embed.text_embedder.linear_2", D, D)
    yield from lin("context_embedder", cfg.joint_attention_dim, cfg.caption_projection_dim)
    for i in range(cfg.num_layers):
        b = f"transformer_blocks.{i}"
        last = i == cfg.num_layers - 1
        dual = i in cfg.dual_attention_layers
        yield from lin(f"{b}.norm1.linear", D, (9 if dual else 6) * D, "w_mod")
        yield from lin(f"{b}.norm1_context.linear", D, (2 if last else 6) * D, "w_mod")
        for n in ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_out.0"):
            yield from lin(f"{b}.attn.{n}", D, D)
        if not last:
            yield from lin(f"{b}.attn.to_add_out", D, D)
        for n in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
            yield f"{b}.attn.{n}.weight", (hd,), "rms"
        if dual:
            for n in ("to_q", "to_k", "to_v", "to_out.0"):
                yield from lin(f"{b}.attn2.{n}", D, D)
            yield f"{b}.attn2.norm_q.weight", (hd,), "rms"
            yield f"{b}.attn2.norm_k.weight", (hd,), "rms"
        yield from lin(f"{b}.ff.net.0.proj", D, mlp)
        yield from lin(f"{b}.ff.net.2", mlp, D)
        if not last:
            yield from lin(f"{b}.ff_context.net.0.proj", D, mlp)
            yield from lin(f"{b}.ff_context.net.2", mlp, D)
    yield from lin("norm_out.linear", D, 2 * D, "w_mod")
    yield from lin("proj_out", D, p * p * cfg.out_channels)


def synth_tensor(name: str, shape, kind: str, seed: int = 0) -> torch.Tensor:
    if kind == "conv":
        std, off = (1.0 / int(np.prod(shape[1:]))) ** 0.5, 0.0
    else:
        std, off = synth_dist(kind, shape)
    return bf16_from_bits(synth.normal_bf16_bits(synth.key_seed("sd3." + name, seed), int(np.prod(shape)), std, off), shape)


def synth_state_dict(cfg: SD3Config, seed: int = 0) -> Dict[str, torch.Tensor]:
    return {name: synth_tensor(name, shape, kind, seed) for name, shape, kind in param_specs(cfg)}


def synth_inputs(cfg: SD3Config, batch: int, height: int, width: int, n_txt: int, seed: int = 0):
    """latents [B, 16, H, W] ~ N(0,1) (H, W = pixels / 8), prompt embeddings U[0,1), pooled U[0,1)."""
    def make(tag, shape, normal=False):
        n = int(np.prod(shape))
        key = synth.key_seed(f"sd3.input.{tag}", seed)
        bits = synth.normal_bf16_bits(key, n, 1.0, 0.0) if normal else synth.uniform_bf16_bits(key, n)
        return bf16_from_bits(bits, shape)

    return dict(hidden_states=make("latents", (batch, cfg.in_channels, height, width), True),
                encoder_hidden_states=make("prompt_embeds", (batch, n_txt, cfg.joint_attention_dim)),
                pooled_projections=make("pooled", (batch, cfg.pooled_projection_dim)))


# ------------------------------------------------------------------ layers
class SD3Attention:
    """fastdm/layer/transformer.py:232-317: joint attention without RoPE (text rows first), or image-only (attn2)."""

    def __init__(self, sd, prefix, heads, head_dim, quant, joint: bool, context_pre_only: bool = False):
        self.heads, self.head_dim, self.joint, self.context_pre_only = heads, head_dim, joint, context_pre_only
        self.inner = heads * head_dim
        self.qkv = L.linear_from(sd, [f"{prefix}.to_q", f"{prefix}.to_k", f"{prefix}.to_v"], quant)
        self.norm_q, self.norm_k = sd[f"{prefix}.norm_q.weight"], sd[f"{prefix}.norm_k.weight"]
        self.to_out = L.linear_from(sd, [f"{prefix}.to_out.0"], quant)
        if joint:
            self.add_qkv = L.linear_from(sd, [f"{prefix}.add_q_proj", f"{prefix}.add_k_proj", f"{prefix}.add_v_proj"], quant)
            self.norm_added_q, self.norm_added_k = sd[f"{prefix}.norm_added_q.weight"], sd[f"{prefix}.norm_added_k.weight"]
            if not context_pre_only:
                self.to_add_out = L.linear_from(sd, [f"{prefix}.to_add_out"], quant)

    def _split(self, fused, wq, wk):
        B = fused.shape[0]
        q, k, v = fused[..., : self.inner], fused[..., self.inner: 2 * self.inner], fused[..., 2 * self.inner:]
        q = ops.rms_norm(q.unflatten(-1, (self.heads, -1)).contiguous(), wq, 1e-6).view(B, -1, self.inner)
        k = ops.rms_norm(k.unflatten(-1, (self.heads, -1)).contiguous(), wk, 1e-6).view(B, -1, self.inner)
        return q, k, v

    def __call__(self, hidden, encoder=None):
        q, k, v = self._split(self.qkv(hidden), self.norm_q, self.norm_k)
        if encoder is not None:
            eq, ek, ev = self._split(self.add_qkv(encoder), self.norm_added_q, self.norm_added_k)
            q, k, v = torch.cat([eq, q], dim=1), torch.cat([ek, k], dim=1), torch.cat([ev, v], dim=1)
        o = ops.scaled_dot_product_attention(q.contiguous(), k.contiguous(), v.contiguous(), self.heads, self.heads, self.head_dim,
                                             scale=self.head_dim ** -0.5)
        if encoder is None:
            return self.to_out(o)
        n_txt = encoder.shape[1]
        enc_o = o[:, :n_txt]
        if not self.context_pre_only:
            enc_o = self.to_add_out(enc_o)
        return self.to_out(o[:, n_txt:]), enc_o


class JointTransformerBlock:
    """fastdm/model/sd35.py:32-200."""

    def __init__(self, sd, i, cfg: SD3Config, quant):
        b = f"transformer_blocks.{i}"
        self.last = i == cfg.num_layers - 1
        self.dual = i in cfg.dual_attention_layers
        self.norm1 = L.linear_from(sd, [f"{b}.norm1.linear"])
        self.norm1_context = L.linear_from(sd, [f"{b}.norm1_context.linear"])
        self.attn = SD3Attention(sd, f"{b}.attn", cfg.num_attention_heads, cfg.attention_head_dim, quant, joint=True, context_pre_only=self.last)
        self.attn2 = SD3Attention(sd, f"{b}.attn2", cfg.num_attention_heads, cfg.attention_head_dim, quant, joint=False) if self.dual else None
        self.ff = L.FeedForwardGeluTanh(sd, f"{b}.ff", quant)
        self.ff_context = None if self.last else L.FeedForwardGeluTanh(sd, f"{b}.ff_context", quant)

    def norm1_modulated(self, hidden, temb):
        """norm1's first output, what TeaCache probes (caching/xcaching.py:181-183)."""
        emb = self.norm1(F.silu(temb))
        chunks = emb.chunk(9 if self.dual else 6, dim=1)
        return L.layer_norm_noaffine(hidden, 1e-5 if self.dual else 1e-6) * (1 + chunks[1][:, None]) + chunks[0][:, None]

    def __call__(self, hidden, encoder, temb, rope=None):
        emb = self.norm1(F.silu(temb))
        if self.dual:   # SD35AdaLayerNormZeroX (normalization.py:45-88), eps 1e-5
            shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp, shift_msa2, scale_msa2, gate_msa2 = emb.chunk(9, dim=1)
            n = L.layer_norm_noaffine(hidden, 1e-5)
            norm_hidden = n * (1 + scale_msa[:, None]) + shift_msa[:, None]
            norm_hidden2 = n * (1 + scale_msa2[:, None]) + shift_msa2[:, None]
        else:           # AdaLayerNormZero, eps 1e-6
            shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = emb.chunk(6, dim=1)
            norm_hidden = L.layer_norm_noaffine(hidden, 1e-6) * (1 + scale_msa[:, None]) + shift_msa[:, None]
        cemb = self.norm1_context(F.silu(temb).to(encoder.dtype) if self.last else F.silu(temb))
        if self.last:   # AdaLayerNormContinuous: (scale, shift)
            c_scale, c_shift = cemb.chunk(2, dim=1)
            norm_enc = L.layer_norm_noaffine(encoder, 1e-6) * (1 + c_scale)[:, None, :] + c_shift[:, None, :]
        else:
            c_shift_msa, c_scale_msa, c_gate_msa, c_shift_mlp, c_scale_mlp, c_gate_mlp = cemb.chunk(6, dim=1)
            norm_enc = L.layer_norm_noaffine(encoder, 1e-6) * (1 + c_scale_msa[:, None]) + c_shift_msa[:, None]
        attn_out, ctx_out = self.attn(norm_hidden, norm_enc)
        hidden = hidden + gate_msa.unsqueeze(1) * attn_out
        if self.dual:
            hidden = hidden + gate_msa2.unsqueeze(1) * self.attn2(norm_hidden2)
        n = L.layer_norm_noaffine(hidden, 1e-6) * (1 + scale_mlp[:, None]) + shift_mlp[:, None]
        hidden = hidden + gate_mlp.unsqueeze(1) * self.ff(n)
        if self.last:
            return None, hidden
        encoder = encoder + c_gate_msa.unsqueeze(1) * ctx_out
        n = L.layer_norm_noaffine(encoder, 1e-6) * (1 + c_scale_mlp[:, None]) + c_shift_mlp[:, None]
        encoder = encoder + c_gate_mlp.unsqueeze(1) * self.ff_context(n)
        return encoder, hidden


class SD3Core:
    """fastdm/model/sd35.py:202-422.  norm_out.linear and proj_out are quantised like the block linears (sd35.py:326-329)."""

    def __init__(self, sd, cfg: SD3Config = SD3Config(), quant=torch.float8_e4m3fn):
        self.cfg, self.quant = cfg, quant
        self.proj_w, self.proj_b, self.pos = sd["pos_embed.proj.weight"], sd["pos_embed.proj.bias"], sd["pos_embed.pos_embed"]
        self.t1 = L.linear_from(sd, ["time_text_embed.timestep_embedder.linear_1"])
        self.t2 = L.linear_from(sd, ["time_text_embed.timestep_embedder.linear_2"])
        self.p1 = L.linear_from(sd, ["time_text_embed.text_embedder.linear_1"])
        self.p2 = L.linear_from(sd, ["time_text_embed.text_embedder.linear_2"])
        self.context_embedder = L.linear_from(sd, ["context_embedder"])
        self.blocks = [JointTransformerBlock(sd, i, cfg, quant) for i in range(cfg.num_layers)]
        self.transformer_blocks = self.blocks        # the name the cache oracles use
        self.cache_fn = None                         # step cache hook: f(core, hidden, encoder, temb) -> hidden (sd35.py:384-392)
        self.norm_out = L.linear_from(sd, ["norm_out.linear"], quant)
        self.proj_out = L.linear_from(sd, ["proj_out"], quant)

    def patch_embed(self, latent):
        """PatchEmbed.forward with pos_embed_max_size cropping (embeddings.py:313-360): pos_embed is fp32-promoted in the add."""
        cfg = self.cfg
        h, w = latent.shape[-2] // cfg.patch_size, latent.shape[-1] // cfg.patch_size
        x = F.conv2d(latent, self.proj_w, self.proj_b, stride=cfg.patch_size).flatten(2).transpose(1, 2)
        m = cfg.pos_embed_max_size
        top, left = (m - h) // 2, (m - w) // 2
        pe = self.pos.reshape(1, m, m, -1)[:, top: top + h, left: left + w, :].reshape(1, h * w, -1)
        return (x + pe).to(x.dtype)

    def forward(self, hidden_states, encoder_hidden_states=None, pooled_projections=None, timestep=None,
                block_controlnet_hidden_states=None, **_):
        cfg = self.cfg
        height, width = hidden_states.shape[-2:]
        x = self.patch_embed(hidden_states)
        te = self.t2(F.silu(self.t1(L.timestep_embedding(timestep).to(pooled_projections.dtype))))
        temb = te + self.p2(F.silu(self.p1(pooled_projections)))
        enc = self.context_embedder(encoder_hidden_states)
        if self.cache_fn is not None:
            x = self.cache_fn(self, x, enc, temb)
        else:
            cn = block_controlnet_hidden_states
            for i, blk in enumerate(self.blocks):
                enc, x = blk(x, enc, temb)
                if cn is not None and not blk.last:                    # sd35.py:401-404
                    x = x + cn[i // (len(self.blocks) // len(cn))]
        emb = self.norm_out(F.silu(temb).to(x.dtype))
        scale, shift = emb.chunk(2, dim=1)
        x = L.layer_norm_noaffine(x, 1e-6) * (1 + scale)[:, None, :] + shift[:, None, :]
        x = self.proj_out(x)
        p = cfg.patch_size
        h, w = height // p, width // p
        x = x.reshape(x.shape[0], h, w, p, p, cfg.out_channels)
        x = torch.einsum("nhwpqc->nchpwq", x)
        return (x.reshape(x.shape[0], cfg.out_channels, h * p, w * p),)
