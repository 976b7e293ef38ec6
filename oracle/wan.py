"""Wan2.x video transformer oracle (torch-CPU restatement of fastdm/model/wan.py, WanAttention in
fastdm/layer/transformer.py:393-534, the Wan pieces of fastdm/layer/embeddings.py and the radial block-sparse mask
generator of fastdm/sparse/xsparse.py).  T2V path (no image conditioning).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Citations relative to /root/reference.

The reference's sparse attention only exists on its CUDA backend (SpargeAttn / SageAttention kernels, int8/fp8 inside;
the torch and triton backends raise): the block-sparse attention ARITHMETIC is therefore **parity unpinned**; the
definition used here is dense bf16 attention with the masked blocks at -inf (oracle/ops.py).  The MASK itself is
generated on the CPU by the reference and pinned bit-exactly (tests/golden/wan_radial_mask.npz).
"""
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import math

import numpy as np
import torch
import torch.nn.functional as F

from . import layers as L
from . import ops
from . import synth
from .flux import bf16_from_bits, synth_dist


@dataclass
class WanConfig:
    """Defaults = Wan2.2-T2V-A14B (fastdm/model/wan.py:154-176)."""
    patch_size: Tuple[int, int, int] = (1, 2, 2)
    num_attention_heads: int = 40
    attention_head_dim: int = 128
    in_channels: int = 16
    out_channels: int = 16
    text_dim: int = 4096
    freq_dim: int = 256
    ffn_dim: int = 13824
    num_layers: int = 40
    eps: float = 1e-6
    rope_max_seq_len: int = 1024
    image_dim: Optional[int] = None           # Wan2.1-I2V: CLIP image embeddings (1280) -> image_embedder (wan.py:167,197)
    added_kv_proj_dim: Optional[int] = None   # ... and their own K/V projections in every cross-attention (wan.py:168,205)

    @property
    def inner_dim(self):
        return self.attention_head_dim * self.num_attention_heads


TINY = WanConfig(num_attention_heads=2, attention_head_dim=64, in_channels=16, out_channels=16, text_dim=96, ffn_dim=256, num_layers=2)
TINY_I2V = WanConfig(num_attention_heads=2, attention_head_dim=64, in_channels=36, out_channels=16, text_dim=96, ffn_dim=256, num_layers=2,
                     image_dim=48, added_kv_proj_dim=128)
TEXT_LEN_I2V = 512     # "512 is the context length of the text encoder, hardcoded for now" (transformer.py:473-474)


def param_specs(cfg: WanConfig):
    """(diffusers key, shape, kind) for every tensor WanTransformer3DModelCore loads (wan.py:225-281), T2V."""
    D, F_ = cfg.inner_dim, cfg.ffn_dim
    pt, ph, pw = cfg.patch_size

    def lin(name, k, n, kind="w"):
        yield f"{name}.weight", (n, k), kind
        yield f"{name}.bias", (n,), "b"

    yield "patch_embedding.weight", (D, cfg.in_channels, pt, ph, pw), "conv"
    yield "patch_embedding.bias", (D,), "b"
    yield from lin("condition_embedder.time_embedder.linear_1", cfg.freq_dim, D)
    yield from lin("condition_embedder.time_embedder.linear_2", D, D)
    yield from lin("condition_embedder.time_proj", D, 6 * D, "w_mod")
    yield from lin("condition_embedder.text_embedder.linear_1", cfg.text_dim, D)
    yield from lin("condition_embedder.text_embedder.linear_2", D, D)
    if cfg.image_dim is not None:                                   # wan.py:234-243
        ie = "condition_embedder.image_embedder"
        yield f"{ie}.norm1.weight", (cfg.image_dim,), "rms"
        yield f"{ie}.norm1.bias", (cfg.image_dim,), "b"
        yield from lin(f"{ie}.ff.net.0.proj", cfg.image_dim, cfg.image_dim)
        yield from lin(f"{ie}.ff.net.2", cfg.image_dim, D)
        yield f"{ie}.norm2.weight", (D,), "rms"
        yield f"{ie}.norm2.bias", (D,), "b"
    for i in range(cfg.num_layers):
        p = f"blocks.{i}"
        for a in ("attn1", "attn2"):
            for n in ("to_q", "to_k", "to_v", "to_out.0"):
                yield from lin(f"{p}.{a}.{n}", D, D)
            yield f"{p}.{a}.norm_q.weight", (D,), "rms"
            yield f"{p}.{a}.norm_k.weight", (D,), "rms"
        if cfg.added_kv_proj_dim is not None:                       # wan.py:262-265
            yield from lin(f"{p}.attn2.add_k_proj", cfg.added_kv_proj_dim, D)
            yield from lin(f"{p}.attn2.add_v_proj", cfg.added_kv_proj_dim, D)
            yield f"{p}.attn2.norm_added_k.weight", (D,), "rms"
        yield f"{p}.norm2.weight", (D,), "rms"
        yield f"{p}.norm2.bias", (D,), "b"
        yield from lin(f"{p}.ffn.net.0.proj", D, F_)
        yield from lin(f"{p}.ffn.net.2", F_, D)
        yield f"{p}.scale_shift_table", (1, 6, D), "table"
    yield from lin("proj_out", D, cfg.out_channels * pt * ph * pw)
    yield "scale_shift_table", (1, 2, D), "table"


def synth_tensor(name: str, shape, kind: str, seed: int = 0) -> torch.Tensor:
    if kind == "conv":
        std, off = (1.0 / int(np.prod(shape[1:]))) ** 0.5, 0.0
    elif kind == "table":
        std, off = (1.0 / shape[-1]) ** 0.5, 0.0          # torch.randn(...) / dim**0.5 (wan.py:65,215)
    else:
        std, off = synth_dist(kind, shape)
    n = int(np.prod(shape))
    return bf16_from_bits(synth.normal_bf16_bits(synth.key_seed("wan." + name, seed), n, std, off), shape)


def synth_state_dict(cfg: WanConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    return {name: synth_tensor(name, shape, kind, seed) for name, shape, kind in param_specs(cfg)}


def synth_inputs(cfg: WanConfig, batch: int, frames: int, height: int, width: int, n_txt: int, seed: int = 0):
    """examples/profiling/wan_profiling.py:13-32: latents [B, 16, F, H, W] ~ N(0,1), text embeddings [B, n_txt, text_dim]."""
    def nrm(tag, shape):
        bits = synth.normal_bf16_bits(synth.key_seed(f"wan.input.{tag}", seed), int(np.prod(shape)), 1.0, 0.0)
        return bf16_from_bits(bits, shape)

    d = dict(hidden_states=nrm("latents", (batch, cfg.in_channels, frames, height, width)),
             encoder_hidden_states=nrm("prompt_embeds", (batch, n_txt, cfg.text_dim)))
    if cfg.image_dim is not None:
        d["encoder_hidden_states_image"] = nrm("image_embeds", (batch, 5, cfg.image_dim))      # 5 image-context tokens (257 for CLIP ViT-H)
    return d


# ------------------------------------------------------------------ embeddings
def wan_rope_cache(cfg: WanConfig, frames: int, height: int, width: int, dtype=torch.bfloat16) -> torch.Tensor:
    """WanRotaryPosEmbed (embeddings.py:859-922) followed by the merge in WanAttention.forward (transformer.py:489-491):
    [tokens, head_dim] = [cos[:, 0::2] | sin[:, 1::2]] cast to `dtype`.  (frames, height, width) = LATENT sizes."""
    hd = cfg.attention_head_dim
    pt, ph, pw = cfg.patch_size
    ppf, pph, ppw = frames // pt, height // ph, width // pw
    h_dim = w_dim = 2 * (hd // 6)
    t_dim = hd - h_dim - w_dim
    cos_l, sin_l = [], []
    for dim in (t_dim, h_dim, w_dim):
        pos = torch.arange(cfg.rope_max_seq_len)
        freqs = 1.0 / (10000.0 ** (torch.arange(0, dim, 2, dtype=torch.float64)[: dim // 2] / dim))
        fr = torch.outer(pos, freqs)
        cos_l.append(fr.cos().repeat_interleave(2, dim=1).float())
        sin_l.append(fr.sin().repeat_interleave(2, dim=1).float())
    fc, fs = torch.cat(cos_l, dim=1), torch.cat(sin_l, dim=1)
    split = [hd - 2 * (hd // 3), hd // 3, hd // 3]
    fc, fs = fc.split(split, dim=1), fs.split(split, dim=1)

    def grid(parts):
        a = parts[0][:ppf].view(ppf, 1, 1, -1).expand(ppf, pph, ppw, -1)
        b = parts[1][:pph].view(1, pph, 1, -1).expand(ppf, pph, ppw, -1)
        c = parts[2][:ppw].view(1, 1, ppw, -1).expand(ppf, pph, ppw, -1)
        return torch.cat([a, b, c], dim=-1).reshape(ppf * pph * ppw, -1)

    cos, sin = grid(fc), grid(fs)
    return torch.cat((cos[:, 0::2], sin[:, 1::2]), dim=-1).to(dtype).contiguous()


class WanConditionEmbedder:
    """WanTimeTextImageEmbedding (embeddings.py:949-992), T2V: sinusoid (scale 1) -> fp32 TimestepEmbedding ->
    type_as(text) ; time_proj(silu(temb)) ; PixArtAlphaTextProjection(gelu_tanh) on the text."""

    def __init__(self, sd, cfg):
        p = "condition_embedder"
        self.w1, self.b1 = sd[f"{p}.time_embedder.linear_1.weight"].float(), sd[f"{p}.time_embedder.linear_1.bias"].float()
        self.w2, self.b2 = sd[f"{p}.time_embedder.linear_2.weight"].float(), sd[f"{p}.time_embedder.linear_2.bias"].float()
        self.time_proj = L.linear_from(sd, [f"{p}.time_proj"])
        self.t1 = L.linear_from(sd, [f"{p}.text_embedder.linear_1"])
        self.t2 = L.linear_from(sd, [f"{p}.text_embedder.linear_2"])
        self.freq_dim = cfg.freq_dim
        self.img = None
        if cfg.image_dim is not None:           # WanImageEmbedding (embeddings.py:925-947), no pos_embed
            ie = f"{p}.image_embedder"
            self.img = dict(n1=(sd[f"{ie}.norm1.weight"].float(), sd[f"{ie}.norm1.bias"].float()), f1=L.linear_from(sd, [f"{ie}.ff.net.0.proj"]),
                            f2=L.linear_from(sd, [f"{ie}.ff.net.2"]), n2=(sd[f"{ie}.norm2.weight"].float(), sd[f"{ie}.norm2.bias"].float()))

    def image_embed(self, x):
        m = self.img
        h = fp32_layer_norm(x, *m["n1"], eps=1e-5).to(x.dtype)               # FP32LayerNorm returns the input dtype (normalization.py:158-160)
        h = m["f2"](F.gelu(m["f1"](h)))                                       # FeedForward(mult=1, activation_fn="gelu"): erf GELU
        return fp32_layer_norm(h, *m["n2"], eps=1e-5).to(h.dtype)

    def __call__(self, timestep, encoder_hidden_states, timestep_seq_len=None):
        ts = L.timestep_embedding(timestep, self.freq_dim)                       # fp32 [cos | sin]
        if timestep_seq_len is not None:                                         # wan2.2 ti2v: one timestep per token (embeddings.py:979-980)
            ts = ts.unflatten(0, (-1, timestep_seq_len))
        temb = (F.silu(ts @ self.w1.t() + self.b1) @ self.w2.t() + self.b2)     # fp32 QLinear: torch.addmm (qlinear.py:76)
        temb = temb.type_as(encoder_hidden_states)
        timestep_proj = self.time_proj(F.silu(temb))
        enc = self.t2(F.gelu(self.t1(encoder_hidden_states), approximate="tanh"))
        return temb, timestep_proj, enc


def fp32_layer_norm(x, weight=None, bias=None, eps=1e-6):
    """FP32LayerNorm.forward (normalization.py:130-160): F.layer_norm on x.float(); the result STAYS fp32."""
    return F.layer_norm(x.float(), (x.shape[-1],), weight, bias, eps)


# ------------------------------------------------------------------ radial block-sparse mask (xsparse.py:44-183)
class RadialMask:
    def __init__(self, block_size=64, decay_factor=0.3, model_type="wan"):
        self.block_size, self.decay_factor, self.model_type = block_size, decay_factor, model_type

    def window_width(self, i, j, tpf):
        dist = abs(i - j)
        if self.model_type == "wan":
            if dist < 1:
                return tpf
            if dist == 1:
                return tpf // 2
        elif self.model_type == "hunyuan":
            if dist <= 1:
                return tpf
        group = dist.bit_length()
        decay_length = 2 ** tpf.bit_length() / 2 ** group * self.decay_factor
        return decay_length if decay_length >= self.block_size else self.block_size

    def diagonal_split(self, i, j, tpf):
        dist = abs(i - j)
        group = dist.bit_length()
        decay_length = 2 ** tpf.bit_length() / 2 ** group
        if decay_length >= self.block_size:
            return True
        split_factor = int(self.block_size / decay_length)
        return dist % split_factor == 0

    def shrink(self, mask):
        bs = self.block_size
        bn = mask.shape[0] // bs
        m = mask[: bn * bs, : bn * bs].view(bn, bs, bn, bs)
        col_dens = m.sum(dim=1) / bs
        nonzero = col_dens > 0
        high = col_dens > 1 / 3
        frac = high.sum(dim=-1) / (nonzero.sum(dim=-1) + 1e-9)
        return frac > 0.6

    def log_mask(self, s: int, video_token_num: int, num_frame: int) -> torch.Tensor:
        """gen_log_mask_shrinked: bool [s/bs, s/bs]."""
        bs = self.block_size
        final = torch.zeros((s // bs, s // bs), dtype=torch.bool)
        tpf = video_token_num // num_frame
        border = video_token_num // bs
        col = torch.arange(0, tpf).view(1, -1)
        row = torch.arange(0, tpf).view(-1, 1)
        final[border:] = True
        final[:, border:] = True
        for i in range(num_frame):
            for j in range(num_frame):
                if j == 0 and self.model_type == "wan":
                    local = torch.ones((tpf, tpf), dtype=torch.bool)
                else:
                    local = torch.abs(col - row) <= self.window_width(i, j, tpf)
                    if not self.diagonal_split(i, j, tpf):
                        local = torch.zeros((tpf, tpf), dtype=torch.bool)
                rr, rc = (i * tpf) % bs, (j * tpf) % bs
                lr = rr + ((tpf - 1) // bs + 1) * bs
                lc = rc + ((tpf - 1) // bs + 1) * bs
                padded = torch.zeros((lr, lc), dtype=torch.bool)
                padded[rr:rr + tpf, rc:rc + tpf] = local
                bm = self.shrink(padded)
                r0, c0 = (i * tpf) // bs, (j * tpf) // bs
                final[r0:r0 + bm.shape[0], c0:c0 + bm.shape[1]] |= bm
        return final

    def kernel_mask(self, log_mask: torch.Tensor) -> torch.Tensor:
        """sparge_mask_convert for a non-sm90 part (xsparse.py:227-250): (BQ, BK) = (128, 64)."""
        if self.block_size == 128:
            return torch.repeat_interleave(log_mask, 2, dim=1)
        nr, nc = log_mask.shape
        return log_mask.view(nr // 2, 2, nc).max(dim=1).values


# ------------------------------------------------------------------ attention / block / core
class WanAttention:
    """transformer.py:393-534; with `added_kv_proj_dim` the cross-attention also attends the image-context tokens (the
    first S_enc - 512 rows of the encoder states) through their own K/V projections and adds that result."""

    def __init__(self, sd, prefix, cfg: WanConfig, quant, cross: bool):
        self.heads, self.head_dim, self.cross, self.eps = cfg.num_attention_heads, cfg.attention_head_dim, cross, cfg.eps
        self.inner = cfg.inner_dim
        if cross:
            self.to_q = L.linear_from(sd, [f"{prefix}.to_q"], quant)
            self.to_kv = L.linear_from(sd, [f"{prefix}.to_k", f"{prefix}.to_v"], quant)
        else:
            self.qkv = L.linear_from(sd, [f"{prefix}.to_q", f"{prefix}.to_k", f"{prefix}.to_v"], quant)
        self.to_out = L.linear_from(sd, [f"{prefix}.to_out.0"], quant)
        self.norm_q, self.norm_k = sd[f"{prefix}.norm_q.weight"], sd[f"{prefix}.norm_k.weight"]
        self.scale = self.head_dim ** -0.5
        self.add_k = self.add_v = None
        if cross and cfg.added_kv_proj_dim is not None:
            self.add_k = L.linear_from(sd, [f"{prefix}.add_k_proj"], quant)
            self.add_v = L.linear_from(sd, [f"{prefix}.add_v_proj"], quant)
            self.norm_added_k = sd[f"{prefix}.norm_added_k.weight"]

    def __call__(self, hidden, encoder=None, rope=None, block_mask=None):
        enc_img = None
        if self.add_k is not None:
            n_img = encoder.shape[1] - TEXT_LEN_I2V
            enc_img, encoder = encoder[:, :n_img], encoder[:, n_img:]
        if self.cross:
            q = self.to_q(hidden)
            kv = self.to_kv(encoder)
            k, v = kv[..., : self.inner], kv[..., self.inner:]
        else:
            qkv = self.qkv(hidden)
            q, k, v = qkv[..., : self.inner], qkv[..., self.inner: 2 * self.inner], qkv[..., 2 * self.inner:]
        q = ops.rms_norm(q.contiguous(), self.norm_q, self.eps)          # across heads: over the whole inner dim
        k = ops.rms_norm(k.contiguous(), self.norm_k, self.eps)
        if rope is not None:
            ops.rotary_pos_embedding(q, k, self.head_dim, rope, is_neox=False)
        o = ops.scaled_dot_product_attention(q, k, v.contiguous(), self.heads, self.heads, self.head_dim, scale=self.scale,
                                             block_mask=block_mask)
        if enc_img is not None:
            k_img = ops.rms_norm(self.add_k(enc_img).contiguous(), self.norm_added_k, self.eps)
            o = o + ops.scaled_dot_product_attention(q, k_img, self.add_v(enc_img).contiguous(), self.heads, self.heads, self.head_dim,
                                                     scale=self.scale)
        return self.to_out(o)


class WanTransformerBlock:
    """wan.py:20-115 (temb [B, 6, D]: wan2.1 / wan2.2-14B)."""

    def __init__(self, sd, i, cfg: WanConfig, quant):
        p = f"blocks.{i}"
        self.attn1 = WanAttention(sd, f"{p}.attn1", cfg, quant, cross=False)
        self.attn2 = WanAttention(sd, f"{p}.attn2", cfg, quant, cross=True)
        self.norm2_w, self.norm2_b = sd[f"{p}.norm2.weight"].float(), sd[f"{p}.norm2.bias"].float()
        self.ffn = L.FeedForwardGeluTanh(sd, f"{p}.ffn", quant)
        self.table = sd[f"{p}.scale_shift_table"]
        self.eps = cfg.eps

    def __call__(self, hidden, encoder, temb6, rope, block_mask=None):
        if temb6.ndim == 4:      # [B, S, 6, D]: per-token modulation (wan2.2 ti2v, wan.py:77-88)
            shift_msa, scale_msa, gate_msa, c_shift, c_scale, c_gate = [t.squeeze(2) for t in
                                                                        (self.table.unsqueeze(0) + temb6.float()).chunk(6, dim=2)]
        else:
            shift_msa, scale_msa, gate_msa, c_shift, c_scale, c_gate = (self.table + temb6.float()).chunk(6, dim=1)
        n = (fp32_layer_norm(hidden, eps=self.eps) * (1 + scale_msa) + shift_msa).type_as(hidden)
        a = self.attn1(n, None, rope, block_mask)
        hidden = (hidden.float() + a * gate_msa).type_as(hidden)
        n = fp32_layer_norm(hidden, self.norm2_w, self.norm2_b, self.eps).type_as(hidden)
        hidden = hidden + self.attn2(n, encoder, None)
        n = (fp32_layer_norm(hidden, eps=self.eps) * (1 + c_scale) + c_shift).type_as(hidden)
        f = self.ffn(n)
        return (hidden.float() + f.float() * c_gate).type_as(hidden)


class WanCore:
    """wan.py:117-379, T2V, no cache.  `sparse` = (RadialMask, dense_layers, dense_steps, step_callback) or None."""

    def __init__(self, sd, cfg: WanConfig = WanConfig(), quant=torch.float8_e4m3fn, sparse=None):
        self.cfg, self.quant, self.sparse = cfg, quant, sparse
        self.patch_w, self.patch_b = sd["patch_embedding.weight"], sd["patch_embedding.bias"]
        self.cond = WanConditionEmbedder(sd, cfg)
        self.blocks = [WanTransformerBlock(sd, i, cfg, quant) for i in range(cfg.num_layers)]
        self.proj_out = L.linear_from(sd, ["proj_out"])
        self.table = sd["scale_shift_table"]
        self._mask = None

    def block_mask(self, layer_index, tokens, frames_pp, batch):
        if self.sparse is None:
            return None
        rm, dense_layers, dense_steps, step_cb = self.sparse
        step = step_cb() if step_cb is not None else 0
        if step < dense_steps or layer_index < dense_layers:
            return None
        if self._mask is None:
            km = rm.kernel_mask(rm.log_mask(tokens, tokens, frames_pp))
            self._mask = km.to(torch.int8)[None, None].expand(batch, self.cfg.num_attention_heads, -1, -1).contiguous()
        return self._mask

    def forward(self, hidden_states, timestep, encoder_hidden_states, encoder_hidden_states_image=None, **_):
        cfg = self.cfg
        B, C, Fr, H, W = hidden_states.shape
        pt, ph, pw = cfg.patch_size
        ppf, pph, ppw = Fr // pt, H // ph, W // pw
        rope = wan_rope_cache(cfg, Fr, H, W, hidden_states.dtype)
        x = F.conv3d(hidden_states, self.patch_w, self.patch_b, cfg.patch_size)
        x = x.flatten(2).transpose(1, 2)
        ts_len = None
        if timestep.ndim == 2:                                       # wan2.2 ti2v (wan.py:319-322)
            ts_len, timestep = timestep.shape[1], timestep.flatten()
        temb, tproj, enc = self.cond(timestep, encoder_hidden_states, timestep_seq_len=ts_len)
        tproj = tproj.unflatten(2, (6, -1)) if ts_len is not None else tproj.unflatten(1, (6, -1))
        if encoder_hidden_states_image is not None:                  # wan.py:333-334
            enc = torch.concat([self.cond.image_embed(encoder_hidden_states_image), enc], dim=1)
        if getattr(self, "cache_fn", None) is not None:      # step cache hook (wan.py:339-349): f(core, hidden, encoder, temb6, rope)
            x = self.cache_fn(self, x, enc, tproj, rope)
        else:
            for li, blk in enumerate(self.blocks):
                x = blk(x, enc, tproj, rope, self.block_mask(li, x.shape[1], ppf, B))
        if temb.ndim == 3:                                           # per token (wan.py:355-359)
            shift, scale = [t.squeeze(2) for t in (self.table.unsqueeze(0) + temb.unsqueeze(2)).chunk(2, dim=2)]
        else:
            shift, scale = (self.table + temb.unsqueeze(1)).chunk(2, dim=1)
        x = (fp32_layer_norm(x, eps=cfg.eps) * (1 + scale) + shift).type_as(x)
        x = self.proj_out(x)
        x = x.reshape(B, ppf, pph, ppw, pt, ph, pw, -1).permute(0, 7, 1, 4, 2, 5, 3, 6)
        return (x.flatten(6, 7).flatten(4, 5).flatten(2, 3),)
