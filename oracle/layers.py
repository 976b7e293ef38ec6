"""Layer-level oracle (torch-CPU restatement of fastdm/layer/*.py for the FLUX path).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Citations relative to /root/reference.
"""
import math
from typing import List, Optional

import torch
import torch.nn.functional as F

from . import ops


class QLinear:
    """fastdm/layer/qlinear.py:6-81.

    Weights arrive in the diffusers layout [out, in]; several are fused along N in the
    given order (qlinear.py:29).  quant in {None, torch.float8_e4m3fn, torch.int8}:
    per-output-channel weight quantisation from the bf16 weights in fp32 (qlinear.py:40-50);
    forward = per-token activation quant -> scaled GEMM (+bias) (qlinear.py:69-76).
    """

    def __init__(self, weights: List[torch.Tensor], biases: List[Optional[torch.Tensor]], quant=None):
        w_nk = torch.cat(list(weights), dim=0).contiguous()          # [N, K]
        self.bias = None if biases[0] is None else torch.cat(list(biases), dim=0).contiguous()
        self.quant = quant
        self.colsum = None
        if quant is None:
            self.w_kn = w_nk.t()                                       # [K, N] view, K contiguous
            self.w_scale = None
        elif quant == torch.float8_e4m3fn:
            q, s = ops.fp8_weight_quantization(w_nk)
            self.w_kn, self.w_scale = q.t(), s
        elif quant == torch.int8:
            q, s = ops.int8_weight_quantization(w_nk)
            self.w_kn, self.w_scale = q.t(), s
            self.colsum = self.w_kn.to(torch.int32).sum(dim=0, keepdim=True, dtype=torch.int32)  # qlinear.py:49
        else:
            raise ValueError(f"Unsupported quantization type: {quant}")

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        if self.quant == torch.float8_e4m3fn:
            xq, xs = ops.quantize_to_fp8(x2)
            y = ops.fp8_matmul(xq, self.w_kn, xs, self.w_scale, x.dtype, self.bias)
        elif self.quant == torch.int8:
            xq, xs, zp = ops.quantize_to_int8(x2, symmetric=False)
            y = ops.int8_matmul(xq, self.w_kn, xs, self.w_scale, x.dtype, self.colsum, zp, self.bias)
        else:
            y = torch.addmm(self.bias, x2, self.w_kn) if self.bias is not None else torch.mm(x2, self.w_kn)
        return y.view(*shp[:-1], self.w_kn.shape[-1])


def linear_from(sd, names, quant=None) -> QLinear:
    """fastdm/model/basemodel.py:33-63: gather `<name>.weight` / `<name>.bias` for each name."""
    ws = [sd[f"{n}.weight"] for n in names]
    bs = [sd.get(f"{n}.bias") for n in names]
    return QLinear(ws, bs, quant)


# ------------------------------------------------------------------ embeddings (a14)
def timestep_embedding(t: torch.Tensor, dim: int = 256) -> torch.Tensor:
    """fastdm/layer/embeddings.py:18-69 with flip_sin_to_cos=True, downscale_freq_shift=0,
    scale=1, max_period=1e4 (embeddings.py:573): fp32, output [N, dim] = [cos | sin]."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half)
    arg = t[:, None].float() * freqs[None, :]
    return torch.cat([torch.cos(arg), torch.sin(arg)], dim=-1)


def flux_pos_embed(ids: torch.Tensor, axes_dim=(16, 56, 56), theta: float = 10000.0):
    """fastdm/layer/embeddings.py:527-549 + 196-209: per axis fp64 freqs, cos/sin cast to
    fp32 and each value repeated twice along the last dim.  Returns (cos, sin) [S, sum(axes)]."""
    pos = ids.float()
    cos_out, sin_out = [], []
    for i, d in enumerate(axes_dim):
        freqs = 1.0 / (theta ** (torch.arange(0, d, 2, dtype=torch.float64)[: d // 2] / d))
        ang = torch.outer(pos[:, i], freqs)                # float32 (x) float64 -> float64
        cos_out.append(ang.cos().repeat_interleave(2, dim=1).float())
        sin_out.append(ang.sin().repeat_interleave(2, dim=1).float())
    return torch.cat(cos_out, dim=-1), torch.cat(sin_out, dim=-1)


def flux_rope_cache(txt_ids, img_ids, axes_dim=(16, 56, 56), dtype=torch.bfloat16):
    """fastdm/model/flux.py:425-428: ids = [txt; img]; cache = [cos[:,0::2] | sin[:,1::2]] in dtype."""
    cos, sin = flux_pos_embed(torch.cat((txt_ids, img_ids), dim=0), axes_dim)
    return torch.cat((cos[:, 0::2], sin[:, 1::2]), dim=-1).to(dtype)


class TimeTextEmbed:
    """CombinedTimestepGuidanceTextProjEmbeddings, fastdm/layer/embeddings.py:569-590
    (TimestepEmbedding :400-410, PixArtAlphaTextProjection :118-147 with act 'silu')."""

    def __init__(self, sd, prefix="time_text_embed"):
        self.t1 = linear_from(sd, [f"{prefix}.timestep_embedder.linear_1"])
        self.t2 = linear_from(sd, [f"{prefix}.timestep_embedder.linear_2"])
        self.g1 = linear_from(sd, [f"{prefix}.guidance_embedder.linear_1"])
        self.g2 = linear_from(sd, [f"{prefix}.guidance_embedder.linear_2"])
        self.p1 = linear_from(sd, [f"{prefix}.text_embedder.linear_1"])
        self.p2 = linear_from(sd, [f"{prefix}.text_embedder.linear_2"])

    def __call__(self, timestep, guidance, pooled):
        te = self.t2(F.silu(self.t1(timestep_embedding(timestep).to(pooled.dtype))))
        ge = self.g2(F.silu(self.g1(timestep_embedding(guidance).to(pooled.dtype))))
        pe = self.p2(F.silu(self.p1(pooled)))
        return (te + ge) + pe


# ------------------------------------------------------------------ AdaLN (a12)
def layer_norm_noaffine(x, eps=1e-6):
    return F.layer_norm(x, (x.shape[-1],), None, None, eps)


def modulate(x, scale, shift):
    """LN(x) * (1 + scale[:,None]) + shift[:,None], each op rounded in x.dtype
    (fastdm/layer/normalization.py:198)."""
    return layer_norm_noaffine(x) * (1 + scale[:, None]) + shift[:, None]


class AdaLayerNormZero:
    """fastdm/layer/normalization.py:162-199: emb = Linear(silu(temb)) -> 6 chunks."""

    def __init__(self, sd, name):
        self.linear = linear_from(sd, [name])

    def __call__(self, x, temb):
        emb = self.linear(F.silu(temb))
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = emb.chunk(6, dim=1)
        return modulate(x, scale_msa, shift_msa), gate_msa, shift_mlp, scale_mlp, gate_mlp


class AdaLayerNormZeroSingle:
    """fastdm/layer/normalization.py:201-236: 3 chunks (shift, scale, gate)."""

    def __init__(self, sd, name):
        self.linear = linear_from(sd, [name])

    def __call__(self, x, temb):
        emb = self.linear(F.silu(temb))
        shift, scale, gate = emb.chunk(3, dim=1)
        return modulate(x, scale, shift), gate


class AdaLayerNormContinuous:
    """fastdm/layer/normalization.py:90-128: emb -> (scale, shift) (scale FIRST)."""

    def __init__(self, sd, name):
        self.linear = linear_from(sd, [name])

    def __call__(self, x, cond):
        emb = self.linear(F.silu(cond).to(x.dtype))
        scale, shift = torch.chunk(emb, 2, dim=1)
        return layer_norm_noaffine(x) * (1 + scale)[:, None, :] + shift[:, None, :]


# ------------------------------------------------------------------ attention / FFN (a11)
class FluxAttention:
    """fastdm/layer/transformer.py:232-317 restricted to the FLUX uses: joint
    (added_kv_proj_dim set, context_pre_only=False) and single-stream (pre_only=True)."""

    def __init__(self, sd, prefix, heads, head_dim, quant, joint: bool):
        self.heads, self.head_dim, self.joint = heads, head_dim, joint
        self.inner = heads * head_dim
        self.qkv = linear_from(sd, [f"{prefix}.to_q", f"{prefix}.to_k", f"{prefix}.to_v"], quant)
        self.norm_q = sd[f"{prefix}.norm_q.weight"]
        self.norm_k = sd[f"{prefix}.norm_k.weight"]
        if joint:
            self.add_qkv = linear_from(sd, [f"{prefix}.add_q_proj", f"{prefix}.add_k_proj", f"{prefix}.add_v_proj"], quant)
            self.norm_added_q = sd[f"{prefix}.norm_added_q.weight"]
            self.norm_added_k = sd[f"{prefix}.norm_added_k.weight"]
            self.to_out = linear_from(sd, [f"{prefix}.to_out.0"], quant)
            self.to_add_out = linear_from(sd, [f"{prefix}.to_add_out"], quant)
        self.eps = 1e-6
        self.scale = head_dim ** -0.5

    def _split_norm(self, fused, wq, wk):
        B = fused.shape[0]
        q, k, v = fused[..., : self.inner], fused[..., self.inner: 2 * self.inner], fused[..., 2 * self.inner:]
        q = ops.rms_norm(q.unflatten(-1, (self.heads, -1)).contiguous(), wq, self.eps).view(B, -1, self.inner)
        k = ops.rms_norm(k.unflatten(-1, (self.heads, -1)).contiguous(), wk, self.eps).view(B, -1, self.inner)
        return q, k, v

    def __call__(self, hidden, encoder=None, rope=None):
        q, k, v = self._split_norm(self.qkv(hidden), self.norm_q, self.norm_k)
        if encoder is not None:
            eq, ek, ev = self._split_norm(self.add_qkv(encoder), self.norm_added_q, self.norm_added_k)
            q = torch.cat([eq, q], dim=1)           # [text ; image], transformer.py:293-295
            k = torch.cat([ek, k], dim=1)
            v = torch.cat([ev, v], dim=1)
        else:
            q, k = q.contiguous(), k.contiguous()
        if rope is not None:
            ops.rotary_pos_embedding(q, k, self.head_dim, rope, is_neox=False)
        o = ops.scaled_dot_product_attention(q, k, v, self.heads, self.heads, self.head_dim, scale=self.scale)
        if encoder is not None:
            n_txt = encoder.shape[1]
            return self.to_out(o[:, n_txt:]), self.to_add_out(o[:, :n_txt])
        return o


class FeedForwardGeluTanh:
    """fastdm/layer/transformer.py:14-62 with activation_fn='gelu-approximate'
    (activations.py:38-41): Linear -> tanh-GELU -> Linear."""

    def __init__(self, sd, prefix, quant):
        self.proj = linear_from(sd, [f"{prefix}.net.0.proj"], quant)
        self.out = linear_from(sd, [f"{prefix}.net.2"], quant)

    def __call__(self, x):
        return self.out(F.gelu(self.proj(x), approximate="tanh"))
