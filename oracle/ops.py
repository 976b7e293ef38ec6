"""Operator-level oracle: the reference's torch backend, restated on torch-CPU.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  All citations are relative to
/root/reference.  Inputs/outputs use the reference's layouts.
"""
import math
from typing import Optional, Tuple

import torch
import torch.nn.functional as F

FP8 = torch.float8_e4m3fn
FP8_MAX = 448.0


# ----------------------------------------------------------------------------- a5
def rms_norm(x: torch.Tensor, weight: Optional[torch.Tensor], eps: float) -> torch.Tensor:
    """fastdm/kernel/torch/norm.py:19-27 (and csrc/elmwise_ops.cu:56-84).

    mean(x^2) in fp32 over the last dim; x*rsqrt(.) is an fp32 product (bf16 * fp32
    promotes), rounded to the weight dtype *before* the weight multiply.
    """
    in_dtype = x.dtype
    var = x.float().square().mean(dim=-1, keepdim=True)
    y = x.float() * torch.rsqrt(var + eps)
    if weight is None:
        return y.to(in_dtype)
    return y.to(weight.dtype) * weight


# ----------------------------------------------------------------------------- a6
def rotary_pos_embedding(query: torch.Tensor, key: torch.Tensor, head_size: int,
                         cos_sin_cache: torch.Tensor, is_neox: bool = False) -> None:
    """fastdm/kernel/torch/rotemb.py:33-65; in place on query and key, returns None.

    positions are arange(seq_len); cache row = [cos(head_size/2) | sin(head_size/2)].
    Arithmetic is carried out in the tensor dtype with one rounding per op
    (x*c, y*s, difference), as torch does for bf16 tensors.
    """
    seq = query.shape[1]
    half = cos_sin_cache.shape[-1] // 2
    cos = cos_sin_cache[:seq, :half].to(query.dtype)[None, :, None, :]
    sin = cos_sin_cache[:seq, half:].to(query.dtype)[None, :, None, :]

    def rot(t):
        shp = t.shape
        v = t.reshape(shp[0], shp[1], -1, head_size)
        if is_neox:
            a, b = v[..., :head_size // 2], v[..., head_size // 2:]
        else:
            a, b = v[..., 0::2], v[..., 1::2]
        ra = a * cos - b * sin
        rb = b * cos + a * sin
        if is_neox:
            out = torch.cat((ra, rb), dim=-1)
        else:
            out = torch.stack((ra, rb), dim=-1).flatten(-2)
        t.copy_(out.reshape(shp))

    rot(query)
    rot(key)
    return None


# ----------------------------------------------------------------------------- a9
def gelu_and_mul(x: torch.Tensor) -> torch.Tensor:
    """fastdm/kernel/torch/gelumul.py:14-16: first half times erf-GELU(second half)."""
    d = x.shape[-1] // 2
    return x[..., :d] * F.gelu(x[..., d:])


# ----------------------------------------------------------------------------- a1
def quantize_to_fp8(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """fastdm/kernel/torch/quantize.py:58-67: per-row absmax (clamped >= 1e-12) / 448;
    q = e4m3fn(clamp(x / scale, +-448)).  Returns (q [M,K] fp8, scale [M,1] f32)."""
    assert x.ndim == 2
    amax = x.abs().amax(dim=1).clamp(min=1e-12)
    scale = amax.float() / FP8_MAX
    q = (x.float() / scale[:, None]).clamp(-FP8_MAX, FP8_MAX).to(FP8)
    return q, scale[:, None]


# ----------------------------------------------------------------------------- a2
def quantize_to_int8(x: torch.Tensor, symmetric: bool = True):
    """fastdm/kernel/torch/quantize.py:21-43.  asymmetric: scale=(max-min)/255,
    zp = -128 - round(min/scale) (int32), q = clamp(round(x/scale + zp), -128, 127)."""
    assert x.ndim == 2
    xf = x.float()
    lo = xf.amin(dim=1)
    hi = xf.amax(dim=1)
    if symmetric:
        scale = torch.maximum(lo.abs(), hi.abs()) / 127
        q = torch.round(xf / scale[:, None]).clamp(-128, 127).to(torch.int8)
        return q, scale[:, None], None
    scale = (hi - lo) / 255
    zp = (-128 - torch.round(lo / scale)).to(torch.int32)
    q = torch.round(xf / scale[:, None] + zp.float()[:, None]).clamp(-128, 127).to(torch.int8)
    return q, scale[:, None], zp[:, None]


# ----------------------------------------------------------------------------- a3
def fp8_matmul(a, b, scale_a, scale_b, out_dtype, bias=None, chunk: int = 2048):
    """fastdm/kernel/torch/matrixmul.py:29-35 = torch._scaled_mm row-wise, which is
    unsupported on CPU; restated from the CUDA epilogue csrc/gemm/hopper_w8a8_fp8.cu:95-142:
    out = T( acc_f32 * sa[M,1] * sb[N,1]^T + bias[N] ) with one final rounding."""
    assert b.shape[0] % 16 == 0 and b.shape[1] % 16 == 0
    bf = b.float()
    sb = scale_b.reshape(1, -1).float()
    outs = []
    for s in range(0, a.shape[0], chunk):
        acc = a[s:s + chunk].float() @ bf
        r = acc * scale_a[s:s + chunk].float() * sb
        if bias is not None:
            r = r + bias.float()
        outs.append(r.to(out_dtype))
    return torch.cat(outs, 0) if len(outs) != 1 else outs[0]


# ----------------------------------------------------------------------------- a4
def int8_matmul(a, b, scale_a, scale_b, out_dtype, azp_adj, azp, bias=None):
    """fastdm/kernel/torch/matrixmul.py:67-74.  The reference accumulates in fp32
    (a.float()@b.float()); here the int product is exact (float64 accumulate, every
    value < 2^53) and converted to fp32 once -- the CUDA kernel's int32 accumulate
    (csrc/gemm/hopper_w8a8_int8.cu) does the same.  Then
    out = T((acc - azp*azp_adj) * (sa*sb^T)); bias is added afterwards in T (second rounding)."""
    assert b.shape[0] % 16 == 0 and b.shape[1] % 16 == 0
    acc = (a.double() @ b.double()) - azp.double() @ azp_adj.double()
    sc = scale_a.float() * scale_b.reshape(1, -1).float()
    out = (acc.float() * sc).to(out_dtype)
    return out + bias if bias is not None else out


# ----------------------------------------------------------------------------- a7 / a8
def scaled_dot_product_attention(query, key, value, num_q_heads, num_kv_heads, head_dim,
                                 is_causal=False, scale=None, block_mask=None, block_q=128, block_k=64):
    """fastdm/kernel/torch/attention.py:34-43 delegates to torch SDPA (third-party
    arithmetic).  Restated the way the reference's own test pins it
    (tests/test_attention.py:23-63): fp32 scores, fp32 softmax, fp32 P@V, one rounding
    to the input dtype.  Layout [B, S, H*D] in and out.

    ``block_mask`` (int8 [B,H,ceil(Sq/block_q),ceil(Sk/block_k)], 1 = compute) gives the
    block-sparse variant (fastdm/kernel/cuda/attention.py:77-147 contract): skipped blocks
    get -inf before the softmax.  That definition is ours: parity unpinned upstream.
    """
    B, Sq, _ = query.shape
    Sk = key.shape[1]
    q = query.view(B, Sq, num_q_heads, head_dim).transpose(1, 2).float()
    k = key.view(B, Sk, num_kv_heads, head_dim).transpose(1, 2).float()
    v = value.view(B, Sk, num_kv_heads, head_dim).transpose(1, 2).float()
    if num_kv_heads != num_q_heads:
        rep = num_q_heads // num_kv_heads
        k = k.repeat_interleave(rep, dim=1)
        v = v.repeat_interleave(rep, dim=1)
    if scale is None:
        scale = 1.0 / math.sqrt(head_dim)
    out = torch.empty(B, num_q_heads, Sq, head_dim, dtype=torch.float32)
    for b in range(B):
        for h in range(num_q_heads):
            s = (q[b, h] @ k[b, h].T) * scale
            if is_causal:
                cm = torch.ones(Sq, Sk, dtype=torch.bool).tril(diagonal=Sk - Sq)
                s = s.masked_fill(~cm, float("-inf"))
            if block_mask is not None:
                bm = block_mask[b, h].bool()
                bm = bm.repeat_interleave(block_q, 0)[:Sq].repeat_interleave(block_k, 1)[:, :Sk]
                s = s.masked_fill(~bm, float("-inf"))
            p = torch.softmax(s, dim=-1)
            out[b, h] = p @ v[b, h]
    return out.transpose(1, 2).reshape(B, Sq, num_q_heads * head_dim).to(query.dtype)


# ----------------------------------------------------------------------------- a10 (load time)
def fp8_weight_quantization(w_nk: torch.Tensor):
    """fastdm/utils/quantization.py:42-63 applied to W^T [N,K] (layer/qlinear.py:40-44):
    per-output-channel scale [N,1] f32 and fp8 weights [N,K]."""
    return quantize_to_fp8(w_nk)


def int8_weight_quantization(w_nk: torch.Tensor):
    """fastdm/utils/quantization.py:5-40 (symmetric=True default, layer/qlinear.py:46-50)."""
    q, s, _ = quantize_to_int8(w_nk, symmetric=True)
    return q, s
