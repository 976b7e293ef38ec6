"""The reference's 9-function operator API (fastdm/kernel/operators_set.py:9-208) on the HIP library.

Same names, argument meaning, layouts, return conventions (rotary_pos_embedding mutates in place
and returns None) and error behaviour (AssertionError for shape/dtype contract violations, as the
reference's shims assert; RuntimeError for anything the native layer rejects, as TORCH_CHECK does).
Single backend: every call goes through the C ABI; nothing here computes on the host.
"""
from typing import Optional, Tuple

import torch

from fastdm_amd import _native as N
from fastdm_amd.kernel.utils import attention_pv_fp8

BF16 = torch.bfloat16
FP8 = torch.float8_e4m3fn


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("fastdm_amd operators need GPU tensors (no CPU fallback)")


def rms_norm(input: torch.Tensor, scale: torch.Tensor, eps: float) -> torch.Tensor:
    """operators_set.py:9-21 / cuda/norm.py:19-21.  Normalises over the last dim of a contiguous tensor."""
    _need_gpu(input, scale)
    assert input.dtype == BF16 and scale.dtype == BF16, "rms_norm: bf16 only"
    assert input.is_contiguous() and scale.is_contiguous() and scale.numel() == input.shape[-1]
    out = torch.empty_like(input)
    hidden = input.shape[-1]
    N.check(N.lib().fdm_rms_norm(N.ptr(out), N.ptr(input), N.ptr(scale), float(eps), input.numel() // hidden, hidden,
                                 N.stream()), "rms_norm")
    return out


def rotary_pos_embedding(query: torch.Tensor, key: torch.Tensor, head_size: int, cos_sin_cache: torch.Tensor,
                         is_neox: bool = False):
    """operators_set.py:23-52 / cuda/rotemb.py:36-37: in place on query/key [B,S,H*hd]; positions arange(S)."""
    _need_gpu(query, key, cos_sin_cache)
    assert query.dtype == BF16 and key.dtype == BF16 and cos_sin_cache.dtype == BF16, "rotary_pos_embedding: bf16 only"
    assert query.ndim == 3 and key.ndim == 3 and query.shape[:2] == key.shape[:2]
    assert query.stride(-1) == 1 and key.stride(-1) == 1 and cos_sin_cache.is_contiguous()
    B, S = query.shape[0], query.shape[1]
    assert query.stride(0) == S * query.stride(1) and key.stride(0) == S * key.stride(1), "batch must be dense over tokens"
    assert query.shape[2] % head_size == 0 and key.shape[2] % head_size == 0 and cos_sin_cache.shape[0] >= S
    N.check(N.lib().fdm_rope_inplace(N.ptr(query), N.ptr(key), N.ptr(cos_sin_cache), B * S, S, query.shape[2] // head_size,
                                     key.shape[2] // head_size, head_size, cos_sin_cache.shape[1], query.stride(1),
                                     key.stride(1), int(bool(is_neox)), N.stream()), "rotary_pos_embedding")
    return


def gelu_and_mul(input: torch.Tensor) -> torch.Tensor:
    """operators_set.py:54-67: x[..., :d] * gelu_erf(x[..., d:]) (the gate is the SECOND half)."""
    _need_gpu(input)
    assert input.dtype == BF16 and input.is_contiguous() and input.shape[-1] % 2 == 0
    d = input.shape[-1] // 2
    out = torch.empty(*input.shape[:-1], d, device=input.device, dtype=input.dtype)
    N.check(N.lib().fdm_gelu_and_mul(N.ptr(out), N.ptr(input), input.numel() // (2 * d), d, N.stream()), "gelu_and_mul")
    return out


def quantize_to_int8(input: torch.Tensor, symmetric: bool = True) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
    """operators_set.py:69-84 / cuda/quantize.py:24-29.  Returns (int8 [M,K], scale f32 [M,1], azp i32 [M,1] | None)."""
    _need_gpu(input)
    assert input.ndim == 2 and input.dtype == BF16 and input.stride(1) == 1
    M, K = input.shape
    q = torch.empty((M, K), device=input.device, dtype=torch.int8)
    s = torch.empty((M, 1), device=input.device, dtype=torch.float32)
    zp = None if symmetric else torch.empty((M, 1), device=input.device, dtype=torch.int32)
    N.check(N.lib().fdm_quant_int8_rowwise(N.ptr(q), N.ptr(s), N.ptr(zp), N.ptr(input), M, K, input.stride(0), N.stream()),
            "quantize_to_int8")
    return q, s, zp


def quantize_to_fp8(input: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """operators_set.py:86-100 / cuda/quantize.py:45-55.  Returns (e4m3fn [M,K], scale f32 [M,1])."""
    _need_gpu(input)
    assert input.ndim == 2 and input.dtype == BF16 and input.stride(1) == 1
    M, K = input.shape
    q = torch.empty((M, K), device=input.device, dtype=FP8)
    s = torch.empty((M, 1), device=input.device, dtype=torch.float32)
    N.check(N.lib().fdm_quant_fp8_rowwise(N.ptr(q), N.ptr(s), N.ptr(input), M, K, input.stride(0), N.stream()), "quantize_to_fp8")
    return q, s


def _check_mm(a, b, scale_a, scale_b, out_dtype, bias, dtype):
    """The TORCH_CHECKs of csrc/torch_bindings.cpp:31-61 (fp8) / :94-123 (int8)."""
    _need_gpu(a, b, scale_a, scale_b, bias)
    assert b.shape[0] % 16 == 0 and b.shape[1] % 16 == 0
    assert out_dtype is BF16, "out_dtype must be bfloat16 in this build"
    assert bias is None or (bias.shape[0] == b.shape[1] and bias.dtype == out_dtype)
    if a.dim() != 2 or b.dim() != 2:
        raise RuntimeError("mat_a / mat_b must be 2D tensors")
    if a.stride(1) != 1:
        raise RuntimeError("mat_a must be a row major tensor")
    if b.stride(0) != 1:
        raise RuntimeError("mat_b must be a column major tensor")
    if a.shape[1] != b.shape[0]:
        raise RuntimeError("mat_a and mat_b shapes cannot be multiplied")
    if a.dtype != dtype or b.dtype != dtype:
        raise RuntimeError(f"mat_a / mat_b must be {dtype}")
    if scale_a.numel() != a.shape[0] or scale_b.numel() != b.shape[1]:
        raise RuntimeError("size of scales is not matched")
    if not (scale_a.is_contiguous() and scale_b.is_contiguous()):
        raise RuntimeError("scales must be contiguous")
    if scale_a.dtype != torch.float32 or scale_b.dtype != torch.float32:
        raise RuntimeError("scales must be Float32")
    if bias is not None and not bias.is_contiguous():
        raise RuntimeError("bias must be contiguous")


def fp8_matmul(a: torch.Tensor, b: torch.Tensor, scale_a: torch.Tensor, scale_b: torch.Tensor, out_dtype: torch.dtype,
               bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """operators_set.py:102-124 / cuda/matrixmul.py:31-35.  a [M,K] row-major, b [K,N] column-major, both e4m3fn."""
    _check_mm(a, b, scale_a, scale_b, out_dtype, bias, FP8)
    M, K = a.shape
    Nn = b.shape[1]
    out = torch.empty((M, Nn), device=a.device, dtype=out_dtype)
    N.check(N.lib().fdm_gemm_w8a8_fp8(N.ptr(out), N.ptr(a), N.ptr(b), N.ptr(scale_a), N.ptr(scale_b), N.ptr(bias), M, Nn, K,
                                      a.stride(0), b.stride(1), Nn, N.EPI_NONE, None, 0, None, 0, N.stream()), "fp8_matmul")
    return out


def int8_matmul(a: torch.Tensor, b: torch.Tensor, scale_a: torch.Tensor, scale_b: torch.Tensor, out_dtype: torch.dtype,
                azp_adj: torch.Tensor, azp: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """operators_set.py:126-152 / cuda/matrixmul.py:65-69.  azp_adj [1,N] i32 (weight column sums), azp [M,1] i32."""
    _check_mm(a, b, scale_a, scale_b, out_dtype, bias, torch.int8)
    _need_gpu(azp_adj, azp)
    M, K = a.shape
    Nn = b.shape[1]
    if not (azp.numel() == M and azp.is_contiguous() and azp_adj.numel() == Nn and azp_adj.is_contiguous()):
        raise RuntimeError("azp / azp_adj size mismatch")
    assert azp.dtype == torch.int32 and azp_adj.dtype == torch.int32
    out = torch.empty((M, Nn), device=a.device, dtype=out_dtype)
    N.check(N.lib().fdm_gemm_w8a8_int8(N.ptr(out), N.ptr(a), N.ptr(b), N.ptr(scale_a), N.ptr(scale_b), N.ptr(azp_adj), N.ptr(azp),
                                       N.ptr(bias), M, Nn, K, a.stride(0), b.stride(1), Nn, N.EPI_NONE, None, 0, None, 0,
                                       N.stream()), "int8_matmul")
    return out


def _attn(query, key, value, num_q_heads, num_kv_heads, head_dim, is_causal, scale, mask, bq, bk):
    _need_gpu(query, key, value, mask)
    if is_causal and mask is not None:
        raise NotImplementedError("is_causal with a block mask: the reference's sparse shim drops the flag (kernel/cuda/attention.py:218-261); not built")
    assert query.dtype == BF16 and key.dtype == BF16 and value.dtype == BF16
    assert query.ndim == 3 and key.ndim == 3 and value.ndim == 3
    B, Sq, _ = query.shape
    Sk = key.shape[1]
    assert query.shape[2] == num_q_heads * head_dim and key.shape[2] == num_kv_heads * head_dim and value.shape == key.shape
    for t, S in ((query, Sq), (key, Sk), (value, Sk)):
        assert t.stride(2) == 1 and (B == 1 or t.stride(0) == S * t.stride(1)), "tokens must be dense within the batch"
    if scale is None:
        scale = head_dim ** -0.5
    out = torch.empty((B, Sq, num_q_heads * head_dim), device=query.device, dtype=query.dtype)
    if is_causal:       # torch's top-left-aligned causal mask (kernel/torch/attention.py:34-43 hands the flag to F.scaled_dot_product_attention); bf16 arithmetic always
        N.check(N.lib().fdm_attn_fwd_causal(N.ptr(out), N.ptr(query), N.ptr(key), N.ptr(value), B, Sq, Sk, num_q_heads, num_kv_heads, head_dim,
                                            query.stride(1), key.stride(1), value.stride(1), out.stride(1), float(scale), N.stream()),
                "scaled_dot_product_attention (causal)")
        return out
    ws = N.attn_workspace(query.device, head_dim)
    if mask is not None and head_dim == 128 and attention_pv_fp8() and -(-Sk // 64) <= 4096:      # opt-in: the reference's SpargeAttn class
        mode = attention_pv_fp8()
        vq = N.attn_pvfp8_workspace(query.device, B, Sk, num_kv_heads, head_dim, mode)
        N.check(N.lib().fdm_attn_fwd_quant_masked(N.ptr(out), N.ptr(query), N.ptr(key), N.ptr(value), B, Sq, Sk, num_q_heads, num_kv_heads,
                                                  head_dim, query.stride(1), key.stride(1), value.stride(1), out.stride(1), float(scale),
                                                  N.ptr(mask), bq, bk, mode, N.ptr(vq), vq.numel(), N.stream()),
                "sparse_scaled_dot_product_attention (fp8)")
        return out
    if mask is None and head_dim == 128 and attention_pv_fp8():      # opt-in: the reference's SageAttention class (kernel/utils.py)
        mode = attention_pv_fp8()
        vq = N.attn_pvfp8_workspace(query.device, B, Sk, num_kv_heads, head_dim, mode)
        fn = N.lib().fdm_attn_fwd_fp8 if mode == 2 else N.lib().fdm_attn_fwd_pvfp8
        N.check(fn(N.ptr(out), N.ptr(query), N.ptr(key), N.ptr(value), B, Sq, Sk, num_q_heads, num_kv_heads,
                                           head_dim, query.stride(1), key.stride(1), value.stride(1), out.stride(1), float(scale),
                                           N.ptr(ws), ws.numel(), N.ptr(vq), vq.numel(), N.stream()), "scaled_dot_product_attention (fp8 P.V)")
        return out
    N.check(N.lib().fdm_attn_fwd(N.ptr(out), N.ptr(query), N.ptr(key), N.ptr(value), B, Sq, Sk, num_q_heads, num_kv_heads,
                                 head_dim, query.stride(1), key.stride(1), value.stride(1), out.stride(1), float(scale),
                                 N.ptr(mask), bq, bk, N.ptr(ws), ws.numel(), N.stream()), "scaled_dot_product_attention")
    return out


def scaled_dot_product_attention(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, num_q_heads: int,
                                 num_kv_heads: int, head_dim: int, is_causal: bool = False,
                                 scale: Optional[float] = None) -> torch.Tensor:
    """operators_set.py:154-179: softmax(q k^T * scale) v, [B,S,H*D] in and out (no transposes)."""
    return _attn(query, key, value, num_q_heads, num_kv_heads, head_dim, is_causal, scale, None, 0, 0)


SPARSE_BLOCK_Q, SPARSE_BLOCK_K = 128, 64  # the reference's off-sm90 SpargeAttn layout (cuda/attention.py:89-94)


def sparse_scaled_dot_product_attention(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, num_q_heads: int,
                                        num_kv_heads: int, head_dim: int, is_causal: bool = False,
                                        scale: Optional[float] = None, sparse_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """operators_set.py:181-208: sparse_mask int8 [B,H,ceil(Sq/128),ceil(Sk/64)], 1 = compute, 0 = skip."""
    if sparse_mask is None:
        return _attn(query, key, value, num_q_heads, num_kv_heads, head_dim, is_causal, scale, None, 0, 0)
    B, Sq, Sk = query.shape[0], query.shape[1], key.shape[1]
    nq, nk = -(-Sq // SPARSE_BLOCK_Q), -(-Sk // SPARSE_BLOCK_K)
    assert sparse_mask.dtype == torch.int8 and sparse_mask.is_contiguous()
    assert tuple(sparse_mask.shape) == (B, num_q_heads, nq, nk), f"sparse_mask must be {(B, num_q_heads, nq, nk)}"
    return _attn(query, key, value, num_q_heads, num_kv_heads, head_dim, is_causal, scale, sparse_mask, SPARSE_BLOCK_Q, SPARSE_BLOCK_K)
