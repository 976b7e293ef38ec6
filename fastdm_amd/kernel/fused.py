"""Python entry points of the fused HIP kernels the FLUX wiring uses (C ABI: include/fastdm_amd.h).

Activations are 2-D row-major bf16 matrices [rows, width]; `rows_per_batch` tells a kernel which
row of a per-batch modulation vector applies.  Nothing here computes on the host.
"""
import os
from typing import Optional, Tuple

import torch

from fastdm_amd import _native as N
from fastdm_amd.kernel.timing import span

BF16 = torch.bfloat16
FP8 = torch.float8_e4m3fn
QUANT_MODE = {None: 0, FP8: 1, torch.int8: 2}


class Quantized:
    """A per-row quantised activation: q [M,K] (fp8 / int8), scale [M,1] f32, azp [M,1] i32 | None."""
    __slots__ = ("q", "scale", "azp")

    def __init__(self, q, scale, azp=None):
        self.q, self.scale, self.azp = q, scale, azp


def _alloc_q(rows, width, quant, device):
    q = torch.empty((rows, width), device=device, dtype=quant)
    s = torch.empty((rows, 1), device=device, dtype=torch.float32)
    zp = torch.empty((rows, 1), device=device, dtype=torch.int32) if quant == torch.int8 else None
    return Quantized(q, s, zp)


def ln_modulate(x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, rows_per_batch: int, quant=None,
                want_bf16: bool = False, res: Optional[torch.Tensor] = None, gate: Optional[torch.Tensor] = None,
                eps: float = 1e-6, scale_has_one: bool = False) -> Tuple[Optional[torch.Tensor], Optional[Quantized], Optional[torch.Tensor]]:
    """y = LN(x)*(1+scale)+shift (torch bf16 rounding order), optionally quantised per row.
    scale_has_one: `scale` is already bf16(1 + scale) (ModulationTable adds the 1 in the GEMV that produces it).

    scale/shift/gate are [B, width] views (any batch stride, unit inner stride).  With res/gate:
    x := res + gate*x first; the updated x is returned as the third value.
    """
    rows, width = x.shape
    assert x.is_contiguous() and scale.stride(-1) == 1 and shift.stride(-1) == 1 and scale.stride(0) == shift.stride(0)
    out = torch.empty_like(x) if (want_bf16 or quant is None) else None
    qz = _alloc_q(rows, width, quant, x.device) if quant is not None else None
    x_new = torch.empty_like(x) if res is not None else None
    if gate is not None:
        assert gate.stride(0) == scale.stride(0) and gate.stride(-1) == 1
    N.check(N.lib().fdm_ln_modulate(N.ptr(out), N.ptr(qz.q) if qz else None, N.ptr(qz.scale) if qz else None,
                                    N.ptr(qz.azp) if (qz and qz.azp is not None) else None, QUANT_MODE[quant], N.ptr(x),
                                    N.ptr(scale), N.ptr(shift), scale.stride(0), N.ptr(res), N.ptr(gate), N.ptr(x_new), rows,
                                    width, rows_per_batch, eps, 1 if scale_has_one else 0, N.stream()), "ln_modulate")
    return out, qz, x_new


def ln_modulate_f32(x: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, rows_per_batch: int, quant=None,
                    want_bf16: bool = False, eps: float = 1e-6) -> Tuple[Optional[torch.Tensor], Optional[Quantized]]:
    """Wan's fp32 modulation: y = bf16(LN_f32(x) * (1 + scale) + shift), scale/shift fp32 [B, width] (or [1, width]
    broadcast over the batch: an affine FP32LayerNorm with scale = weight - 1, shift = bias), one rounding."""
    rows, width = x.shape
    assert x.is_contiguous() and scale.dtype == torch.float32 and shift.dtype == torch.float32
    assert scale.stride(-1) == 1 and shift.stride(-1) == 1 and scale.shape == shift.shape
    bstride = 0 if scale.shape[0] == 1 else scale.stride(0)
    assert bstride == (0 if shift.shape[0] == 1 else shift.stride(0))
    out = torch.empty_like(x) if (want_bf16 or quant is None) else None
    qz = _alloc_q(rows, width, quant, x.device) if quant is not None else None
    N.check(N.lib().fdm_ln_modulate_f32(N.ptr(out), N.ptr(qz.q) if qz else None, N.ptr(qz.scale) if qz else None,
                                        N.ptr(qz.azp) if (qz and qz.azp is not None) else None, QUANT_MODE[quant], N.ptr(x),
                                        N.ptr(scale), N.ptr(shift), bstride, rows, width, rows_per_batch, eps, N.stream()),
            "ln_modulate_f32")
    return out, qz


def rms_norm_rows(x: torch.Tensor, weight: torch.Tensor, eps: float, rope: Optional[torch.Tensor] = None, seq_len: int = 0,
                  head_dim: int = 0) -> torch.Tensor:
    """rms_norm over the last dim of a [rows, hidden] matrix whose rows may be column slices of a wider matrix.
    rope ([>= seq_len, head_dim] cos | sin table): the GPT-J rotary embedding of `rotary_pos_embedding(..., is_neox=False)` with
    positions row % seq_len is applied to the normalised row in the same pass (same bits as the two operators)."""
    rows, hidden = x.shape
    assert x.stride(1) == 1 and weight.is_contiguous() and weight.numel() == hidden
    out = torch.empty((rows, hidden), device=x.device, dtype=x.dtype)
    if rope is not None:
        assert rope.dtype == BF16 and rope.is_contiguous() and rope.shape[1] == head_dim and rope.shape[0] >= seq_len > 0 and rows % seq_len == 0
        N.check(N.lib().fdm_rms_norm_rope_strided(N.ptr(out), N.ptr(x), N.ptr(weight), eps, rows, hidden, x.stride(0), hidden, N.ptr(rope),
                                                  seq_len, head_dim, N.stream()), "rms_norm_rope_strided")
        return out
    N.check(N.lib().fdm_rms_norm_strided(N.ptr(out), N.ptr(x), N.ptr(weight), eps, rows, hidden, x.stride(0), hidden, N.stream()),
            "rms_norm_strided")
    return out


def gate_residual(res: torch.Tensor, y: torch.Tensor, gate: torch.Tensor, rows_per_batch: int) -> torch.Tensor:
    """res + gate[b]*y (model/flux.py:153-154)."""
    rows, width = res.shape
    assert res.is_contiguous() and y.is_contiguous() and gate.stride(-1) == 1
    out = torch.empty_like(res)
    N.check(N.lib().fdm_gate_residual(N.ptr(out), N.ptr(res), N.ptr(y), N.ptr(gate), gate.stride(0), rows, width, rows_per_batch,
                                      N.stream()), "gate_residual")
    return out


def gelu_and_mul_quant(x: torch.Tensor, quant) -> Quantized:
    """gelu_and_mul(x) quantised per token in one pass (the GEGLU feed-forward's proj2 input); bit-identical to the two operators."""
    rows, d2 = x.shape
    assert x.is_contiguous() and x.dtype == BF16 and d2 % 16 == 0
    qz = _alloc_q(rows, d2 // 2, quant, x.device)
    N.check(N.lib().fdm_gelu_and_mul_quant(N.ptr(qz.q), N.ptr(qz.scale), N.ptr(qz.azp), QUANT_MODE[quant], N.ptr(x), rows, d2 // 2, N.stream()),
            "gelu_and_mul_quant")
    return qz


_gn_ws = N.StreamScratch("GroupNorm statistics workspace", min_bytes=1 << 16)


def group_norm(x: torch.Tensor, groups: int, gamma: torch.Tensor, beta: torch.Tensor, eps: float, addend: Optional[torch.Tensor] = None,
               silu: bool = False) -> torch.Tensor:
    """F.group_norm(x [+ addend[:, :, None, None]], groups, gamma, beta, eps) [-> F.silu] on an NCHW-contiguous bf16 tensor in two
    kernels (statistics per slice, merge + apply) instead of torch's add / moments / fused-parameters / apply / SiLU passes
    (the UNet's call sites: layer/resnetblock.py, layer/unetblock.py, model/sdxl.py).  Channels-last tensors, spatial sizes that are not
    a multiple of 8 and FDM_GN_TORCH=1 (A/B runs) take the torch calls -- the same library ops the reference makes there."""
    B, C, H, W = x.shape
    hw = H * W
    if (x.dtype == BF16 and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last) and C % groups == 0 and C % 8 == 0
            and (C // groups) % 2 == 0 and groups <= 64 and C <= 4096 and B <= 65535 and os.environ.get("FDM_GN_TORCH", "0") != "1"
            and all(t.dtype == BF16 and t.is_contiguous() and t.numel() == C and t.device == x.device for t in (gamma, beta))):
        # channels-last activations (the layout of the HIP convolution): the NHWC form of the same two kernels
        if addend is not None:
            assert addend.shape == (B, C) and addend.dtype == BF16 and addend.is_contiguous()
        ws = _gn_ws.get(x.device, N.lib().fdm_group_norm_nhwc_workspace_bytes(B, C, hw, groups))
        out = torch.empty_like(x)          # preserves the channels_last strides
        N.check(N.lib().fdm_group_norm_nhwc(N.ptr(out), N.ptr(x), N.ptr(gamma), N.ptr(beta), N.ptr(addend), B, C, hw, groups, eps, 1 if silu else 0,
                                            N.ptr(ws), ws.numel(), N.stream()), "group_norm_nhwc")
        return out
    affine_ok = all(t.dtype == BF16 and t.is_contiguous() and t.numel() == C and t.device == x.device for t in (gamma, beta))   # read as raw bf16[C]
    if (x.dtype != BF16 or not x.is_contiguous() or (hw & 7) or C % groups or B * groups > 65535 or (C // groups) * hw >= (1 << 31)
            or not affine_ok or os.environ.get("FDM_GN_TORCH", "0") == "1"):
        import torch.nn.functional as F
        y = F.group_norm(x if addend is None else x + addend[:, :, None, None], groups, gamma, beta, eps)
        return F.silu(y) if silu else y
    if addend is not None:
        assert addend.shape == (B, C) and addend.dtype == BF16 and addend.is_contiguous()
    ws = _gn_ws.get(x.device, N.lib().fdm_group_norm_workspace_bytes(B, C, hw, groups))
    out = torch.empty_like(x)
    N.check(N.lib().fdm_group_norm(N.ptr(out), N.ptr(x), N.ptr(gamma), N.ptr(beta), N.ptr(addend), B, C, hw, groups, eps, 1 if silu else 0,
                                   N.ptr(ws), ws.numel(), N.stream()), "group_norm")
    return out


def conv2d(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], stride: int = 1, padding: int = 0, upsample: bool = False,
           res: Optional[torch.Tensor] = None) -> torch.Tensor:
    """F.conv2d of the UNet's call sites (reference layer/resnetblock.py:29-47, layer/unetblock.py:203-240, model/sdxl.py:417,470; cuDNN there)
    [+ the nearest 2x upsampling in front of it] [+ the residual add behind it].  On channels-last bf16 tensors:
      * 3 x 3 / padding 1 (stride 1 or 2): the implicit GEMM on the HIP tile kernel (fdm_conv2d_3x3_nhwc_bf16: fp32 accumulation over all
        9 C_in products, bias added in fp32, ONE rounding).  C_in < 32 (conv_in's 4 latent channels) is zero-padded to 32, C_out that is not a
        multiple of 8 (conv_out's 4) to 8 -- weights once (cached), activations per call (tiny tensors);
      * 1 x 1 / stride 1 (the ResNet shortcuts): fdm_gemm_bf16 on the [B*H*W, C] pixel matrix.
    Everything else -- NCHW activations, FDM_CONV_TORCH=1 (A/B) -- takes the library call the reference makes.  `weight` is the [O, I, kh, kw]
    tensor in channels_last memory format: its bytes ARE the [O, kh, kw, I] layout the kernels read."""
    import torch.nn.functional as F
    B, C, H, W = x.shape
    O = weight.shape[0]
    cl = torch.channels_last
    common = (x.dtype == BF16 and weight.dtype == BF16 and weight.dim() == 4 and weight.shape[1] == C and x.is_contiguous(memory_format=cl)
              and weight.is_contiguous(memory_format=cl) and (bias is None or (bias.dtype == BF16 and bias.is_contiguous()))
              and (res is None or (res.dtype == BF16 and res.is_contiguous(memory_format=cl))) and os.environ.get("FDM_CONV_TORCH", "0") != "1")
    k = tuple(weight.shape[2:])
    if common and k == (1, 1) and stride == 1 and padding == 0 and not upsample and res is None and C % 8 == 0 and O % 8 == 0:
        out = torch.empty((B, O, H, W), device=x.device, dtype=BF16, memory_format=cl)
        M = B * H * W
        with span("gemm_bf16", 2.0 * M * O * C):
            N.check(N.lib().fdm_gemm_bf16(N.ptr(out), N.ptr(x), N.ptr(weight), N.ptr(bias), M, O, C, C, C, O, N.EPI_NONE, None, 0, None, 0, N.stream()),
                    "conv2d_1x1_nhwc")
        return out
    if not (common and k == (3, 3) and padding == 1 and stride in (1, 2) and not (upsample and stride != 1)):
        if upsample:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        y = F.conv2d(x, weight, bias, stride, padding)
        return y if res is None else res + y
    Cp, Op = (C + 31) // 32 * 32, (O + 7) // 8 * 8
    if Cp != C or Op != O:
        key = (weight.data_ptr(), None if bias is None else bias.data_ptr(), Cp, Op)
        pad = _conv_pad.get(key)
        if pad is None:
            wp = torch.zeros((Op, Cp, 3, 3), device=x.device, dtype=BF16).contiguous(memory_format=cl)
            wp[:O, :C] = weight
            bp = None
            if bias is not None:
                bp = torch.zeros(Op, device=x.device, dtype=BF16)
                bp[:O] = bias
            pad = _conv_pad[key] = (wp, bp, weight, bias)      # (the originals are kept alive: the key holds their addresses)
        weight, bias = pad[0], pad[1]
        if Cp != C:
            xp = torch.zeros((B, Cp, H, W), device=x.device, dtype=BF16).contiguous(memory_format=cl)
            xp[:, :C] = x
            x = xp
        assert res is None or Op == O
    Ho, Wo = (2 * H, 2 * W) if upsample else ((H - 1) // stride + 1, (W - 1) // stride + 1)
    out = torch.empty((B, Op, Ho, Wo), device=x.device, dtype=BF16, memory_format=cl)
    ones = None
    if res is not None:
        assert res.shape == out.shape
        ones = _conv_ones.get((x.device, Op))
        if ones is None:
            ones = _conv_ones[(x.device, Op)] = torch.ones(Op, device=x.device, dtype=BF16)
    ws = N.gemm_int8_workspace(x.device)      # the per-(device, stream) split-K scratch of the tile GEMMs (pinned once a capture has seen it)
    with span("conv3x3_bf16", 2.0 * B * Ho * Wo * Op * 9 * Cp):
        N.check(N.lib().fdm_conv2d_3x3_nhwc_bf16(N.ptr(out), N.ptr(x), N.ptr(weight), N.ptr(bias), N.ptr(res), N.ptr(ones), B, H, W, Cp, Op, stride,
                                                 1 if upsample else 0, N.ptr(ws), ws.numel(), N.stream()), "conv2d_3x3_nhwc")
    return out if Op == O else out[:, :O]


_conv_pad = {}
_conv_ones = {}


def gelu(x: torch.Tensor, approximate: str = "none") -> torch.Tensor:
    assert x.is_contiguous() and x.dtype == BF16
    out = torch.empty_like(x)
    N.check(N.lib().fdm_gelu(N.ptr(out), N.ptr(x), x.numel(), 1 if approximate == "tanh" else 2, N.stream()), "gelu")
    return out


def quantize_rows(x0: torch.Tensor, x1: Optional[torch.Tensor], quant, act0: int = 0, act1: int = 0, out: Optional[Quantized] = None) -> Quantized:
    """Per-row quantisation of [x0 | x1] without materialising the concatenation (model/flux.py:69).
    act0 / act1 (N.EPI_GELU_TANH / N.EPI_GELU_ERF): that source holds PRE-activations -- bf16(gelu(x)) is applied on the way in, the
    arithmetic of the GEMM's GELU epilogue followed by this quantiser (bit-identical), at the price of an HBM-bound pass's idle VALU slots
    instead of an MFMA-bound kernel's drain (layer/activations.py:38-41 + layer/qlinear.py:62-66 of the reference)."""
    rows, w0 = x0.shape
    w1 = 0 if x1 is None else x1.shape[1]
    assert x0.stride(1) == 1 and (x1 is None or (x1.stride(1) == 1 and x1.shape[0] == rows))
    assert act0 in (0, N.EPI_GELU_TANH, N.EPI_GELU_ERF) and act1 in (0, N.EPI_GELU_TANH, N.EPI_GELU_ERF)
    # out: rows of a larger Quantized (the batch elements of a CFG batch quantised one by one into ONE GEMM operand)
    qz = _alloc_q(rows, w0 + w1, quant, x0.device) if out is None else out
    assert qz.q.shape == (rows, w0 + w1) and qz.q.is_contiguous() and qz.scale.is_contiguous()
    N.check(N.lib().fdm_quant_rowwise_cat2_act(N.ptr(qz.q), N.ptr(qz.scale), N.ptr(qz.azp), QUANT_MODE[quant], N.ptr(x0), w0,
                                               x0.stride(0), act0, N.ptr(x1), w1, 0 if x1 is None else x1.stride(0), act1, rows,
                                               N.stream()), "quant_rowwise")
    return qz


def qk_norm_rope(qkv: torch.Tensor, wq: torch.Tensor, wk: torch.Tensor, rope: Optional[torch.Tensor], q_out, k_out, v_out,
                 rows_per_batch: int, seq_total: int, seq_offset: int, heads: int, head_dim: int, eps: float = 1e-6):
    """RMS-norm(q,k) -> RoPE -> scatter into the joint [text ; image] q/k/v buffers [B*seq_total, heads*hd]."""
    rows = qkv.shape[0]
    assert qkv.is_contiguous() and qkv.shape[1] == 3 * heads * head_dim
    N.check(N.lib().fdm_qk_norm_rope(N.ptr(q_out), N.ptr(k_out), N.ptr(v_out), N.ptr(qkv), N.ptr(wq), N.ptr(wk), N.ptr(rope), rows,
                                     rows_per_batch, seq_total, seq_offset, heads, head_dim, eps, N.stream()), "qk_norm_rope")


def qk_norm_rope_inplace(q_rows: torch.Tensor, k_rows: torch.Tensor, wq: torch.Tensor, wk: torch.Tensor, rope: Optional[torch.Tensor],
                         rows: int, rows_per_batch: int, batch_row_stride: int, pos_offset: int, heads: int, head_dim: int,
                         eps: float = 1e-6):
    """RMS-norm(q,k) -> RoPE in place.  `q_rows` / `k_rows` are views of the same allocation starting at the first row
    to rewrite (equal row strides): planes of a plane-split projection or thirds of an interleaved qkv buffer."""
    assert q_rows.stride(-1) == 1 and k_rows.stride(-1) == 1 and q_rows.stride(-2) == k_rows.stride(-2)
    k_off = (k_rows.data_ptr() - q_rows.data_ptr()) // q_rows.element_size()
    N.check(N.lib().fdm_qk_norm_rope_inplace(N.ptr(q_rows), N.ptr(wq), N.ptr(wk), N.ptr(rope), rows, rows_per_batch, batch_row_stride,
                                             q_rows.stride(-2), k_off, pos_offset, heads, head_dim, eps, N.stream()), "qk_norm_rope_inplace")


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, batch: int, heads: int, head_dim: int, scale: float,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Dense attention on [B*S, H*D] buffers (same kernel as operators_set.scaled_dot_product_attention).
    out: a [B*S, H*D] view whose rows may be strided (the left columns of the [attention | mlp] operand of FLUX's single-block proj_out: no torch.cat)."""
    S = q.shape[0] // batch
    Sk = k.shape[0] // batch
    if out is None:
        out = torch.empty((q.shape[0], heads * head_dim), device=q.device, dtype=q.dtype)  # q/k/v may be strided views
    assert out.shape == (q.shape[0], heads * head_dim) and out.stride(1) == 1 and out.dtype == q.dtype
    ws = N.attn_workspace(q.device, head_dim)
    from fastdm_amd.kernel.utils import attention_pv_fp8
    if head_dim == 128 and attention_pv_fp8():       # opt-in quantised attention (kernel/utils.py); its own timing span
        mode = attention_pv_fp8()
        vq = N.attn_pvfp8_workspace(q.device, batch, Sk, heads, head_dim, mode)
        fn = N.lib().fdm_attn_fwd_fp8 if mode == 2 else N.lib().fdm_attn_fwd_pvfp8
        with span("attn_fwd_fp8" if mode == 2 else "attn_fwd_pvfp8", 4.0 * batch * S * Sk * heads * head_dim):
            N.check(fn(N.ptr(out), N.ptr(q), N.ptr(k), N.ptr(v), batch, S, Sk, heads, heads, head_dim,
                                               q.stride(0), k.stride(0), v.stride(0), out.stride(0), float(scale), N.ptr(ws), ws.numel(),
                                               N.ptr(vq), vq.numel(), N.stream()),
                    "attention (fp8 P.V)")
        return out
    with span("attn_fwd_bf16" if head_dim == 128 else "attn_fwd_bf16_hd64", 4.0 * batch * S * Sk * heads * head_dim):      # (two kernels: bench.py HIP_KERNEL)
        N.check(N.lib().fdm_attn_fwd(N.ptr(out), N.ptr(q), N.ptr(k), N.ptr(v), batch, S, Sk, heads, heads, head_dim,
                                     q.stride(0), k.stride(0), v.stride(0), out.stride(0), float(scale), None, 0, 0, N.ptr(ws), ws.numel(),
                                     N.stream()),
                "attention")
    return out


def absdiff_sums(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """[sum|a-b|, sum|b|] in fp32 on the device (TeaCache metric, caching/xcaching.py:214)."""
    assert a.is_contiguous() and b.is_contiguous() and a.shape == b.shape
    assert a.dtype == BF16 and b.dtype == BF16, "fdm_absdiff_sums reads its inputs as bf16"
    out = torch.zeros(2, device=a.device, dtype=torch.float32)
    N.check(N.lib().fdm_absdiff_sums(N.ptr(out), N.ptr(a), N.ptr(b), a.numel(), N.stream()), "absdiff_sums")
    return out


def synth_fill(out: torch.Tensor, seed: int, kind: int, scale: float = 0.0, offset: float = 0.0):
    assert out.dtype == BF16 and out.is_contiguous()
    N.check(N.lib().fdm_synth_fill_bf16(N.ptr(out), out.numel(), seed, kind, scale, offset, N.stream()), "synth_fill")
    return out


def clock_probe(device: torch.device, ms: float = 40.0, n_blocks: int = 256) -> dict:
    """Shader clock (MHz) the chip sustains under an fp8-MFMA + LDS loop (fdm_clock_probe): a short calibration launch sizes the
    loop for about `ms` milliseconds, a second launch is the measurement; clock of a workgroup = delta s_memtime / delta
    s_memrealtime x 100 MHz.  Returns the median / min / max over the workgroups.  A measurement aid for bench.py's
    `device_state`, never part of a step."""
    with torch.cuda.device(device):
        out = torch.zeros(2 * n_blocks, dtype=torch.int64, device=device)
        sink = torch.zeros(256, dtype=torch.float32, device=device)

        def run(iters):
            N.check(N.lib().fdm_clock_probe(N.ptr(out), N.ptr(sink), n_blocks, int(iters), N.stream()), "clock_probe")
            torch.cuda.synchronize()
            return out.cpu().view(n_blocks, 2).double()

        cal = run(20000)
        ticks = float(cal[:, 1].median())                        # 100 MHz ticks for 20000 rounds
        iters = max(1000, int(20000 * (ms * 1e5) / max(ticks, 1.0)))
        r = run(iters)
        clk = (r[:, 0] / r[:, 1].clamp(min=1.0) * 100.0).sort().values
        return {"median": round(float(clk[n_blocks // 2]), 1), "min": round(float(clk[0]), 1), "max": round(float(clk[-1]), 1),
                "probe_ms": round(float(r[:, 1].median()) / 1e5, 2), "rounds": iters, "workgroups": n_blocks}
