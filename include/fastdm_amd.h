/*
 * fastdm_amd C ABI  --  MI355X (gfx950) kernels for the FLUX / DiT denoise-step hot path.
 *
 * This header is the drop-in boundary that replaces the reference's native module
 * `fastdm.cuda_ops` (pybind exports: /root/reference/csrc/torch_bindings.cpp:191-201,
 * declarations: /root/reference/csrc/include/ops.h:9-33).  Every entry point takes raw device
 * pointers, sizes/strides and a stream, allocates nothing, never synchronises, and returns
 * 0 (FDM_OK) or a negative error code (no exceptions cross the ABI).  `stream` is a
 * hipStream_t passed as void* (NULL = the default stream).  Activations are bf16 unless
 * stated; "fp8" is OCP e4m3fn (torch.float8_e4m3fn).
 */
#ifndef FASTDM_AMD_H
#define FASTDM_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* fdm_stream_t;

enum {
  FDM_OK = 0,
  FDM_ERR_INVALID = -1,      /* bad pointer / size / alignment                       */
  FDM_ERR_UNSUPPORTED = -2,  /* shape or mode this build has no kernel for            */
  FDM_ERR_LAUNCH = -3        /* hipGetLastError() reported a launch failure           */
};

/* GEMM epilogue selector (applied after scale/bias, before the final store). */
enum {
  FDM_EPI_NONE = 0,
  FDM_EPI_GELU_TANH = 1,     /* bf16(gelu_tanh(bf16(y)))  -- layer/activations.py:38-41 */
  FDM_EPI_GELU_ERF = 2,      /* bf16(gelu_erf(bf16(y)))   -- model/flux.py:61            */
  FDM_EPI_GATE_RESIDUAL = 3, /* bf16(res + bf16(gate[n] * bf16(y))) -- model/flux.py:153-154 */
  FDM_EPI_GATE_RESIDUAL_F32 = 4 /* bf16(res + gate_f32[n] * bf16(y)), fp32 gate, ONE rounding -- model/wan.py:102,113 */
};

int fdm_version(void);
const char* fdm_error_string(int code);
/* HIP runtime message behind the most recent FDM_ERR_LAUNCH returned to the CALLING THREAD (the library keeps no other
 * mutable state: per-device constants are cached per device id, and no environment variable changes what a kernel does). */
const char* fdm_last_hip_error_string(void);

/* Bumped whenever an entry point changes its argument list: a binding built against another revision of this header must
 * refuse to run instead of passing arguments at the wrong positions. */
#define FDM_ABI_VERSION 20
int fdm_abi_version(void);

/* Provenance the binary carries itself (no stamp file beside it): the content digest (64 hex digits) of every source the
 * library was built from -- the .hip / .h / .sh / .inc / .py files of csrc + this header, the manifest rule of csrc/build.sh -- and the
 * extra hipcc flags of an A/B build ("" for the product library; build.sh refuses flags in fastdm_amd/_lib/).  The single
 * build identity plays the role of the reference's one-arch guard (/root/reference/setup.py:99-111).  The same two strings sit
 * in the file as "@(#)fdm_source_digest=..." / "@(#)fdm_build_flags=..." so a loader can check them before dlopen. */
const char* fdm_source_digest(void);
const char* fdm_build_flags(void);

/* Measurement aid (bench.py `device_state`; no reference counterpart, no product kernel carries a stamp): n_blocks workgroups
 * (one per CU) each run `iters` rounds of 4 LDS reads + 16 fp8 16x16x128 MFMAs on random finite data and write
 * out_pairs[2 b] = delta s_memtime (shader cycles), out_pairs[2 b + 1] = delta s_memrealtime (100 MHz ticks) of block b, both
 * uint64: clock = cycles / ticks x 100 MHz, median over blocks.  `sink`: 256 floats, never written in practice. */
int fdm_clock_probe(void* out_pairs, void* sink, int n_blocks, int iters, fdm_stream_t stream);

/* ---- replaces cuda_ops.rms_norm_ (ops.h:17-20; csrc/elmwise_ops.cu:56-84,433-449) -------------
 * out[r,:] = bf16(bf16(x[r,:] * rsqrt(mean(x^2)+eps)) * w[:]);  rows x hidden, contiguous.  Any hidden >= 1: widths that
 * are multiples of 8 on 16-byte aligned buffers take the vector kernels, everything else a scalar kernel (same arithmetic). */
int fdm_rms_norm(void* out, const void* in, const void* weight, float eps,
                 int64_t rows, int hidden, fdm_stream_t stream);
/* Same arithmetic on rows that are slices of wider matrices (Wan: q / k columns of the fused qkv projection,
 * layer/transformer.py:478-485): row strides in elements, multiples of 8.                         */
int fdm_rms_norm_strided(void* out, const void* in, const void* weight, float eps, int64_t rows, int hidden,
                         int64_t in_row_stride, int64_t out_row_stride, fdm_stream_t stream);
/* The same with the GPT-J rotary embedding of fdm_rope_inplace (is_neox = 0, rot_dim == head_size, positions row % seq_len) applied to
 * the normalised row before it is written: Wan's across-head q / k norm + rotary_emb_ (layer/transformer.py:478-499) in one pass.
 * Bit-identical to fdm_rms_norm_strided followed by fdm_rope_inplace.  hidden % head_size == 0, head_size % 8 == 0.                  */
int fdm_rms_norm_rope_strided(void* out, const void* in, const void* weight, float eps, int64_t rows, int hidden,
                              int64_t in_row_stride, int64_t out_row_stride, const void* cos_sin_cache, int64_t seq_len,
                              int head_size, fdm_stream_t stream);

/* ---- replaces cuda_ops.rotary_emb_ (ops.h:22-33; elmwise_ops.cu:86-168,451-522) ---------------
 * In place on q [tokens, q_heads*head_size] and k [tokens, k_heads*head_size] (row strides in
 * elements).  Position of token t is t % seq_len (the reference passes arange(S).repeat(B)).
 * cos_sin_cache [>=seq_len, rot_dim] bf16: first half cos, second half sin.                   */
int fdm_rope_inplace(void* q, void* k, const void* cos_sin_cache,
                     int64_t tokens, int64_t seq_len, int q_heads, int k_heads, int head_size,
                     int rot_dim, int64_t q_row_stride, int64_t k_row_stride, int is_neox,
                     fdm_stream_t stream);

/* ---- replaces cuda_ops.fp8_quant_ (ops.h:14-16; elmwise_ops.cu:250-306,524-547) ---------------
 * Per-row dynamic quantisation, torch-backend semantics (kernel/torch/quantize.py:58-67):
 * scale = max(absmax, bf16(1e-12))/448;  q = e4m3fn(clamp(x/scale, +-448)).                     */
int fdm_quant_fp8_rowwise(void* q_out, float* scale_out, const void* in,
                          int64_t rows, int hidden, int64_t in_row_stride, fdm_stream_t stream);

/* ---- replaces cuda_ops.int8_quant_ (ops.h:11-13; elmwise_ops.cu:308-398) ----------------------
 * azp_out == NULL: symmetric (scale = absmax/127).  Otherwise asymmetric, torch-backend
 * semantics (kernel/torch/quantize.py:21-43): scale=(max-min)/255, zp=-128-round(min/scale),
 * q = clamp(round(x/scale + zp), -128, 127).                                                    */
int fdm_quant_int8_rowwise(void* q_out, float* scale_out, int32_t* azp_out, const void* in,
                           int64_t rows, int hidden, int64_t in_row_stride, fdm_stream_t stream);

/* ---- gelu_and_mul (kernel/operators_set.py:54-67; kernel/triton/gelumul.py:7-68) --------------
 * out[r, :d] = bf16(x[r, :d] * bf16(gelu_erf(x[r, d:2d])))                                      */
int fdm_gelu_and_mul(void* out, const void* in, int64_t rows, int d, fdm_stream_t stream);

/* ---- replaces cuda_ops.fp8_scaled_mm_ (torch_bindings.cpp:24-84; csrc/gemm/hopper_w8a8_fp8.cu) -
 * out[M,N] (bf16, row stride ldc) = epi( acc_f32(A[M,K] fp8 row-major lda  x  B[K,N] fp8 column-
 * major ldb) * sa[M] * sb[N] + bias[N] ).  K % 16 == 0, N % 8 == 0 (the check in fill_problem; 16-byte output rows).  bias may be NULL.
 * epi_gate / epi_res [M,N] (row stride ldc; may alias out) are only read for
 * FDM_EPI_GATE_RESIDUAL; gate is indexed as gate[(m / epi_rows_per_batch) * epi_gate_stride + n]. */
int fdm_gemm_w8a8_fp8(void* out, const void* a, const void* b, const float* scale_a,
                      const float* scale_b, const void* bias,
                      int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                      int epilogue, const void* epi_gate, int64_t epi_gate_stride,
                      const void* epi_res, int64_t epi_rows_per_batch, fdm_stream_t stream);

/* Two independent fp8 GEMMs in ONE persistent launch (e.g. the image- and text-stream projections of a
 * double block: same op, different weights).  The small problem's tiles fill the last partial round of the
 * large one instead of running at < 10 % occupancy on their own.  Fields as in fdm_gemm_w8a8_fp8.        */
typedef struct fdm_gemm_fp8_desc {
  void* out;
  const void* a;
  const void* b;
  const float* scale_a;
  const float* scale_b;
  const void* bias;
  int64_t M, N, K, lda, ldb, ldc;
  int64_t epilogue;
  const void* epi_gate;
  int64_t epi_gate_stride;
  const void* epi_res;
  int64_t epi_rows_per_batch;
  /* Plane-split output (0 = off): output column n is written to out + (n / out_plane_cols) * out_plane_stride
   * + row*ldc + n % out_plane_cols, so a fused q|k|v projection (layer/transformer.py:264-266 `chunk(3)`) lands in
   * three compact [rows, out_plane_cols] planes; ldc is then the plane's row stride.  Not with residual epilogues. */
  int64_t out_plane_cols;
  int64_t out_plane_stride;
} fdm_gemm_fp8_desc;
int fdm_gemm_w8a8_fp8_x2(const fdm_gemm_fp8_desc* d0, const fdm_gemm_fp8_desc* d1, fdm_stream_t stream);

/* ---- replaces cuda_ops.int8_scaled_mm_ (torch_bindings.cpp:86-158; hopper_w8a8_int8.cu;
 *      epilogue include/cutlass_extensions/epilogue/scaled_mm_epilogues_c3x.hpp:130-197) ---------
 * out = bf16( bf16( f32(acc_i32 - azp[M]*azp_adj[N]) * (sa[M]*sb[N]) ) + bias[N] )
 * (torch-backend rounding order, kernel/torch/matrixmul.py:67-74).  azp/azp_adj may both be NULL
 * (symmetric).                                                                                   */
int fdm_gemm_w8a8_int8(void* out, const void* a, const void* b, const float* scale_a,
                       const float* scale_b, const int32_t* azp_adj, const int32_t* azp,
                       const void* bias, int64_t M, int64_t N, int64_t K,
                       int64_t lda, int64_t ldb, int64_t ldc,
                       int epilogue, const void* epi_gate, int64_t epi_gate_stride,
                       const void* epi_res, int64_t epi_rows_per_batch, fdm_stream_t stream);

/* Two independent int8 problems in one launch (as fdm_gemm_w8a8_fp8_x2).  Fields as in fdm_gemm_w8a8_int8.   */
typedef struct fdm_gemm_int8_desc {
  void* out;
  const void* a;
  const void* b;
  const float* scale_a;
  const float* scale_b;
  const int32_t* azp_adj;
  const int32_t* azp;
  const void* bias;
  int64_t M, N, K, lda, ldb, ldc;
  int64_t epilogue;
  const void* epi_gate;
  int64_t epi_gate_stride;
  const void* epi_res;
  int64_t epi_rows_per_batch;
  int64_t out_plane_cols;   /* as in fdm_gemm_fp8_desc */
  int64_t out_plane_stride;
} fdm_gemm_int8_desc;
int fdm_gemm_w8a8_int8_x2(const fdm_gemm_int8_desc* d0, const fdm_gemm_int8_desc* d1, fdm_stream_t stream);

/* The same one or two int8 problems (d1 may be NULL) with a split-K tail: when the last round of 256 x 256 output tiles
 * (one per CU at a time) would fill at most half of the device, those tiles are cut into up to 8 K ranges that run side by
 * side, write int32 partial accumulators into `workspace` and are merged -- sum of the parts, then the usual epilogue -- by a
 * second launch on the same stream.  int32 sums are exact: the output is BIT-IDENTICAL to fdm_gemm_w8a8_int8[_x2]
 * whatever the split (no such entry point exists for fp8, whose fp32 accumulation order would change).
 * workspace: 16-byte aligned device memory of fdm_gemm_int8_workspace_bytes() (64 MiB; smaller is allowed and limits the
 * split; NULL = never split), owned by the caller, reusable by consecutive calls on ONE stream.                            */
int64_t fdm_gemm_int8_workspace_bytes(void);
int fdm_gemm_w8a8_int8_x2_ws(const fdm_gemm_int8_desc* d0, const fdm_gemm_int8_desc* d1, void* workspace,
                             int64_t workspace_bytes, fdm_stream_t stream);

/* Diagnostic: which kernel the LAST GEMM entry point called on this host thread launched (the dispatch policy of gemm.hip picks
 * per launch): 0 fdm::gemm_kernel (8 waves), 1 fdm::gemm_w4_kernel (4 waves), 2 fdm::gemm_p4_kernel (persistent, hand-written
 * assembly), 3 fdm::gemv_bf16_kernel, 4 fdm::gemm_kernel<.., 2> (half tiles), -1 nothing launched yet.  bench.py names the kernel of a timing span with it; the parity
 * tests use it to prove which kernel they compared.  No reference counterpart.                                                  */
#define FDM_GEMM_KERNEL_W8 0
#define FDM_GEMM_KERNEL_W4 1
#define FDM_GEMM_KERNEL_P4 2
#define FDM_GEMM_KERNEL_GEMV 3
#define FDM_GEMM_KERNEL_W8_HALF 4    /* fdm::gemm_kernel<.., 2>: the 8-wave kernel's wave code on 256 x 128 half tiles (ABI 19) */
#define FDM_GEMM_KERNEL_W4_224 5     /* fdm::gemm_w4_kernel<.., 7>: the 4-wave kernel on 224-row tiles (ABI 20) */
int fdm_gemm_last_kernel(void);

/* A/B and parity-test control of the GEMM dispatch policy (process-wide, atomic; default = auto everywhere).  The library reads no environment
 * variable: fastdm_amd/_native.py translates FDM_GEMM_KERNEL / FDM_GEMM_P4 / FDM_GEMM_HALF_TILES into one call of this when it loads the
 * library; a host that binds the C ABI directly never needs it.  No reference counterpart.
 *   kernel     -1 auto | 0 the 8-wave kernel for every launch | 1 the 4-wave kernel wherever it runs | 2 auto + every epilogue on the persistent kernel
 *   p4         -1 default (plain / tanh-GELU launches) | 0 never | 1 default | 2 every epilogue it implements      (ignored when kernel is 0 / 1)
 *   half_tiles -1 the cost model | 0 never | 1 always (8-wave kernel only)                                                               */
int fdm_set_gemm_dispatch(int kernel, int p4, int half_tiles);
/* Tile height of the 4-wave kernel (ABI 20): -1 the policy (default: 224-row tiles where rounds x rows promises >= 10 % -- a launch whose 256-row tile
 * count sits just above a whole number of rounds, 408 tiles on 256 CUs -- and never in place of the int8 split-K tail) | 256 | 224.  Both forms run the
 * same MFMAs in the same K order per output element: results are bit-identical.  Measured: 9-14 % per launch back to back, neutral inside
 * the power-limited FLUX step, -2 % on the SD3.5-large step.  _native.py translates FDM_GEMM_TILE_ROWS.  No reference counterpart.                    */
int fdm_set_gemm_tile_rows(int rows);

/* ---- 3 x 3 convolution, NHWC bf16, as an implicit GEMM on the 8-wave tile kernel (ABI 19) ----------------------------------------
 * The UNet's convolutions (/root/reference/fastdm/layer/resnetblock.py:29-47 conv1 / conv2, fastdm/layer/unetblock.py:203-218 downsample
 * (stride 2), :232-240 upsample (nearest 2x, then the convolution), F.conv2d there = cuDNN) with fp32 accumulation over all 9 * c_in products
 * and ONE rounding: out[b, y, x, :] = bf16(sum_{dy, dx, c} x[b, y * stride + dy - 1, x * stride + dx - 1, c] * w[:, dy, dx, c] + bias)
 * with zero padding 1; upsample2x: x is read through a nearest-neighbour 2x upsampling (stride 1 only), so the upsampled tensor never exists.
 *   x    [batch, h_in, w_in, c_in]   NHWC bf16 (a channels_last torch tensor's memory), c_in % 32 == 0
 *   w    [c_out, 3, 3, c_in]         bf16 (the checkpoint's OIHW weight permuted once at load), c_out % 8 == 0
 *   out  [batch, h_out, w_out, c_out], h_out = (h_in - 1) / stride + 1 (2 * h_in with upsample2x)
 *   res / res_gate (both or neither): out = bf16(res + bf16(res_gate[c] * y)), res like out, res_gate bf16 [c_out] -- a ResNet block's skip
 *   connection with a vector of ones (x + h, resnetblock.py:47), rounded like the two torch ops.
 *   workspace (may be NULL: never split): 16-byte aligned scratch, fdm_gemm_int8_workspace_bytes() suffices; a convolution of few pixels and a
 *   long K (at most half a round of 256 x 256 tiles) is cut into K ranges whose fp32 partial sums go through it and are merged, in a fixed
 *   order, by a second launch.  Owned by the caller, reusable by consecutive calls on ONE stream.
 * FDM_ERR_UNSUPPORTED for other strides / channel counts (the caller keeps F.conv2d for conv_in's 4 channels).                          */
int fdm_conv2d_3x3_nhwc_bf16(void* out, const void* x, const void* w, const void* bias, const void* res, const void* res_gate,
                             int64_t batch, int64_t h_in, int64_t w_in, int64_t c_in, int64_t c_out, int stride, int upsample2x,
                             void* workspace, int64_t workspace_bytes, fdm_stream_t stream);

/* ---- bf16 GEMM (QLinear's un-quantised branch: torch.addmm, layer/qlinear.py:76) --------------
 * out[M,N] = bf16(acc_f32(A[M,K] bf16 row-major x B[K,N] bf16 column-major) + bias[N])          */
int fdm_gemm_bf16(void* out, const void* a, const void* b, const void* bias,
                  int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc,
                  int epilogue, const void* epi_gate, int64_t epi_gate_stride,
                  const void* epi_res, int64_t epi_rows_per_batch, fdm_stream_t stream);

/* Two independent bf16 GEMMs in ONE launch (ABI 20; as fdm_gemm_w8a8_fp8_x2: the image- and text-stream projection of a double block, or one problem
 * with an empty second one), with the plane-split output for fused q|k|v projections.  Fields as in fdm_gemm_bf16 / fdm_gemm_fp8_desc.  Until round 6
 * the unquantised model ran the text stream's launches on their own (M = 512: 24 tiles on 256 CUs, 0.07 of peak: 9 of the 54 ms of GEMM time of a
 * FLUX bf16 1024 x 1024 step) and a fused q|k|v projection as one launch per plane.  Same kernels, same arithmetic as fdm_gemm_bf16.            */
typedef struct fdm_gemm_bf16_desc {
  void* out;
  const void* a;
  const void* b;
  const void* bias;
  int64_t M, N, K, lda, ldb, ldc;
  int64_t epilogue;
  const void* epi_gate;
  int64_t epi_gate_stride;
  const void* epi_res;
  int64_t epi_rows_per_batch;
  int64_t out_plane_cols;   /* as in fdm_gemm_fp8_desc */
  int64_t out_plane_stride;
} fdm_gemm_bf16_desc;
int fdm_gemm_bf16_x2(const fdm_gemm_bf16_desc* d0, const fdm_gemm_bf16_desc* d1, fdm_stream_t stream);

/* The same linear for M <= 8 rows (AdaLN modulation / embedder projections: one weight-streaming GEMV launch, the kernel
 * fdm_gemm_bf16 itself picks for such shapes), with an optional second addend applied AFTER the output rounding:
 * out[m,n] = bf16(acc + bias[n]); where post_add != NULL and post_add[n] != 0: out[m,n] = bf16(out[m,n] + post_add[n]).
 * With post_add = 1 on the `scale` columns of a concatenated AdaLN projection the output holds the reference's
 * `1 + scale` tensor (layer/normalization.py:197,233: bf16 add) and fdm_ln_modulate takes it with scale_has_one = 1.
 * FDM_ERR_UNSUPPORTED when the shape is not GEMV-shaped (M > 8, K > 16384 or not a multiple of 8, unaligned): use fdm_gemm_bf16. */
int fdm_linear_skinny_bf16(void* out, const void* a, const void* b, const void* bias, const void* post_add,
                           int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, fdm_stream_t stream);

/* ---- scaled_dot_product_attention / sparse_... (kernel/operators_set.py:154-208) --------------
 * q [B,Sq,Hq*D], k/v [B,Sk,Hkv*D], o [B,Sq,Hq*D] bf16, token strides in elements (batch stride =
 * S * token stride).  Non-causal softmax(q k^T * scale) v with fp32 softmax state.  D in {64,128}.
 * block_mask (may be NULL): int8 [B,Hq,ceil(Sq/block_q),ceil(Sk/block_k)], 1 = compute
 * (kernel/cuda/attention.py:77-147); requires block_q % 128 == 0 and block_k % 64 == 0.
 * o must be 16-byte aligned with o_tok_stride % 8 == 0.                                            */
int fdm_attn_fwd(void* o, const void* q, const void* k, const void* v,
                 int64_t batch, int64_t sq, int64_t sk, int q_heads, int kv_heads, int head_dim,
                 int64_t q_tok_stride, int64_t k_tok_stride, int64_t v_tok_stride,
                 int64_t o_tok_stride, float scale,
                 const int8_t* block_mask, int block_q, int block_k,
                 void* workspace, int64_t workspace_bytes, fdm_stream_t stream);
/* The same with causal masking (`is_causal=True` of scaled_dot_product_attention, kernel/operators_set.py:154-179; the reference's torch backend
 * hands the flag to F.scaled_dot_product_attention, kernel/torch/attention.py:34-43 -- no model of the reference sets it): key j contributes to
 * query row i iff j <= i (torch's top-left alignment when Sq != Sk).  bf16 arithmetic of fdm_attn_fwd; D in {64,128}; no block mask (ABI 20).   */
int fdm_attn_fwd_causal(void* o, const void* q, const void* k, const void* v,
                        int64_t batch, int64_t sq, int64_t sk, int q_heads, int kv_heads, int head_dim,
                        int64_t q_tok_stride, int64_t k_tok_stride, int64_t v_tok_stride,
                        int64_t o_tok_stride, float scale, fdm_stream_t stream);
/* Optional scratch for fdm_attn_fwd (NULL = never split; a non-NULL workspace that is not 16-byte aligned is
 * FDM_ERR_INVALID, never a silent fall-back): when the last round of equal-sized q-block work
 * items would fill less than half the chip, those items are split along KV and merged (split-KV tail
 * balancing).  Returns the number of bytes that is always sufficient for the given head_dim.                 */
int64_t fdm_attn_fwd_workspace_bytes(int head_dim);

/* OPT-IN quantised attention: the same dense softmax(q k^T * scale) v with the P.V product on the fp8 matrix cores
 * (v_mfma_f32_32x32x64_f8f6f4) -- the class of the reference's preferred CUDA path, SageAttention (int8 QK^T / fp8 P.V:
 * /root/reference/fastdm/kernel/cuda/attention.py:20,35-75), which its published numbers were taken with.  Never the default:
 * fdm_attn_fwd is the reference TORCH backend's bf16 arithmetic and stays the headline.  What is quantised: P = exp2(s - ref) to
 * e4m3 (row reference kept so that 8 <= P_max < 2^7.7); V to e4m3 per (batch, kv head, channel): centred on the channel's mean
 * c over the keys, scale = max|v - c| / 448 (a pre-pass of this call writes the image into vq_workspace; c is added
 * back in the epilogue -- sum_j p_j (v_j - c) / l + c = sum_j p_j v_j / l because the weights sum to 1 -- so a common offset of V costs
 * no precision: the "smoothing" of SageAttention 2); scale * log2(e) is folded into Q with one more bf16 rounding.  QK^T and the softmax
 * state are those of fdm_attn_fwd; the row sums are taken from the ROUNDED weights (on the matrix core), so numerator and denominator see
 * the same P and uniform attention is exact.  Error against an
 * fp64 softmax on N(0, 1) inputs: mean 5e-4 / max 3e-3 at S = 8704, mean 1.5e-3 / max 1e-2 at S = 1000 (the reference's own
 * tolerance for its kernels: atol 1.8e-2, tests/test_attention.py:66-100); a row dominated by ONE key inherits that key's V rounding
 * (up to 6 % of |v - c|).  head_dim 128, no block mask (FDM_ERR_UNSUPPORTED otherwise).
 * vq_workspace: 16-byte aligned, >= fdm_attn_pvfp8_workspace_bytes(batch, sk, kv_heads, head_dim) bytes, owned by the caller,
 * rewritten by every call (reusable by consecutive calls on ONE stream).  workspace: as fdm_attn_fwd.                            */
int64_t fdm_attn_pvfp8_workspace_bytes(int64_t batch, int64_t sk, int kv_heads, int head_dim);
int fdm_attn_fwd_pvfp8(void* o, const void* q, const void* k, const void* v,
                       int64_t batch, int64_t sq, int64_t sk, int q_heads, int kv_heads, int head_dim,
                       int64_t q_tok_stride, int64_t k_tok_stride, int64_t v_tok_stride,
                       int64_t o_tok_stride, float scale,
                       void* workspace, int64_t workspace_bytes, void* vq_workspace, int64_t vq_workspace_bytes,
                       fdm_stream_t stream);

/* The same opt-in with QK^T on the fp8 matrix cores as well (SageAttention quantises Q and K too -- to int8; here e4m3):
 * v_mfma_scale_f32_32x32x64_f8f6f4, K as an e4m3 image -- centred per channel on its mean over the keys (q.(k_j - c) = q.k_j - q.c
 * shifts every score of a row by the same amount: the softmax does not change) and scaled by one power of two per (batch, kv head) (a
 * pre-pass of this call) --, Q
 * converted per 64-row wave with scale * log2(e) and its own power of two folded in; both powers of two are undone exactly by the
 * MFMA's E8M0 block scales, so the softmax sees scores in the same units as fdm_attn_fwd.  8 + 8 (+ 2 row-sum) MFMAs per 64-key tile
 * instead of 32 + 32.  Error against an fp64 softmax on N(0, 1) inputs: mean 7.4e-4 / max 3.7e-3 at S = 8704, mean 2.2e-3 / max 1.2e-2
 * at S = 1000.  Same restrictions as fdm_attn_fwd_pvfp8; fp8_workspace >= fdm_attn_fp8_workspace_bytes(...) (V image + K image). */
int64_t fdm_attn_fp8_workspace_bytes(int64_t batch, int64_t sk, int kv_heads, int head_dim);
int fdm_attn_fwd_fp8(void* o, const void* q, const void* k, const void* v,
                     int64_t batch, int64_t sq, int64_t sk, int q_heads, int kv_heads, int head_dim,
                     int64_t q_tok_stride, int64_t k_tok_stride, int64_t v_tok_stride,
                     int64_t o_tok_stride, float scale,
                     void* workspace, int64_t workspace_bytes, void* fp8_workspace, int64_t fp8_workspace_bytes,
                     fdm_stream_t stream);

/* The block-sparse form of the two quantised modes (mode 1: fp8 P.V, mode 2: fp8 QK^T + fp8 P.V) -- the class of the reference's sparse
 * CUDA path, SpargeAttn = block-sparse SageAttention (/root/reference/fastdm/kernel/cuda/attention.py:77-147).  block_mask as in
 * fdm_attn_fwd (int8 [B,Hq,ceil(Sq/block_q),ceil(Sk/block_k)], block_q % 128 == 0, block_k % 64 == 0, at most 4096 64-key tiles);
 * fp8_workspace as fdm_attn_fwd_pvfp8 (mode 1) / fdm_attn_fwd_fp8 (mode 2).  Opt-in like the dense forms.                             */
int fdm_attn_fwd_quant_masked(void* o, const void* q, const void* k, const void* v,
                              int64_t batch, int64_t sq, int64_t sk, int q_heads, int kv_heads, int head_dim,
                              int64_t q_tok_stride, int64_t k_tok_stride, int64_t v_tok_stride,
                              int64_t o_tok_stride, float scale,
                              const int8_t* block_mask, int block_q, int block_k, int mode,
                              void* fp8_workspace, int64_t fp8_workspace_bytes, fdm_stream_t stream);

/* ================= fused kernels used by the FLUX block wiring (no reference export) ========== */

/* AdaLN modulate (layer/normalization.py:196-198,232-234):
 * y = bf16(bf16(LN(x) * bf16(1+scale[b,:])) + shift[b,:]), LN without affine, fp32 statistics.
 * Any of the three outputs may be NULL: out_bf16 [rows,hidden]; out_q + out_scale = the per-row
 * fp8 (quant_mode 1) or asymmetric int8 (quant_mode 2, also out_azp) quantisation of y.
 * Optional fused residual: if res != NULL, x := bf16(res + bf16(gate[b,:] * x)) first and, when
 * x_out != NULL, that updated x is written there (model/flux.py:153-156).
 * scale_has_one = 1: `scale` already holds bf16(1 + scale) (fdm_linear_skinny_bf16's post_add wrote it so), the kernel
 * multiplies by it as is -- the same value the reference's `1 + scale` tensor holds, computed once per column
 * instead of once per element.                                                                    */
/* Wan's fp32 modulation (model/wan.py:99,109-111,362): y = bf16(LN_f32(x) * (1 + scale_f32[b,:]) + shift_f32[b,:]) with ONE
 * rounding (FP32LayerNorm, then fp32 multiply-add, then type_as).  scale/shift are fp32 [B, hidden] with
 * mod_batch_stride in floats (0 broadcasts one row: an affine FP32LayerNorm is scale = weight - 1, shift = bias).
 * Outputs as fdm_ln_modulate.                                                                                        */
int fdm_ln_modulate_f32(void* out_bf16, void* out_q, float* out_scale, int32_t* out_azp, int quant_mode,
                        const void* x, const float* scale, const float* shift, int64_t mod_batch_stride,
                        int64_t rows, int hidden, int64_t rows_per_batch, float eps, fdm_stream_t stream);

int fdm_ln_modulate(void* out_bf16, void* out_q, float* out_scale, int32_t* out_azp, int quant_mode,
                    const void* x, const void* scale, const void* shift, int64_t mod_batch_stride,
                    const void* res, const void* gate, void* x_out,
                    int64_t rows, int hidden, int64_t rows_per_batch, float eps, int scale_has_one, fdm_stream_t stream);

/* out = bf16(res + bf16(gate[b,:] * y))  (model/flux.py:153-154,161-163,70-72). */
int fdm_gate_residual(void* out, const void* res, const void* y, const void* gate,
                      int64_t gate_batch_stride, int64_t rows, int hidden, int64_t rows_per_batch,
                      fdm_stream_t stream);

/* GroupNorm of an NCHW-contiguous bf16 tensor [batch, channels, hw] with the element-wise neighbours of its call sites fused in -- the
 * F.group_norm / F.silu pairs of the UNet (/root/reference/fastdm/layer/resnetblock.py:29-47, fastdm/model/sdxl.py:469) and the
 * spatial transformer's F.group_norm (fastdm/layer/unetblock.py:315).  Rounding order of the separate torch calls:
 *   v   = addend ? bf16(x + addend[b, c]) : x              (addend: bf16 [batch, channels], the time-embedding offset; may be NULL)
 *   y   = bf16(a * v + (beta[c] - a * mean)),  a = rstd * gamma[c]      (fp32; biased variance over the group's channels/groups * hw values)
 *   out = act == 1 ? bf16(y / (1 + exp(-y))) : y           (act 1 = SiLU)
 * hw % 8 == 0 (FDM_ERR_UNSUPPORTED otherwise); x / out 16-byte aligned; out must not alias x (a slice is read by two kernels).
 * workspace: 16-byte aligned, >= fdm_group_norm_workspace_bytes(batch, channels, hw, groups) bytes (per-slice statistics), rewritten by
 * every call.                                                                                                                         */
int64_t fdm_group_norm_workspace_bytes(int64_t batch, int64_t channels, int64_t hw, int groups);
int fdm_group_norm(void* out, const void* x, const void* gamma, const void* beta, const void* addend,
                   int64_t batch, int64_t channels, int64_t hw, int groups, float eps, int act,
                   void* workspace, int64_t workspace_bytes, fdm_stream_t stream);

/* The same GroupNorm on an NHWC tensor [batch, hw, channels] (ABI 19; the layout of fdm_conv2d_3x3_nhwc_bf16): same rounding order, same
 * arguments.  channels % 8 == 0, an even number of channels per group, groups <= 64, channels <= 4096 (FDM_ERR_UNSUPPORTED otherwise); gamma /
 * beta / addend 16-byte aligned.  Statistics: fp32 partial sums of <= 64 pixels per channel, merged in double in a fixed order (deterministic). */
int64_t fdm_group_norm_nhwc_workspace_bytes(int64_t batch, int64_t channels, int64_t hw, int groups);
int fdm_group_norm_nhwc(void* out, const void* x, const void* gamma, const void* beta, const void* addend,
                        int64_t batch, int64_t channels, int64_t hw, int groups, float eps, int act,
                        void* workspace, int64_t workspace_bytes, fdm_stream_t stream);

/* gelu_and_mul followed by the per-token quantisation of its output (fdm_gelu_and_mul + fdm_quant_*_rowwise in one pass: the GEGLU
 * feed-forward of the UNet, /root/reference/fastdm/layer/unetblock.py:213-224): in [rows, 2d] bf16 -> y = bf16(in[:, :d] * bf16(gelu_erf(in[:, d:])))
 * -> q_out [rows, d] + scale_out [rows] (+ azp_out [rows] for quant_mode 2); quant_mode 1 = fp8, 2 = int8-asym.  Bit-identical to the two calls.
 * d % 8 == 0, d <= 16384.                                                                                                              */
int fdm_gelu_and_mul_quant(void* q_out, float* scale_out, int32_t* azp_out, int quant_mode, const void* in,
                           int64_t rows, int d, fdm_stream_t stream);

/* Element-wise GELU on bf16 (mode 1 = tanh approximation, 2 = erf), F.gelu semantics. */
int fdm_gelu(void* out, const void* in, int64_t n, int mode, fdm_stream_t stream);

/* Per-row quantisation of the virtual concatenation [src0 (w0 cols) | src1 (w1 cols)] (kills the
 * torch.cat in model/flux.py:69).  src1 may be NULL (w1 = 0).  quant_mode 1 = fp8, 2 = int8-asym. */
int fdm_quant_rowwise_cat2(void* q_out, float* scale_out, int32_t* azp_out, int quant_mode,
                           const void* src0, int w0, int64_t stride0,
                           const void* src1, int w1, int64_t stride1,
                           int64_t rows, fdm_stream_t stream);

/* The same with an activation applied to a source's values on the way in (ABI 19): act 0 = none, 1 = tanh-GELU, 2 = erf-GELU:
 *   x := bf16(gelu(x))   -- /root/reference/fastdm/layer/activations.py:38-41 (GELU = proj, then F.gelu), fastdm/model/flux.py:61 (act_mlp)
 * followed by the reference's per-token quantisation of the next QLinear's input (fastdm/layer/qlinear.py:62-66).  A W8A8 feed-forward
 * runs its first GEMM with FDM_EPI_NONE and defers the activation to this pass: the rounding points are those of the GEMM's GELU epilogue
 * followed by fdm_quant_rowwise_cat2 (bit-identical), but the activation's VALU work hides under an HBM-bound kernel instead of extending
 * an MFMA-bound one.  Instantiated: a GELU on the only source, or on the second of two ([attention | gelu(mlp)], FLUX single block);
 * quant_mode 1 / 2; w0 + w1 <= 16384.  Other combinations: FDM_ERR_UNSUPPORTED.                                                         */
int fdm_quant_rowwise_cat2_act(void* q_out, float* scale_out, int32_t* azp_out, int quant_mode,
                               const void* src0, int w0, int64_t stride0, int act0,
                               const void* src1, int w1, int64_t stride1, int act1,
                               int64_t rows, fdm_stream_t stream);

/* q/k RMS-norm + RoPE + scatter into the joint [text ; image] q/k/v buffers
 * (layer/transformer.py:264-298).  qkv [rows, 3*heads*hd] is one stream's fused projection;
 * rows r of batch b land at token (b*seq_total + seq_offset + r%rows_per_batch) of q_out/k_out/
 * v_out [B*seq_total, heads*hd].  cos_sin_cache row = position within seq_total.
 * cos_sin_cache == NULL skips the rotation.                                                      */
int fdm_qk_norm_rope(void* q_out, void* k_out, void* v_out, const void* qkv,
                     const void* wq, const void* wk, const void* cos_sin_cache,
                     int64_t rows, int64_t rows_per_batch, int64_t seq_total, int64_t seq_offset,
                     int heads, int head_dim, float eps, fdm_stream_t stream);

/* The same q/k RMS-norm + RoPE (layer/transformer.py:264-298) applied IN PLACE, for when the fused-qkv GEMMs of the
 * text and image stream have already written q, k, v where fdm_attn_fwd reads them -- three compact planes
 * (fdm_gemm_*_desc.out_plane_cols: row_stride = heads*hd, k_offset = plane stride) or one interleaved
 * [rows, 3*heads*hd] buffer (row_stride = 3*heads*hd, k_offset = heads*hd).  Rewrites the q and k heads of
 * `rows` = B*rows_per_batch rows (batch b's rows start at row b*batch_row_stride of `qkv`, rope position =
 * pos_offset + r); v is not touched.  Replaces the reference's chunk + per-tensor norm + cat with a third of the
 * scatter kernel's HBM traffic.                                                                      */
int fdm_qk_norm_rope_inplace(void* qkv, const void* wq, const void* wk, const void* cos_sin_cache,
                             int64_t rows, int64_t rows_per_batch, int64_t batch_row_stride,
                             int64_t row_stride, int64_t k_offset, int64_t pos_offset,
                             int heads, int head_dim, float eps, fdm_stream_t stream);

/* TeaCache metric (caching/xcaching.py:214): out[0] = sum|a-b|, out[1] = sum|b| (fp32 atomics;
 * caller zeroes out[0..1]).                                                                      */
int fdm_absdiff_sums(float* out2, const void* a, const void* b, int64_t n, fdm_stream_t stream);

/* Counter-based synthetic data (bit-identical twin of oracle/synth.py): kind 0 = offset +
 * N(0,std^2) via Irwin-Hall(4); kind 1 = U[0,1).  Output bf16.                                   */
int fdm_synth_fill_bf16(void* out, int64_t n, uint32_t seed, int kind, float scale, float offset,
                        fdm_stream_t stream);

/* Debug/verification helpers (used by tests to pin hardware conversions). */
int fdm_debug_cvt_fp8(void* out_u8, const float* in, int64_t n, int use_sw, fdm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FASTDM_AMD_H */
