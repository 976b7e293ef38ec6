// HBM-bound row / element-wise kernels of the denoise step (bf16 I/O, fp32 math).
// Every kernel streams 16-byte vectors per lane; rows are owned by one wave (hidden <= 4096) or
// one 256-thread block, and a row's values stay in registers between the statistics pass and the
// output pass so each byte is read from HBM once.
#include "common.h"

namespace fdm {

#ifdef FDM_DEBUG_BUILD
bool g_fp8_sw_cvt = false;
#endif

constexpr int MAXV = 8;  // 16-byte vectors held per thread

// Row loops come in two shapes.  NIT = 0: any width, MAXV guarded iterations (`v < nvec` around every load: hipcc keeps each load in
// its own exec-masked block, reuses one register quad and waits for every load before it issues the next -- one 1 KB request in
// flight per wave, a third of what the HBM latency x bandwidth product needs at full occupancy).  NIT = ceil(nvec / THREADS) known
// at compile time: NIT unguarded loads issued back to back (the last one clamps its vector index into the row, so it is always a
// legal address), and only the last iteration's arithmetic / stores are predicated.  The host picks NIT; NIT = 0 stays as the
// fallback for the software-fp8 debug path.
template <int NIT>
struct RowIt {
  static constexpr int N = NIT ? NIT : MAXV;
  // may iteration i of this thread touch vector v?
  static __device__ __forceinline__ bool live(int i, int v, int nvec) { return (NIT && i < NIT - 1) || v < nvec; }
  // vector index to LOAD for iteration i (always inside the row when NIT != 0)
  static __device__ __forceinline__ int ld(int i, int v, int nvec) { return (NIT && i == NIT - 1) ? min(v, nvec - 1) : v; }
};

// Load this thread's slice of a row: vectors t, t+THREADS, ...  (nvec = hidden/8)
template <int THREADS, int NIT = 0>
__device__ __forceinline__ void load_row(Vec8 (&x)[MAXV], const bf16_t* row, int nvec) {
  using It = RowIt<NIT>;
#pragma unroll
  for (int i = 0; i < It::N; ++i) {
    int v = threadIdx.x % THREADS + i * THREADS;
    if (NIT || v < nvec) x[i] = load8(row + (size_t)It::ld(i, v, nvec) * 8);
  }
}

// ------------------------------------------------------------------------------------ rms_norm
// Small hidden (<= 512): LPR lanes per row, several rows per wave.
template <int LPR>
__global__ void __launch_bounds__(256) rms_norm_small_kernel(bf16_t* __restrict__ out, const bf16_t* __restrict__ in,
                                                             const bf16_t* __restrict__ w, float eps, int64_t rows,
                                                             int hidden) {
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t row = wave * RPW + lane / LPR;
  const int sub = lane % LPR;
  const bool ok = row < rows;
  Vec8 x;
  if (ok) x = load8(in + row * hidden + sub * 8);
  float ss = 0.f;
  if (ok) {
#pragma unroll
    for (int j = 0; j < 8; ++j) ss += x.v[j] * x.v[j];
  }
#pragma unroll
  for (int off = LPR / 2; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);
  if (!ok) return;
  const float rs = rsqrtf(ss / (float)hidden + eps);
  Vec8 wv = load8(w + sub * 8), o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o.v[j] = rbf(x.v[j] * rs) * wv.v[j];
  store8(out + row * hidden + sub * 8, o);
}

// ROPE: the GPT-J rotation of rope_kernel<false> (rot_dim == head_size) applied to the normalised row before it is written -- Wan's
// across-head q / k norm followed by rotary_pos_embedding (layer/transformer.py:478-499) in one pass over the row.  The normalised value
// is rounded to bf16 first, as the store of the separate norm kernel does: same bits as the two kernels.
struct RopeArgs {
  const bf16_t* cache;   // [seq_len rows of rot_dim]: cos | sin halves
  int64_t seq_len;
  int head_size;
};
__device__ __forceinline__ Vec8 rope_gptj8(const Vec8& x, const bf16_t* crow, int half, int c) {
  Vec8 o;
  const i32x2 cr = *reinterpret_cast<const i32x2*>(crow + c * 4);
  const i32x2 sr = *reinterpret_cast<const i32x2*>(crow + half + c * 4);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t cw = (uint32_t)cr[j >> 1], sw = (uint32_t)sr[j >> 1];
    const float cs = (j & 1) ? __builtin_bit_cast(float, cw & 0xffff0000u) : __builtin_bit_cast(float, cw << 16);
    const float sn = (j & 1) ? __builtin_bit_cast(float, sw & 0xffff0000u) : __builtin_bit_cast(float, sw << 16);
    const float a = x.v[2 * j], b = x.v[2 * j + 1];
    o.v[2 * j] = rbf(a * cs) - rbf(b * sn);
    o.v[2 * j + 1] = rbf(b * cs) + rbf(a * sn);
  }
  return o;
}

template <int THREADS, int NIT = 0, bool ROPE = false>
__global__ void __launch_bounds__(THREADS) rms_norm_row_kernel(bf16_t* __restrict__ out, const bf16_t* __restrict__ in,
                                                               const bf16_t* __restrict__ w, float eps, int hidden,
                                                               int64_t in_stride, int64_t out_stride, const RopeArgs ra) {
  using It = RowIt<NIT>;
  __shared__ float smem[THREADS / 64];
  const int64_t row = blockIdx.x;
  const int nvec = hidden >> 3;
  const bf16_t* src = in + row * in_stride;
  Vec8 x[MAXV];
  float ss = 0.f;
  const bool cached = NIT || nvec <= MAXV * THREADS;
  i32x4 wr[MAXV];
  if (cached) {
    load_row<THREADS, NIT>(x, src, nvec);
    if constexpr (NIT != 0) {   // the weight vectors travel under the reduction
#pragma unroll
      for (int i = 0; i < It::N; ++i) wr[i] = *reinterpret_cast<const i32x4*>(w + (size_t)It::ld(i, threadIdx.x + i * THREADS, nvec) * 8);
    }
#pragma unroll
    for (int i = 0; i < It::N; ++i)
      if (It::live(i, (int)threadIdx.x + i * THREADS, nvec))
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += x[i].v[j] * x[i].v[j];
  } else {
    for (int v = threadIdx.x; v < nvec; v += THREADS) {
      Vec8 t = load8(src + (size_t)v * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += t.v[j] * t.v[j];
    }
  }
  ss = block_reduce<THREADS>(ss, OpSum(), smem);
  const float rs = rsqrtf(ss / (float)hidden + eps);
  bf16_t* dst = out + row * out_stride;
  if (cached) {
#pragma unroll
    for (int i = 0; i < It::N; ++i) {
      int v = threadIdx.x + i * THREADS;
      if (It::live(i, v, nvec)) {
        Vec8 wv, o;
        if constexpr (NIT != 0) wv = unpack8(wr[i]);
        else wv = load8(w + (size_t)v * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) o.v[j] = rbf(x[i].v[j] * rs) * wv.v[j];
        if constexpr (ROPE) {
#pragma unroll
          for (int j = 0; j < 8; ++j) o.v[j] = rbf(o.v[j]);
          o = rope_gptj8(o, ra.cache + (row % ra.seq_len) * ra.head_size, ra.head_size >> 1, v % (ra.head_size >> 3));
        }
        store8(dst + (size_t)v * 8, o);
      }
    }
  } else {
    for (int v = threadIdx.x; v < nvec; v += THREADS) {
      Vec8 t = load8(src + (size_t)v * 8), wv = load8(w + (size_t)v * 8), o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o.v[j] = rbf(t.v[j] * rs) * wv.v[j];
      if constexpr (ROPE) {
#pragma unroll
        for (int j = 0; j < 8; ++j) o.v[j] = rbf(o.v[j]);
        o = rope_gptj8(o, ra.cache + (row % ra.seq_len) * ra.head_size, ra.head_size >> 1, v % (ra.head_size >> 3));
      }
      store8(dst + (size_t)v * 8, o);
    }
  }
}

// host side: iteration count -> template instance (widths beyond MAXV iterations use the generic two-pass form)
template <int THREADS, bool ROPE = false>
static void launch_rms_row(bf16_t* o, const bf16_t* i, const bf16_t* w, float eps, int64_t rows, int hidden, int64_t is, int64_t os,
                           hipStream_t s, const RopeArgs ra = RopeArgs{nullptr, 1, 8}) {
#define FDM_RMS(NITV) rms_norm_row_kernel<THREADS, NITV, ROPE><<<(int)rows, THREADS, 0, s>>>(o, i, w, eps, hidden, is, os, ra)
  switch (((hidden >> 3) + THREADS - 1) / THREADS) {
    case 1: FDM_RMS(1); break; case 2: FDM_RMS(2); break; case 3: FDM_RMS(3); break; case 4: FDM_RMS(4); break;
    case 5: FDM_RMS(5); break; case 6: FDM_RMS(6); break; case 7: FDM_RMS(7); break; case 8: FDM_RMS(8); break;
    default: FDM_RMS(0); break;
  }
#undef FDM_RMS
}

// Any width / alignment (the reference kernel takes any last_dim, csrc/elmwise_ops.cu:56-84): one wave per row, scalar bf16
// accesses.  Same arithmetic as the vector kernels: fp32 sum of squares, rsqrt(mean + eps), bf16(x * rs) * w -> bf16.
__global__ void __launch_bounds__(256) rms_norm_scalar_kernel(bf16_t* __restrict__ out, const bf16_t* __restrict__ in,
                                                              const bf16_t* __restrict__ w, float eps, int64_t rows, int hidden) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const bf16_t* src = in + row * hidden;
  float ss = 0.f;
  for (int c = lane; c < hidden; c += 64) {
    const float x = bf2f(src[c]);
    ss += x * x;
  }
  ss = wave_reduce(ss, OpSum());
  const float rs = rsqrtf(ss / (float)hidden + eps);
  bf16_t* dst = out + row * hidden;
  for (int c = lane; c < hidden; c += 64) dst[c] = f2bf(rbf(bf2f(src[c]) * rs) * bf2f(w[c]));
}

// ------------------------------------------------------------------------------------ rope
// One thread per 8 consecutive elements (4 pairs GPT-J / 8 lows or highs NeoX) of q or k.
template <bool NEOX>
__global__ void __launch_bounds__(256) rope_kernel(bf16_t* __restrict__ q, bf16_t* __restrict__ k,
                                                   const bf16_t* __restrict__ cache, int64_t tokens, int64_t seq_len,
                                                   int q_heads, int k_heads, int head_size, int rot_dim,
                                                   int64_t q_stride, int64_t k_stride) {
  const int half = rot_dim >> 1;
  // work items per token: NEOX: (heads * half/8) items each touching 8 lows + 8 highs; GPT-J: heads*rot_dim/8
  const int per_head = NEOX ? (half >> 3) : (rot_dim >> 3);
  const int per_tok = (q_heads + k_heads) * per_head;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= tokens * per_tok) return;
  const int64_t tok = gid / per_tok;
  int it = (int)(gid - tok * per_tok);
  bf16_t* base;
  if (it < q_heads * per_head) {
    base = q + tok * q_stride;
  } else {
    it -= q_heads * per_head;
    base = k + tok * k_stride;
  }
  const int head = it / per_head, c = it % per_head;
  const bf16_t* crow = cache + (tok % seq_len) * rot_dim;
  bf16_t* p = base + (size_t)head * head_size;
  if constexpr (NEOX) {
    Vec8 lo = load8(p + c * 8), hi = load8(p + half + c * 8);
    Vec8 cs = load8(crow + c * 8), sn = load8(crow + half + c * 8), olo, ohi;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      olo.v[j] = rbf(lo.v[j] * cs.v[j]) - rbf(hi.v[j] * sn.v[j]);
      ohi.v[j] = rbf(hi.v[j] * cs.v[j]) + rbf(lo.v[j] * sn.v[j]);
    }
    store8(p + c * 8, olo);
    store8(p + half + c * 8, ohi);
  } else {
    Vec8 x = load8(p + c * 8), o;
    const i32x2 cr = *reinterpret_cast<const i32x2*>(crow + c * 4);
    const i32x2 sr = *reinterpret_cast<const i32x2*>(crow + half + c * 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t cw = (uint32_t)cr[j >> 1], sw = (uint32_t)sr[j >> 1];
      float cs = (j & 1) ? __builtin_bit_cast(float, cw & 0xffff0000u) : __builtin_bit_cast(float, cw << 16);
      float sn = (j & 1) ? __builtin_bit_cast(float, sw & 0xffff0000u) : __builtin_bit_cast(float, sw << 16);
      float a = x.v[2 * j], b = x.v[2 * j + 1];
      o.v[2 * j] = rbf(a * cs) - rbf(b * sn);
      o.v[2 * j + 1] = rbf(b * cs) + rbf(a * sn);
    }
    store8(p + c * 8, o);
  }
}

// ------------------------------------------------------------------------------------ division by a row-uniform scale
// The quantising kernels are VALU-bound, not HBM-bound (a wave64 fp32 instruction occupies a SIMD for 4 clocks; 11 instructions per
// IEEE division x 48 elements per lane was 40 % of ln_modulate's instruction stream).  hipcc expands a / b into
//   b' = div_scale(b), a' = div_scale(a); y0 = rcp(b'); e = fma(-b', y0, 1); y = fma(e, y0, y0);
//   q0 = a' * y; r0 = fma(-b', q0, a'); q1 = fma(r0, y, q0); r1 = fma(-b', q1, a'); q = div_fmas(r1, y, q1); div_fixup(q, b, a)
// The reciprocal chain depends only on b, and div_scale / div_fmas / div_fixup are identities unless an operand or the quotient is
// zero, denormal, infinite, NaN or within 2^-103 of underflow.  RowDiv hoists y once per row and runs the SAME five-instruction
// residual chain per element, so the quotient is bit-identical to `a / b` whenever |a / b| >= 2^-43; below that (and for a = +-0)
// both paths round to a zero byte, and the sign of the quotient is copied from a (b > 0).  Rows whose scale lies outside
// [2^-60, 2^60] (or is NaN) take the compiler's division: `ok` is wave-uniform.
struct RowDiv {
  float b, y;
  bool ok;
  __device__ __forceinline__ explicit RowDiv(float scale) : b(scale) {
#ifdef FDM_ROWDIV_OFF
    ok = false;   // A/B builds (tools/micro/ew_bench.hip): the compiler's division everywhere
#else
    ok = scale >= 0x1p-60f && scale <= 0x1p60f;
#endif
    const float y0 = __builtin_amdgcn_rcpf(scale);
    const float e = __builtin_fmaf(-scale, y0, 1.0f);
    y = __builtin_fmaf(e, y0, y0);
  }
  __device__ __forceinline__ float fast(float a) const {
    const float q0 = a * y;
    const float r0 = __builtin_fmaf(-b, q0, a);
    const float q1 = __builtin_fmaf(r0, y, q0);
    const float r1 = __builtin_fmaf(-b, q1, a);
    return __builtin_copysignf(__builtin_fmaf(r1, y, q1), a);
  }
};

// eight cached values -> eight quantised bytes (FAST: RowDiv::fast, else the compiler's division)
template <int MODE, bool SW, bool FAST>
__device__ __forceinline__ i32x2 quant8(const Vec8& t, const RowDiv& dv, float zpf) {
  uint32_t w[2] = {0, 0};
#pragma unroll
  for (int j = 0; j < 8; j += 2) {
    const float qa = FAST ? dv.fast(t.v[j]) : t.v[j] / dv.b, qb = FAST ? dv.fast(t.v[j + 1]) : t.v[j + 1] / dv.b;
    uint32_t two;
    if constexpr (MODE == 1) {
      // FAST rows have a finite scale = RN(amax / 448) and |a| <= amax, so |a / scale| <= 448 (1 + 2^-23): the hardware conversion
      // rounds that to 448 by itself (e4m3 overflows from 464 up) and the reference's clamp is the identity
      if constexpr (FAST && !SW) two = cvt2_e4m3<SW>(qa, qb);
      else two = cvt2_e4m3<SW>(clamp448(qa), clamp448(qb));
    } else {
      float ra = __builtin_rintf(qa + zpf), rb = __builtin_rintf(qb + zpf);
      int ia = (int)fmaxf(-128.f, fminf(ra, 127.f)), ib = (int)fmaxf(-128.f, fminf(rb, 127.f));
      two = ((uint32_t)ia & 0xffu) | (((uint32_t)ib & 0xffu) << 8);
    }
    w[j >> 2] |= two << (8 * (j & 3));
  }
  i32x2 o;
  o[0] = (int)w[0];
  o[1] = (int)w[1];
  return o;
}

// ------------------------------------------------------------------------------------ row quant
// Shared body: given this thread's cached row slice x[], emit fp8 / int8 bytes + per-row params.
// MODE 1: fp8 (torch/quantize.py:58-67)   MODE 2: int8 asymmetric (:36-41)   MODE 3: int8 symmetric (:31-35)
template <int THREADS, int MODE, bool SW, int NIT = 0>
__device__ __forceinline__ void quant_row_cached(const Vec8 (&x)[MAXV], int nvec, uint8_t* qdst, float* scale_out,
                                                 int32_t* azp_out, int64_t row, float* smem) {
  using It = RowIt<NIT>;
  float amax = 0.f, mn = 3.0e38f, mx = -3.0e38f;
#pragma unroll
  for (int i = 0; i < It::N; ++i)
    if (It::live(i, (int)(threadIdx.x % THREADS) + i * THREADS, nvec)) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        amax = fmaxf(amax, fabsf(x[i].v[j]));
        mn = fminf(mn, x[i].v[j]);
        mx = fmaxf(mx, x[i].v[j]);
      }
    }
  float scale, zpf = 0.f;
  if constexpr (MODE == 1) {
    amax = block_reduce<THREADS>(amax, OpMax(), smem);
    amax = fmaxf(amax, __builtin_bit_cast(float, 0x2B8D0000u));  // bf16(1e-12): the clamp runs on the bf16 tensor
    scale = amax / 448.0f;
  } else if constexpr (MODE == 3) {
    amax = block_reduce<THREADS>(amax, OpMax(), smem);
    scale = amax / 127.0f;
  } else {
    mn = block_reduce<THREADS>(mn, OpMin(), smem);
    mx = block_reduce<THREADS>(mx, OpMax(), smem);
    scale = (mx - mn) / 255.0f;
    zpf = -128.0f - __builtin_rintf(mn / scale);
  }
  if (threadIdx.x % THREADS == 0) {
    scale_out[row] = scale;
    if constexpr (MODE == 2) azp_out[row] = (int32_t)zpf;
  }
  const RowDiv dv(scale);
  if (dv.ok) {
#pragma unroll
    for (int i = 0; i < It::N; ++i) {
      int v = (int)(threadIdx.x % THREADS) + i * THREADS;
      if (It::live(i, v, nvec)) *reinterpret_cast<i32x2*>(qdst + (size_t)v * 8) = quant8<MODE, SW, true>(x[i], dv, zpf);
    }
  } else {
#pragma unroll
    for (int i = 0; i < It::N; ++i) {
      int v = (int)(threadIdx.x % THREADS) + i * THREADS;
      if (It::live(i, v, nvec)) *reinterpret_cast<i32x2*>(qdst + (size_t)v * 8) = quant8<MODE, SW, false>(x[i], dv, zpf);
    }
  }
}

// Virtual concatenation of two sources so proj_out's K=15360 row never exists in bf16.
// ACT (round 6): an activation applied to a source's values on the way in -- bits 0-1 source 0, bits 2-3 source 1; 0 none, 1 tanh-GELU,
// 2 erf-GELU: x := bf16(gelu(x)), the arithmetic of the GEMM epilogues (common.h: fast_gelu_*), so a W8A8 feed-forward can run its first
// GEMM with the PLAIN epilogue and leave the activation to this HBM-bound pass.  Why: in the GEMMs (one wave per SIMD, MFMA-issue-bound) the
// GELU drain of a 256 x 256 tile is 9-20 k cycles that one K step's MFMAs cannot cover (FLUX: 323 vs 265 us per launch at N = 12288); here
// the same VALU work hides under the row's HBM / Infinity-Cache reads.  Same roundings in the same order: bit-identical to epilogue + quantiser.
template <int ACT>
__device__ __forceinline__ void act8(Vec8& x) {
  if constexpr (ACT == 1) {
#pragma unroll
    for (int j = 0; j < 8; ++j) x.v[j] = rbf(fast_gelu_tanh(x.v[j]));
  } else if constexpr (ACT == 2) {
#pragma unroll
    for (int j = 0; j < 8; ++j) x.v[j] = rbf(fast_gelu_erf(x.v[j]));
  }
}

template <int THREADS, int MODE, bool SW, int NIT = 0, int ACT = 0>
__global__ void __launch_bounds__(THREADS) quant_rowwise_kernel(uint8_t* __restrict__ q, float* __restrict__ scale,
                                                                int32_t* __restrict__ azp,
                                                                const bf16_t* __restrict__ s0, int w0, int64_t st0,
                                                                const bf16_t* __restrict__ s1, int w1, int64_t st1) {
  using It = RowIt<NIT>;
  __shared__ float smem[THREADS / 64 + 1];
  const int64_t row = blockIdx.x;
  const int nv0 = w0 >> 3, nvec = (w0 + w1) >> 3;
  Vec8 x[MAXV];
#pragma unroll
  for (int i = 0; i < It::N; ++i) {
    int v = threadIdx.x + i * THREADS;
    if (NIT || v < nvec) {
      v = It::ld(i, v, nvec);
      // one address, one load: a select between two loads would issue both
      const bf16_t* src = (v < nv0) ? s0 + row * st0 + (size_t)v * 8 : s1 + row * st1 + (size_t)(v - nv0) * 8;
      x[i] = load8(src);
    }
  }
  if constexpr (ACT != 0) {
    // after ALL loads are in flight (the activation of vector 0 runs while vector NIT - 1 is still on its way)
#pragma unroll
    for (int i = 0; i < It::N; ++i) {
      const int v = It::ld(i, (int)threadIdx.x + i * THREADS, nvec);
      if (NIT || (int)threadIdx.x + i * THREADS < nvec) {
        if (v < nv0) act8<(ACT & 3)>(x[i]);
        else act8<((ACT >> 2) & 3)>(x[i]);
      }
    }
  }
  quant_row_cached<THREADS, MODE, SW, NIT>(x, nvec, q + row * (int64_t)(w0 + w1), scale, azp, row, smem);
}

// Fallback for very wide rows (> MAXV*256*8 = 16384 elements): two passes, second read from L2.
template <int MODE, bool SW>
__global__ void __launch_bounds__(256) quant_rowwise_wide_kernel(uint8_t* __restrict__ q, float* __restrict__ scale,
                                                                 int32_t* __restrict__ azp, const bf16_t* __restrict__ in,
                                                                 int hidden, int64_t stride) {
  __shared__ float smem[5];
  const int64_t row = blockIdx.x;
  const bf16_t* src = in + row * stride;
  const int nvec = hidden >> 3;
  float amax = 0.f, mn = 3.0e38f, mx = -3.0e38f;
  for (int v = threadIdx.x; v < nvec; v += 256) {
    Vec8 t = load8(src + (size_t)v * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      amax = fmaxf(amax, fabsf(t.v[j]));
      mn = fminf(mn, t.v[j]);
      mx = fmaxf(mx, t.v[j]);
    }
  }
  float sc, zpf = 0.f;
  if constexpr (MODE == 1) {
    amax = fmaxf(block_reduce<256>(amax, OpMax(), smem), __builtin_bit_cast(float, 0x2B8D0000u));
    sc = amax / 448.0f;
  } else if constexpr (MODE == 3) {
    sc = block_reduce<256>(amax, OpMax(), smem) / 127.0f;
  } else {
    mn = block_reduce<256>(mn, OpMin(), smem);
    mx = block_reduce<256>(mx, OpMax(), smem);
    sc = (mx - mn) / 255.0f;
    zpf = -128.0f - __builtin_rintf(mn / sc);
  }
  if (threadIdx.x == 0) {
    scale[row] = sc;
    if constexpr (MODE == 2) azp[row] = (int32_t)zpf;
  }
  uint8_t* dst = q + row * (int64_t)hidden;
  const RowDiv dv(sc);
  if (dv.ok) {
    for (int v = threadIdx.x; v < nvec; v += 256)
      *reinterpret_cast<i32x2*>(dst + (size_t)v * 8) = quant8<MODE, SW, true>(load8(src + (size_t)v * 8), dv, zpf);
  } else {
    for (int v = threadIdx.x; v < nvec; v += 256)
      *reinterpret_cast<i32x2*>(dst + (size_t)v * 8) = quant8<MODE, SW, false>(load8(src + (size_t)v * 8), dv, zpf);
  }
}

// ------------------------------------------------------------------------------------ AdaLN modulate (+residual, +quant)
// y = bf16(bf16(LN(x) * bf16(1+scale)) + shift); optional x := bf16(res + bf16(gate*x)) first.
// S1: `scale` already holds bf16(1 + scale) (the producer of the modulation vectors added the 1: one add + one rounding per ELEMENT
// of every row become one per column of the step).
template <int THREADS, int QMODE, bool SW, int NIT = 0, bool S1 = false>
__global__ void __launch_bounds__(THREADS) ln_modulate_kernel(
    bf16_t* __restrict__ out_bf16, uint8_t* __restrict__ out_q, float* __restrict__ out_scale,
    int32_t* __restrict__ out_azp, const bf16_t* __restrict__ xin, const bf16_t* __restrict__ scale,
    const bf16_t* __restrict__ shift, int64_t mod_bstride, const bf16_t* __restrict__ res,
    const bf16_t* __restrict__ gate, bf16_t* __restrict__ x_out, int hidden, int64_t rows_per_batch, float eps) {
  using It = RowIt<NIT>;
  __shared__ float smem[THREADS / 64 + 1];
  const int64_t row = blockIdx.x;
  const int64_t b = row / rows_per_batch;
  const int nvec = hidden >> 3;
  Vec8 x[MAXV];
  load_row<THREADS, NIT>(x, xin + row * hidden, nvec);
  // the modulation vectors do not depend on the row statistics: request them now (16 B raw per lane and iteration), so they
  // travel under the two reductions instead of one L2 round trip per iteration after them
  i32x4 scr[MAXV], shr[MAXV];
#pragma unroll
  for (int i = 0; i < It::N; ++i) {
    const int v = threadIdx.x + i * THREADS;
    if (NIT || v < nvec) {
      const size_t off = b * mod_bstride + (size_t)It::ld(i, v, nvec) * 8;
      scr[i] = *reinterpret_cast<const i32x4*>(scale + off);
      shr[i] = *reinterpret_cast<const i32x4*>(shift + off);
    }
  }
  if (res != nullptr) {
    i32x4 rr[MAXV], gr[MAXV];
#pragma unroll
    for (int i = 0; i < It::N; ++i) {
      const int v = threadIdx.x + i * THREADS;
      if (NIT || v < nvec) {
        const int vl = It::ld(i, v, nvec);
        rr[i] = *reinterpret_cast<const i32x4*>(res + row * hidden + (size_t)vl * 8);
        gr[i] = *reinterpret_cast<const i32x4*>(gate + b * mod_bstride + (size_t)vl * 8);
      }
    }
#pragma unroll
    for (int i = 0; i < It::N; ++i) {
      const int v = threadIdx.x + i * THREADS;
      if (NIT || v < nvec) {
        const Vec8 r = unpack8(rr[i]), g = unpack8(gr[i]);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[i].v[j] = rbf(r.v[j] + rbf(g.v[j] * x[i].v[j]));
      }
    }
    if (x_out != nullptr) {
#pragma unroll
      for (int i = 0; i < It::N; ++i) {
        const int v = threadIdx.x + i * THREADS;
        if (It::live(i, v, nvec)) store8(x_out + row * hidden + (size_t)v * 8, x[i]);
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < It::N; ++i)
    if (It::live(i, (int)threadIdx.x + i * THREADS, nvec))
#pragma unroll
      for (int j = 0; j < 8; ++j) s += x[i].v[j];
  const float mean = block_reduce<THREADS>(s, OpSum(), smem) / (float)hidden;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < It::N; ++i)
    if (It::live(i, (int)threadIdx.x + i * THREADS, nvec))
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float d = x[i].v[j] - mean;
        ss += d * d;
      }
  const float var = block_reduce<THREADS>(ss, OpSum(), smem) / (float)hidden;
  const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int i = 0; i < It::N; ++i) {
    const int v = threadIdx.x + i * THREADS;
    if (NIT || v < nvec) {
      const Vec8 sc = unpack8(scr[i]), sh = unpack8(shr[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float n = rbf((x[i].v[j] - mean) * rstd);
        x[i].v[j] = rbf(rbf(n * (S1 ? sc.v[j] : rbf(1.0f + sc.v[j]))) + sh.v[j]);
      }
    }
  }
  if (out_bf16 != nullptr) {
#pragma unroll
    for (int i = 0; i < It::N; ++i) {
      const int v = threadIdx.x + i * THREADS;
      if (It::live(i, v, nvec)) store8(out_bf16 + row * hidden + (size_t)v * 8, x[i]);
    }
  }
  if constexpr (QMODE != 0) {
    quant_row_cached<THREADS, QMODE, SW, NIT>(x, nvec, out_q + row * (int64_t)hidden, out_scale, out_azp, row, smem);
  }
}

// fp32 modulation (Wan): y = bf16(((x - mean) * rstd) * (1 + scale_f32) + shift_f32), one rounding.
template <int THREADS, int QMODE, bool SW, int NIT = 0>
__global__ void __launch_bounds__(THREADS) ln_modulate_f32_kernel(
    bf16_t* __restrict__ out_bf16, uint8_t* __restrict__ out_q, float* __restrict__ out_scale, int32_t* __restrict__ out_azp,
    const bf16_t* __restrict__ xin, const float* __restrict__ scale, const float* __restrict__ shift, int64_t mod_bstride,
    int hidden, int64_t rows_per_batch, float eps) {
  using It = RowIt<NIT>;
  __shared__ float smem[THREADS / 64 + 1];
  const int64_t row = blockIdx.x;
  const int64_t b = row / rows_per_batch;
  const int nvec = hidden >> 3;
  Vec8 x[MAXV];
  load_row<THREADS, NIT>(x, xin + row * hidden, nvec);
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < It::N; ++i)
    if (It::live(i, (int)threadIdx.x + i * THREADS, nvec))
#pragma unroll
      for (int j = 0; j < 8; ++j) s += x[i].v[j];
  const float mean = block_reduce<THREADS>(s, OpSum(), smem) / (float)hidden;
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < It::N; ++i)
    if (It::live(i, (int)threadIdx.x + i * THREADS, nvec))
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float d = x[i].v[j] - mean;
        ss += d * d;
      }
  const float var = block_reduce<THREADS>(ss, OpSum(), smem) / (float)hidden;
  const float rstd = 1.0f / sqrtf(var + eps);
#pragma unroll
  for (int i = 0; i < It::N; ++i) {
    int v = threadIdx.x + i * THREADS;
    if (NIT || v < nvec) {
      const int vl = It::ld(i, v, nvec);
      const float* sc = scale + b * mod_bstride + (size_t)vl * 8;
      const float* sh = shift + b * mod_bstride + (size_t)vl * 8;
      const f32x4 sc0 = *reinterpret_cast<const f32x4*>(sc), sc1 = *reinterpret_cast<const f32x4*>(sc + 4);
      const f32x4 sh0 = *reinterpret_cast<const f32x4*>(sh), sh1 = *reinterpret_cast<const f32x4*>(sh + 4);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float n = (x[i].v[j] - mean) * rstd;
        const float scj = j < 4 ? sc0[j & 3] : sc1[j & 3], shj = j < 4 ? sh0[j & 3] : sh1[j & 3];
        x[i].v[j] = rbf(n * (1.0f + scj) + shj);
      }
    }
  }
  if (out_bf16 != nullptr) {   // stores after the arithmetic: a branch per iteration would pin every modulation load behind it
#pragma unroll
    for (int i = 0; i < It::N; ++i) {
      const int v = threadIdx.x + i * THREADS;
      if (It::live(i, v, nvec)) store8(out_bf16 + row * hidden + (size_t)v * 8, x[i]);
    }
  }
  if constexpr (QMODE != 0) {
    quant_row_cached<THREADS, QMODE, SW, NIT>(x, nvec, out_q + row * (int64_t)hidden, out_scale, out_azp, row, smem);
  }
}

// ------------------------------------------------------------------------------------ simple element-wise
__global__ void __launch_bounds__(256) gate_residual_kernel(bf16_t* __restrict__ out, const bf16_t* __restrict__ res,
                                                            const bf16_t* __restrict__ y, const bf16_t* __restrict__ gate,
                                                            int64_t gate_bstride, int64_t nvec_total, int nvec_row,
                                                            int64_t rows_per_batch) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec_total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / nvec_row;
    const int v = (int)(i - row * nvec_row);
    const int64_t b = row / rows_per_batch;
    Vec8 r = load8(res + i * 8), yy = load8(y + i * 8), g = load8(gate + b * gate_bstride + (size_t)v * 8), o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.v[j] = r.v[j] + rbf(g.v[j] * yy.v[j]);
    store8(out + i * 8, o);
  }
}

template <int MODE>
__global__ void __launch_bounds__(256) gelu_kernel(bf16_t* __restrict__ out, const bf16_t* __restrict__ in, int64_t nvec) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    Vec8 x = load8(in + i * 8), o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.v[j] = (MODE == 1) ? gelu_tanh_f(x.v[j]) : gelu_erf_f(x.v[j]);
    store8(out + i * 8, o);
  }
}

__global__ void __launch_bounds__(256) gelu_and_mul_kernel(bf16_t* __restrict__ out, const bf16_t* __restrict__ in,
                                                           int64_t rows, int d) {
  const int nv = d >> 3;
  const int64_t total = rows * nv;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / nv;
    const int v = (int)(i - row * nv);
    Vec8 a = load8(in + row * 2 * d + (size_t)v * 8), g = load8(in + row * 2 * d + d + (size_t)v * 8), o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o.v[j] = a.v[j] * rbf(gelu_erf_f(g.v[j]));
    store8(out + row * d + (size_t)v * 8, o);
  }
}

// GEGLU + per-token quantisation of its output in one pass (the UNet's feed-forward: proj1 -> a * gelu(g) -> quantise -> proj2): the row of
// bf16(a * bf16(gelu_erf(g))) stays in registers between the product and the quantiser, so the [rows, d] bf16 tensor never exists.
template <int MODE>
__global__ void __launch_bounds__(256) gelu_and_mul_quant_kernel(uint8_t* __restrict__ q, float* __restrict__ scale, int32_t* __restrict__ azp,
                                                                 const bf16_t* __restrict__ in, int d) {
  __shared__ float smem[256 / 64 + 1];
  const int64_t row = blockIdx.x;
  const int nvec = d >> 3;
  const bf16_t* src = in + row * 2 * (int64_t)d;
  Vec8 x[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int v = threadIdx.x + i * 256;
    if (v < nvec) {
      const Vec8 a = load8(src + (size_t)v * 8), g = load8(src + d + (size_t)v * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) x[i].v[j] = rbf(a.v[j] * rbf(gelu_erf_f(g.v[j])));
    }
  }
  quant_row_cached<256, MODE, false, 0>(x, nvec, q + row * (int64_t)d, scale, azp, row, smem);
}

// ------------------------------------------------------------------------------------ q/k norm + rope + scatter
// x: 8 consecutive elements of one head (lane `sub` of the head's HD/8 lanes); wv: the norm weight's 8 elements; cr / sr: this
// lane's 4 cos / 4 sin factors of the row's position (ignored without a cache)
template <int HD>
__device__ __forceinline__ Vec8 qk_norm_rope_apply_pre(const Vec8& x, float rs, const Vec8& wv, bool rope, const i32x2 cr, const i32x2 sr) {
  Vec8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o.v[j] = rbf(rbf(x.v[j] * rs) * wv.v[j]);
  if (rope) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint32_t cw = (uint32_t)cr[j >> 1], sw = (uint32_t)sr[j >> 1];
      float cs = (j & 1) ? __builtin_bit_cast(float, cw & 0xffff0000u) : __builtin_bit_cast(float, cw << 16);
      float sn = (j & 1) ? __builtin_bit_cast(float, sw & 0xffff0000u) : __builtin_bit_cast(float, sw << 16);
      float a = o.v[2 * j], bb = o.v[2 * j + 1];
      o.v[2 * j] = rbf(a * cs) - rbf(bb * sn);
      o.v[2 * j + 1] = rbf(bb * cs) + rbf(a * sn);
    }
  }
  return o;
}
template <int HD>
__device__ __forceinline__ Vec8 qk_norm_rope_apply(const Vec8& x, float rs, const bf16_t* __restrict__ w,
                                                 const bf16_t* __restrict__ cache, int64_t pos, int sub) {
  i32x2 cr = {0, 0}, sr = {0, 0};
  if (cache != nullptr) {
    const bf16_t* crow = cache + pos * HD;
    cr = *reinterpret_cast<const i32x2*>(crow + sub * 4);
    sr = *reinterpret_cast<const i32x2*>(crow + HD / 2 + sub * 4);
  }
  return qk_norm_rope_apply_pre<HD>(x, rs, load8(w + sub * 8), cache != nullptr, cr, sr);
}

// One wave handles one (token, head) pair-of-8-lanes... layout: 16 lanes x 8 elements = one 128-wide head
// (8 lanes for hd 64).  Each wave covers 64/LPH heads of q, then k; v is copied.
template <int HD>
__global__ void __launch_bounds__(256) qk_norm_rope_kernel(bf16_t* __restrict__ qo, bf16_t* __restrict__ ko,
                                                           bf16_t* __restrict__ vo, const bf16_t* __restrict__ qkv,
                                                           const bf16_t* __restrict__ wq, const bf16_t* __restrict__ wk,
                                                           const bf16_t* __restrict__ cache, int64_t rows,
                                                           int64_t rows_per_batch, int64_t seq_total, int64_t seq_offset,
                                                           int heads, float eps) {
  constexpr int LPH = HD / 8;  // lanes per head
  const int inner = heads * HD;
  const int items_per_row = 3 * heads;  // (q|k|v, head)
  const int64_t gid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LPH;
  const int sub = threadIdx.x % LPH;
  const bool ok = gid < rows * items_per_row;
  const int64_t row = ok ? gid / items_per_row : 0;
  const int it = ok ? (int)(gid - row * items_per_row) : 0;
  const int which = it / heads, head = it % heads;
  const int64_t b = row / rows_per_batch, r = row - b * rows_per_batch;
  const int64_t pos = seq_offset + r;
  const int64_t orow = b * seq_total + pos;
  Vec8 x;
  if (ok) x = load8(qkv + row * 3 * inner + (size_t)which * inner + head * HD + sub * 8);
  float ss = 0.f;
  if (ok && which < 2) {
#pragma unroll
    for (int j = 0; j < 8; ++j) ss += x.v[j] * x.v[j];
  }
#pragma unroll
  for (int off = LPH / 2; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);
  if (!ok) return;
  if (which == 2) {
    store8(vo + orow * inner + head * HD + sub * 8, x);
    return;
  }
  const float rs = rsqrtf(ss / (float)HD + eps);
  const Vec8 o = qk_norm_rope_apply<HD>(x, rs, which == 0 ? wq : wk, cache, pos, sub);
  store8((which == 0 ? qo : ko) + orow * inner + head * HD + sub * 8, o);
}

// In-place variant: the fused-qkv GEMMs have already written q, k and v where the attention kernel reads them (three
// compact planes via the GEMM's plane-split output, or one interleaved [rows, 3*inner] buffer); q and k heads are
// normalised + rotated where they lie and v is not touched.  4/3 x rows x inner x 2 B of HBM traffic instead of the
// scatter kernel's 4 x.  Row r's q head lives at qkv + r*row_stride, its k head k_offset elements further.  Each
// 8/16-lane group reads its head before it writes it, so aliasing is safe.
template <int HD>
__global__ void __launch_bounds__(256) qk_norm_rope_inplace_kernel(bf16_t* qkv, const bf16_t* __restrict__ wq,
                                                                   const bf16_t* __restrict__ wk,
                                                                   const bf16_t* __restrict__ cache, int64_t rows,
                                                                   int64_t rows_per_batch, int64_t row_stride_batch,
                                                                   int64_t row_stride, int64_t k_offset,
                                                                   int64_t pos_offset, int heads, float eps) {
  constexpr int LPH = HD / 8;
  const int items_per_row = 2 * heads;  // (q|k, head)
  const int64_t gid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / LPH;
  const int sub = threadIdx.x % LPH;
  const bool ok = gid < rows * items_per_row;
  const int64_t row = ok ? gid / items_per_row : 0;
  const int it = ok ? (int)(gid - row * items_per_row) : 0;
  const int which = it / heads, head = it % heads;
  const int64_t b = row / rows_per_batch, r = row - b * rows_per_batch;
  bf16_t* p = qkv + (b * row_stride_batch + r) * row_stride + (which ? k_offset : 0) + head * HD + sub * 8;
  Vec8 x;
  if (ok) x = load8(p);
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) ss += x.v[j] * x.v[j];
#pragma unroll
  for (int off = LPH / 2; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);
  if (!ok) return;
  const float rs = rsqrtf(ss / (float)HD + eps);
  store8(p, qk_norm_rope_apply<HD>(x, rs, which == 0 ? wq : wk, cache, pos_offset + r, sub));
}

// The same arithmetic with one block per row: the (q | k, head) items of a row are walked in NI = ceil(2 * heads / (256 / LPH))
// unrolled passes, so a thread has NI 16-byte loads in flight instead of one, the norm weights and the row's cos / sin factors
// are loaded once per thread instead of once per item, and the per-thread 64-bit divisions of the flat mapping (item -> row,
// head, batch: two thirds of that kernel's instruction stream) become one scalar division per block.
template <int HD, int NI>
__global__ void __launch_bounds__(256) qk_norm_rope_inplace_row_kernel(bf16_t* qkv, const bf16_t* __restrict__ wq,
                                                                       const bf16_t* __restrict__ wk,
                                                                       const bf16_t* __restrict__ cache, int64_t rows_per_batch,
                                                                       int64_t row_stride_batch, int64_t row_stride,
                                                                       int64_t k_offset, int64_t pos_offset, int heads, float eps) {
  constexpr int LPH = HD / 8, IPP = 256 / LPH;
  const int64_t row = blockIdx.x;
  const int64_t b = row / rows_per_batch, r = row - b * rows_per_batch;
  const int sub = threadIdx.x % LPH, it0 = threadIdx.x / LPH;
  const int items = 2 * heads;
  bf16_t* base = qkv + (b * row_stride_batch + r) * row_stride + sub * 8;
  i32x4 raw[NI];
  bf16_t* ptr[NI];
  bool isk[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int item = min(it0 + i * IPP, items - 1);   // past the end: re-read the last item (never stored)
    isk[i] = item >= heads;
    ptr[i] = base + (isk[i] ? k_offset + (int64_t)(item - heads) * HD : (int64_t)item * HD);
    raw[i] = *reinterpret_cast<const i32x4*>(ptr[i]);
  }
  const Vec8 wqv = load8(wq + sub * 8), wkv = load8(wk + sub * 8);
  i32x2 cr = {0, 0}, sr = {0, 0};
  if (cache != nullptr) {
    const bf16_t* crow = cache + (pos_offset + r) * HD;
    cr = *reinterpret_cast<const i32x2*>(crow + sub * 4);
    sr = *reinterpret_cast<const i32x2*>(crow + HD / 2 + sub * 4);
  }
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const Vec8 x = unpack8(raw[i]);
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) ss += x.v[j] * x.v[j];
#pragma unroll
    for (int off = LPH / 2; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);
    const float rs = rsqrtf(ss / (float)HD + eps);
    const Vec8 o = qk_norm_rope_apply_pre<HD>(x, rs, isk[i] ? wkv : wqv, cache != nullptr, cr, sr);
    if (it0 + i * IPP < items) store8(ptr[i], o);
  }
}

// ------------------------------------------------------------------------------------ TeaCache metric
__global__ void __launch_bounds__(256) absdiff_sums_kernel(float* __restrict__ out2, const bf16_t* __restrict__ a,
                                                           const bf16_t* __restrict__ b, int64_t nvec) {
  __shared__ float smem[5];
  float sd = 0.f, sb = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    Vec8 x = load8(a + i * 8), y = load8(b + i * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sd += fabsf(rbf(x.v[j] - y.v[j]));  // (a-b) is a bf16 tensor in the reference before .abs()
      sb += fabsf(y.v[j]);
    }
  }
  sd = block_reduce<256>(sd, OpSum(), smem);
  sb = block_reduce<256>(sb, OpSum(), smem);
  if (threadIdx.x == 0) {
    atomicAdd(out2, sd);
    atomicAdd(out2 + 1, sb);
  }
}

// ------------------------------------------------------------------------------------ synthetic data (twin of oracle/synth.py)
__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__global__ void __launch_bounds__(256) synth_fill_kernel(bf16_t* __restrict__ out, int64_t n, uint32_t seed, int kind,
                                                         float scale, float offset) {
  const uint64_t s0 = (uint64_t)seed * 0x9E3779B97F4A7C15ull;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t z = splitmix64(s0 + (uint64_t)i);
    float v;
    if (kind == 0) {
      int tot = (int)(z & 0xffff) + (int)((z >> 16) & 0xffff) + (int)((z >> 32) & 0xffff) + (int)(z >> 48);
      v = __fmul_rn((float)(tot - 131070), scale);
      if (offset != 0.0f) v = __fadd_rn(v, offset);
    } else {
      v = __fmul_rn((float)(uint32_t)(z >> 40), 5.9604644775390625e-08f);
    }
    out[i] = f2bf(v);
  }
}

template <bool SW>
__global__ void debug_cvt_fp8_kernel(uint8_t* out, const float* in, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (uint8_t)(cvt2_e4m3<SW>(clamp448(in[i]), 0.f) & 0xff);
}


// ------------------------------------------------------------------------------------ GroupNorm (+ addend in front, + SiLU behind), NCHW
// The UNet's F.group_norm / F.silu call sites (reference layer/resnetblock.py:29-47, layer/unetblock.py:315, model/sdxl.py:469):
//   v = bf16(x + addend[b, c])                       (optional: the time-embedding offset added in front of norm2)
//   y = bf16(a * v + b'),  a = rstd * gamma[c],  b' = beta[c] - a * mean      (torch's fused-parameter form; fp32, statistics over a group)
//   out = bf16(y / (1 + exp(-y)))                    (optional SiLU on the rounded y, as the separate F.silu call does)
// A group (b, g) is ONE contiguous run of L = C/G * HW elements.  With 2 x 32 groups per tensor a block per group would leave 3/4 of the
// chip idle, so a group is cut into P slices: the statistics kernel keeps its slice in registers (exact two-pass mean / M2 per slice, no
// E[x^2] - mean^2 cancellation) and writes (n, mean, M2); the apply kernel merges a group's P partials (Chan's formula, in double), re-reads
// its slice -- same block -> same XCD: the slice is still in that L2 -- and writes the output.  Each byte crosses HBM once per direction.
constexpr int GN_THREADS = 256;
template <int NV>      // 16-byte vectors per thread (slice <= NV * 256 vectors)
__global__ void __launch_bounds__(GN_THREADS) group_norm_stats_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ addend, float* __restrict__ part,
                                                                      int64_t group_elems, int hw, int hw_shift, int cpg, int channels, int groups, int P) {
  __shared__ float smem[GN_THREADS / 64 + 1];
  const int slice = blockIdx.x, bg = blockIdx.y;      // bg = b * groups + g
  const int b = bg / groups, g = bg - b * groups;
  const int64_t nvec = group_elems >> 3;
  const int v0 = (int)(nvec * slice / P), v1 = (int)(nvec * (slice + 1) / P);      // (group_elems < 2^31: checked by the host)
  const bf16_t* src = x + (int64_t)bg * group_elems;
  auto chan_of = [&](int v) { return hw_shift >= 0 ? (int)(((uint32_t)v << 3) >> hw_shift) : (int)(((uint32_t)v << 3) / (uint32_t)hw); };
  i32x4 raw[NV];      // the slice as loaded (4 registers per vector); with an addend the sum is formed again in the second pass
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int v = v0 + (int)threadIdx.x + i * GN_THREADS;
    if (v < v1) raw[i] = *reinterpret_cast<const i32x4*>(src + (int64_t)v * 8);
  }
  auto value = [&](int i, int v) {
    Vec8 xv = unpack8(raw[i]);
    if (addend != nullptr) {      // hw % 8 == 0: a vector lies inside one channel
      const float ad = bf2f(addend[(int64_t)b * channels + g * cpg + chan_of(v)]);
#pragma unroll
      for (int j = 0; j < 8; ++j) xv.v[j] = rbf(xv.v[j] + ad);
    }
    return xv;
  };
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int v = v0 + (int)threadIdx.x + i * GN_THREADS;
    if (v < v1) {
      const Vec8 xv = value(i, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += xv.v[j];
    }
  }
  const float n = (float)((v1 - v0) * 8);
  const float mean = block_reduce<GN_THREADS>(s, OpSum(), smem) / n;
  float m2 = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int v = v0 + (int)threadIdx.x + i * GN_THREADS;
    if (v < v1) {
      const Vec8 xv = value(i, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = xv.v[j] - mean;
        m2 += d * d;
      }
    }
  }
  m2 = block_reduce<GN_THREADS>(m2, OpSum(), smem);
  if (threadIdx.x == 0) {
    float* o = part + ((int64_t)bg * P + slice) * 4;
    o[0] = n;
    o[1] = mean;
    o[2] = m2;
  }
}

template <int NV, bool SILU>
__global__ void __launch_bounds__(GN_THREADS) group_norm_apply_kernel(bf16_t* __restrict__ out, const bf16_t* __restrict__ x, const bf16_t* __restrict__ addend,
                                                                      const bf16_t* __restrict__ gamma, const bf16_t* __restrict__ beta,
                                                                      const float* __restrict__ part, int64_t group_elems, int hw, int hw_shift, int cpg,
                                                                      int channels, int groups, int P, float eps) {
  __shared__ float stat[2];
  const int slice = blockIdx.x, bg = blockIdx.y;
  const int b = bg / groups, g = bg - b * groups;
  const int64_t nvec = group_elems >> 3;
  const int v0 = (int)(nvec * slice / P), v1 = (int)(nvec * (slice + 1) / P);
  auto chan_of = [&](int v) { return hw_shift >= 0 ? (int)(((uint32_t)v << 3) >> hw_shift) : (int)(((uint32_t)v << 3) / (uint32_t)hw); };
  const bf16_t* src = x + (int64_t)bg * group_elems;
  bf16_t* dst = out + (int64_t)bg * group_elems;
  i32x4 raw[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {      // the slice is requested before the partials are merged
    const int v = v0 + (int)threadIdx.x + i * GN_THREADS;
    if (v < v1) raw[i] = *reinterpret_cast<const i32x4*>(src + (int64_t)v * 8);
  }
  if (threadIdx.x == 0) {
    const float* pp = part + (int64_t)bg * P * 4;
    double n = 0.0, mean = 0.0, m2 = 0.0;
    for (int i = 0; i < P; ++i) {      // Chan et al.: pairwise update of (n, mean, M2), fixed order
      const double ni = pp[4 * i], mi = pp[4 * i + 1], qi = pp[4 * i + 2];
      const double nt = n + ni, d = mi - mean;
      mean += d * (ni / nt);
      m2 += qi + d * d * (n * ni / nt);
      n = nt;
    }
    const float var = (float)(m2 / n);      // biased, as F.group_norm
    stat[0] = (float)mean;
    stat[1] = 1.0f / sqrtf(var + eps);
  }
  __syncthreads();
  const float mean = stat[0], rstd = stat[1];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int v = v0 + (int)threadIdx.x + i * GN_THREADS;
    if (v < v1) {
      const int c = g * cpg + chan_of(v);
      const float a = rstd * bf2f(gamma[c]);
      const float bb = bf2f(beta[c]) - a * mean;
      Vec8 xv = unpack8(raw[i]);
      if (addend != nullptr) {
        const float ad = bf2f(addend[(int64_t)b * channels + c]);
#pragma unroll
        for (int j = 0; j < 8; ++j) xv.v[j] = rbf(xv.v[j] + ad);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float y = rbf(__builtin_fmaf(a, xv.v[j], bb));
        if constexpr (SILU) y = y / (1.0f + expf(-y));
        xv.v[j] = y;
      }
      store8(dst + (int64_t)v * 8, xv);
    }
  }
}

// ---- the same GroupNorm (+ addend, + SiLU) on an NHWC tensor [batch, hw, C] (round 6: the layout of fdm_conv2d_3x3_nhwc_bf16).  A group's
// channels are cpg = C / groups neighbours in EVERY pixel, so a thread owns one 8-channel vector position and walks down the pixels of its
// slice: its affine coefficients are loop constants.  Statistics: per-thread fp32 (sum, sum of squares) over <= 64 pixels, reduced over the
// block's pixel lanes and the group's channels in a fixed order, merged over the slices in double by the apply kernel (deterministic; the
// partial sums are short enough that E[x^2] - mean^2 is formed in double from exact-to-fp32 parts).  cpg is even (a vector never splits a
// channel pair) but not a multiple of 8 (C = 320: 10), so a vector may straddle two groups: per-channel bookkeeping, not per-vector.
constexpr int GNH_MAXT = 512;
constexpr int GNH_ROWS = 8;      // pixels a thread holds in registers: a slice is at most GNH_ROWS * rpi pixels
__global__ void __launch_bounds__(GNH_MAXT) group_norm_nhwc_stats_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ addend, float* __restrict__ part,
                                                                        int hw, int channels, int groups, int P, int rpi) {
  extern __shared__ float gsm[];      // [rpi][channels] sums, then [rpi][channels] sums of squares
  const int nv = channels >> 3, cpg = channels / groups;
  const int slice = blockIdx.x, b = blockIdx.y;
  const int r0 = (int)((int64_t)hw * slice / P), r1 = (int)((int64_t)hw * (slice + 1) / P);
  const int rsub = (int)threadIdx.x / nv, vec = (int)threadIdx.x - rsub * nv;
  float s[8], q[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = q[j] = 0.f;
  Vec8 ad;
#pragma unroll
  for (int j = 0; j < 8; ++j) ad.v[j] = 0.f;
  if (addend != nullptr) ad = load8(addend + (int64_t)b * channels + vec * 8);
  const bf16_t* src = x + ((int64_t)b * hw) * channels + vec * 8;
  // a thread's <= GNH_ROWS pixels are requested back to back (one load in flight per thread left the kernel latency-bound: 74 us per launch)
  i32x4 raw[GNH_ROWS];
#pragma unroll
  for (int k = 0; k < GNH_ROWS; ++k) {
    const int r = r0 + rsub + k * rpi;
    if (r < r1) raw[k] = *reinterpret_cast<const i32x4*>(src + (int64_t)r * channels);
  }
#pragma unroll
  for (int k = 0; k < GNH_ROWS; ++k) {
    const int r = r0 + rsub + k * rpi;
    if (r < r1) {
      Vec8 xv = unpack8(raw[k]);
      if (addend != nullptr) {
#pragma unroll
        for (int j = 0; j < 8; ++j) xv.v[j] = rbf(xv.v[j] + ad.v[j]);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        s[j] += xv.v[j];
        q[j] = __builtin_fmaf(xv.v[j], xv.v[j], q[j]);
      }
    }
  }
  float* ssum = gsm;
  float* ssq = gsm + rpi * channels;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    ssum[rsub * channels + vec * 8 + j] = s[j];
    ssq[rsub * channels + vec * 8 + j] = q[j];
  }
  __syncthreads();
  if ((int)threadIdx.x < groups) {      // group g: its channels in order, each channel's pixel lanes in order
    const int g = threadIdx.x;
    double ts = 0.0, tq = 0.0;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c)
      for (int rr = 0; rr < rpi; ++rr) {
        ts += (double)ssum[rr * channels + c];
        tq += (double)ssq[rr * channels + c];
      }
    float* o = part + (((int64_t)b * P + slice) * groups + g) * 2;
    o[0] = (float)ts;
    o[1] = (float)tq;
  }
}

// (mean, rstd) per (image, group) from the slices' partial sums: 8 threads per group take every 8th slice with loads issued in batches (a
// serial walk over the ~340 slices of a 128 x 128 image inside EVERY apply block was 70 of that kernel's 97 us), then the 8 sub-sums are added in
// a fixed order -- deterministic, in double.
__global__ void __launch_bounds__(512) group_norm_nhwc_finalize_kernel(float* __restrict__ stat, const float* __restrict__ part, int hw, int cpg, int groups, int P,
                                                                      float eps) {
  __shared__ double sh[2][64][8];
  const int b = blockIdx.x, g = (int)threadIdx.x >> 3, k = (int)threadIdx.x & 7;
  double ts = 0.0, tq = 0.0;
  if (g < groups) {
    for (int i0 = k; i0 < P; i0 += 64) {      // 8 slices per batch of loads
      float a[8], c[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + 8 * u;
        const float* pp = part + (((int64_t)b * P + (i < P ? i : 0)) * groups + g) * 2;
        a[u] = i < P ? pp[0] : 0.f;
        c[u] = i < P ? pp[1] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        ts += (double)a[u];
        tq += (double)c[u];
      }
    }
    sh[0][g][k] = ts;
    sh[1][g][k] = tq;
  }
  __syncthreads();
  if (g < groups && k == 0) {
    double s0 = 0.0, s1 = 0.0;
    for (int u = 0; u < 8; ++u) {
      s0 += sh[0][g][u];
      s1 += sh[1][g][u];
    }
    const double n = (double)hw * cpg, mean = s0 / n;
    double var = s1 / n - mean * mean;      // biased, as F.group_norm
    if (var < 0.0) var = 0.0;
    stat[((int64_t)b * groups + g) * 2] = (float)mean;
    stat[((int64_t)b * groups + g) * 2 + 1] = 1.0f / sqrtf((float)var + eps);
  }
}

template <bool SILU>
__global__ void __launch_bounds__(GNH_MAXT) group_norm_nhwc_apply_kernel(bf16_t* __restrict__ out, const bf16_t* __restrict__ x, const bf16_t* __restrict__ addend,
                                                                        const bf16_t* __restrict__ gamma, const bf16_t* __restrict__ beta,
                                                                        const float* __restrict__ stat, int hw, int channels, int groups, int P, int rpi) {
  __shared__ float smean[64], srstd[64];
  const int nv = channels >> 3, cpg = channels / groups;
  const int slice = blockIdx.x, b = blockIdx.y;
  const int r0 = (int)((int64_t)hw * slice / P), r1 = (int)((int64_t)hw * (slice + 1) / P);
  const int rsub = (int)threadIdx.x / nv, vec = (int)threadIdx.x - rsub * nv;
  i32x4 raw[GNH_ROWS];      // the slice is requested before the partials are merged
  {
    const bf16_t* src0 = x + ((int64_t)b * hw) * channels + vec * 8;
#pragma unroll
    for (int k = 0; k < GNH_ROWS; ++k) {
      const int r = r0 + rsub + k * rpi;
      if (r < r1) raw[k] = *reinterpret_cast<const i32x4*>(src0 + (int64_t)r * channels);
    }
  }
  if ((int)threadIdx.x < groups) {      // (mean, rstd) of this image's groups: group_norm_nhwc_finalize_kernel
    smean[threadIdx.x] = stat[((int64_t)b * groups + threadIdx.x) * 2];
    srstd[threadIdx.x] = stat[((int64_t)b * groups + threadIdx.x) * 2 + 1];
  }
  __syncthreads();
  float a[8], bb[8];
  Vec8 ad;
  {
    const Vec8 gm = load8(gamma + vec * 8), bt = load8(beta + vec * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int g = (vec * 8 + j) / cpg;
      a[j] = srstd[g] * gm.v[j];
      bb[j] = bt.v[j] - a[j] * smean[g];
      ad.v[j] = 0.f;
    }
    if (addend != nullptr) ad = load8(addend + (int64_t)b * channels + vec * 8);
  }
  const bf16_t* src = x + ((int64_t)b * hw) * channels + vec * 8;
  bf16_t* dst = out + ((int64_t)b * hw) * channels + vec * 8;
#pragma unroll
  for (int k = 0; k < GNH_ROWS; ++k) {
    const int r = r0 + rsub + k * rpi;
    if (r < r1) {
      Vec8 xv = unpack8(raw[k]);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = xv.v[j];
        if (addend != nullptr) v = rbf(v + ad.v[j]);
        float y = rbf(__builtin_fmaf(a[j], v, bb[j]));
        if constexpr (SILU) y = y / (1.0f + expf(-y));
        xv.v[j] = y;
      }
      store8(dst + (int64_t)r * channels, xv);
    }
  }
}

}  // namespace fdm

using namespace fdm;

static inline int grid_for(int64_t work, int threads, int cap = 256 * 8) {
  int64_t g = (work + threads - 1) / threads;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <int MODE, int ACT = 0>
static int launch_quant(void* q, float* scale, int32_t* azp, const bf16_t* s0, int w0, int64_t st0, const bf16_t* s1, int w1,
                        int64_t st1, int64_t rows, hipStream_t s) {
  const int hidden = w0 + w1;
  const bool sw = g_fp8_sw_cvt;   // constexpr false in the product build
  auto qq = (uint8_t*)q;
#define FDM_QNIT(T, NITV) quant_rowwise_kernel<T, MODE, false, NITV, ACT><<<(unsigned)rows, T, 0, s>>>(qq, scale, azp, s0, w0, st0, s1, w1, st1)
#define FDM_QROW(T)                                                                                                      \
  if (sw) quant_rowwise_kernel<T, MODE, true, 0, ACT><<<(unsigned)rows, T, 0, s>>>(qq, scale, azp, s0, w0, st0, s1, w1, st1);     \
  else switch (((hidden >> 3) + T - 1) / T) {                                                                             \
    case 1: FDM_QNIT(T, 1); break; case 2: FDM_QNIT(T, 2); break; case 3: FDM_QNIT(T, 3); break; case 4: FDM_QNIT(T, 4); break; \
    case 5: FDM_QNIT(T, 5); break; case 6: FDM_QNIT(T, 6); break; case 7: FDM_QNIT(T, 7); break; default: FDM_QNIT(T, 8); break; \
  }
  if (hidden <= 4096 && ACT == 0) {
    FDM_QROW(64)
  } else if (hidden <= MAXV * 256 * 8) {
    FDM_QROW(256)
#undef FDM_QROW
#undef FDM_QNIT
  } else {
    if (s1 != nullptr || ACT != 0) return FDM_ERR_UNSUPPORTED;
    if (sw) quant_rowwise_wide_kernel<MODE, true><<<(unsigned)rows, 256, 0, s>>>(qq, scale, azp, s0, hidden, st0);
    else quant_rowwise_wide_kernel<MODE, false><<<(unsigned)rows, 256, 0, s>>>(qq, scale, azp, s0, hidden, st0);
  }
  FDM_CHECK_LAUNCH();
  return FDM_OK;
}

template <int THREADS>
static int launch_ln(void* out_bf16, void* out_q, float* out_scale, int32_t* out_azp, int qm, const void* x, const void* scale,
                     const void* shift, int64_t mbs, const void* res, const void* gate, void* x_out, int64_t rows, int hidden,
                     int64_t rpb, float eps, bool s1, hipStream_t s) {
  const bool sw = g_fp8_sw_cvt;   // constexpr false in the product build
#define FDM_LNARGS                                                                                                        \
  (bf16_t*)out_bf16, (uint8_t*)out_q, out_scale, out_azp, (const bf16_t*)x, (const bf16_t*)scale, (const bf16_t*)shift, mbs, \
      (const bf16_t*)res, (const bf16_t*)gate, (bf16_t*)x_out, hidden, rpb, eps
#define FDM_LNK(QM, SWV, NITV)                                                                                           \
  if (s1 && !(SWV)) ln_modulate_kernel<THREADS, QM, false, NITV, true><<<(unsigned)rows, THREADS, 0, s>>>(FDM_LNARGS);   \
  else ln_modulate_kernel<THREADS, QM, SWV, NITV, false><<<(unsigned)rows, THREADS, 0, s>>>(FDM_LNARGS)
#define FDM_LN(QM, SWV)                                                                                                  \
  if (SWV) FDM_LNK(QM, SWV, 0);                                                                                          \
  else switch (((hidden >> 3) + THREADS - 1) / THREADS) {                                                                 \
    case 1: FDM_LNK(QM, false, 1); break; case 2: FDM_LNK(QM, false, 2); break; case 3: FDM_LNK(QM, false, 3); break;    \
    case 4: FDM_LNK(QM, false, 4); break; case 5: FDM_LNK(QM, false, 5); break; case 6: FDM_LNK(QM, false, 6); break;    \
    case 7: FDM_LNK(QM, false, 7); break; default: FDM_LNK(QM, false, 8); break;                                         \
  }
  if (s1 && sw) return FDM_ERR_UNSUPPORTED;
  if (qm == 0) { FDM_LN(0, false) }
  else if (qm == 1) { if (sw) { FDM_LN(1, true) } else { FDM_LN(1, false) } }
  else if (qm == 2) { FDM_LN(2, false) }
  else return FDM_ERR_UNSUPPORTED;
#undef FDM_LNK
#undef FDM_LNARGS
#undef FDM_LN
  FDM_CHECK_LAUNCH();
  return FDM_OK;
}

template <int THREADS>
static int launch_ln_f32(void* out_bf16, void* out_q, float* out_scale, int32_t* out_azp, int qm, const void* x, const float* scale,
                         const float* shift, int64_t mbs, int64_t rows, int hidden, int64_t rpb, float eps, hipStream_t s) {
  const bool sw = g_fp8_sw_cvt;   // constexpr false in the product build
#define FDM_LNFK(QM, SWV, NITV)                                                                                  \
  ln_modulate_f32_kernel<THREADS, QM, SWV, NITV><<<(unsigned)rows, THREADS, 0, s>>>(                             \
      (bf16_t*)out_bf16, (uint8_t*)out_q, out_scale, out_azp, (const bf16_t*)x, scale, shift, mbs, hidden, rpb, eps)
#define FDM_LNF(QM, SWV)                                                                                         \
  if (SWV) FDM_LNFK(QM, SWV, 0);                                                                                 \
  else switch (((hidden >> 3) + THREADS - 1) / THREADS) {                                                         \
    case 1: FDM_LNFK(QM, false, 1); break; case 2: FDM_LNFK(QM, false, 2); break; case 3: FDM_LNFK(QM, false, 3); break; \
    case 4: FDM_LNFK(QM, false, 4); break; case 5: FDM_LNFK(QM, false, 5); break; case 6: FDM_LNFK(QM, false, 6); break; \
    case 7: FDM_LNFK(QM, false, 7); break; default: FDM_LNFK(QM, false, 8); break;                                \
  }
  if (qm == 0) { FDM_LNF(0, false) }
  else if (qm == 1) { if (sw) { FDM_LNF(1, true) } else { FDM_LNF(1, false) } }
  else if (qm == 2) { FDM_LNF(2, false) }
  else return FDM_ERR_UNSUPPORTED;
#undef FDM_LNFK
#undef FDM_LNF
  FDM_CHECK_LAUNCH();
  return FDM_OK;
}

extern "C" {

int fdm_ln_modulate_f32(void* out_bf16, void* out_q, float* out_scale, int32_t* out_azp, int quant_mode, const void* x,
                        const float* scale, const float* shift, int64_t mod_batch_stride, int64_t rows, int hidden,
                        int64_t rows_per_batch, float eps, fdm_stream_t stream) {
  (void)hipGetLastError();  // drop any stale error of an earlier, unrelated runtime call: FDM_CHECK_LAUNCH must see only ours
  if (!x || !scale || !shift || rows < 0 || hidden <= 0 || (hidden & 7) || rows_per_batch <= 0 || (mod_batch_stride & 3))
    return FDM_ERR_INVALID;
  if (quant_mode != 0 && (!out_q || !out_scale)) return FDM_ERR_INVALID;
  if (quant_mode == 2 && !out_azp) return FDM_ERR_INVALID;
  if (!aligned16(x) || !aligned16(scale) || !aligned16(shift) || !aligned16(out_bf16) || (reinterpret_cast<uintptr_t>(out_q) & 7))
    return FDM_ERR_INVALID;
  if (hidden > MAXV * 256 * 8) return FDM_ERR_UNSUPPORTED;
  if (rows == 0) return FDM_OK;
  hipStream_t s = to_stream(stream);
  if (hidden <= 4096)
    return launch_ln_f32<64>(out_bf16, out_q, out_scale, out_azp, quant_mode, x, scale, shift, mod_batch_stride, rows, hidden,
                             rows_per_batch, eps, s);
  return launch_ln_f32<256>(out_bf16, out_q, out_scale, out_azp, quant_mode, x, scale, shift, mod_batch_stride, rows, hidden,
                            rows_per_batch, eps, s);
}

int fdm_rms_norm(void* out, const void* in, const void* weight, float eps, int64_t rows, int hidden, fdm_stream_t stream) {
  (void)hipGetLastError();  // drop any stale error of an earlier, unrelated runtime call: FDM_CHECK_LAUNCH must see only ours
  if (!out || !in || !weight || rows < 0 || hidden <= 0) return FDM_ERR_INVALID;
  if (rows == 0) return FDM_OK;
  hipStream_t s = to_stream(stream);
  auto o = (bf16_t*)out;
  auto i = (const bf16_t*)in;
  auto w = (const bf16_t*)weight;
  if ((hidden & 7) || !aligned16(out) || !aligned16(in) || !aligned16(weight)) {   // odd widths / unaligned rows: scalar path
    if ((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(weight)) & 1) return FDM_ERR_INVALID;
    rms_norm_scalar_kernel<<<(unsigned)((rows + 3) / 4), 256, 0, s>>>(o, i, w, eps, rows, hidden);
    FDM_CHECK_LAUNCH();
    return FDM_OK;
  }
  const int lpr = hidden / 8;
  if (lpr <= 64 && (lpr & (lpr - 1)) == 0) {
    const int rpw = 64 / lpr;
    const int64_t waves = (rows + rpw - 1) / rpw;
    const int grid = (int)((waves + 3) / 4);
    switch (lpr) {
      case 1: rms_norm_small_kernel<1><<<grid, 256, 0, s>>>(o, i, w, eps, rows, hidden); break;
      case 2: rms_norm_small_kernel<2><<<grid, 256, 0, s>>>(o, i, w, eps, rows, hidden); break;
      case 4: rms_norm_small_kernel<4><<<grid, 256, 0, s>>>(o, i, w, eps, rows, hidden); break;
      case 8: rms_norm_small_kernel<8><<<grid, 256, 0, s>>>(o, i, w, eps, rows, hidden); break;
      case 16: rms_norm_small_kernel<16><<<grid, 256, 0, s>>>(o, i, w, eps, rows, hidden); break;
      case 32: rms_norm_small_kernel<32><<<grid, 256, 0, s>>>(o, i, w, eps, rows, hidden); break;
      default: rms_norm_small_kernel<64><<<grid, 256, 0, s>>>(o, i, w, eps, rows, hidden); break;
    }
  } else if (hidden <= 4096) {
    launch_rms_row<64>(o, i, w, eps, rows, hidden, hidden, hidden, s);
  } else {
    launch_rms_row<256>(o, i, w, eps, rows, hidden, hidden, hidden, s);
  }
  FDM_CHECK_LAUNCH();
  return FDM_OK;
}

int fdm_rms_norm_strided(void* out, const void* in, const void* weight, float eps, int64_t rows, int hidden, int64_t in_row_stride,
                         int64_t out_row_stride, fdm_stream_t stream) {
  (void)hipGetLastError();  // drop any stale error of an earlier, unrelated runtime call: FDM_CHECK_LAUNCH must see only ours
  if (rows == 0 && hidden > 0 && !(hidden & 7)) return FDM_OK;
  if (!out || !in || !weight || rows < 0 || hidden <= 0 || (hidden & 7) || !aligned16(out) || !aligned16(in) || !aligned16(weight) ||
      in_row_stride < hidden || out_row_stride < hidden || (in_row_stride & 7) || (out_row_stride & 7))
    return FDM_ERR_INVALID;
  hipStream_t s = to_stream(stream);
  if (hidden <= 4096) launch_rms_row<64>((bf16_t*)out, (const bf16_t*)in, (const bf16_t*)weight, eps, rows, hidden, in_row_stride, out_row_stride, s);
  else launch_rms_row<256>((bf16_t*)out, (const bf16_t*)in, (const bf16_t*)weight, eps, rows, hidden, in_row_stride, out_row_stride, s);
  FDM_CHECK_LAUNCH();
  return FDM_OK;
}

int fdm_rms_norm_rope_strided(void* out, const void* in, const void* weight, float eps, int64_t rows, int hidden, int64_t in_row_stride,
                              int64_t out_row_stride, const void* cos_sin_cache, int64_t seq_len, int head_size, fdm_stream_t stream) {
  (void)hipGetLastError();  // drop any stale error of an earlier, unrelated runtime call: FDM_CHECK_LAUNCH must see only ours
  if (rows == 0 && hidden > 0 && !(hidden & 7)) return FDM_OK;
  if (!out || !in || !weight || !cos_sin_cache || rows < 0 || hidden <= 0 || (hidden & 7) || !aligned16(out) || !aligned16(in) || !aligned16(weight) ||
      in_row_stride < hidden || out_row_stride < hidden || (in_row_stride & 7) || (out_row_stride & 7) || seq_len <= 0)
    return FDM_ERR_INVALID;
  if (head_size <= 0 || (head_size & 7) || hidden % head_size || (reinterpret_cast<uintptr_t>(cos_sin_cache) & 7)) return FDM_ERR_UNSUPPORTED;
  hipStream_t s = to_stream(stream);
  const RopeArgs ra{(const bf16_t*)cos_sin_cache, seq_len, head_size};
  if (hidden <= 4096)
    launch_rms_row<64, true>((bf16_t*)out, (const bf16_t*)in, (const bf16_t*)weight, eps, rows, hidden, in_row_stride, out_row_stride, s, ra);
  else
    launch_rms_row<256, true>((bf16_t*)out, (const bf16_t*)in, (const bf16_t*)weight, eps, rows, hidden, in_row_stride, out_row_stride, s, ra);
  FDM_CHECK_LAUNCH();
  return FDM_OK;
}

int fdm_rope_inplace(void* q, void* k, const void* cos_sin_cache, int64_t tokens, int64_t seq_len, int q_heads,
                     int k_heads, int head_size, int rot_dim, int64_t q_row_stride, int64_t k_row_stride, int is_neox,
                     fdm_stream_t stream) {
  (void)hipGetLastError();  // drop any stale error of an earlier, unrelated runtime call: FDM_CHECK_LAUNCH must see only ours
  if (!q || !k || !cos_sin_cache || tokens < 0 || seq_len <= 0 || q_heads < 0 || k_heads < 0 || rot_dim > head_size)
    return FDM_ERR_INVALID;
  if ((head_size & 7) || (q_row_stride & 7) || (k_row_stride & 7) || !aligned16(q) || !aligned16(k)) return FDM_ERR_INVALID;
  if (is_neox ? (rot_dim % 16) : (rot_dim % 8)) return FDM_ERR_UNSUPPORTED;
  if (tokens == 0) return FDM_OK;
  const int per_head = is_neox ? (rot_dim / 16) : (rot_dim / 8);
  const int64_t work = tokens * (int64_t)(q_heads + k_heads) * per_head;
  const int64_t grid = (work + 255) / 256;
  hipStream_t s = to_stream(stream);
  if (is_neox)
    rope_kernel<true><<<(unsigned)grid, 256, 0, s>>>((bf16_t*)q, (bf16_t*)k, (const bf16_t*)cos_sin_cache, tokens, seq_len,
                                                     q_heads, k_heads, head_size, rot_dim, q_row_stride, k_row_stride);
  else
    rope_kernel<false><<<(unsigned)grid, 256, 0, s>>>((bf16_t*)q, (bf16_t*)k, (const bf16_t*)cos_sin_cache, tokens, seq_len,
                                                      q_heads, k_heads, head_size, rot_dim, q_row_stride, k_row_stride);
  FDM_CHECK_LAUNCH();
  return FDM_OK;
}

int fdm_quant_rowwise_cat2_act(void* q_out, float* scale_out, int32_t* azp_out, int quant_mode, const void* src0, int w0,
                               int64_t stride0, int act0, const void* src1, int w1, int64_t stride1, int act1, int64_t rows,
                               fdm_stream_t stream) {
  (void)hipGetLastError();  // drop any stale error of an earlier, unrelated runtime call: FDM_CHECK_LAUNCH must see only ours
  if (act0 < 0 || act0 > 2 || act1 < 0 || act1 > 2) return FDM_ERR_INVALID;
  if (rows == 0 && w0 > 0 && !(w0 & 7) && w1 >= 0 && !(w1 & 7)) return FDM_OK;
  if (!q_out || !scale_out || !src0 || w0 <= 0 || (w0 & 7) || (w1 & 7) || w1 < 0 || (stride0 & 7) || (stride1 & 7) || rows < 0)
    return FDM_ERR_INVALID;
  if ((w1 > 0) != (src1 != nullptr)) return FDM_ERR_INVALID;
  if (!aligned16(src0) || !aligned16(src1) || (reinterpret_cast<uintptr_t>(q_out) & 7)) return FDM_ERR_INVALID;
  if (quant_mode == 2 && !azp_out) return FDM_ERR_INVALID;
  if (rows == 0) return FDM_OK;
  if (w1 == 0) act1 = 0;
  hipStream_t s = to_stream(stream);
  auto a = (const bf16_t*)src0;
  auto b = (const bf16_t*)src1;
  const int act = act0 | (act1 << 2);
  // the activation forms are instantiated for the call sites that exist: a GELU on the only source (feed-forward) or on the second of two
  // (FLUX single block: [attention | gelu(mlp)]); fp8 and int8-asym
#define FDM_QACT(M)                                                                                                       \
  switch (act) {                                                                                                          \
    case 0: return launch_quant<M, 0>(q_out, scale_out, azp_out, a, w0, stride0, b, w1, stride1, rows, s);                \
    case 1: return launch_quant<M, 1>(q_out, scale_out, azp_out, a, w0, stride0, b, w1, stride1, rows, s);                \
    case 2: return launch_quant<M, 2>(q_out, scale_out, azp_out, a, w0, stride0, b, w1, stride1, rows, s);                \
    case 4: return launch_quant<M, 4>(q_out, scale_out, azp_out, a, w0, stride0, b, w1, stride1, rows, s);                \
    case 8: return launch_quant<M, 8>(q_out, scale_out, azp_out, a, w0, stride0, b, w1, stride1, rows, s);                \
    default: return FDM_ERR_UNSUPPORTED;                                                                                  \
  }
  switch (quant_mode) {
    case 1: FDM_QACT(1)
    case 2: FDM_QACT(2)
    case 3: if (act) return FDM_ERR_UNSUPPORTED; return launch_quant<3>(q_out, scale_out, azp_out, a, w0, stride0, b, w1, stride1, rows, s);
    default: return FDM_ERR_UNSUPPORTED;
  }
#undef FDM_QACT
}

int fdm_quant_rowwise_cat2(void* q_out, float* scale_out, int32_t* azp_out, int quant_mode, const void* src0, int w0,
                           int64_t stride0, const void* src1, int w1, int64_t stride1, int64_t rows, fdm_stream_t stream) {
  return fdm_quant_rowwise_cat2_act(q_out, scale_out, azp_out, quant_mode, src0, w0, stride0, 0, src1, w1, stride1, 0, rows, stream);
}

int fdm_quant_fp8_rowwise(void* q_out, float* scale_out, const void* in, int64_t rows, int hidden, int64_t in_row_stride,
                          fdm_stream_t stream) {
  (void)hipGetLastError();  // drop any stale error of an earlier, unrelated runtime call: FDM_CHECK_LAUNCH must see only ours
  return fdm_quant_rowwise_cat2(q_out, scale_out, nullptr, 1, in, hidden, in_row_stride, nullptr, 0, 0, rows, stream);
}

int fdm_quant_int8_rowwise(void* q_out, float* scale_out, int32_t* azp_out, const void* in, int64_t rows, int hidden,
                           int64_t in_row_stride, fdm_stream_t stream) {
  (void)hipGetLastError();  // drop any stale error of an earlier, unrelated runtime call: FDM_CHECK_LAUNCH must see only ours
  return fdm_quant_rowwise_cat2(q_out, scale_out, azp_out, azp_out ? 2 : 3, in, hidden, in_row_stride, nullptr, 0, 0, rows,
                                stream);
}

int fdm_gelu_and_mul(void* out, const void* in, int64_t rows, int d, fdm_stream_t stream) {
  (void)hipGetLastError();  // drop any stale error of an earlier, unrelated runtime call: FDM_CHECK_LAUNCH must see only ours
  if (!out || !in || rows < 0 || d <= 0 || (d & 7) || !aligned16(out) || !aligned16(in)) return FDM_ERR_INVALID;
  if (rows == 0) return FDM_OK;
  gelu_and_mul_kernel<<<grid_for(rows * (d / 8), 256), 256, 0, to_stream(stream)>>>((bf16_t*)out, (const bf16_t*)in, rows, d);
  FDM_CHECK_LAUNCH();
  return FDM_OK;
}

int fdm_gelu_and_mul_quant(void* q_out, float* scale_out, int32_t* azp_out, int quant_mode, const void* in, int64_t rows, int d, fdm_stream_t stream) {
  (void)hipGetLastError();  // drop any stale error of an earlier, unrelated runtime call: FDM_CHECK_LAUNCH must see only ours
  if (!q_out || !scale_out || !in || rows < 0 || d <= 0 || (d & 7) || !aligned16(in) || (reinterpret_cast<uintptr_t>(q_out) & 7)) return FDM_ERR_INVALID;
  if (quant_mode != 1 && quant_mode != 2) return FDM_ERR_UNSUPPORTED;
  if (quant_mode == 2 && !azp_out) return FDM_ERR_INVALID;
  if (d > MAXV * 256 * 8) return FDM_ERR_UNSUPPORTED;      // a row lives in the registers of one block
  if (rows == 0) return FDM_OK;
  if (quant_mode == 1) gelu_and_mul_quant_kernel<1><<<(unsigned)rows, 256, 0, to_stream(stream)>>>((uint8_t*)q_out, scale_out, azp_out, (const bf16_t*)in, d);
  else gelu_and_mul_quant_kernel<2><<<(unsigned)rows, 256, 0, to_stream(stream)>>>((uint8_t*)q_out, scale_out, azp_out, (const bf16_t*)in, d);
  FDM_CHECK_LAUNCH();
  return FDM_OK;
}

// slices per group: enough blocks for ~4 per CU, at most 8 vectors of a slice per thread
static int gn_slices(int64_t batch, int groups, int64_t group_elems) {
  const int64_t nvec = group_elems >> 3;
  int64_t P = (1024 + batch * groups - 1) / (batch * groups);
  const int64_t pmin = (nvec + 8 * GN_THREADS - 1) / (8 * GN_THREADS);
  if (P < pmin) P = pmin;
  const int64_t pmax = (nvec + GN_THREADS - 1) / GN_THREADS;      // at least one vector per thread
  if (P > pmax) P = pmax;
  return (int)(P < 1 ? 1 : P);
}

int64_t fdm_group_norm_workspace_bytes(int64_t batch, int64_t channels, int64_t hw, int groups) {
  if (batch <= 0 || channels <= 0 || hw <= 0 || groups <= 0 || channels % groups) return 0;
  return batch * groups * (int64_t)gn_slices(batch, groups, channels / groups * hw) * 16;
}

int fdm_group_norm(void* out, const void* x, const void* gamma, const void* beta, const void* addend, int64_t batch, int64_t channels, int64_t hw,
                   int groups, float eps, int act, void* workspace, int64_t workspace_bytes, fdm_stream_t stream) {
  (void)hipGetLastError();  // drop any stale error of an earlier, unrelated runtime call: FDM_CHECK_LAUNCH must see only ours
  if (!out || !x || !gamma || !beta || batch < 0 || channels <= 0 || hw <= 0 || groups <= 0 || channels % groups) return FDM_ERR_INVALID;
  if (!aligned16(out) || !aligned16(x)) return FDM_ERR_INVALID;
  if (act != 0 && act != 1) return FDM_ERR_UNSUPPORTED;
  if (hw & 7) return FDM_ERR_UNSUPPORTED;      // 16-byte vectors must not straddle channels
  if (batch == 0) return FDM_OK;
  if (batch * groups > 65535) return FDM_ERR_UNSUPPORTED;
  const int cpg = (int)(channels / groups);
  const int64_t group_elems = (int64_t)cpg * hw;
  if (group_elems >= ((int64_t)1 << 31)) return FDM_ERR_UNSUPPORTED;
  const int hw_shift = (hw & (hw - 1)) == 0 ? __builtin_ctzll((unsigned long long)hw) : -1;
  const int P = gn_slices(batch, groups, group_elems);
  if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 15) || workspace_bytes < batch * groups * (int64_t)P * 16) return FDM_ERR_INVALID;
  const int64_t nvec = group_elems >> 3;
  const int64_t per_thread = ((nvec + P - 1) / P + GN_THREADS - 1) / GN_THREADS;      // the largest slice of the floor partition has ceil(nvec / P) vectors
  if (per_thread > 8) return FDM_ERR_UNSUPPORTED;      // (cannot happen: gn_slices)
  const dim3 grid((unsigned)P, (unsigned)(batch * groups));
  hipStream_t s = to_stream(stream);
  float* part = reinterpret_cast<float*>(workspace);
#define FDM_GN_LAUNCH(NV)                                                                                                                              \
  do {                                                                                                                                                 \
    group_norm_stats_kernel<NV><<<grid, GN_THREADS, 0, s>>>((const bf16_t*)x, (const bf16_t*)addend, part, group_elems, (int)hw, hw_shift, cpg,         \
                                                            (int)channels, groups, P);                                                                                \
    FDM_CHECK_LAUNCH();                                                                                                                                \
    if (act) group_norm_apply_kernel<NV, true><<<grid, GN_THREADS, 0, s>>>((bf16_t*)out, (const bf16_t*)x, (const bf16_t*)addend, (const bf16_t*)gamma, \
                                                                           (const bf16_t*)beta, part, group_elems, (int)hw, hw_shift, cpg, (int)channels, \
                                                                           groups, P, eps);                                                                     \
    else group_norm_apply_kernel<NV, false><<<grid, GN_THREADS, 0, s>>>((bf16_t*)out, (const bf16_t*)x, (const bf16_t*)addend, (const bf16_t*)gamma,    \
                                                                        (const bf16_t*)beta, part, group_elems, (int)hw, hw_shift, cpg, (int)channels,  \
                                                                        groups, P, eps);                                                                           \
  } while (0)
  if (per_thread <= 4) FDM_GN_LAUNCH(4);
  else FDM_GN_LAUNCH(8);
#undef FDM_GN_LAUNCH
  FDM_CHECK_LAUNCH();
  return FDM_OK;
}

// slices of the pixel axis for the NHWC form: enough blocks for ~4 per CU, at least 64 pixels per pixel lane of a block
static int gnh_slices(int64_t batch, int64_t hw, int rpi) {
  (void)batch;
  const int64_t per = (int64_t)rpi * GNH_ROWS;      // pixels a block holds in registers
  return (int)((hw + per - 1) / per);
}
static int gnh_rpi(int64_t channels) {
  const int nv = (int)(channels >> 3);
  int rpi = 256 / nv;
  return rpi < 1 ? 1 : rpi;
}

int64_t fdm_group_norm_nhwc_workspace_bytes(int64_t batch, int64_t channels, int64_t hw, int groups) {
  if (batch <= 0 || channels <= 0 || hw <= 0 || groups <= 0 || channels % groups) return 0;
  return batch * (int64_t)gnh_slices(batch, hw, gnh_rpi(channels)) * groups * 8 + batch * (int64_t)groups * 8;      // slice partials + (mean, rstd)
}

int fdm_group_norm_nhwc(void* out, const void* x, const void* gamma, const void* beta, const void* addend, int64_t batch, int64_t channels, int64_t hw,
                        int groups, float eps, int act, void* workspace, int64_t workspace_bytes, fdm_stream_t stream) {
  (void)hipGetLastError();  // drop any stale error of an earlier, unrelated runtime call: FDM_CHECK_LAUNCH must see only ours
  if (!out || !x || !gamma || !beta || batch < 0 || channels <= 0 || hw <= 0 || groups <= 0 || channels % groups) return FDM_ERR_INVALID;
  if (!aligned16(out) || !aligned16(x) || !aligned16(gamma) || !aligned16(beta) || !aligned16(addend)) return FDM_ERR_INVALID;
  if (act != 0 && act != 1) return FDM_ERR_UNSUPPORTED;
  const int cpg = (int)(channels / groups);
  if ((channels & 7) || (cpg & 1) || groups > 64 || channels > 8 * GNH_MAXT || batch > 65535 || hw >= ((int64_t)1 << 31)) return FDM_ERR_UNSUPPORTED;
  if (batch == 0) return FDM_OK;
  const int nv = (int)(channels >> 3), rpi = gnh_rpi(channels), threads = nv * rpi;
  const int P = gnh_slices(batch, hw, rpi);
  if (!workspace || (reinterpret_cast<uintptr_t>(workspace) & 15) || workspace_bytes < batch * (int64_t)P * groups * 8 + batch * (int64_t)groups * 8)
    return FDM_ERR_INVALID;
  if (threads < groups) return FDM_ERR_UNSUPPORTED;
  const dim3 grid((unsigned)P, (unsigned)batch);
  hipStream_t s = to_stream(stream);
  float* part = reinterpret_cast<float*>(workspace);
  const size_t shm = (size_t)2 * rpi * channels * sizeof(float);
  group_norm_nhwc_stats_kernel<<<grid, threads, shm, s>>>((const bf16_t*)x, (const bf16_t*)addend, part, (int)hw, (int)channels, groups, P, rpi);
  FDM_CHECK_LAUNCH();
  float* stat = part + batch * (int64_t)P * groups * 2;
  group_norm_nhwc_finalize_kernel<<<(unsigned)batch, 512, 0, s>>>(stat, part, (int)hw, cpg, groups, P, eps);
  FDM_CHECK_LAUNCH();
  if (act) group_norm_nhwc_apply_kernel<true><<<grid, threads, 0, s>>>((bf16_t*)out, (const bf16_t*)x, (const bf16_t*)addend, (const bf16_t*)gamma,
                                                                       (const bf16_t*)beta, stat, (int)hw, (int)channels, groups, P, rpi);
  else group_norm_nhwc_apply_kernel<false><<<grid, threads, 0, s>>>((bf16_t*)out, (const bf16_t*)x, (const bf16_t*)addend, (const bf16_t*)gamma,
                                                                    (const bf16_t*)beta, stat, (int)hw, (int)channels, groups, P, rpi);
  FDM_CHECK_LAUNCH();
  return FDM_OK;
}

int fdm_gelu(void* out, const void* in, int64_t n, int mode, fdm_stream_t stream) {
  (void)hipGetLastError();  // drop any stale error of an earlier, unrelated runtime call: FDM_CHECK_LAUNCH must see only ours
  if (!out || !in || n < 0 || (n & 7) || !aligned16(out) || !aligned16(in)) return FDM_ERR_INVALID;
  if (mode != 1 && mode != 2) return FDM_ERR_UNSUPPORTED;
  if (n == 0) return FDM_OK;
  if (mode == 1) gelu_kernel<1><<<grid_for(n / 8, 256), 256, 0, to_stream(stream)>>>((bf16_t*)out, (const bf16_t*)in, n / 8);
  else gelu_kernel<2><<<grid_for(n / 8, 256), 256, 0, to_stream(stream)>>>((bf16_t*)out, (const bf16_t*)in, n / 8);
  FDM_CHECK_LAUNCH();
  return FDM_OK;
}

int fdm_gate_residual(void* out, const void* res, const void* y, const void* gate, int64_t gate_batch_stride, int64_t rows,
                      int hidden, int64_t rows_per_batch, fdm_stream_t stream) {
  (void)hipGetLastError();  // drop any stale error of an earlier, unrelated runtime call: FDM_CHECK_LAUNCH must see only ours
  if (!out || !res || !y || !gate || rows < 0 || hidden <= 0 || (hidden & 7) || rows_per_batch <= 0 || (gate_batch_stride & 7))
    return FDM_ERR_INVALID;
  if (!aligned16(out) || !aligned16(res) || !aligned16(y) || !aligned16(gate)) return FDM_ERR_INVALID;
  if (rows == 0) return FDM_OK;
  const int64_t nvec = rows * (hidden / 8);
  gate_residual_kernel<<<grid_for(nvec, 256), 256, 0, to_stream(stream)>>>((bf16_t*)out, (const bf16_t*)res, (const bf16_t*)y,
                                                                         (const bf16_t*)gate, gate_batch_stride, nvec,
                                                                         hidden / 8, rows_per_batch);
  FDM_CHECK_LAUNCH();
  return FDM_OK;
}

int fdm_ln_modulate(void* out_bf16, void* out_q, float* out_scale, int32_t* out_azp, int quant_mode, const void* x,
                    const void* scale, const void* shift, int64_t mod_batch_stride, const void* res, const void* gate,
                    void* x_out, int64_t rows, int hidden, int64_t rows_per_batch, float eps, int scale_has_one, fdm_stream_t stream) {
  (void)hipGetLastError();  // drop any stale error of an earlier, unrelated runtime call: FDM_CHECK_LAUNCH must see only ours
  if (scale_has_one != 0 && scale_has_one != 1) return FDM_ERR_INVALID;
  if (!x || !scale || !shift || rows < 0 || hidden <= 0 || (hidden & 7) || rows_per_batch <= 0 || (mod_batch_stride & 7))
    return FDM_ERR_INVALID;
  if (quant_mode != 0 && (!out_q || !out_scale)) return FDM_ERR_INVALID;
  if (quant_mode == 2 && !out_azp) return FDM_ERR_INVALID;
  if ((res != nullptr) != (gate != nullptr)) return FDM_ERR_INVALID;
  if (!aligned16(x) || !aligned16(scale) || !aligned16(shift) || !aligned16(res) || !aligned16(gate) || !aligned16(out_bf16) ||
      !aligned16(x_out) || (reinterpret_cast<uintptr_t>(out_q) & 7))
    return FDM_ERR_INVALID;
  if (hidden > MAXV * 256 * 8) return FDM_ERR_UNSUPPORTED;
  if (rows == 0) return FDM_OK;
  hipStream_t s = to_stream(stream);
  if (hidden <= 4096)
    return launch_ln<64>(out_bf16, out_q, out_scale, out_azp, quant_mode, x, scale, shift, mod_batch_stride, res, gate, x_out,
                         rows, hidden, rows_per_batch, eps, scale_has_one != 0, s);
  return launch_ln<256>(out_bf16, out_q, out_scale, out_azp, quant_mode, x, scale, shift, mod_batch_stride, res, gate, x_out,
                        rows, hidden, rows_per_batch, eps, scale_has_one != 0, s);
}

int fdm_qk_norm_rope(void* q_out, void* k_out, void* v_out, const void* qkv, const void* wq, const void* wk,
                     const void* cos_sin_cache, int64_t rows, int64_t rows_per_batch, int64_t seq_total, int64_t seq_offset,
                     int heads, int head_dim, float eps, fdm_stream_t stream) {
  (void)hipGetLastError();  // drop any stale error of an earlier, unrelated runtime call: FDM_CHECK_LAUNCH must see only ours
  if (!q_out || !k_out || !v_out || !qkv || !wq || !wk || rows < 0 || rows_per_batch <= 0 || heads <= 0) return FDM_ERR_INVALID;
  if (seq_offset < 0 || seq_offset + rows_per_batch > seq_total) return FDM_ERR_INVALID;
  if (!aligned16(q_out) || !aligned16(k_out) || !aligned16(v_out) || !aligned16(qkv) || !aligned16(wq) || !aligned16(wk))
    return FDM_ERR_INVALID;
  if (head_dim != 64 && head_dim != 128 && head_dim != 32) return FDM_ERR_UNSUPPORTED;
  if (rows == 0) return FDM_OK;
  const int lph = head_dim / 8;
  const int64_t threads = rows * 3 * heads * lph;
  const int64_t grid = (threads + 255) / 256;
  hipStream_t s = to_stream(stream);
#define FDM_QK(HD)                                                                                                         \
  qk_norm_rope_kernel<HD><<<(unsigned)grid, 256, 0, s>>>((bf16_t*)q_out, (bf16_t*)k_out, (bf16_t*)v_out, (const bf16_t*)qkv, \
                                                         (const bf16_t*)wq, (const bf16_t*)wk, (const bf16_t*)cos_sin_cache, \
                                                         rows, rows_per_batch, seq_total, seq_offset, heads, eps)
  if (head_dim == 128) FDM_QK(128);
  else if (head_dim == 64) FDM_QK(64);
  else FDM_QK(32);
#undef FDM_QK
  FDM_CHECK_LAUNCH();
  return FDM_OK;
}

int fdm_qk_norm_rope_inplace(void* qkv, const void* wq, const void* wk, const void* cos_sin_cache, int64_t rows,
                             int64_t rows_per_batch, int64_t batch_row_stride, int64_t row_stride, int64_t k_offset,
                             int64_t pos_offset, int heads, int head_dim, float eps, fdm_stream_t stream) {
  (void)hipGetLastError();  // drop any stale error of an earlier, unrelated runtime call: FDM_CHECK_LAUNCH must see only ours
  if (!qkv || !wq || !wk || rows < 0 || rows_per_batch <= 0 || batch_row_stride < rows_per_batch || pos_offset < 0 ||
      heads <= 0 || row_stride < (int64_t)heads * head_dim || (row_stride & 7) || k_offset < (int64_t)heads * head_dim || (k_offset & 7))
    return FDM_ERR_INVALID;
  if (head_dim != 128 && head_dim != 64 && head_dim != 32) return FDM_ERR_UNSUPPORTED;
  if (!aligned16(qkv) || !aligned16(wq) || !aligned16(wk) || (cos_sin_cache && !aligned16(cos_sin_cache))) return FDM_ERR_INVALID;
  if (rows == 0) return FDM_OK;
  const int64_t threads = rows * 2 * heads * (head_dim / 8);
  const int64_t grid = (threads + 255) / 256;
  hipStream_t s = to_stream(stream);
#define FDM_QKI(HD)                                                                                                  \
  qk_norm_rope_inplace_kernel<HD><<<(unsigned)grid, 256, 0, s>>>((bf16_t*)qkv, (const bf16_t*)wq, (const bf16_t*)wk, \
                                                                 (const bf16_t*)cos_sin_cache, rows, rows_per_batch,  \
                                                                 batch_row_stride, row_stride, k_offset, pos_offset, heads, eps)
#define FDM_QKR(HD, NIV)                                                                                                    \
  qk_norm_rope_inplace_row_kernel<HD, NIV><<<(unsigned)rows, 256, 0, s>>>((bf16_t*)qkv, (const bf16_t*)wq, (const bf16_t*)wk,   \
                                                                          (const bf16_t*)cos_sin_cache, rows_per_batch,        \
                                                                          batch_row_stride, row_stride, k_offset, pos_offset, heads, eps)
#define FDM_QKSEL(HD)                                                                          \
  switch ((2 * heads + (2048 / HD) - 1) / (2048 / HD)) {   /* passes of 256 / (HD / 8) items */  \
    case 1: FDM_QKR(HD, 1); break; case 2: FDM_QKR(HD, 2); break; case 3: FDM_QKR(HD, 3); break; \
    case 4: FDM_QKR(HD, 4); break; default: FDM_QKI(HD); break;                                  \
  }
  if (head_dim == 128) { FDM_QKSEL(128) }
  else if (head_dim == 64) { FDM_QKSEL(64) }
  else { FDM_QKSEL(32) }
#undef FDM_QKSEL
#undef FDM_QKR
#undef FDM_QKI
  FDM_CHECK_LAUNCH();
  return FDM_OK;
}

int fdm_absdiff_sums(float* out2, const void* a, const void* b, int64_t n, fdm_stream_t stream) {
  (void)hipGetLastError();  // drop any stale error of an earlier, unrelated runtime call: FDM_CHECK_LAUNCH must see only ours
  if (!out2 || !a || !b || n < 0 || (n & 7) || !aligned16(a) || !aligned16(b)) return FDM_ERR_INVALID;
  if (n == 0) return FDM_OK;
  absdiff_sums_kernel<<<grid_for(n / 8, 256, 1024), 256, 0, to_stream(stream)>>>(out2, (const bf16_t*)a, (const bf16_t*)b, n / 8);
  FDM_CHECK_LAUNCH();
  return FDM_OK;
}

int fdm_synth_fill_bf16(void* out, int64_t n, uint32_t seed, int kind, float scale, float offset, fdm_stream_t stream) {
  (void)hipGetLastError();  // drop any stale error of an earlier, unrelated runtime call: FDM_CHECK_LAUNCH must see only ours
  if (!out || n < 0 || (kind != 0 && kind != 1)) return FDM_ERR_INVALID;
  if (n == 0) return FDM_OK;
  synth_fill_kernel<<<grid_for(n, 256, 256 * 16), 256, 0, to_stream(stream)>>>((bf16_t*)out, n, seed, kind, scale, offset);
  FDM_CHECK_LAUNCH();
  return FDM_OK;
}

int fdm_debug_cvt_fp8(void* out_u8, const float* in, int64_t n, int use_sw, fdm_stream_t stream) {
  (void)hipGetLastError();  // drop any stale error of an earlier, unrelated runtime call: FDM_CHECK_LAUNCH must see only ours
  if (!out_u8 || !in || n < 0) return FDM_ERR_INVALID;
  if (n == 0) return FDM_OK;
  const unsigned grid = (unsigned)((n + 255) / 256);
  if (use_sw) debug_cvt_fp8_kernel<true><<<grid, 256, 0, to_stream(stream)>>>((uint8_t*)out_u8, in, n);
  else debug_cvt_fp8_kernel<false><<<grid, 256, 0, to_stream(stream)>>>((uint8_t*)out_u8, in, n);
  FDM_CHECK_LAUNCH();
  return FDM_OK;
}

}  // extern "C"
