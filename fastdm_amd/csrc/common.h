// Shared device helpers for the gfx950 kernels (wave64 everywhere).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/fastdm_amd.h"

#define FDM_CHECK_LAUNCH()                                         \
  do {                                                             \
    hipError_t e_ = hipGetLastError();                             \
    if (e_ != hipSuccess) {                                        \
      fdm::g_last_hip_error = (int)e_;                             \
      return FDM_ERR_LAUNCH;                                       \
    }                                                              \
  } while (0)

namespace fdm {

extern thread_local int g_last_hip_error;  // HIP error code behind this thread's most recent FDM_ERR_LAUNCH (fdm_last_hip_error_string)

// Compute units of the CURRENT device, cached per device id (two devices of one process may differ; threads may race
// here, all writing the same value).
int device_cu_count();

typedef uint16_t bf16_t;  // raw bf16 bits
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float bf2f(bf16_t b) { return __builtin_bit_cast(float, (uint32_t)b << 16); }
// RNE, NaN-preserving (v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
// round an fp32 value to bf16 precision and keep it as fp32 (one "torch bf16 op" rounding)
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }

// two floats -> packed bf16x2 with ONE v_cvt_pk_bf16_f32 (RNE, NaN-preserving)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

// 8 bf16 (16 bytes) <-> 8 floats
struct Vec8 {
  float v[8];
};
__device__ __forceinline__ Vec8 unpack8(const i32x4 r) {
  Vec8 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint32_t w = (uint32_t)r[i];
    o.v[2 * i] = __builtin_bit_cast(float, w << 16);
    o.v[2 * i + 1] = __builtin_bit_cast(float, w & 0xffff0000u);
  }
  return o;
}
__device__ __forceinline__ Vec8 load8(const bf16_t* p) { return unpack8(*reinterpret_cast<const i32x4*>(p)); }
__device__ __forceinline__ void store8(bf16_t* p, const Vec8& x) {
  i32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = (int)pack2bf(x.v[2 * i], x.v[2 * i + 1]);
  *reinterpret_cast<i32x4*>(p) = r;
}

// ---- wave / block reductions ------------------------------------------------------------------
template <typename Op>
__device__ __forceinline__ float wave_reduce(float v, Op op) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = op(v, __shfl_xor(v, off, 64));
  return v;
}
struct OpSum { __device__ __forceinline__ float operator()(float a, float b) const { return a + b; } };
struct OpMax { __device__ __forceinline__ float operator()(float a, float b) const { return fmaxf(a, b); } };
struct OpMin { __device__ __forceinline__ float operator()(float a, float b) const { return fminf(a, b); } };

// All threads of the block get the reduced value.  smem: >= THREADS/64 floats, reusable after return.
template <int THREADS, typename Op>
__device__ __forceinline__ float block_reduce(float v, Op op, float* smem) {
  v = wave_reduce(v, op);
  if constexpr (THREADS > 64) {
    constexpr int NW = THREADS / 64;
    const int w = threadIdx.x >> 6;
    __syncthreads();  // protect smem reuse across consecutive reductions
    if ((threadIdx.x & 63) == 0) smem[w] = v;
    __syncthreads();
    float r = smem[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) r = op(r, smem[i]);
    v = r;
  }
  return v;
}

// ---- fp8 e4m3fn conversion ----------------------------------------------------------------------
// Software RNE conversion of a finite |x| <= 448 float (callers clamp first); NaN -> 0x7f.
__device__ __forceinline__ uint8_t f32_to_e4m3_sw(float x) {
  uint32_t u = __builtin_bit_cast(uint32_t, x);
  uint32_t sign = (u >> 24) & 0x80u;
  uint32_t a = u & 0x7fffffffu;
  if (a > 0x7f800000u) return (uint8_t)(sign | 0x7f);
  float ax = __builtin_bit_cast(float, a);
  if (ax >= 0.015625f) {  // normal range of e4m3 (2^-6 ..)
    // keep 3 mantissa bits: add rounding bias at bit 20, RNE
    uint32_t r = a + 0x7ffffu + ((a >> 20) & 1u);
    uint32_t e = (r >> 23) - 127 + 7;
    uint32_t m = (r >> 20) & 7u;
    uint32_t v = (e << 3) | m;
    if (v > 0x7e) v = 0x7e;  // saturate at 448
    return (uint8_t)(sign | v);
  }
  // subnormal: value = m * 2^-9, m in 0..7 (rounding to 8 gives the smallest normal 0x08)
  float scaled = ax * 512.0f;
  uint32_t m = (uint32_t)__builtin_rintf(scaled);
  return (uint8_t)(sign | m);
}

// two floats -> two e4m3 bytes in the low 16 bits
template <bool SW>
__device__ __forceinline__ uint32_t cvt2_e4m3(float a, float b) {
  if constexpr (SW) {
    return (uint32_t)f32_to_e4m3_sw(a) | ((uint32_t)f32_to_e4m3_sw(b) << 8);
  } else {
    return (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false) & 0xffffu;
  }
}

__device__ __forceinline__ float clamp448(float x) { return fmaxf(-448.0f, fminf(x, 448.0f)); }

// GELU variants with torch's fp32 formulas (ATen/native/cpu/Gelu.h)
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float gelu_tanh_f(float x) {
  const float kBeta = 0.79788456080286535588f;  // sqrt(2/pi)
  const float kKappa = 0.044715f;
  float inner = kBeta * (x + kKappa * x * x * x);
  return 0.5f * x * (1.0f + tanhf(inner));
}

// Cheap forms of the two GELUs for the fused paths (GEMM epilogues, gemm.hip; the activation deferred into the row quantiser,
// elementwise.hip): libm tanhf / erff cost ~40 VALU each.  |error| <= ~2e-7 absolute before the bf16 rounding.
__device__ __forceinline__ float fast_gelu_tanh(float x) {
  // 0.5 x (1 + tanh(u)) = x - x / (exp(2u) + 1), 2u log2(e) = x (c0 + c1 x^2): 4 plain VALU + exp2 + rcp.  Rounded to bf16 this
  // equals the textbook form for every bf16 input (checked exhaustively on the host).
  const float c0 = 0.79788456080286535588f * 2.885390081777927f, c1 = c0 * 0.044715f;
  const float e = __builtin_amdgcn_exp2f(x * __builtin_fmaf(x * x, c1, c0));
  return __builtin_fmaf(-x, __builtin_amdgcn_rcpf(e + 1.0f), x);
}
__device__ __forceinline__ float fast_gelu_erf(float x) {
  // Abramowitz-Stegun 7.1.26 (|err| <= 1.5e-7): erf(z) = 1 - poly(t) exp(-z^2), t = 1/(1 + p z), z >= 0.
  // The polynomial is evaluated with explicit FMAs (the file is built with -ffp-contract=off for the torch-faithful
  // op chains elsewhere; here contraction only changes an approximation's last bit).
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
  float poly = __builtin_fmaf(t, 1.061405429f, -1.453152027f);
  poly = __builtin_fmaf(t, poly, 1.421413741f);
  poly = __builtin_fmaf(t, poly, -0.284496736f);
  poly = __builtin_fmaf(t, poly, 0.254829592f);
  poly *= t;
  const float er = __builtin_fmaf(-poly, __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f), 1.0f);
  const float erf_x = x < 0.f ? -er : er;
  return 0.5f * x * (1.0f + erf_x);
}

// Bring-up switches exist only in builds made with -DFDM_DEBUG_BUILD (tools/micro); the product library has none: no
// environment variable can change what a kernel computes or stores.
#ifdef FDM_DEBUG_BUILD
extern bool g_fp8_sw_cvt;  // env FDM_FP8_SW_CVT: software e4m3 converter instead of v_cvt_pk_fp8_f32
#else
constexpr bool g_fp8_sw_cvt = false;
#endif

inline hipStream_t to_stream(fdm_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace fdm
