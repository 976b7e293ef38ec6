// Pure-MFMA micro-benchmark: measured ceiling of each MFMA flavour on this chip at the clock it sustains
// (random operands, 2 waves per SIMD, independent accumulators, no memory traffic in the loop).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ void __launch_bounds__(512, 2) k(const int* __restrict__ src, float* __restrict__ out, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = src[(t * 8 + i) & 0xffff]; b[i] = src[(t * 8 + i + 4096) & 0xffff]; }
  if constexpr (MODE <= 2) {
    f32x16 acc[8];
    for (int j = 0; j < 8; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    i32x16 iacc[8];
    for (int j = 0; j < 8; ++j) for (int e = 0; e < 16; ++e) iacc[j][e] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if constexpr (MODE == 0) acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        if constexpr (MODE == 1) {
          i32x4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a4), __builtin_bit_cast(bf16x8, b4), acc[j], 0, 0, 0);
        }
        if constexpr (MODE == 2) {
          i32x4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
          iacc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a4, b4, iacc[j], 0, 0, 0);
        }
      }
    }
    float s = 0.f;
    for (int j = 0; j < 8; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e] + (float)iacc[j][e];
    out[t] = s;
  } else {
    f32x4 acc[16];
    for (int j = 0; j < 16; ++j) for (int e = 0; e < 4; ++e) acc[j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if constexpr (MODE == 3) acc[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        if constexpr (MODE == 4) {
          i32x4 a4 = {a[0], a[1], a[2], a[3]}, b4 = {b[0], b[1], b[2], b[3]};
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a4), __builtin_bit_cast(bf16x8, b4), acc[j], 0, 0, 0);
        }
        if constexpr (MODE == 5) {
          long al = ((long)a[1] << 32) | (unsigned)a[0], bl = ((long)b[1] << 32) | (unsigned)b[0];
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(al, bl, acc[j], 0, 0, 0);
        }
      }
    }
    float s = 0.f;
    for (int j = 0; j < 16; ++j) for (int e = 0; e < 4; ++e) s += acc[j][e];
    out[t] = s;
  }
}

template <int MODE>
void run(const char* name, double flop_per_mfma, int per_iter, const int* src, float* out, int waves_per_simd) {
  const int iters = 20000, blocks = 256 * (waves_per_simd == 2 ? 1 : 1), threads = waves_per_simd == 2 ? 512 : 256;
  hipEvent_t s, e;
  hipEventCreate(&s); hipEventCreate(&e);
  k<MODE><<<blocks, threads>>>(src, out, 2000);
  hipDeviceSynchronize();
  hipEventRecord(s);
  k<MODE><<<blocks, threads>>>(src, out, iters);
  hipEventRecord(e);
  hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  double waves = (double)blocks * threads / 64;
  double tf = waves * iters * per_iter * flop_per_mfma / (ms * 1e-3) / 1e12;
  double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * per_iter * (threads / 256));
  printf("%-34s %d waves/SIMD: %8.1f TFLOP/s   %.1f cycles@2.4GHz per MFMA per SIMD   (%.2f ms)\n", name, threads / 256, tf, cyc, ms);
}

int main() {
  int* src; float* out;
  hipMalloc(&src, 65536 * 4); hipMalloc(&out, 256 * 512 * 4);
  std::vector<int> h(65536);
  srand(1);
  for (auto& v : h) {  // random fp8-safe bytes (no NaN patterns 0x7f/0xff), also fine as bf16/int8
    unsigned x = 0; for (int b = 0; b < 4; ++b) { unsigned byte = rand() & 0xff; if ((byte & 0x7f) == 0x7f) byte ^= 1; if ((byte & 0x7f) > 0x58) byte &= 0xbf; x |= byte << (8 * b); } v = (int)x; }
  hipMemcpy(src, h.data(), 65536 * 4, hipMemcpyHostToDevice);
  for (int w = 2; w >= 1; --w) {
    run<0>("mfma_scale 32x32x64 f8f6f4 (e4m3)", 2.0 * 32 * 32 * 64, 8, src, out, w);
    run<3>("mfma_scale 16x16x128 f8f6f4 (e4m3)", 2.0 * 16 * 16 * 128, 16, src, out, w);
    run<5>("mfma 16x16x32 fp8_fp8 (non-scaled)", 2.0 * 16 * 16 * 32, 16, src, out, w);
    run<1>("mfma 32x32x16 bf16", 2.0 * 32 * 32 * 16, 8, src, out, w);
    run<4>("mfma 16x16x32 bf16", 2.0 * 16 * 16 * 32, 16, src, out, w);
    run<2>("mfma 32x32x32 i8", 2.0 * 32 * 32 * 32, 8, src, out, w);
  }
  return 0;
}
