// Do MFMA and VALU work of two waves on the same SIMD overlap on gfx950?  (What bounds the attention kernel: its softmax is
// ~2000 VALU cycles per 128-key tile per wave next to ~2048 MFMA cycles.)
// A block is 8 waves: waves w and w+4 share SIMD w.  Roles: M = bf16 32x32x16 MFMA loop (8 independent accumulators),
// V = softmax-like VALU loop (fma, exp2, add, max on 16 independent values), B = both interleaved in one wave, - = exit.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void mfma8(f32x16 (&acc)[8], bf16x8 a, bf16x8 b) {
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
}
__device__ __forceinline__ void valu16(float (&v)[16], float (&s)[16], float (&m)[16], float c) {
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    float p = __builtin_amdgcn_exp2f(__builtin_fmaf(v[j], c, -m[j]));
    s[j] += p;
    m[j] = fmaxf(m[j], p);
    v[j] = p * 0.5f + v[j];
  }
}

// role: 0 exit, 1 MFMA, 2 VALU, 3 both in one wave
__global__ void __launch_bounds__(512, 1) k(const int* __restrict__ src, float* __restrict__ out, int iters, int role_lo, int role_hi) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int wave = threadIdx.x >> 6;
  const int role = wave < 4 ? role_lo : role_hi;
  if (role == 0) return;
  i32x4 a4, b4;
  for (int i = 0; i < 4; ++i) { a4[i] = src[(t * 4 + i) & 0xffff]; b4[i] = src[(t * 4 + i + 4096) & 0xffff]; }
  const bf16x8 a = __builtin_bit_cast(bf16x8, a4), b = __builtin_bit_cast(bf16x8, b4);
  f32x16 acc[8];
  for (int j = 0; j < 8; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  float v[16], s[16], m[16];
  for (int j = 0; j < 16; ++j) { v[j] = (float)(src[(t + j) & 0xffff] & 0xff) * 0.01f; s[j] = 0.f; m[j] = 0.f; }
  const float c = 0.001f * (float)(src[0] & 3);
  if (role == 1) {
    for (int it = 0; it < iters; ++it) mfma8(acc, a, b);
  } else if (role == 2) {
    for (int it = 0; it < iters; ++it) valu16(v, s, m, c);
  } else {
    for (int it = 0; it < iters; ++it) { mfma8(acc, a, b); valu16(v, s, m, c); }
  }
  float r = 0.f;
  for (int j = 0; j < 8; ++j) for (int e = 0; e < 16; ++e) r += acc[j][e];
  for (int j = 0; j < 16; ++j) r += v[j] + s[j] + m[j];
  out[t] = r;
}

static float run(const int* src, float* out, int iters, int lo, int hi) {
  hipEvent_t s, e;
  hipEventCreate(&s); hipEventCreate(&e);
  k<<<256, 512>>>(src, out, 200, lo, hi);
  hipDeviceSynchronize();
  hipEventRecord(s);
  k<<<256, 512>>>(src, out, iters, lo, hi);
  hipEventRecord(e);
  hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  return ms;
}

int main() {
  int* src; float* out;
  hipMalloc(&src, 65536 * 4); hipMalloc(&out, 256 * 512 * 4);
  int* h = new int[65536];
  for (int i = 0; i < 65536; ++i) h[i] = (i * 2654435761u) & 0x3f803f80;
  hipMemcpy(src, h, 65536 * 4, hipMemcpyHostToDevice);
  const int iters = 20000;
  const char* names[4] = {"-", "M", "V", "B"};
  const int cfg[][2] = {{1, 0}, {2, 0}, {1, 1}, {2, 2}, {1, 2}, {3, 0}, {3, 3}};
  for (auto& c : cfg) {
    float ms = run(src, out, iters, c[0], c[1]);
    // per iteration per wave: 8 MFMA (8 x 32 = 256 matrix cycles) and/or 16 x (fma, exp2, add, max, fma) VALU
    printf("waves 0-3: %s   waves 4-7: %s   %8.3f ms   %.0f cycles@2.4GHz per iteration\n", names[c[0]], names[c[1]], ms,
           ms * 1e-3 * 2.4e9 / iters);
  }
  return 0;
}
