// Which K position of operand A does v_mfma_f32_16x16x128_f8f6f4 pair with which K position of operand B?
// Lane (i, q) register byte p is given "logical" K = 32 q + p in BOTH operands; one-hot A[0][k0] x one-hot B[k1][0].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
  const int lane = threadIdx.x, i = lane & 15, q = lane >> 4;
  for (int k0 = 0; k0 < 128; ++k0)
    for (int k1 = 0; k1 < 128; ++k1) {
      i32x8 a, b;
#pragma unroll
      for (int v = 0; v < 8; ++v) {
        a[v] = (i == 0 && (k0 >> 5) == q && ((k0 & 31) >> 2) == v) ? (0x38 << (8 * (k0 & 3))) : 0;
        b[v] = (i == 0 && (k1 >> 5) == q && ((k1 & 31) >> 2) == v) ? (0x38 << (8 * (k1 & 3))) : 0;
      }
      f32x4 c = {0.f, 0.f, 0.f, 0.f};
      c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      out[(size_t)(k0 * 128 + k1) * 64 + lane] = c[0];
    }
}
int main() {
  float* d; hipMalloc(&d, 128 * 128 * 4 * 64);
  k<<<1, 64>>>(d);
  std::vector<float> hh(128 * 128 * 64), h(128 * 128);
  hipMemcpy(hh.data(), d, hh.size() * 4, hipMemcpyDeviceToHost);
  for (int x = 0; x < 128 * 128; ++x) h[x] = hh[(size_t)x * 64];
  int bad = 0;
  for (int k0 = 0; k0 < 128; ++k0) {
    int n = 0, first = -1;
    for (int k1 = 0; k1 < 128; ++k1) if (h[k0 * 128 + k1] != 0.f) { ++n; if (first < 0) first = k1; }
    if (n != 1 || first != k0) { if (bad < 40) printf("A k=%d pairs with B k=%d (%d matches)\n", k0, first, n); ++bad; }
  }
  printf("%d of 128 K positions do not pair position-wise\n", bad);
  for (int k0 : {0, 31, 32, 33, 48, 64, 100}) {
    printf("k0=%3d:", k0);
    for (int k1 = 0; k1 < 128; ++k1) if (h[k0 * 128 + k1] != 0.f) printf(" [%d]=%g", k1, h[k0 * 128 + k1]);
    printf("\n");
  }
  return 0;
}
