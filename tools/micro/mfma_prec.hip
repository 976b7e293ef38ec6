// Is v_mfma_f32_16x16x128_f8f6f4 as exact as v_mfma_f32_32x32x64_f8f6f4?  One MFMA each on random e4m3 operands (lane (i, q)
// holds K = 32 q .. 32 q + 31 of row i), compared with the exact dot products on the host.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k16(const int* a, const int* b, float* out) {
  const int lane = threadIdx.x;
  i32x8 av, bv;
  for (int v = 0; v < 8; ++v) { av[v] = a[lane * 8 + v]; bv[v] = b[lane * 8 + v]; }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
}
__global__ void k16asm(const int* a, const int* b, float* out) {
  const int lane = threadIdx.x;
  i32x8 av, bv;
  for (int v = 0; v < 8; ++v) { av[v] = a[lane * 8 + v]; bv[v] = b[lane * 8 + v]; }
  float c0, c1, c2, c3;
  asm volatile("v_accvgpr_write_b32 a0, 0\n v_accvgpr_write_b32 a1, 0\n v_accvgpr_write_b32 a2, 0\n v_accvgpr_write_b32 a3, 0\n s_nop 4\n"
               "v_mfma_f32_16x16x128_f8f6f4 a[0:3], %4, %5, a[0:3]\n s_nop 15\n s_nop 15\n"
               "v_accvgpr_read_b32 %0, a0\n v_accvgpr_read_b32 %1, a1\n v_accvgpr_read_b32 %2, a2\n v_accvgpr_read_b32 %3, a3\n"
               : "=&v"(c0), "=&v"(c1), "=&v"(c2), "=&v"(c3) : "v"(av), "v"(bv) : "a0", "a1", "a2", "a3");
  out[lane * 4 + 0] = c0; out[lane * 4 + 1] = c1; out[lane * 4 + 2] = c2; out[lane * 4 + 3] = c3;
}
__global__ void k32(const int* a, const int* b, float* out) {
  const int lane = threadIdx.x;
  i32x8 av, bv;
  for (int v = 0; v < 8; ++v) { av[v] = a[lane * 8 + v]; bv[v] = b[lane * 8 + v]; }
  f32x16 c; for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  for (int r = 0; r < 16; ++r) out[lane * 16 + r] = c[r];
}
static double e4m3(uint8_t v) {
  const int sgn = v >> 7, e = (v >> 3) & 15, m = v & 7;
  const double mag = e == 0 ? m * 0.001953125 : (1.0 + m / 8.0) * pow(2.0, e - 7);
  return sgn ? -mag : mag;
}
int main() {
  std::vector<uint8_t> ha(64 * 32), hb(64 * 32);
  uint64_t x = 88172645463325252ull;
  for (int mode = 0; mode < 2; ++mode) {
    for (auto* h : {&ha, &hb}) for (auto& v : *h) {
      x ^= x << 13; x ^= x >> 7; x ^= x << 17; uint8_t c = (uint8_t)(x >> 32);
      if (mode == 0) { if ((c & 0x7f) >= 0x78) c &= 0xc7; }            // any finite code
      else c = (c & 0x87) | (uint8_t)((6 + ((x >> 40) % 3)) << 3);       // exponents 6..8 only: magnitudes 0.5 .. 3.75
      v = c;
    }
    int *a, *b; float* o;
    hipMalloc(&a, 2048); hipMalloc(&b, 2048); hipMalloc(&o, 64 * 16 * 4);
    hipMemcpy(a, ha.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 2048, hipMemcpyHostToDevice);
    std::vector<float> r(64 * 16);
    // 16x16x128: lane l = (i = l % 16, q = l / 16); D[row = 4 (l / 16) + r][col = l % 16], rows <- operand a, cols <- operand b
    k16<<<1, 64>>>(a, b, o); hipMemcpy(r.data(), o, 64 * 4 * 4, hipMemcpyDeviceToHost);
    double worst = 0, mean = 0;
    for (int l = 0; l < 64; ++l) for (int rr = 0; rr < 4; ++rr) {
      const int row = 4 * (l / 16) + rr, col = l % 16;
      double ref = 0;
      for (int q = 0; q < 4; ++q) for (int p = 0; p < 32; ++p) ref += e4m3(ha[(row + 16 * q) * 32 + p]) * e4m3(hb[(col + 16 * q) * 32 + p]);
      const double err = fabs(r[l * 4 + rr] - ref) / (fabs(ref) + 1e-30);
      worst = fmax(worst, err); mean += err / 256;
    }
    printf("mode %d 16x16x128: relative error mean %.3g max %.3g\n", mode, mean, worst);
    k16asm<<<1, 64>>>(a, b, o); hipMemcpy(r.data(), o, 64 * 4 * 4, hipMemcpyDeviceToHost);
    worst = 0; mean = 0; double byr[4] = {0, 0, 0, 0};
    for (int l = 0; l < 64; ++l) for (int rr = 0; rr < 4; ++rr) {
      const int row = 4 * (l / 16) + rr, col = l % 16;
      double ref = 0;
      for (int q = 0; q < 4; ++q) for (int p = 0; p < 32; ++p) ref += e4m3(ha[(row + 16 * q) * 32 + p]) * e4m3(hb[(col + 16 * q) * 32 + p]);
      const double err = fabs(r[l * 4 + rr] - ref) / (fabs(ref) + 1e-30);
      worst = fmax(worst, err); mean += err / 256; byr[rr] += err / 64;
    }
    printf("mode %d 16x16x128 asm/AGPR: relative error mean %.3g max %.3g  by register %.3g %.3g %.3g %.3g\n", mode, mean, worst, byr[0], byr[1], byr[2], byr[3]);
    // 32x32x64: lane l = (i = l % 32, g = l / 32) holds K = 32 g ..; D[row = 8 (r / 4) + 4 (l / 32) + r % 4][col = l % 32]
    k32<<<1, 64>>>(a, b, o); hipMemcpy(r.data(), o, 64 * 16 * 4, hipMemcpyDeviceToHost);
    worst = 0; mean = 0;
    for (int l = 0; l < 64; ++l) for (int rr = 0; rr < 16; ++rr) {
      const int row = 8 * (rr / 4) + 4 * (l / 32) + rr % 4, col = l % 32;
      double ref = 0;
      for (int g = 0; g < 2; ++g) for (int p = 0; p < 32; ++p) ref += e4m3(ha[(row + 32 * g) * 32 + p]) * e4m3(hb[(col + 32 * g) * 32 + p]);
      const double err = fabs(r[l * 16 + rr] - ref) / (fabs(ref) + 1e-30);
      worst = fmax(worst, err); mean += err / 1024;
    }
    printf("mode %d 32x32x64 : relative error mean %.3g max %.3g\n", mode, mean, worst);
  }
  return 0;
}
