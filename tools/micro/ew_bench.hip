// Standalone timing of the fp8 element-wise hot kernels (device-side: hip events around back-to-back launches).
#include "../../fastdm_amd/csrc/elementwise.hip"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
namespace fdm { thread_local int g_last_hip_error = 0; int device_cu_count() { return 256; } }
static uint16_t f2b(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
// ceiling probes: same traffic as the row quantiser (16 B in, 8 B out per lane) without any reduction
__global__ void __launch_bounds__(256) stream_cvt_kernel(uint8_t* __restrict__ q, const uint16_t* __restrict__ x, int64_t nvec, float rinv) {
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * 256) {
    const fdm::i32x4 w = *reinterpret_cast<const fdm::i32x4*>(x + v * 8);
    uint32_t c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const uint32_t u = (uint32_t)w[k];
      c[k] = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(__builtin_bit_cast(float, u << 16) * rinv, __builtin_bit_cast(float, u & 0xffff0000u) * rinv, 0, false) & 0xffffu;
    }
    fdm::i32x2 o; o[0] = (int)(c[0] | (c[1] << 16)); o[1] = (int)(c[2] | (c[3] << 16));
    *reinterpret_cast<fdm::i32x2*>(q + v * 8) = o;
  }
}
__global__ void __launch_bounds__(256) stream_copy16_kernel(uint16_t* __restrict__ y, const uint16_t* __restrict__ x, int64_t nvec) {
  for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * 256)
    *reinterpret_cast<fdm::i32x4*>(y + v * 8) = *reinterpret_cast<const fdm::i32x4*>(x + v * 8);
}
int main() {
  const int64_t R = 8704, W = 15360;
  uint16_t *x, *y, *sc, *sh; uint8_t* q; float* s;
  hipMalloc(&x, R * W * 2); hipMalloc(&y, R * 3072 * 2); hipMalloc(&q, R * W); hipMalloc(&s, R * 4); hipMalloc(&sc, 3072 * 2); hipMalloc(&sh, 3072 * 2);
  std::vector<uint16_t> h(R * W);
  uint64_t r = 88172645463325252ull;
  {   // gaussian activations (a 2^20-entry Box-Muller table, tiled with a rolling offset): continuous values like real ones
    std::vector<uint16_t> tab(1 << 20);
    for (auto& e : tab) {
      r ^= r << 13; r ^= r >> 7; r ^= r << 17; const double u1 = ((r >> 11) + 0.5) / 9007199254740992.0;
      r ^= r << 13; r ^= r >> 7; r ^= r << 17; const double u2 = ((r >> 11) + 0.5) / 9007199254740992.0;
      e = f2b((float)(sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2)));
    }
    for (size_t i = 0; i < h.size(); ++i) h[i] = tab[(i * 2654435761u + (i >> 20) * 7919u) & ((1 << 20) - 1)];
  }
  hipMemcpy(x, h.data(), R * W * 2, hipMemcpyHostToDevice); hipMemcpy(y, h.data(), R * 3072 * 2, hipMemcpyHostToDevice);
  hipMemcpy(sc, h.data(), 3072 * 2, hipMemcpyHostToDevice); hipMemcpy(sh, h.data() + 5000, 3072 * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](const char* name, double bytes, auto f) {
    for (int i = 0; i < 3; ++i) f();
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < 20; ++i) f();
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    printf("%-40s %8.1f us  %6.2f TB/s\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e12);
  };
  for (int64_t rows : {8192, 8704, 512}) {
    char nm[64]; snprintf(nm, 64, "ln_modulate+fp8 [%ld,3072]", (long)rows);
    time(nm, rows * 3072 * 3.0, [&] { fdm_ln_modulate(nullptr, q, s, nullptr, 1, y, sc, sh, 3072, nullptr, nullptr, nullptr, rows, 3072, rows, 1e-6f, (getenv("EW_S1") ? 1 : 0), nullptr); });
    snprintf(nm, 64, "ln_modulate+int8 (generic) [%ld,3072]", (long)rows);
    time(nm, rows * 3072 * 3.0, [&] { fdm_ln_modulate(nullptr, q, s, nullptr, 3, y, sc, sh, 3072, nullptr, nullptr, nullptr, rows, 3072, rows, 1e-6f, (getenv("EW_S1") ? 1 : 0), nullptr); });
  }
  time("quant fp8 cat2 [8704,3072+12288]", R * 15360 * 3.0, [&] { fdm_quant_rowwise_cat2(q, s, nullptr, 1, y, 3072, 3072, x, 12288, W, R, nullptr); });
  time("quant int8 (generic) cat2", R * 15360 * 3.0, [&] { fdm_quant_rowwise_cat2(q, s, nullptr, 3, y, 3072, 3072, x, 12288, W, R, nullptr); });
  time("quant fp8 [8704,12288]", R * 12288 * 3.0, [&] { fdm_quant_rowwise_cat2(q, s, nullptr, 1, x, 12288, W, nullptr, 0, 0, R, nullptr); });
  time("quant fp8 [8704,3072]", R * 3072 * 3.0, [&] { fdm_quant_rowwise_cat2(q, s, nullptr, 1, y, 3072, 3072, nullptr, 0, 0, R, nullptr); });
  time("quant int8 (generic) [8704,3072]", R * 3072 * 3.0, [&] { fdm_quant_rowwise_cat2(q, s, nullptr, 3, y, 3072, 3072, nullptr, 0, 0, R, nullptr); });
  {   // q|k|v planes [3][8704][3072]; norm + rope of q and k in place
    uint16_t *planes, *wq, *cache; hipMalloc(&planes, 3 * R * 3072 * 2); hipMalloc(&wq, 256); hipMalloc(&cache, R * 128 * 2);
    for (int i = 0; i < 3; ++i) hipMemcpy(planes + i * R * 3072, h.data() + i * 1000, R * 3072 * 2, hipMemcpyHostToDevice);
    hipMemcpy(wq, h.data(), 256, hipMemcpyHostToDevice); hipMemcpy(cache, h.data() + 777, R * 128 * 2, hipMemcpyHostToDevice);
    time("qk_norm_rope_inplace flat kernel (r02)", R * 3072 * 2 * 4.0, [&] { fdm::qk_norm_rope_inplace_kernel<128><<<(unsigned)((R * 48 * 16 + 255) / 256), 256>>>(planes, wq, wq, cache, R, R, R, 3072, R * 3072, 0, 24, 1e-6f); });
    time("qk_norm_rope_inplace [8704, 2x24x128]", R * 3072 * 2 * 4.0, [&] { fdm_qk_norm_rope_inplace(planes, wq, wq, cache, R, R, R, 3072, R * 3072, 0, 24, 128, 1e-6f, nullptr); });
  }
  for (int g : {2048, 4096, 8704, 32768}) {
    char nm[64]; snprintf(nm, 64, "stream cvt 16B->8B grid %d", g);
    time(nm, R * W * 3.0, [&] { stream_cvt_kernel<<<g, 256>>>(q, x, R * W / 8, 0.37f); });
  }
  uint16_t* x2; hipMalloc(&x2, R * W * 2);
  time("stream copy 16B->16B grid 8192 (267 MB)", R * W * 4.0, [&] { stream_copy16_kernel<<<8192, 256>>>(x2, x, R * W / 8); });
  return 0;
}
