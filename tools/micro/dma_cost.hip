// What does feeding a 256x256 fp8 GEMM tile cost next to the MFMAs?  8 waves per CU (2 per SIMD), each step a wave
// issues 8 MFMAs (32x32x64 fp8) and NLOAD 1-KiB vector-memory instructions of one flavour:
//   MODE 0  global_load_lds_dwordx4 (LDS-DMA)      MODE 1  global_load_dwordx4 -> VGPR
//   MODE 2  buffer_load_dwordx4 -> VGPR            MODE 3  LDS-DMA + 12 ds_read_b128 (fragment reads) per step
//   MODE 4  register staging: global_load_dwordx4 -> VGPR, ds_write_b128 of the previous step's registers, + 12 ds_read_b128
//   W4: one wave per SIMD (256 threads), 16 MFMAs and twice the loads / fragment reads per wave and step
// Loads stay 2 steps in flight (counted vmcnt), sources are L2-resident.  Prints step time per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int MODE, int NLOAD, bool BARRIER, bool M16 = false, bool W4 = false>
__global__ void __launch_bounds__(W4 ? 256 : 512, W4 ? 1 : 2) k(const uint8_t* __restrict__ src, float* __restrict__ out, int iters) {
  __shared__ __attribute__((aligned(1024))) uint8_t smem[128 * 1024];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = ((const int*)src)[(tid * 8 + i) & 0xffff]; b[i] = ((const int*)src)[(tid * 8 + i + 4096) & 0xffff]; }
  f32x16 acc[8];
  for (int j = 0; j < 8; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  const uint8_t* g = src + ((size_t)(blockIdx.x & 31) << 18) + tid * 16;   // 256 KiB window per block-group, L2 resident
  i32x4 r[NLOAD > 8 ? NLOAD : 8];
  for (int i = 0; i < (NLOAD > 8 ? NLOAD : 8); ++i) r[i] = i32x4{0, 0, 0, 0};
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
  i32x4 fr[12];
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 4) {
      wait_vm<0>();                       // the registers loaded during the previous step
#pragma unroll
      for (int i = 0; i < NLOAD; ++i)
        *reinterpret_cast<i32x4*>(smem + ((it & 3) * 32768) + (i & 7) * 4096 + (tid & 255) * 16) = r[i];
    } else if constexpr (NLOAD > 0) wait_vm<NLOAD>();
    if constexpr (BARRIER) __builtin_amdgcn_s_barrier();
    const int off = (it & 15) * 8192 * 2;
#pragma unroll
    for (int i = 0; i < NLOAD; ++i) {
      if constexpr (MODE == 0 || MODE == 3)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + off + i * 8192),
                                         (__attribute__((address_space(3))) void*)(smem + ((it & 3) * 32768) + i * 8192 + wave * 1024), 16, 0, 0);
      if constexpr (MODE == 1 || MODE == 4) {
        i32x4 v;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(g + off + i * 8192) : "memory");
        r[i] = v;
      }
      if constexpr (MODE == 2) {
        i32x4 v;
        const int vo = (int)(((blockIdx.x & 31) << 18) + tid * 16 + i * 8192);
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(vo), "s"(rs), "s"(off) : "memory");
        r[i] = v;
      }
    }
    if constexpr (MODE == 3 || MODE == 4) {
#pragma unroll
      for (int rep = 0; rep < (W4 ? 2 : 1); ++rep)
#pragma unroll
        for (int i = 0; i < 12; ++i) fr[i] = *reinterpret_cast<const i32x4*>(smem + (((it + 1) & 3) * 32768) + i * 1024 + rep * 16384 + ((lane * 16) ^ ((i & 3) << 4)));
    }
    if constexpr (!M16) {
#pragma unroll
      for (int j = 0; j < (W4 ? 16 : 8); ++j) {
        acc[j & 7] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[j & 7], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
      }
    } else {
      // same flops per step with 16x16x128: 16 MFMAs on 4-register accumulators (32 of them live in acc[8][16])
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        f32x4 c = {acc[j >> 1][(j & 1) * 8 + 0], acc[j >> 1][(j & 1) * 8 + 1], acc[j >> 1][(j & 1) * 8 + 2], acc[j >> 1][(j & 1) * 8 + 3]};
        c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        acc[j >> 1][(j & 1) * 8 + 0] = c[0]; acc[j >> 1][(j & 1) * 8 + 1] = c[1]; acc[j >> 1][(j & 1) * 8 + 2] = c[2]; acc[j >> 1][(j & 1) * 8 + 3] = c[3];
      }
    }
    if constexpr (MODE == 3 || MODE == 4) {
#pragma unroll
      for (int i = 0; i < 12; ++i) asm volatile("" ::"v"(fr[i]));
    }
    if constexpr (MODE == 1 || MODE == 2 || MODE == 4) {
      // keep the loaded registers live across the step boundary without consuming them early
#pragma unroll
      for (int i = 0; i < NLOAD; ++i) asm volatile("" : "+v"(r[i]));
    }
  }
  wait_vm<0>();
  float s = 0.f;
  for (int j = 0; j < 8; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
  for (int i = 0; i < 8; ++i) s += (float)r[i][0];
  out[blockIdx.x * 512 + tid] = s + smem[tid];
}

template <int MODE, int NLOAD, bool BARRIER, bool M16 = false, bool W4 = false>
void run(const char* name, const uint8_t* src, float* out) {
  const int iters = 4000;
  hipEvent_t s, e;
  hipEventCreate(&s); hipEventCreate(&e);
  k<MODE, NLOAD, BARRIER, M16, W4><<<256, W4 ? 256 : 512>>>(src, out, 500);
  hipDeviceSynchronize();
  hipEventRecord(s);
  k<MODE, NLOAD, BARRIER, M16, W4><<<256, W4 ? 256 : 512>>>(src, out, iters);
  hipEventRecord(e);
  hipEventSynchronize(e);
  float ms; hipEventElapsedTime(&ms, s, e);
  const double ns_step = ms * 1e6 / iters;
  const double tf = 256.0 * 8 * iters * 8 * 131072.0 / (ms * 1e-3) / 1e12;
  printf("%-44s loads/wave/step %d barrier %d: %7.1f ns/step  %7.1f TF/s\n", name, NLOAD, (int)BARRIER, ns_step, tf);
}

int main() {
  uint8_t* src; float* out;
  hipMalloc(&src, 16 << 20); hipMalloc(&out, 256 * 512 * 4);
  std::vector<uint8_t> h(16 << 20);
  uint64_t x = 88172645463325252ull;
  for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; uint8_t b = (uint8_t)(x >> 32); if ((b & 0x7f) >= 0x78) b &= 0xc7; v = b; }
  hipMemcpy(src, h.data(), h.size(), hipMemcpyHostToDevice);
  run<0, 0, false>("MFMA only", src, out);
  run<0, 0, true>("MFMA only", src, out);
  run<0, 2, false>("LDS-DMA", src, out);
  run<0, 4, false>("LDS-DMA", src, out);
  run<0, 4, true>("LDS-DMA", src, out);
  run<0, 8, false>("LDS-DMA", src, out);
  run<1, 2, false>("global_load -> VGPR", src, out);
  run<1, 4, false>("global_load -> VGPR", src, out);
  run<1, 8, false>("global_load -> VGPR", src, out);
  run<2, 4, false>("buffer_load -> VGPR", src, out);
  run<2, 8, false>("buffer_load -> VGPR", src, out);
  run<3, 0, false>("12 ds_read_b128 only", src, out);
  run<3, 4, false>("LDS-DMA + 12 ds_read_b128", src, out);
  run<3, 4, true>("LDS-DMA + 12 ds_read_b128", src, out);
  run<0, 0, false, true>("16x16x128: MFMA only", src, out);
  run<0, 4, true, true>("16x16x128: LDS-DMA", src, out);
  run<3, 4, true, true>("16x16x128: LDS-DMA + 12 ds_read_b128", src, out);
  run<3, 4, true>("32x32x64: LDS-DMA + 12 ds_read_b128", src, out);
  run<3, 4, true, true>("16x16x128: LDS-DMA + 12 ds_read_b128", src, out);
  printf("-- register staging against LDS-DMA, same fragment reads\n");
  run<3, 4, true>("8 waves: LDS-DMA + 12 ds_read", src, out);
  run<4, 4, true>("8 waves: VGPR staging + 12 ds_read", src, out);
  run<0, 0, true, false, true>("W4: MFMA only", src, out);
  run<3, 8, true, false, true>("W4: LDS-DMA + 24 ds_read", src, out);
  run<4, 8, true, false, true>("W4: VGPR staging + 24 ds_read", src, out);
  run<3, 4, true, false, true>("W4: LDS-DMA x4 + 24 ds_read", src, out);
  run<4, 4, true, false, true>("W4: VGPR staging x4 + 24 ds_read", src, out);
  return 0;
}
