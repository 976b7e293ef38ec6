// gfx950 software hazard, measured (profiles/r04_experiments.md): a VGPR written by a VALU instruction needs one wait state before
// v_readfirstlane_b32 reads it -- hipcc inserts it, hand-written assembly must.  Variant a (no pad) stores the SGPR the SALU then sees:
// the stale register content, not tid >> 6.  Build: hipcc --offload-arch=gfx950 -O3 readlane_hazard_probe.hip -o readlane_hazard_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
struct Args { uint32_t* out; uint32_t a, b; uint32_t pad[44]; };
__global__ void __launch_bounds__(256, 1) ka(const Args args) {
  const auto ka = __builtin_amdgcn_kernarg_segment_ptr();
  asm volatile(
      "s_mov_b64 s[8:9], %0\n\t"
      "s_mov_b32 s10, %1\n\t"
      "v_and_b32 v0, 0xff, %2\n\t"
      "s_load_dwordx2 s[16:17], s[8:9], 0\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_and_b32 v1, 63, v0\n\t"
      "v_lshrrev_b32 v4, 6, v0\n\t"
      "v_readfirstlane_b32 s11, v4\n\t"
      "s_nop 4\n\t"
      "s_lshl_b32 s52, s10, 13\n\t"
      "v_lshlrev_b32 v6, 2, v0\n\t"
      "v_add_u32 v6, s52, v6\n\t"
      "v_mov_b32 v5, s11\n\t"
      "global_store_dword v6, v5, s[16:17]\n\t"
      "global_store_dword v6, v4, s[16:17] offset:1024\n\t"
      "global_store_dword v6, v0, s[16:17] offset:2048\n\t"
      "s_waitcnt vmcnt(0)\n\t"
      :: "s"(ka), "s"(blockIdx.x), "v"(threadIdx.x) : "v1", "v4", "v5", "v6", "s8", "s9", "s10", "s11", "s16", "s17", "s20", "s52", "s53", "memory");
}
__global__ void __launch_bounds__(256, 1) kb(const Args args) {
  const auto ka = __builtin_amdgcn_kernarg_segment_ptr();
  asm volatile(
      "s_mov_b64 s[8:9], %0\n\t"
      "s_mov_b32 s10, %1\n\t"
      "v_and_b32 v0, 0xff, %2\n\t"
      "s_load_dwordx2 s[16:17], s[8:9], 0\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_and_b32 v1, 63, v0\n\t"
      "v_lshrrev_b32 v4, 6, v0\n\t"
      "v_readfirstlane_b32 s20, v4\n\t"
      "s_nop 15\n\t"
      "s_nop 15\n\t"
      "s_lshl_b32 s52, s10, 13\n\t"
      "s_lshl_b32 s53, s20, 11\n\t"
      "s_add_u32 s52, s52, s53\n\t"
      "v_lshlrev_b32 v6, 2, v1\n\t"
      "v_add_u32 v6, s52, v6\n\t"
      "v_mov_b32 v5, s10\n\t"
      "global_store_dword v6, v5, s[16:17]\n\t"
      "s_waitcnt vmcnt(0)\n\t"
      :: "s"(ka), "s"(blockIdx.x), "v"(threadIdx.x) : "v1", "v4", "v5", "v6", "s8", "s9", "s10", "s11", "s16", "s17", "s20", "s52", "s53", "memory");
}

int main(int argc, char** argv) {
  uint32_t* d;
  const size_t n = 256 * 4 * 8 * 64;
  hipMalloc(&d, n * 4);
  Args a{};
  a.out = d;
  uint32_t* h = new uint32_t[n];
  const char w = argc > 1 ? argv[1][0] : 'a';
  hipMemset(d, 0xee, n * 4);
  if (w == 'a') ka<<<256, 256>>>(a);
  if (w == 'b') kb<<<256, 256>>>(a);
  printf("variant %c: launch %d sync %d;\n", w, (int)hipGetLastError(), (int)hipDeviceSynchronize());
  hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
  if (w == 'a') for (int b : {0, 1, 255}) { printf(" block %d:", b); for (int t : {0, 63, 64, 128, 255}) printf("  tid %d: s11 %08x v4 %08x v0 %08x", t, h[b * 2048 + t], h[b * 2048 + 256 + t], h[b * 2048 + 512 + t]); printf("\n"); }
  else printf(" h[0] %08x, block 1 wave 0..3: %08x %08x %08x %08x\n", h[0], h[4 * 512], h[5 * 512], h[6 * 512], h[7 * 512]);
  return 0;
}
