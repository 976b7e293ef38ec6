// Standalone fp8 GEMM micro-benchmark: compiles gemm.hip directly so that kernel variants (-D macros) can be
// A/B-tested in one GPU call without Python.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off
//   tools/micro/gemm_bench.hip -o tools/micro/gemm_bench[_variant] [-DFDM_...]
#include "../../fastdm_amd/csrc/gemm.hip"
#include <cstdio>
#include <cstring>
namespace fdm { thread_local int g_last_hip_error = 0; int device_cu_count() { return 256; } }
#include <vector>

static uint64_t rng = 0x9E3779B97F4A7C15ull;
static inline uint32_t nextu() {
  rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17;
  return (uint32_t)(rng >> 32);
}

int main(int argc, char** argv) {
  struct Shape { const char* name; int64_t M, K, N; int epi; int cnt; };
  const bool sweep = argc > 1 && argv[1][0] == 's';
  std::vector<Shape> shapes;
  if (sweep) {
    static char names[16][32];
    int n = 0;
    for (int epi : {0, 1, 3})
      for (int64_t K : {512, 1024, 2048, 4096, 8192}) {
        snprintf(names[n], 32, "8192^2 K=%ld epi%d", (long)K, epi);
        shapes.push_back({names[n], 8192, K, 8192, epi, 0});
        ++n;
      }
  } else shapes = {{"dbl qkv img", 8192, 3072, 9216, 0, 19},   {"dbl to_out img", 8192, 3072, 3072, 3, 19},
                               {"dbl ff1 img", 8192, 3072, 12288, 1, 19},  {"dbl ff2 img", 8192, 12288, 3072, 3, 19},
                               {"sgl qkv", 8704, 3072, 9216, 0, 38},       {"sgl mlp", 8704, 3072, 12288, 2, 38},
                               {"sgl out", 8704, 15360, 3072, 3, 38}};
  if (argc > 5 && argv[1][0] == 'c') {   // custom: c M K N epi [M K N epi ...]
    shapes.clear();
    static char cn[32][48];
    for (int a = 2, n = 0; a + 3 < argc && n < 32; a += 4, ++n) {
      snprintf(cn[n], 48, "M%s K%s N%s e%s", argv[a], argv[a + 1], argv[a + 2], argv[a + 3]);
      shapes.push_back({cn[n], atol(argv[a]), atol(argv[a + 1]), atol(argv[a + 2]), atoi(argv[a + 3]), 0});
    }
  }
  const bool nohash = argc > 1 && (argv[1][0] == 'n' || argv[1][0] == 'c');
  const int64_t maxA = 8704 * 15360, maxB = 12288 * 3072 * 2, maxC = 8704 * 12288;
  uint8_t *A, *B; uint16_t *C, *R, *G, *bias; float *sa, *sb;
  hipMalloc(&A, maxA); hipMalloc(&B, maxB); hipMalloc(&C, maxC * 2); hipMalloc(&R, maxC * 2);
  hipMalloc(&G, 12288 * 2); hipMalloc(&bias, 12288 * 2); hipMalloc(&sa, 8704 * 4); hipMalloc(&sb, 12288 * 4);
  {
    std::vector<uint8_t> h(maxA);
    for (auto& x : h) { uint8_t v = (uint8_t)nextu(); if ((v & 0x7f) >= 0x78) v &= 0xc7; x = v; }   // finite e4m3, random sign/exp
    hipMemcpy(A, h.data(), maxA, hipMemcpyHostToDevice);
    h.resize(maxB);
    for (auto& x : h) { uint8_t v = (uint8_t)nextu(); if ((v & 0x7f) >= 0x78) v &= 0xc7; x = v; }
    hipMemcpy(B, h.data(), maxB, hipMemcpyHostToDevice);
    std::vector<uint16_t> r(maxC);
    for (auto& x : r) x = 0x3c00 + (nextu() & 0x3ff);
    hipMemcpy(R, r.data(), maxC * 2, hipMemcpyHostToDevice);
    hipMemcpy(G, r.data(), 12288 * 2, hipMemcpyHostToDevice);
    hipMemcpy(bias, r.data() + 20000, 12288 * 2, hipMemcpyHostToDevice);
    std::vector<float> s(12288, 1e-4f);
    hipMemcpy(sa, s.data(), 8704 * 4, hipMemcpyHostToDevice);
    hipMemcpy(sb, s.data(), 12288 * 4, hipMemcpyHostToDevice);
  }
#ifdef FDM_GEMM_STAMPS
  hipMalloc(&fdm::g_gemm_stamps, 2048 * 4 * 8);
  hipMemset(fdm::g_gemm_stamps, 0, 2048 * 4 * 8);
#endif
  const bool use_int8 = getenv("GEMM_DT") && !strcmp(getenv("GEMM_DT"), "int8");
  int32_t *azp, *azp_adj;
  hipMalloc(&azp, 8704 * 4); hipMalloc(&azp_adj, 12288 * 4);
  hipMemset(azp, 0, 8704 * 4); hipMemset(azp_adj, 1, 12288 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  double tot = 0;
  for (auto& s : shapes) {
    auto run = [&]() {
      if (use_int8) return fdm_gemm_w8a8_int8(C, A, B, sa, sb, azp_adj, azp, bias, s.M, s.N, s.K, s.K, s.K, s.N, s.epi, G, s.N, R, s.M, nullptr);
      return fdm_gemm_w8a8_fp8(C, A, B, sa, sb, bias, s.M, s.N, s.K, s.K, s.K, s.N, s.epi, G, s.N, R, s.M, nullptr);
    };
    for (int i = 0; i < 3; ++i) if (run() != 0) { printf("launch failed\n"); return 1; }
    hipDeviceSynchronize();
    const int iters = 20;
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) run();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= iters;
    uint64_t hsh = 1469598103934665603ull;
    if (!nohash && !sweep) {
      std::vector<uint16_t> hc((size_t)s.M * s.N);
      hipMemcpy(hc.data(), C, hc.size() * 2, hipMemcpyDeviceToHost);
      for (size_t i = 0; i < hc.size(); i += 7) hsh = (hsh ^ hc[i]) * 1099511628211ull;
    }
    printf("%-16s %8.1f us %8.1f TF/s  hash %016llx\n", s.name, ms * 1e3, 2.0 * s.M * s.N * s.K / (ms * 1e-3) / 1e12, (unsigned long long)hsh);
    tot += ms * s.cnt;
#ifdef FDM_GEMM_STAMPS
    {
      hipDeviceSynchronize();
      std::vector<unsigned long long> st(2048 * 4);
      hipMemcpy(st.data(), fdm::g_gemm_stamps, st.size() * 8, hipMemcpyDeviceToHost);
      for (int b : {0, 100, 255, 256, 700}) {
        unsigned long long* q = &st[(size_t)b * 4];
        if (q[3] == 0) continue;
        printf("   block %4d: start +%llu  prologue %llu  main %llu  epilogue(+drain) %llu\n", b, q[0] - st[0], q[1] - q[0], q[2] - q[1], q[3] - q[2]);
      }
      hipMemset(fdm::g_gemm_stamps, 0, st.size() * 8);
    }
#endif
  }
  printf("TOTAL img-stream GEMM per step: %.2f ms\n", tot);
  return 0;
}
