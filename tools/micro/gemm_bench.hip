// Standalone fp8 GEMM micro-benchmark: compiles gemm.hip directly so that kernel variants (-D macros) can be
// A/B-tested in one GPU call without Python.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off
//   tools/micro/gemm_bench.hip -o tools/micro/gemm_bench[_variant] [-DFDM_...]
#ifndef GEMM_SRC
#define GEMM_SRC "../../fastdm_amd/csrc/gemm.hip"   // A/B builds point this at another revision of the file (-DGEMM_SRC=... -I fastdm_amd/csrc)
#endif
#include GEMM_SRC
#include <cstdio>
#include <cstring>
#include <cmath>
namespace fdm { thread_local int g_last_hip_error = 0; int device_cu_count() { return 256; } }
#ifdef FDM_GEMM_AB
namespace fdm { extern int g_gemm_use_w4; extern int g_gemm_use_p4; }
#endif
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
    // GEMM_DATA=gauss: e4m3 codes of N(0, 1) * 448 / 4 (what a per-row-scaled activation looks like); default: random finite
    // codes (uniform over sign / exponent / mantissa: 35 binades of product range)
    const bool gauss = getenv("GEMM_DATA") && !strcmp(getenv("GEMM_DATA"), "gauss");
    const bool narrow = getenv("GEMM_DATA") && !strcmp(getenv("GEMM_DATA"), "narrow");   // exponents 6..8: every partial sum exact in fp32
    auto enc = [](float f) -> uint8_t {   // nearest e4m3fn code, saturating
      const uint8_t sgn = f < 0 ? 0x80 : 0; float a = fabsf(f); if (a > 448.f) a = 448.f;
      int e; float m = frexpf(a, &e);   // a = m 2^e, m in [0.5, 1)
      if (a < 0.0009765625f) return sgn;
      int E = e - 1 + 7; float frac;
      if (E <= 0) { frac = a / 0.001953125f; const int q = (int)lrintf(frac); return sgn | (uint8_t)(q > 7 ? 8 : q); }
      frac = (m * 2.f - 1.f) * 8.f; int q = (int)lrintf(frac); if (q == 8) { q = 0; ++E; }
      if (E > 15 || (E == 15 && q > 6)) { E = 15; q = 6; }
      return sgn | (uint8_t)(E << 3) | (uint8_t)q;
    };
    auto gfill = [&](std::vector<uint8_t>& h) {
      for (size_t i = 0; i < h.size(); i += 2) {
        const float u1 = (nextu() + 1.0f) / 4294967808.0f, u2 = nextu() / 4294967296.0f;
        const float r = sqrtf(-2.f * logf(u1)) * 112.f;
        h[i] = enc(r * cosf(6.2831853f * u2));
        if (i + 1 < h.size()) h[i + 1] = enc(r * sinf(6.2831853f * u2));
      }
    };
    std::vector<uint8_t> h(maxA);
    if (gauss) gfill(h);
    else for (auto& x : h) { uint8_t v = (uint8_t)nextu(); if ((v & 0x7f) >= 0x78) v &= 0xc7; x = v; }   // finite e4m3, random sign/exp
    if (getenv("GEMM_DATA") && !strcmp(getenv("GEMM_DATA"), "ident")) {   // A[m][k] = 1.0 at k == m % 128 (row stride: IDENT_K), else 0
      const int64_t ik = atol(getenv("IDENT_K"));
      for (size_t i = 0; i < h.size(); ++i) h[i] = ((int64_t)(i % ik) == (int64_t)((i / ik) % ik)) ? 0x38 : 0;
    }
    hipMemcpy(A, h.data(), maxA, hipMemcpyHostToDevice);
    h.resize(maxB);
    if (gauss) gfill(h);
    else if (narrow) for (auto& x : h) { const uint32_t u = nextu(); x = (uint8_t)((u & 0x87) | ((6 + (u >> 8) % 3) << 3)); }
    else for (auto& x : h) { uint8_t v = (uint8_t)nextu(); if ((v & 0x7f) >= 0x78) v &= 0xc7; x = v; }
    if (getenv("GEMM_DATA") && !strcmp(getenv("GEMM_DATA"), "identb")) {   // B[n][k] = 1.0 at k == n % IDENT_K
      const int64_t ik = atol(getenv("IDENT_K"));
      const int64_t dd = getenv("IDENT_D") ? atol(getenv("IDENT_D")) : 0;
      for (size_t i = 0; i < h.size(); ++i) {
        const int64_t k = i % ik, n = (i / ik) % ik;
        h[i] = (k == n || (dd && k == (n + dd) % ik)) ? 0x38 : 0;
      }
    }
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
  printf("buffers: A %p (+%ld) B %p (+%ld) C %p (+%ld) R %p G %p bias %p sa %p sb %p\n", (void*)A, (long)maxA, (void*)B, (long)maxB, (void*)C, (long)maxC * 2,
         (void*)R, (void*)G, (void*)bias, (void*)sa, (void*)sb);
  fflush(stdout);
#ifdef FDM_GEMM_STAMPS
  hipMalloc(&fdm::g_gemm_stamps, 2048 * 8 * 8);
  hipMemset(fdm::g_gemm_stamps, 0, 2048 * 8 * 8);
#endif
  const bool use_int8 = getenv("GEMM_DT") && !strcmp(getenv("GEMM_DT"), "int8");
  const bool use_bf16 = getenv("GEMM_DT") && !strcmp(getenv("GEMM_DT"), "bf16");   // K counts bf16 elements (2 K bytes per row); the byte patterns of A / B are reinterpreted: cycle counts only
  int32_t *azp, *azp_adj;
  hipMalloc(&azp, 8704 * 4); hipMalloc(&azp_adj, 12288 * 4);
  hipMemset(azp, 0, 8704 * 4); hipMemset(azp_adj, 1, 12288 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  double tot = 0;
  for (auto& s : shapes) {
    auto run = [&]() {
      if (use_bf16) return fdm_gemm_bf16(C, A, B, bias, s.M, s.N, s.K, s.K, s.K, s.N, s.epi, G, s.N, R, s.M, nullptr);
      if (use_int8) return fdm_gemm_w8a8_int8(C, A, B, sa, sb, azp_adj, azp, bias, s.M, s.N, s.K, s.K, s.K, s.N, s.epi, G, s.N, R, s.M, nullptr);
      return fdm_gemm_w8a8_fp8(C, A, B, sa, sb, bias, s.M, s.N, s.K, s.K, s.K, s.N, s.epi, G, s.N, R, s.M, nullptr);
    };
#ifdef FDM_GEMM_AB
    {   // element-wise comparison of the two kernels on this shape (bf16 outputs)
      std::vector<uint16_t> h0((size_t)s.M * s.N), h1((size_t)s.M * s.N);
      fdm::g_gemm_use_w4 = 0; run(); hipMemcpy(h0.data(), C, h0.size() * 2, hipMemcpyDeviceToHost);
      hipMemset(C, 0xff, h0.size() * 2);
      fdm::g_gemm_use_w4 = 1; run(); hipMemcpy(h1.data(), C, h1.size() * 2, hipMemcpyDeviceToHost);
      size_t ndiff = 0, nbig = 0; int maxulp = 0;
      for (size_t i = 0; i < h0.size(); ++i) if (h0[i] != h1[i]) {
        ++ndiff;
        const int d = abs((int)(h0[i] & 0x7fff) - (int)(h1[i] & 0x7fff));
        const bool sign = (h0[i] ^ h1[i]) & 0x8000;
        const int u = sign ? (h0[i] & 0x7fff) + (h1[i] & 0x7fff) : d;
        if (u > maxulp) maxulp = u;
        if (u > 1) ++nbig;
      }
      if (getenv("GEMM_DUMP")) {
        int shown = 0;
        std::vector<int> per_n(s.N, 0), per_m(s.M, 0);
        std::vector<uint8_t> ha2((size_t)s.M * s.K), hb2((size_t)s.N * s.K); std::vector<uint16_t> hbias2(s.N);
        hipMemcpy(ha2.data(), A, ha2.size(), hipMemcpyDeviceToHost); hipMemcpy(hb2.data(), B, hb2.size(), hipMemcpyDeviceToHost);
        hipMemcpy(hbias2.data(), bias, s.N * 2, hipMemcpyDeviceToHost);
        auto e4 = [](uint8_t v) -> double { const int sg = v >> 7, e = (v >> 3) & 15, m = v & 7; const double mag = e == 0 ? m * 0.001953125 : (1.0 + m / 8.0) * pow(2.0, e - 7); return sg ? -mag : mag; };
        for (size_t i = 0; i < h0.size(); ++i) if (h0[i] != h1[i]) {
          ++per_n[i % s.N]; ++per_m[i / s.N];
          if (shown++ < 12) {
            const size_t m = i / s.N, n = i % s.N; double acc = 0;
            for (int64_t k = 0; k < s.K; ++k) acc += e4(ha2[m * s.K + k]) * e4(hb2[n * s.K + k]);
            const float f1 = (float)acc * 1e-4f, f2 = f1 * 1e-4f; uint32_t bb = (uint32_t)hbias2[n] << 16; float bf_; memcpy(&bf_, &bb, 4);
            const float f3 = f2 + bf_; uint32_t u; memcpy(&u, &f3, 4);
            printf("      m=%zu n=%zu old %04x w4 %04x | acc %.10g fp32 chain %08x bias %04x\n", m, n, h0[i], h1[i], acc, u, hbias2[n]);
          }
        }
        {
          long hn[16] = {0}, hm[16] = {0}, hnb[8] = {0}, hmb[8] = {0};
          for (size_t i = 0; i < h0.size(); ++i) if (h0[i] != h1[i]) { const size_t m = i / s.N, n = i % s.N; ++hn[n & 15]; ++hm[m & 15]; ++hnb[(n >> 4) & 7]; ++hmb[(m >> 4) & 7]; }
          printf("      by n%%16:"); for (int x = 0; x < 16; ++x) printf(" %ld", hn[x]); printf("\n      by m%%16:"); for (int x = 0; x < 16; ++x) printf(" %ld", hm[x]);
          printf("\n      by nb:"); for (int x = 0; x < 8; ++x) printf(" %ld", hnb[x]); printf("\n      by mb:"); for (int x = 0; x < 8; ++x) printf(" %ld", hmb[x]); printf("\n");
        }
        printf("      mismatches per n (first 64):"); for (int n = 0; n < 256 && n < s.N; ++n) printf(" %d", per_n[n]); printf("\n");
        printf("      mismatches per m (first 64):"); for (int m = 0; m < 256 && m < s.M; ++m) printf(" %d", per_m[m]); printf("\n");
      }
      printf("   old vs w4: %zu of %zu elements differ, %zu by more than 1 bf16 ulp, max %d ulp\n", ndiff, h0.size(), nbig, maxulp);
      if (s.epi == 0 && !use_int8) {   // both against an exact host reference on sampled elements (fp8 products are exact in double)
        auto e4m3 = [](uint8_t v) -> double {
          const int sgn = v >> 7, e = (v >> 3) & 15, m = v & 7;
          const double mag = e == 0 ? m * 0.001953125 : (1.0 + m / 8.0) * pow(2.0, e - 7);
          return sgn ? -mag : mag;
        };
        auto bf = [](uint16_t h) -> double { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; };
        std::vector<uint8_t> ha((size_t)s.M * s.K), hb((size_t)s.N * s.K);
        std::vector<uint16_t> hbias(s.N);
        hipMemcpy(ha.data(), A, ha.size(), hipMemcpyDeviceToHost); hipMemcpy(hb.data(), B, hb.size(), hipMemcpyDeviceToHost);
        hipMemcpy(hbias.data(), bias, s.N * 2, hipMemcpyDeviceToHost);
        double e0 = 0, e1 = 0, m0 = 0, m1 = 0; int cnt = 0;
        for (int smp = 0; smp < 2000; ++smp) {
          const size_t m = nextu() % s.M, n = nextu() % s.N;
          double acc = 0;
          for (int64_t k = 0; k < s.K; ++k) acc += e4m3(ha[m * s.K + k]) * e4m3(hb[n * s.K + k]);
          const double ref = acc * 1e-4f * 1e-4f + bf(hbias[n]);
          const double ulp = fabs(ref) * 0.0078125 + 1e-30;
          const double d0 = fabs(bf(h0[m * s.N + n]) - ref) / ulp, d1 = fabs(bf(h1[m * s.N + n]) - ref) / ulp;
          if (getenv("GEMM_DUMP") && smp < 24) {
            // partial sums over the four 32-byte K groups of the first 128-byte step
            double part[4] = {0, 0, 0, 0};
            for (int64_t k = 0; k < 128 && k < s.K; ++k) part[k / 32] += e4m3(ha[m * s.K + k]) * e4m3(hb[n * s.K + k]);
            printf("      m=%zu n=%zu exact acc %.6g (groups %.6g %.6g %.6g %.6g) | old acc %.6g | w4 acc %.6g\n", m, n, acc, part[0], part[1], part[2], part[3],
                   (bf(h0[m * s.N + n]) - bf(hbias[n])) * 1e8, (bf(h1[m * s.N + n]) - bf(hbias[n])) * 1e8);
          }
          e0 += d0; e1 += d1; if (d0 > m0) m0 = d0; if (d1 > m1) m1 = d1; ++cnt;
        }
        printf("   vs exact (2000 samples, units of 2^-7 |ref|): old mean %.4f max %.3f | w4 mean %.4f max %.3f\n", e0 / cnt, m0, e1 / cnt, m1);
      }
      {   // the persistent assembly kernel against gemm_w4_kernel: same MFMAs, same order, same epilogue arithmetic -> every bit
        std::vector<uint16_t> h2((size_t)s.M * s.N);
        hipMemset(C, 0xee, h2.size() * 2);
        fdm::g_gemm_use_w4 = 1; fdm::g_gemm_use_p4 = 1;
        const int rc = run();
        const hipError_t er = hipDeviceSynchronize();
        hipMemcpy(h2.data(), C, h2.size() * 2, hipMemcpyDeviceToHost);
        size_t nd = 0, first = (size_t)-1;
        for (size_t i = 0; i < h2.size(); ++i) if (h2[i] != h1[i]) { if (first == (size_t)-1) first = i; ++nd; }
        if (getenv("GEMM_DUMP_P4")) {   // P4_DEBUG=8 builds: [block * 4 + wave][slot][64 lanes] dwords at C
          const int nslots = atoi(getenv("GEMM_DUMP_P4"));
          const uint32_t* w = reinterpret_cast<const uint32_t*>(h2.data());
          for (int bw : {0, 1, 2, 3, 4, 37 * 4 + 2, 255 * 4 + 3}) {
            printf("   block %d wave %d:\n", bw / 4, bw % 4);
            for (int k = 0; k < nslots; ++k) {
              const uint32_t* q = w + ((size_t)bw * nslots + k) * 64;
              printf("     slot %2d: lane0 %08x lane1 %08x lane17 %08x lane63 %08x\n", k, q[0], q[1], q[17], q[63]);
            }
          }
        }
        printf("   w4 vs p4 (persistent asm): rc %d sync %d, %zu of %zu elements differ", rc, (int)er, nd, h2.size());
        if (nd) {
          printf(" (first at m=%zu n=%zu: w4 %04x p4 %04x)", first / s.N, first % s.N, h1[first], h2[first]);
          // which tiles / rows / columns are wrong
          long tiles_bad = 0; std::vector<char> tb((s.M / 256) * (s.N / 256), 0);
          std::vector<long> by_m16(16, 0), by_n16(16, 0), by_mb(16, 0), by_nb(16, 0);
          for (size_t i = 0; i < h2.size(); ++i) if (h2[i] != h1[i]) { const size_t m = i / s.N, n = i % s.N; tb[(m / 256) * (s.N / 256) + n / 256] = 1; ++by_m16[m & 15]; ++by_n16[n & 15]; ++by_mb[(m >> 4) & 15]; ++by_nb[(n >> 4) & 15]; }
          for (char c : tb) tiles_bad += c;
          printf("\n      wrong tiles %ld of %zu; by m%%16:", tiles_bad, tb.size()); for (long x : by_m16) printf(" %ld", x);
          printf("\n      by n%%16:"); for (long x : by_n16) printf(" %ld", x);
          printf("\n      by (m/16)%%16:"); for (long x : by_mb) printf(" %ld", x);
          printf("\n      by (n/16)%%16:"); for (long x : by_nb) printf(" %ld", x);
          printf("\n      first bad tiles:"); int shown = 0; for (size_t t = 0; t < tb.size() && shown < 24; ++t) if (tb[t]) { printf(" (%zu,%zu)", t / (s.N / 256), t % (s.N / 256)); ++shown; }
        }
        printf("\n");
        fdm::g_gemm_use_p4 = getenv("GEMM_P4") ? 1 : 0;
      }
      fdm::g_gemm_use_w4 = getenv("GEMM_OLD") ? 0 : 1;
    }
#endif
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
      std::vector<unsigned long long> st(2048 * 8);
      hipMemcpy(st.data(), fdm::g_gemm_stamps, st.size() * 8, hipMemcpyDeviceToHost);
#ifdef FDM_GEMM_AB
      const int stride = fdm::g_gemm_use_w4 ? 8 : 4;
#else
      const int stride = getenv("GEMM_OLD") ? 4 : 8;
#endif
      for (int b : {0, 100, 255, 256, 700}) {
        unsigned long long* q = &st[(size_t)b * stride];
        if (q[3] == 0) continue;
        printf("   block %4d: start +%llu  prologue %llu  main %llu  epilogue(+drain) %llu", b, q[0] - st[0], q[1] - q[0], q[2] - q[1], q[3] - q[2]);
        if (stride == 8 && q[4]) printf("  [pass 1 %llu, store issue %llu, drain %llu]", q[4] - q[2], q[5] - q[4], q[3] - q[5]);
        if (stride == 8 && q[7] > q[6]) printf("  clock %.0f MHz", (double)(q[3] - q[0]) / (double)(q[7] - q[6]) * 100.0);
        printf("\n");
      }
      hipMemset(fdm::g_gemm_stamps, 0, 2048 * 8 * 8);
    }
#endif
  }
  printf("TOTAL img-stream GEMM per step: %.2f ms\n", tot);
  return 0;
}
