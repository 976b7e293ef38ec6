// Interleaved A/B timing of attention-kernel variants in ONE process on ONE device (cdna guide, methodology rule 24):
//   attn_ab libA.so libB.so ...   each .so exports fdm_attn_fwd / fdm_attn_fwd_workspace_bytes (build_attn_variants.sh)
// Runs R rounds; in each round every variant runs `iters` back-to-back launches; prints median / min per variant and an
// fp64 spot check of the last variant-independent output rows.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
typedef int (*attn_fn)(void*, const void*, const void*, const void*, int64_t, int64_t, int64_t, int, int, int, int64_t, int64_t, int64_t, int64_t, float,
                       const int8_t*, int, int, void*, int64_t, void*);
typedef int64_t (*ws_fn)(int);
static uint16_t f2b(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static float b2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
static uint64_t rs = 88172645463325252ull;
static double ur() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return ((rs >> 11) + 0.5) * (1.0 / 9007199254740992.0); }
static float nr() { return (float)(sqrt(-2.0 * log(ur())) * cos(6.283185307179586 * ur())); }

int main(int argc, char** argv) {
  const int64_t B = 1, S = getenv("AB_S") ? atoll(getenv("AB_S")) : 8704, H = 24, D = 128, C = H * D;
  const int rounds = getenv("AB_ROUNDS") ? atoi(getenv("AB_ROUNDS")) : 7, iters = 10;
  std::vector<attn_fn> fns; std::vector<const char*> names;
  int64_t wsb = 0;
  for (int i = 1; i < argc; ++i) {
    void* h = dlopen(argv[i], RTLD_NOW | RTLD_LOCAL);
    if (!h) { printf("dlopen %s: %s\n", argv[i], dlerror()); return 1; }
    fns.push_back((attn_fn)dlsym(h, "fdm_attn_fwd")); names.push_back(argv[i]);
    wsb = std::max(wsb, ((ws_fn)dlsym(h, "fdm_attn_fwd_workspace_bytes"))(128));
  }
  const int64_t n = B * S * C;
  std::vector<uint16_t> hq(n), hk(n), hv(n), ho(n);
  for (auto* b : {&hq, &hk, &hv}) for (auto& e : *b) e = f2b(nr());
  uint16_t *q, *k, *v, *o; void* ws;
  hipMalloc(&q, n * 2); hipMalloc(&k, n * 2); hipMalloc(&v, n * 2); hipMalloc(&o, n * 2); hipMalloc(&ws, wsb);
  hipMemcpy(q, hq.data(), n * 2, hipMemcpyHostToDevice); hipMemcpy(k, hk.data(), n * 2, hipMemcpyHostToDevice); hipMemcpy(v, hv.data(), n * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  std::vector<std::vector<float>> t(fns.size());
  auto run = [&](attn_fn f) { return f(o, q, k, v, B, S, S, (int)H, (int)H, (int)D, C, C, C, C, 1.0f / sqrtf((float)D), nullptr, 0, 0, ws, wsb, nullptr); };
  for (size_t i = 0; i < fns.size(); ++i) for (int w = 0; w < 3; ++w) if (run(fns[i])) { printf("launch failed %s\n", names[i]); return 1; }
  hipDeviceSynchronize();
  for (int r = 0; r < rounds; ++r)
    for (size_t i = 0; i < fns.size(); ++i) {
      hipEventRecord(e0, 0);
      for (int it = 0; it < iters; ++it) run(fns[i]);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); t[i].push_back(ms / iters * 1e3f);
    }
  for (size_t i = 0; i < fns.size(); ++i) {
    // accuracy: rows of head 0 / 11 / 23 against fp64
    hipMemset(o, 0xff, n * 2); run(fns[i]); hipDeviceSynchronize(); hipMemcpy(ho.data(), o, n * 2, hipMemcpyDeviceToHost);
    double maxerr = 0; int64_t nan = 0; std::vector<double> sc(S);
    for (int64_t hh : {(int64_t)0, (int64_t)11, (int64_t)23}) for (int64_t r : {(int64_t)0, (int64_t)63, (int64_t)64, S / 2, S - 1}) {
      const uint16_t* qr = &hq[r * C + hh * D]; double mx = -1e300;
      for (int64_t j = 0; j < S; ++j) { double a = 0; for (int d = 0; d < D; ++d) a += (double)b2f(qr[d]) * b2f(hk[j * C + hh * D + d]); sc[j] = a / sqrt((double)D); mx = fmax(mx, sc[j]); }
      double l = 0; for (int64_t j = 0; j < S; ++j) { sc[j] = exp(sc[j] - mx); l += sc[j]; }
      for (int d = 0; d < D; ++d) { double a = 0; for (int64_t j = 0; j < S; ++j) a += sc[j] * b2f(hv[j * C + hh * D + d]);
        const float got = b2f(ho[r * C + hh * D + d]); if (!(got == got)) ++nan; else maxerr = fmax(maxerr, fabs(a / l - got)); }
    }
    std::sort(t[i].begin(), t[i].end());
    printf("%-28s median %7.1f us  min %7.1f  max %7.1f   %7.1f TF/s (median)   max|err| %.2e nan %ld\n", names[i], t[i][t[i].size() / 2], t[i].front(), t[i].back(),
           4.0 * B * S * S * H * D / (t[i][t[i].size() / 2] * 1e-6) / 1e12, maxerr, (long)nan);
  }
  return 0;
}
