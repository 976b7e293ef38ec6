// W8A8 (fp8 / int8) and bf16 GEMM for gfx950:  C[M,N] = epi(A[M,K] . B[N,K]^T)
//
// Both operands are K-contiguous (A row-major, B = the reference's column-major [K,N] weight,
// i.e. [N,K] row-major), which is exactly the MFMA operand order, so tiles go global -> LDS with
// 16-byte LDS-DMA (global_load_lds_dwordx4) and LDS -> VGPR with ds_read_b128, no transposes.
//
//   tile      256 x 256, K-step 64 bytes; 8 waves = 2 (M) x 4 (N), wave tile 128 x 64
//   fp8       v_mfma_scale_f32_32x32x64_f8f6f4 with unit E8M0 scales (2x the bf16 MFMA rate)
//   int8      v_mfma_i32_32x32x32_i8           bf16: v_mfma_f32_32x32x16_bf16
//   pipeline  4-stage LDS ring of 32 KiB (A 16 KiB + B 16 KiB): while step t runs on registers, the
//             fragments of step t+1 are read from LDS and the DMAs of steps t+2, t+3 are in flight.
//             One raw s_barrier per step behind a COUNTED s_waitcnt vmcnt(4) (never 0 in the loop), so
//             LDS-DMA stays in flight across barriers.
//   LDS       rows are 64 B; chunk c of row r lives in slot c ^ ((r >> 2) & 3): conflict-free for the
//             ds_read_b128 lane groups.  LDS-DMA writes are lane-linear, so the swizzle is applied to
//             the per-lane SOURCE address.
//   operands  swapped (D^T = B . A^T): a lane then owns ONE output row m and 4 consecutive columns per
//             register quad -> per-row scale is a lane constant and stores/residual loads are 8 bytes.
//   grid      one block per tile; block ids are remapped so each XCD (private L2) owns a contiguous
//             range of tiles, rastered 4 (M) x n inside it.
//   edges     M/N: source rows clamped, stores masked.  K tail: lanes past K read a zero page.
#include <type_traits>
#include "common.h"

namespace fdm {

enum { DT_FP8 = 0, DT_INT8 = 1, DT_BF16 = 2 };

constexpr int BM = 256, BN = 256, BKB = 64;  // BKB = K-step in bytes
constexpr int GEMM_THREADS = 512;
constexpr int NSTAGE = 4;
constexpr int STAGE_BYTES = (BM + BN) * BKB;  // 32 KiB
constexpr int GROUP_M = 4;

__device__ __attribute__((aligned(16))) uint8_t g_zero_page[16] = {0};

struct GemmParams {
  const uint8_t* A;
  const uint8_t* B;
  bf16_t* C;
  const float* sa;
  const float* sb;
  const bf16_t* bias;
  const int32_t* azp_adj;
  const int32_t* azp;
  const bf16_t* gate;
  const bf16_t* res;
  int64_t rows_per_batch;
  int64_t gate_stride;
  int M, N;
  int64_t Kb;        // K in bytes
  int64_t lda, ldb;  // bytes
  int64_t ldc;       // elements
  int tiles_m, tiles_n;
};

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

template <int DT>
struct Mma;
template <>
struct Mma<DT_FP8> {
  using Acc = f32x16;
  static constexpr int KS = 64;  // bytes of K per MFMA
  using Frag = i32x8;
  static __device__ __forceinline__ Acc run(const Frag& a, const Frag& b, const Acc& c) {
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  }
};
template <>
struct Mma<DT_INT8> {
  using Acc = i32x16;
  static constexpr int KS = 32;
  using Frag = i32x4;
  static __device__ __forceinline__ Acc run(const Frag& a, const Frag& b, const Acc& c) {
    return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
  }
};
template <>
struct Mma<DT_BF16> {
  using Acc = f32x16;
  static constexpr int KS = 32;
  using Frag = i32x4;
  static __device__ __forceinline__ Acc run(const Frag& a, const Frag& b, const Acc& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};

template <typename Frag>
__device__ __forceinline__ Frag lds_frag(const uint8_t* p0, const uint8_t* p1);
template <>
__device__ __forceinline__ i32x8 lds_frag<i32x8>(const uint8_t* p0, const uint8_t* p1) {
  i32x4 lo = *reinterpret_cast<const i32x4*>(p0);
  i32x4 hi = *reinterpret_cast<const i32x4*>(p1);
  i32x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}
template <>
__device__ __forceinline__ i32x4 lds_frag<i32x4>(const uint8_t* p0, const uint8_t*) {
  return *reinterpret_cast<const i32x4*>(p0);
}

__device__ __forceinline__ void glds16(const uint8_t* gsrc, uint8_t* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int DT>
struct FragsA {
  static constexpr int NS = BKB / Mma<DT>::KS;  // MFMA k-slices per step (1 for fp8, 2 for int8/bf16)
  typename Mma<DT>::Frag a[NS][4];
};
template <int DT>
struct FragsB {
  static constexpr int NS = BKB / Mma<DT>::KS;
  typename Mma<DT>::Frag b[NS][2];
};

template <int DT, int EPI>
__global__ void __launch_bounds__(GEMM_THREADS, 2) gemm_kernel(const GemmParams p) {
  using M = Mma<DT>;
  using Acc = typename M::Acc;
  using Frag = typename M::Frag;
  constexpr int NS = FragsA<DT>::NS;
  __shared__ __attribute__((aligned(1024))) uint8_t smem[NSTAGE * STAGE_BYTES];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  // ---- block -> tile (XCD-contiguous, then grouped raster)
  int tm, tn;
  {
    const int nwg = p.tiles_m * p.tiles_n;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int in_group = GROUP_M * p.tiles_n;
    const int gid = t / in_group;
    const int first_m = gid * GROUP_M;
    const int gsz = min(p.tiles_m - first_m, GROUP_M);
    const int rem = t - gid * in_group;
    tm = first_m + rem % gsz;
    tn = rem / gsz;
  }
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- staging: thread t, round i moves the 16-byte chunk that lands at (row = i*128 + t/4, slot = t%4)
  const int srow = tid >> 2;
  const int schunk = (tid & 3) ^ ((srow >> 2) & 3);  // source chunk for that slot
  const uint8_t* ga[2];
  const uint8_t* gb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int ra = min(m0 + i * 128 + srow, p.M - 1);
    int rb = min(n0 + i * 128 + srow, p.N - 1);
    ga[i] = p.A + (int64_t)ra * p.lda + schunk * 16;
    gb[i] = p.B + (int64_t)rb * p.ldb + schunk * 16;
  }
  const int nk = (int)((p.Kb + BKB - 1) / BKB);

  // Every step issues exactly 4 DMAs per wave, also past the end of K (then from a 16-byte zero page, which
  // zero-fills the stage): the loop body is branch-free, s_waitcnt counts are constants, the K tail and the
  // odd-step padding both multiply zeros.
  const uint8_t* zero_page = g_zero_page;
  auto stage = [&](int kt) {
    uint8_t* sA = smem + (kt & (NSTAGE - 1)) * STAGE_BYTES + wave * 1024;
    uint8_t* sB = sA + BM * BKB;
    const int64_t koff = (int64_t)kt * BKB;
    const bool ok = koff + schunk * 16 < p.Kb;
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(ok ? ga[i] + koff : zero_page, sA + i * 8192);
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(ok ? gb[i] + koff : zero_page, sB + i * 8192);
  };

  // ---- fragment read addresses (byte offsets inside a stage)
  const int lrow = lane & 31, g = lane >> 5;
  const int sw = (lrow >> 2) & 3;
  const int a_row_off = (wm * 128 + lrow) * BKB;
  const int b_row_off = BM * BKB + (wn * 64 + lrow) * BKB;

  auto frag_cols = [&](int s, int& c0, int& c1) {
    if constexpr (M::KS == 64) {
      c0 = ((g * 2) ^ sw) * 16;
      c1 = ((g * 2 + 1) ^ sw) * 16;
    } else {
      c0 = ((s * 2 + g) ^ sw) * 16;
      c1 = c0;
    }
  };
  auto read_a = [&](FragsA<DT>& f, int kt, int i) {
    const uint8_t* st = smem + (kt & (NSTAGE - 1)) * STAGE_BYTES + a_row_off + i * 32 * BKB;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      int c0, c1;
      frag_cols(s, c0, c1);
      f.a[s][i] = lds_frag<Frag>(st + c0, st + c1);
    }
  };
  auto read_b = [&](FragsB<DT>& f, int kt) {
    const uint8_t* st = smem + (kt & (NSTAGE - 1)) * STAGE_BYTES + b_row_off;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      int c0, c1;
      frag_cols(s, c0, c1);
#pragma unroll
      for (int j = 0; j < 2; ++j) f.b[s][j] = lds_frag<Frag>(st + j * 32 * BKB + c0, st + j * 32 * BKB + c1);
    }
  };

  Acc acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;

  // ---- prologue: DMAs of steps 0..2 in flight, fragments of step 0 in registers
  stage(0);
  stage(1);
  stage(2);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  FragsA<DT> fa;
  FragsB<DT> fb0, fb1;
#pragma unroll
  for (int i = 0; i < 4; ++i) read_a(fa, 0, i);
  read_b(fb0, 0);

  // One step: wait for DMA(t+1) (leaving DMA(t+2) in flight), barrier, issue DMA(t+3), then MFMA(t) with the
  // fragments of step t+1 streaming in behind it: B fragments are double buffered, each A fragment is replaced
  // in place as soon as its two MFMAs have issued (keeps the kernel under 256 VGPRs).
  auto step = [&](FragsB<DT>& bc, FragsB<DT>& bn, int t) {
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    stage(t + 3);
    read_b(bn, t + 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = M::run(bc.b[s][j], fa.a[s][i], acc[i][j]);  // swapped: rows = n, cols = m
      read_a(fa, t + 1, i);
    }
  };
  const int nk2 = (nk + 1) & ~1;
  for (int t = 0; t < nk2; t += 2) {
    step(fb0, fb1, t);
    step(fb1, fb0, t + 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // retire the padding DMAs before the LDS is released

  // ---- epilogue.  Swapped operands: lane holds row m = m0 + wm*128 + i*32 + (lane&31) and columns
  //      n = n0 + wn*64 + j*32 + 8*q + 4*(lane>>5) + (0..3) for register quad q = e>>2.
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 128 + i * 32 + lrow;
    const bool m_ok = m < p.M;
    const int mc = m_ok ? m : 0;
    float sam = 1.0f;
    int azpm = 0;
    if constexpr (DT != DT_BF16) sam = p.sa[mc];
    if constexpr (DT == DT_INT8) {
      if (p.azp != nullptr) azpm = p.azp[mc];
    }
    const bf16_t* gate_row = nullptr;
    if constexpr (EPI == FDM_EPI_GATE_RESIDUAL) gate_row = p.gate + (mc / p.rows_per_batch) * p.gate_stride;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + wn * 64 + j * 32 + 8 * q + 4 * g;
        if (!m_ok || n >= p.N) continue;  // N % 4 == 0 is guaranteed (N % 16 == 0), so a quad is all-in or all-out
        float sb4[4] = {1.f, 1.f, 1.f, 1.f}, bias4[4] = {0.f, 0.f, 0.f, 0.f};
        int adj4[4] = {0, 0, 0, 0};
        if constexpr (DT != DT_BF16) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(p.sb + n);
          sb4[0] = v[0]; sb4[1] = v[1]; sb4[2] = v[2]; sb4[3] = v[3];
        }
        if (p.bias != nullptr) {
          const i32x2 v = *reinterpret_cast<const i32x2*>(p.bias + n);
          bias4[0] = __builtin_bit_cast(float, (uint32_t)v[0] << 16);
          bias4[1] = __builtin_bit_cast(float, (uint32_t)v[0] & 0xffff0000u);
          bias4[2] = __builtin_bit_cast(float, (uint32_t)v[1] << 16);
          bias4[3] = __builtin_bit_cast(float, (uint32_t)v[1] & 0xffff0000u);
        }
        if constexpr (DT == DT_INT8) {
          if (p.azp_adj != nullptr) {
            const i32x4 v = *reinterpret_cast<const i32x4*>(p.azp_adj + n);
            adj4[0] = v[0]; adj4[1] = v[1]; adj4[2] = v[2]; adj4[3] = v[3];
          }
        }
        float y[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int e = 4 * q + c;
          if constexpr (DT == DT_FP8) {
            float v = acc[i][j][e] * sam * sb4[c];
            if (p.bias != nullptr) v += bias4[c];
            y[c] = rbf(v);
          } else if constexpr (DT == DT_INT8) {
            const int a = acc[i][j][e] - azpm * adj4[c];
            float v = rbf((float)a * (sam * sb4[c]));
            if (p.bias != nullptr) v = rbf(v + bias4[c]);
            y[c] = v;
          } else {
            float v = acc[i][j][e];
            if (p.bias != nullptr) v += bias4[c];
            y[c] = rbf(v);
          }
          if constexpr (EPI == FDM_EPI_GELU_TANH) y[c] = gelu_tanh_f(y[c]);
          if constexpr (EPI == FDM_EPI_GELU_ERF) y[c] = gelu_erf_f(y[c]);
        }
        if constexpr (EPI == FDM_EPI_GATE_RESIDUAL) {
          const i32x2 gv = *reinterpret_cast<const i32x2*>(gate_row + n);
          const i32x2 rv = *reinterpret_cast<const i32x2*>(p.res + (int64_t)m * p.ldc + n);
          const float g4[4] = {__builtin_bit_cast(float, (uint32_t)gv[0] << 16), __builtin_bit_cast(float, (uint32_t)gv[0] & 0xffff0000u),
                               __builtin_bit_cast(float, (uint32_t)gv[1] << 16), __builtin_bit_cast(float, (uint32_t)gv[1] & 0xffff0000u)};
          const float r4[4] = {__builtin_bit_cast(float, (uint32_t)rv[0] << 16), __builtin_bit_cast(float, (uint32_t)rv[0] & 0xffff0000u),
                               __builtin_bit_cast(float, (uint32_t)rv[1] << 16), __builtin_bit_cast(float, (uint32_t)rv[1] & 0xffff0000u)};
#pragma unroll
          for (int c = 0; c < 4; ++c) y[c] = r4[c] + rbf(g4[c] * y[c]);
        }
        i32x2 o;
        o[0] = (int)pack2bf(y[0], y[1]);
        o[1] = (int)pack2bf(y[2], y[3]);
        *reinterpret_cast<i32x2*>(p.C + (int64_t)m * p.ldc + n) = o;
      }
    }
  }
}

template <int DT>
static int launch_gemm(const GemmParams& p, int epi, hipStream_t s) {
  const int grid = p.tiles_m * p.tiles_n;
  switch (epi) {
    case FDM_EPI_NONE: gemm_kernel<DT, FDM_EPI_NONE><<<grid, GEMM_THREADS, 0, s>>>(p); break;
    case FDM_EPI_GELU_TANH: gemm_kernel<DT, FDM_EPI_GELU_TANH><<<grid, GEMM_THREADS, 0, s>>>(p); break;
    case FDM_EPI_GELU_ERF: gemm_kernel<DT, FDM_EPI_GELU_ERF><<<grid, GEMM_THREADS, 0, s>>>(p); break;
    case FDM_EPI_GATE_RESIDUAL: gemm_kernel<DT, FDM_EPI_GATE_RESIDUAL><<<grid, GEMM_THREADS, 0, s>>>(p); break;
    default: return FDM_ERR_UNSUPPORTED;
  }
  FDM_CHECK_LAUNCH();
  return FDM_OK;
}

static int fill_params(GemmParams& p, void* out, const void* a, const void* b, int64_t M, int64_t N, int64_t K, int64_t lda,
                       int64_t ldb, int64_t ldc, int elem_bytes, int epi, const void* gate, int64_t gate_stride, const void* res,
                       int64_t rpb) {
  if (M == 0 && N > 0 && K > 0) return FDM_OK;  // callers re-check M == 0 before launching
  if (!out || !a || !b || M < 0 || N <= 0 || K <= 0) return FDM_ERR_INVALID;
  if (M > 0x7fffffff || N > 0x7fffffff) return FDM_ERR_UNSUPPORTED;
  if ((K * elem_bytes) % 16 || (lda * elem_bytes) % 16 || (ldb * elem_bytes) % 16 || lda < K || ldb < K || ldc < N)
    return FDM_ERR_INVALID;
  if ((reinterpret_cast<uintptr_t>(a) & 15) || (reinterpret_cast<uintptr_t>(b) & 15) || (reinterpret_cast<uintptr_t>(out) & 7) || (ldc & 3) || (N & 3))
    return FDM_ERR_INVALID;
  if (epi == FDM_EPI_GATE_RESIDUAL && ((reinterpret_cast<uintptr_t>(gate) & 7) || (reinterpret_cast<uintptr_t>(res) & 7) || (gate_stride & 3)))
    return FDM_ERR_INVALID;
  if (epi == FDM_EPI_GATE_RESIDUAL && (!gate || !res || rpb <= 0)) return FDM_ERR_INVALID;
  p.A = (const uint8_t*)a;
  p.B = (const uint8_t*)b;
  p.C = (bf16_t*)out;
  p.gate = (const bf16_t*)gate;
  p.res = (const bf16_t*)res;
  p.rows_per_batch = rpb > 0 ? rpb : 1;
  p.gate_stride = gate_stride;
  p.M = (int)M;
  p.N = (int)N;
  p.Kb = K * elem_bytes;
  p.lda = lda * elem_bytes;
  p.ldb = ldb * elem_bytes;
  p.ldc = ldc;
  p.tiles_m = (int)((M + BM - 1) / BM);
  p.tiles_n = (int)((N + BN - 1) / BN);
  p.sa = p.sb = nullptr;
  p.bias = nullptr;
  p.azp = p.azp_adj = nullptr;
  return FDM_OK;
}

}  // namespace fdm

using namespace fdm;

extern "C" {

int fdm_gemm_w8a8_fp8(void* out, const void* a, const void* b, const float* scale_a, const float* scale_b, const void* bias,
                      int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int epilogue,
                      const void* epi_gate, int64_t epi_gate_stride, const void* epi_res, int64_t epi_rows_per_batch,
                      fdm_stream_t stream) {
  GemmParams p;
  int rc = fill_params(p, out, a, b, M, N, K, lda, ldb, ldc, 1, epilogue, epi_gate, epi_gate_stride, epi_res, epi_rows_per_batch);
  if (rc != FDM_OK) return rc;
  if (M == 0) return FDM_OK;
  if (!scale_a || !scale_b) return FDM_ERR_INVALID;
  if (M == 0) return FDM_OK;
  p.sa = scale_a;
  p.sb = scale_b;
  p.bias = (const bf16_t*)bias;
  return launch_gemm<DT_FP8>(p, epilogue, to_stream(stream));
}

int fdm_gemm_w8a8_int8(void* out, const void* a, const void* b, const float* scale_a, const float* scale_b,
                       const int32_t* azp_adj, const int32_t* azp, const void* bias, int64_t M, int64_t N, int64_t K,
                       int64_t lda, int64_t ldb, int64_t ldc, int epilogue, const void* epi_gate, int64_t epi_gate_stride,
                       const void* epi_res, int64_t epi_rows_per_batch, fdm_stream_t stream) {
  GemmParams p;
  int rc = fill_params(p, out, a, b, M, N, K, lda, ldb, ldc, 1, epilogue, epi_gate, epi_gate_stride, epi_res, epi_rows_per_batch);
  if (rc != FDM_OK) return rc;
  if (M == 0) return FDM_OK;
  if (!scale_a || !scale_b || ((azp == nullptr) != (azp_adj == nullptr))) return FDM_ERR_INVALID;
  if (M == 0) return FDM_OK;
  p.sa = scale_a;
  p.sb = scale_b;
  p.bias = (const bf16_t*)bias;
  p.azp = azp;
  p.azp_adj = azp_adj;
  return launch_gemm<DT_INT8>(p, epilogue, to_stream(stream));
}

int fdm_gemm_bf16(void* out, const void* a, const void* b, const void* bias, int64_t M, int64_t N, int64_t K, int64_t lda,
                  int64_t ldb, int64_t ldc, int epilogue, const void* epi_gate, int64_t epi_gate_stride, const void* epi_res,
                  int64_t epi_rows_per_batch, fdm_stream_t stream) {
  GemmParams p;
  int rc = fill_params(p, out, a, b, M, N, K, lda, ldb, ldc, 2, epilogue, epi_gate, epi_gate_stride, epi_res, epi_rows_per_batch);
  if (rc != FDM_OK) return rc;
  if (M == 0) return FDM_OK;
  if (M == 0) return FDM_OK;
  p.bias = (const bf16_t*)bias;
  return launch_gemm<DT_BF16>(p, epilogue, to_stream(stream));
}

}  // extern "C"
