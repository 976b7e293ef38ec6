// W8A8 (fp8 / int8) and bf16 GEMM for gfx950:  C[M,N] = epi(A[M,K] . B[N,K]^T)
//
// Both operands are K-contiguous (A row-major, B = the reference's column-major [K,N] weight,
// i.e. [N,K] row-major), which is exactly the MFMA operand order, so tiles go global -> LDS with
// 16-byte LDS-DMA (global_load_lds_dwordx4) and LDS -> VGPR with ds_read_b128, no transposes.
//
//   tile      256 x 256 x (128 bytes of K)    8 waves = 2 (M) x 4 (N), wave tile 128 x 64
//   fp8       v_mfma_scale_f32_32x32x64_f8f6f4 with unit E8M0 scales (2x the bf16 MFMA rate)
//   int8      v_mfma_i32_32x32x32_i8           bf16: v_mfma_f32_32x32x16_bf16
//   LDS       2 stages x (32 KiB A + 32 KiB B) = 128 KiB; rows are 128 B, chunk c of row r lives in
//             slot c ^ ((r >> 1) & 7): conflict-free for the ds_read_b128 lane groups.  LDS-DMA
//             writes are lane-linear, so the swizzle is applied to the per-lane SOURCE address.
//   grid      one block per tile; block ids are remapped so each XCD (private L2) owns a contiguous
//             range of tiles, rastered 4 (M) x n inside it.
//   edges     M/N: source rows clamped, stores masked.  K tail: lanes past K read a zero page.
#include "common.h"

namespace fdm {

enum { DT_FP8 = 0, DT_INT8 = 1, DT_BF16 = 2 };

constexpr int BM = 256, BN = 256, BKB = 128;  // BKB = K-tile in bytes
constexpr int GEMM_THREADS = 512;
constexpr int STAGE_BYTES = (BM + BN) * BKB;  // 64 KiB
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

template <int DT, int EPI>
__global__ void __launch_bounds__(GEMM_THREADS, 2) gemm_kernel(const GemmParams p) {
  using M = Mma<DT>;
  using Acc = typename M::Acc;
  using Frag = typename M::Frag;
  __shared__ __attribute__((aligned(1024))) uint8_t smem[2 * STAGE_BYTES];

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

  // ---- staging: thread t, round i moves 16-byte chunk (row = i*64 + t/8, slot = t%8)
  const int srow = tid >> 3;
  const int schunk = (tid & 7) ^ ((srow >> 1) & 7);  // source chunk that lands in slot tid%8
  const uint8_t* ga[4];
  const uint8_t* gb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int ra = min(m0 + i * 64 + srow, p.M - 1);
    int rb = min(n0 + i * 64 + srow, p.N - 1);
    ga[i] = p.A + (int64_t)ra * p.lda + schunk * 16;
    gb[i] = p.B + (int64_t)rb * p.ldb + schunk * 16;
  }
  const int nk = (int)((p.Kb + BKB - 1) / BKB);

  auto stage = [&](int buf, int kt) {
    uint8_t* sA = smem + buf * STAGE_BYTES + wave * 1024;
    uint8_t* sB = sA + BM * BKB;
    const int64_t koff = (int64_t)kt * BKB;
    const bool tail = (koff + schunk * 16 >= p.Kb);  // only possible in the last tile
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(tail ? g_zero_page : ga[i] + koff, sA + i * 8192);
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(tail ? g_zero_page : gb[i] + koff, sB + i * 8192);
  };

  // ---- fragment read addresses (byte offsets inside a stage)
  const int lrow = lane & 31, g = lane >> 5;
  const int sw = (lrow >> 1) & 7;
  const int a_row_off = (wm * 128 + lrow) * BKB;
  const int b_row_off = BM * BKB + (wn * 64 + lrow) * BKB;

  Acc acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;

  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
    const uint8_t* st = smem + cur * STAGE_BYTES;
#pragma unroll
    for (int s = 0; s < BKB / M::KS; ++s) {
      Frag af[4], bfr[2];
      int c0, c1;
      if constexpr (M::KS == 64) {
        c0 = ((s * 4 + g * 2) ^ sw) * 16;
        c1 = ((s * 4 + g * 2 + 1) ^ sw) * 16;
      } else {
        c0 = ((s * 2 + g) ^ sw) * 16;
        c1 = c0;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = lds_frag<Frag>(st + a_row_off + i * 32 * BKB + c0, st + a_row_off + i * 32 * BKB + c1);
#pragma unroll
      for (int j = 0; j < 2; ++j) bfr[j] = lds_frag<Frag>(st + b_row_off + j * 32 * BKB + c0, st + b_row_off + j * 32 * BKB + c1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = M::run(af[i], bfr[j], acc[i][j]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue.  Lane holds column n = n0 + wn*64 + j*32 + (lane&31) and rows
  //      m = m0 + wm*128 + i*32 + (e&3) + 8*(e>>2) + 4*(lane>>5)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn * 64 + j * 32 + lrow;
    const bool n_ok = n < p.N;
    const int nc = n_ok ? n : 0;
    float sbn = 1.0f, biasn = 0.0f;
    int adjn = 0;
    if constexpr (DT != DT_BF16) sbn = p.sb[nc];
    if (p.bias != nullptr) biasn = bf2f(p.bias[nc]);
    if constexpr (DT == DT_INT8) {
      if (p.azp_adj != nullptr) adjn = p.azp_adj[nc];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wm * 128 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * g;
        if (m < p.M && n_ok) {
          float y;
          if constexpr (DT == DT_FP8) {
            y = acc[i][j][e] * p.sa[m] * sbn;
            if (p.bias != nullptr) y += biasn;
            y = rbf(y);
          } else if constexpr (DT == DT_INT8) {
            int a = acc[i][j][e];
            if (p.azp != nullptr) a -= p.azp[m] * adjn;
            y = rbf((float)a * (p.sa[m] * sbn));
            if (p.bias != nullptr) y = rbf(y + biasn);
          } else {
            y = acc[i][j][e];
            if (p.bias != nullptr) y += biasn;
            y = rbf(y);
          }
          if constexpr (EPI == FDM_EPI_GELU_TANH) y = gelu_tanh_f(y);
          if constexpr (EPI == FDM_EPI_GELU_ERF) y = gelu_erf_f(y);
          if constexpr (EPI == FDM_EPI_GATE_RESIDUAL) {
            const int64_t b = m / p.rows_per_batch;
            const float gt = bf2f(p.gate[b * p.gate_stride + n]);
            y = bf2f(p.res[(int64_t)m * p.ldc + n]) + rbf(gt * y);
          }
          p.C[(int64_t)m * p.ldc + n] = f2bf(y);
        }
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
  if ((reinterpret_cast<uintptr_t>(a) & 15) || (reinterpret_cast<uintptr_t>(b) & 15) || (reinterpret_cast<uintptr_t>(out) & 1))
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
