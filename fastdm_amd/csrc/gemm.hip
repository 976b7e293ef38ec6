// W8A8 (fp8 / int8) and bf16 GEMM for gfx950:  C[M,N] = epi(A[M,K] . B[N,K]^T), 1 or 2 problems per launch.
//
// Both operands are K-contiguous (A row-major, B = the reference's column-major [K,N] weight,
// i.e. [N,K] row-major), which is exactly the MFMA operand order, so tiles go global -> LDS with
// 16-byte LDS-DMA (global_load_lds_dwordx4) and LDS -> VGPR with ds_read_b128, no transposes.
//
//   tile      256 x 256, K-step 64 bytes; 8 waves = 2 (M) x 4 (N), wave tile 128 x 64
//   fp8       v_mfma_scale_f32_32x32x64_f8f6f4 with unit E8M0 scales (2x the bf16 MFMA rate)
//   int8      v_mfma_i32_32x32x32_i8           bf16: v_mfma_f32_32x32x16_bf16
//   grid      one block per tile; block ids are remapped so each XCD (private L2) owns a contiguous
//             range of tiles, rastered 4 (M) x n inside it.  Two problems (e.g. the image- and text-stream
//             projections of a double block, different weights) can share one launch: the small one's
//             tiles fill the last partial round of the large one.
//   pipeline  4-stage LDS ring of 32 KiB: while step t runs on registers, the fragments of step t+1 are
//             read from LDS and the DMAs of steps t+2, t+3 are in flight.  One raw s_barrier per step
//             behind a COUNTED s_waitcnt vmcnt(4) (never 0 in the loop), so LDS-DMA stays in flight across
//             barriers.  B fragments are double buffered, each A fragment is replaced in place as soon as
//             its MFMAs have issued (fits 256 VGPRs).  Measured alternatives on MI355X, both slower: a
//             two-phase schedule with waves 4-7 one phase behind waves 0-3 (two barriers per step: -35 %),
//             and a persistent grid with a continuous DMA stream across tile seams (cursor bookkeeping in
//             the step body: -15 %).
//   LDS       rows are 64 B; chunk c of row r lives in slot c ^ ((r >> 2) & 3): conflict-free for the
//             ds_read_b128 lane groups.  LDS-DMA writes are lane-linear, so the swizzle is applied to
//             the per-lane SOURCE address.
//   operands  swapped (D^T = B . A^T): a lane owns ONE output row m and 4 consecutive columns per
//             register quad -> the per-row scale is a lane constant.
//   epilogue  scale/bias/activation in registers, rounded to bf16, then each wave transposes its 128x64
//             sub-tile through its own 16 KiB of the (now dead) LDS ring so that every global access is a
//             16-byte row-contiguous vector (8 full 128-byte rows per wave-instruction): 16 stores per
//             lane instead of 32 scattered 8-byte ones; residual / gate are read the same way, all loads
//             before the first store (vmcnt retires in order).
//   edges     M/N: source rows clamped, stores masked.  K tail / stream end: lanes fetch a zero page.
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <atomic>

#include "common.h"

namespace fdm {

enum { DT_FP8 = 0, DT_INT8 = 1, DT_BF16 = 2 };

constexpr int BM = 256, BN = 256, BKB = 64;  // BKB = K-step in bytes
constexpr int GEMM_THREADS = 512;
constexpr int NSTAGE = 4;
constexpr int STAGE_BYTES = (BM + BN) * BKB;  // 32 KiB
constexpr int GROUP_M = 4;
#ifndef FDM_GEMM_SPREAD_DMA
#define FDM_GEMM_SPREAD_DMA 0
#endif
#ifndef FDM_GEMM_W4
#define FDM_GEMM_W4 1
#endif

__device__ __attribute__((aligned(16))) uint8_t g_zero_page[16] = {0};

struct GemmProblem {
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
  int epi;           // FDM_EPI_* (uniform per problem); FDM_EPI_GATE_RESIDUAL_F32 is stored as GATE_RESIDUAL + gate_f32
  int gate_f32;      // the gate vector is fp32 and the epilogue rounds once
  int plane_cols;    // > 0: output column n lands in plane n / plane_cols at column n % plane_cols (fused-qkv -> q|k|v planes)
  int64_t plane_stride;  // elements between planes
  // 3 x 3 convolution as an implicit GEMM (gemm_kernel<DT_BF16, RES, NWN, true>, fdm_conv2d_3x3_nhwc_bf16): A is an NHWC image
  // [batch, cv_hin, cv_win, cv_cin] and row m = (b, y, x) of the virtual A matrix is the 3 x 3 x cv_cin patch around input pixel
  // (y * stride + dy - 1, x * stride + dx - 1), K ordered (tap, channel); cv_up: the input is read through a nearest-neighbour 2x upsampling
  int cv_hin, cv_win, cv_hout, cv_wout, cv_cin, cv_stride, cv_up;
};

struct GemmGroup {
  GemmProblem p[2];
  int n;            // 1 or 2 problems
  int total_tiles;
  // int8 split-K tail (exact: int32 partial sums): the last `split_tiles` tiles of the concatenated tile list are cut into
  // `splits` K ranges each; their blocks write int32 accumulators to `ws` ([tile][part][32 quads][512 threads] x 16 B) and a
  // second launch of the same kernel (mode 1, one block per tail tile) sums the parts and runs the epilogue.
  int split_tiles, splits, mode, n_blocks, spread_shift;
  int32_t* ws;
#ifdef FDM_DEBUG_BUILD
  int debug;        // bring-up ablation (env FDM_GEMM_DEBUG, debug builds only): 4 = no stores
#endif
#ifdef FDM_GEMM_STAMPS
  unsigned long long* stamps;  // [block][4] shader-clock stamps: start / prologue landed / main loop done / epilogue done
#endif
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

// LDS-DMA in the saddr + 32-bit voffset form (wave-uniform 64-bit base in SGPRs, one VGPR per lane): hipcc, given a 64-bit
// pointer, keeps a VGPR PAIR per DMA alive.
__device__ __forceinline__ void glds16_sv(const uint8_t* sbase, uint32_t voff, uint32_t lds_addr) {
  // M0 is written and read inside this statement and NOT restored: hipcc's own LDS-DMAs (the builtin, tail path) set M0
  // themselves right before each use, nothing else in these kernels reads it.  (An "m0" clobber -- ADVICE r3 -- is not honoured by
  // hipcc 7.2: "reserved registers on the clobber list may not be preserved", one warning per statement; what holds the property
  // is tools/micro/audit_gemm_asm.sh, which build.sh now runs on every build's own assembly and which fails the build.)
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %0" : : "s"(sbase), "v"(voff), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ uint32_t lds_addr_of(const uint8_t* p) {
  return (uint32_t)reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) uint8_t*)p);
}

// (fast_gelu_tanh / fast_gelu_erf, the cheap activations of the epilogues, live in common.h: the row-quantisation kernels of
// elementwise.hip apply the same two functions when the activation is deferred to them -- fdm_quant_rowwise_cat2_act)

// The same activations over 16 independent values, written stage by stage: hipcc packs neighbouring scalar f32 ops into
// v_pk_* instructions, and a v_pk_* that consumes the previous instruction's result costs a wait state (s_nop); with 16
// independent lanes of work per stage every packed op has an independent neighbour.
template <int EPI>
__device__ __forceinline__ void gelu16(float (&v)[16]) {
  if constexpr (EPI == FDM_EPI_GELU_TANH) {
    const float c0 = 0.79788456080286535588f * 2.885390081777927f, c1 = c0 * 0.044715f;
    float e[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) e[k] = v[k] * v[k];
#pragma unroll
    for (int k = 0; k < 16; ++k) e[k] = __builtin_fmaf(e[k], c1, c0);
#pragma unroll
    for (int k = 0; k < 16; ++k) e[k] = __builtin_amdgcn_exp2f(v[k] * e[k]);
#pragma unroll
    for (int k = 0; k < 16; ++k) e[k] = __builtin_amdgcn_rcpf(e[k] + 1.0f);
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = __builtin_fmaf(-v[k], e[k], v[k]);
  } else if constexpr (EPI == FDM_EPI_GELU_ERF) {
    float z[16], t[16], pl[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) z[k] = fabsf(v[k]) * 0.70710678118654752440f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t[k] = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z[k], 1.0f));
#pragma unroll
    for (int k = 0; k < 16; ++k) pl[k] = __builtin_fmaf(t[k], 1.061405429f, -1.453152027f);
#pragma unroll
    for (int k = 0; k < 16; ++k) pl[k] = __builtin_fmaf(t[k], pl[k], 1.421413741f);
#pragma unroll
    for (int k = 0; k < 16; ++k) pl[k] = __builtin_fmaf(t[k], pl[k], -0.284496736f);
#pragma unroll
    for (int k = 0; k < 16; ++k) pl[k] = __builtin_fmaf(t[k], pl[k], 0.254829592f) * t[k];
#pragma unroll
    for (int k = 0; k < 16; ++k) z[k] = __builtin_amdgcn_exp2f(-z[k] * z[k] * 1.4426950408889634f);
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float er = __builtin_fmaf(-pl[k], z[k], 1.0f);
      v[k] = 0.5f * v[k] * (1.0f + (v[k] < 0.f ? -er : er));
    }
  }
}

__device__ __forceinline__ void unpack4(const i32x2 v, float (&o)[4]) {
  o[0] = __builtin_bit_cast(float, (uint32_t)v[0] << 16);
  o[1] = __builtin_bit_cast(float, (uint32_t)v[0] & 0xffff0000u);
  o[2] = __builtin_bit_cast(float, (uint32_t)v[1] << 16);
  o[3] = __builtin_bit_cast(float, (uint32_t)v[1] & 0xffff0000u);
}

// RES: the launch contains a FDM_EPI_GATE_RESIDUAL problem (a separate instantiation keeps that epilogue's
// residual registers out of the common kernels).
// NWN = 4: the 256 x 256 tile of 8 waves (2 x 4 wave tiles of 128 x 64), one block per CU.  NWN = 2 (round 5): a 256 x 128 HALF tile of 4 waves
// (2 x 2 of the same wave tiles), 3 stages of 24 KiB so that TWO blocks share a CU -- the same 8 waves per CU, scheduled as two independent units:
// launches whose tile count leaves most of a round empty (SD3.5's N = 2432: 350 tiles on 256 CUs; the SDXL level-2 layers: 40-120 tiles)
// get twice the number of work items at half the size.  Same per-wave code: fragments, MFMA order, epilogue.
// CONV (round 6, bf16 only): the A operand is gathered from an NHWC image, 3 x 3 taps, zero padding 1 (GemmProblem::cv_*): a K step of 64 bytes
// is 32 channels of ONE tap (cv_cin % 32 == 0), so a staging round's source row is the patch pixel of that tap or the zero page.  The SDXL
// UNet's convolutions: fp32 accumulation over all 9 C_in products (MIOpen's bf16 kernels were the whole distance of BASELINE configs[0] from
// the two-implementation envelope, profiles/r06g_sdxl_isolate.txt) on the 8-wave kernel's checked staging; same fragments, MFMAs and epilogue.
template <int DT, bool RES, int NWN = 4, bool CONV = false>
__global__ void __launch_bounds__(128 * NWN, 2) gemm_kernel(const GemmGroup grp) {
  static_assert(!CONV || DT == DT_BF16, "implicit-GEMM convolution: bf16");
  // K ranges of a tile on several blocks, partial accumulators through a workspace, merged by a second launch (mode 1): the int8 GEMMs' exact
  // split-K tail, and -- round 6 -- the convolutions whose M is too small to fill the chip (SDXL level 2: 2 x 32 x 32 pixels = 8 M tiles;
  // fp32 partials summed in a fixed order: deterministic)
  constexpr bool SPLITK = (DT == DT_INT8 || CONV) && NWN == 4;
  using M = Mma<DT>;
  using Acc = typename M::Acc;
  using Frag = typename M::Frag;
  constexpr int NS = BKB / M::KS;  // MFMA k-slices per step (1 for fp8, 2 for int8/bf16)
  constexpr int BNT = 64 * NWN;                    // tile width
  constexpr int RPR = 32 * NWN;                    // rows one staging round of the block moves (threads / 4)
  constexpr int RA = BM / RPR, RB = BNT / RPR;     // staging rounds per stage: A (2 / 4), B (2 / 2)
  constexpr int RBYTES = RPR * BKB;                // LDS bytes of a round
  constexpr int NSTG = NWN == 4 ? NSTAGE : 3;
  constexpr int SBYTES = (BM + BNT) * BKB;
  static_assert(NSTG * SBYTES >= 2 * NWN * 16384, "the epilogue's wave-private tiles live in the dead ring");
  __shared__ __attribute__((aligned(1024))) uint8_t smem[NSTG * SBYTES];
  auto slot_of = [](int kt) { return NSTG == 4 ? (kt & 3) : kt % 3; };
  // s_waitcnt vmcnt(n) alone (gfx9 encoding: vmcnt = bits 15:14 | 3:0, expcnt 6:4 and lgkmcnt 11:8 left at "no wait")
  constexpr int VM1 = RA + RB, VM2 = 2 * (RA + RB);
  constexpr int WAIT_VM1 = (VM1 & 15) | ((VM1 >> 4) << 14) | 0x0F70, WAIT_VM2 = (VM2 & 15) | ((VM2 >> 4) << 14) | 0x0F70;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;

  // ---- block -> (problem, tile): XCD-contiguous ranges of the concatenated tile list, grouped raster inside a problem
  int prob = 0, m0, n0;
  int part = 0, part_slot = 0;   // int8 split-K tail: this block's K range and workspace slot (is_part)
  bool is_part = false;
  {
    const int nwg = grp.total_tiles;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    if constexpr (SPLITK) {
      if (grp.mode == 1) {
        t = grp.total_tiles - grp.split_tiles + (bid >> grp.spread_shift);   // merge launch: 1 << spread_shift blocks per tail tile
      } else if (grp.splits > 1) {
        // Blocks are dealt to the XCDs round-robin, so every XCD gets the same NUMBER of blocks: give each one an equal
        // share of whole tiles (a contiguous range, dispatched first) and an equal share of the K parts (its last round).
        const int tf = grp.total_tiles - grp.split_tiles, nb = grp.n_blocks;
        const int fq = tf / 8 + (xcd < tf % 8 ? 1 : 0), fbase = xcd * (tf / 8) + min(xcd, tf % 8);
        const int cbase = xcd * (nb / 8) + min(xcd, nb % 8), local = bid >> 3;
        if (local < fq) {
          t = fbase + local;
        } else {
          const int pi = (cbase - fbase) + (local - fq);   // part index 0 .. split_tiles * splits - 1
          t = tf + pi / grp.splits;
          part = pi % grp.splits;
          part_slot = pi;          // = (tail tile) * splits + part
          is_part = true;
        }
      }
    }
    const int t0 = grp.p[0].tiles_m * grp.p[0].tiles_n;
    if (t >= t0) {
      prob = 1;
      t -= t0;
    }
    const GemmProblem& pp = grp.p[prob];
    const int in_group = GROUP_M * pp.tiles_n;
    const int gid = t / in_group;
    const int first_m = gid * GROUP_M;
    const int gsz = min(pp.tiles_m - first_m, GROUP_M);
    const int rem = t - gid * in_group;
    m0 = (first_m + rem % gsz) * BM;
    n0 = (rem / gsz) * BNT;
  }
  const GemmProblem& p = grp.p[prob];
#ifdef FDM_GEMM_STAMPS
  unsigned long long* stamp = grp.stamps + (size_t)blockIdx.x * 4;
  if (tid == 0) stamp[0] = __builtin_readcyclecounter();
#endif

  int64_t Kb = p.Kb, k_off = 0;
  int kt_first = 0;      // CONV: the K step this block starts at (its tap / channel-slice counters start there)
  if constexpr (SPLITK) {
    if (is_part) {   // K range of this part, in whole 64-byte steps; the last part takes the K tail
      const int nk_all = (int)((p.Kb + BKB - 1) / BKB);
      const int kt0 = (int)((int64_t)part * nk_all / grp.splits), kt1 = (int)((int64_t)(part + 1) * nk_all / grp.splits);
      k_off = (int64_t)kt0 * BKB;
      Kb = min(p.Kb - k_off, (int64_t)(kt1 - kt0) * BKB);
      kt_first = kt0;
    }
  }

  // ---- staging: thread t, round i moves the 16-byte chunk that lands at (row = i*128 + t/4, slot = t%4)
  const int srow = tid >> 2;
  const int schunk = (tid & 3) ^ ((srow >> 2) & 3);  // source chunk for that slot
  const uint8_t* ga[RA];
  const uint8_t* gb[RB];
#pragma unroll
  for (int i = 0; i < RA; ++i) ga[i] = p.A + (int64_t)min(m0 + i * RPR + srow, p.M - 1) * p.lda + k_off + schunk * 16;
#pragma unroll
  for (int i = 0; i < RB; ++i) gb[i] = p.B + (int64_t)min(n0 + i * RPR + srow, p.N - 1) * p.ldb + k_off + schunk * 16;
  const int nk = (int)((Kb + BKB - 1) / BKB);

  // Every step issues exactly 4 DMAs per wave, also past the end of K (then from a 16-byte zero page, which
  // zero-fills the stage): the loop body is branch-free, s_waitcnt counts are constants, the K tail and the
  // odd-step padding both multiply zeros.
  const uint8_t* zero_page = g_zero_page;
  // CONV: this lane's output pixel of staging round i -- (y * stride - 1, x * stride - 1) of the patch origin in (upsampled) input
  // coordinates and the image's first pixel; rows past M have an origin no tap can bring inside the image
  int cy[RA], cx[RA], cb[RA];
  int c_tap = 0, c_cs = 0;       // tap and 32-channel slice of the NEXT stage() call (stage() is called with kt = 0, 1, 2, ... in order)
  if constexpr (CONV) {
    c_tap = kt_first / (p.cv_cin / 32);
    c_cs = kt_first - c_tap * (p.cv_cin / 32);
    const int hw = p.cv_hout * p.cv_wout;
#pragma unroll
    for (int i = 0; i < RA; ++i) {
      const int m = m0 + i * RPR + srow;
      const int b = m / hw, rem = m - b * hw, y = rem / p.cv_wout, x = rem - y * p.cv_wout;
      cy[i] = m < p.M ? y * p.cv_stride - 1 : -(1 << 20);
      cx[i] = x * p.cv_stride - 1;
      cb[i] = b * p.cv_hin * p.cv_win;
    }
  }
  auto stage = [&](int kt) {
    uint8_t* sA = smem + slot_of(kt) * SBYTES + wave * 1024;
    uint8_t* sB = sA + BM * BKB;
    const int64_t koff = (int64_t)kt * BKB;
    const bool ok = koff + schunk * 16 < Kb;
    if constexpr (CONV) {
      const int dy = c_tap / 3, dx = c_tap - 3 * dy;       // (wave-uniform; taps >= 9 only past the end of K: ok is false there)
      const int hlim = p.cv_up ? 2 * p.cv_hin : p.cv_hin, wlim = p.cv_up ? 2 * p.cv_win : p.cv_win;
      const int64_t coff = (int64_t)c_cs * BKB + schunk * 16;
#pragma unroll
      for (int i = 0; i < RA; ++i) {
        int iy = cy[i] + dy, ix = cx[i] + dx;
        const bool in = ok && (unsigned)iy < (unsigned)hlim && (unsigned)ix < (unsigned)wlim;
        if (p.cv_up) {
          iy >>= 1;
          ix >>= 1;
        }
        const int64_t pix = (int64_t)cb[i] + (int64_t)iy * p.cv_win + ix;
        glds16(in ? p.A + pix * ((int64_t)p.cv_cin * 2) + coff : zero_page, sA + i * RBYTES);
      }
      if (++c_cs == p.cv_cin / 32) {
        c_cs = 0;
        ++c_tap;
      }
    } else {
#pragma unroll
      for (int i = 0; i < RA; ++i) glds16(ok ? ga[i] + koff : zero_page, sA + i * RBYTES);
    }
#pragma unroll
    for (int i = 0; i < RB; ++i) glds16(ok ? gb[i] + koff : zero_page, sB + i * RBYTES);
  };
  // Steps whose whole 64-byte K slice lies inside K need no per-lane bounds select: the source is a wave-uniform base
  // (tile origin + K offset, SALU) plus a loop-invariant 32-bit lane offset, which the compiler can issue in the
  // saddr + voffset form -- no 64-bit VALU address math or v_cndmask in the steady state.
  const uint8_t* baseA = p.A + (int64_t)min(m0, p.M - 1) * p.lda + k_off;
  const uint8_t* baseB = p.B + (int64_t)min(n0, p.N - 1) * p.ldb + k_off;
  uint32_t oa[RA], ob[RB];
#pragma unroll
  for (int i = 0; i < RA; ++i) oa[i] = (uint32_t)((min(m0 + i * RPR + srow, p.M - 1) - min(m0, p.M - 1)) * p.lda) + schunk * 16;
#pragma unroll
  for (int i = 0; i < RB; ++i) ob[i] = (uint32_t)((min(n0 + i * RPR + srow, p.N - 1) - min(n0, p.N - 1)) * p.ldb) + schunk * 16;
  auto stage_inner = [&](int kt) {
    uint8_t* sA = smem + slot_of(kt) * SBYTES + wave * 1024;
    uint8_t* sB = sA + BM * BKB;
    const uint8_t* ka = baseA + (int64_t)kt * BKB;
    const uint8_t* kb = baseB + (int64_t)kt * BKB;
#pragma unroll
    for (int i = 0; i < RA; ++i) glds16(ka + oa[i], sA + i * RBYTES);
#pragma unroll
    for (int i = 0; i < RB; ++i) glds16(kb + ob[i], sB + i * RBYTES);
  };

  // ---- fragment read addresses (byte offsets inside a stage)
  const int lrow = lane & 31, g = lane >> 5;
  const int sw = (lrow >> 2) & 3;
  const int a_row_off = (wm * 128 + lrow) * BKB;
  const int b_row_off = BM * BKB + (wn * 64 + lrow) * BKB;
  Frag fa[NS][4], fb0[NS][2], fb1[NS][2];
  auto frag_cols = [&](int s, int& c0, int& c1) {
    if constexpr (M::KS == 64) {
      c0 = ((g * 2) ^ sw) * 16;
      c1 = ((g * 2 + 1) ^ sw) * 16;
    } else {
      c0 = ((s * 2 + g) ^ sw) * 16;
      c1 = c0;
    }
  };
  auto read_a = [&](int kt, int i) {
    const uint8_t* st = smem + slot_of(kt) * SBYTES + a_row_off + i * 32 * BKB;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      int c0, c1;
      frag_cols(s, c0, c1);
      fa[s][i] = lds_frag<Frag>(st + c0, st + c1);
    }
  };
  auto read_b = [&](Frag (&fb)[NS][2], int kt) {
    const uint8_t* st = smem + slot_of(kt) * SBYTES + b_row_off;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      int c0, c1;
      frag_cols(s, c0, c1);
#pragma unroll
      for (int j = 0; j < 2; ++j) fb[s][j] = lds_frag<Frag>(st + j * 32 * BKB + c0, st + j * 32 * BKB + c1);
    }
  };

  Acc acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;

  // int8 split-K tail, merge launch: the accumulators are the sum of the parts' int32 partials (exact), no K loop
  bool merged = false;
  if constexpr (SPLITK) {      // (the split-K tail's workspace layout is the 8-wave block's; half tiles are never split)
    if (grp.mode == 1) {
      // A CU pulls ~11 B/clk, so reading S x 256 KB through one CU costs as much as a good part of the K loop it replaces:
      // a tile is merged by 1, 2 or 4 blocks (as many as fit one round), each running the waves of its share of the tile
      // (the epilogue below is wave-private).
      merged = true;
      const int32_t* src = grp.ws + ((int64_t)(blockIdx.x >> grp.spread_shift) * grp.splits) * (32 * 512 * 4) + tid * 4;
      const int n_parts = (wave >> (3 - grp.spread_shift)) == (int)(blockIdx.x & ((1 << grp.spread_shift) - 1)) ? grp.splits : 0;
      for (int sp = 0; sp < n_parts; ++sp, src += 32 * 512 * 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const i32x4 v = *reinterpret_cast<const i32x4*>(src + ((i * 2 + j) * 4 + q) * (512 * 4));
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                if constexpr (DT == DT_INT8) {
                  acc[i][j][4 * q + c] += v[c];
                } else {      // fp32 partials, parts in order: deterministic
                  const f32x4 vf = __builtin_bit_cast(f32x4, v);
                  acc[i][j][4 * q + c] += vf[c];
                }
              }
            }
      }
    }
  }
  if (!merged) {
  // ---- prologue: DMAs of steps 0..2 in flight, fragments of step 0 in registers
    stage(0);
    stage(1);
    stage(2);
    __builtin_amdgcn_s_waitcnt(WAIT_VM2);      // stage 0 has landed (stages 1, 2 in flight)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) read_a(0, i);
    read_b(fb0, 0);
#ifdef FDM_GEMM_STAMPS
    if (tid == 0) stamp[1] = __builtin_readcyclecounter();
#endif

    // One step: wait for DMA(t+1) (leaving DMA(t+2) in flight), barrier, issue DMA(t+3), then MFMA(t) with the
    // fragments of step t+1 streaming in behind it.
    auto step = [&](auto inner_c, Frag (&bc)[NS][2], Frag (&bn)[NS][2], int t) {
      if constexpr (NSTG == 3) {
        // three stages: the DMAs issued below refill the slot whose fragments were requested during the previous step -- every wave's reads
        // must have returned before any wave passes the barrier (with four stages that slot has been idle for a whole step)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_waitcnt(WAIT_VM1);      // stage t + 1 has landed (stage t + 2 stays in flight)
      asm volatile("" ::: "memory");
      __builtin_amdgcn_s_barrier();
#if FDM_GEMM_SPREAD_DMA
      // (experiment) one DMA behind each of the four MFMA groups instead of a burst of four behind the barrier
      static_assert(NWN == 4, "FDM_GEMM_SPREAD_DMA hard-codes the 8-wave geometry (4 stages, 2 A + 2 B rounds): not for the half tiles");
      if constexpr (decltype(inner_c)::value) {
        uint8_t* sA = smem + ((t + 3) & (NSTAGE - 1)) * STAGE_BYTES + wave * 1024;
        uint8_t* sB = sA + BM * BKB;
        const uint8_t* ka = baseA + (int64_t)(t + 3) * BKB;
        const uint8_t* kb = baseB + (int64_t)(t + 3) * BKB;
        read_b(bn, t + 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = M::run(bc[s][j], fa[s][i], acc[i][j]);
          read_a(t + 1, i);
          if (i < 2) glds16(ka + oa[i], sA + i * 8192);
          else glds16(kb + ob[i - 2], sB + (i - 2) * 8192);
          __builtin_amdgcn_sched_barrier(0);
        }
        return;
      }
#endif
      if constexpr (decltype(inner_c)::value) stage_inner(t + 3);
      else stage(t + 3);
      read_b(bn, t + 1);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = M::run(bc[s][j], fa[s][i], acc[i][j]);  // swapped: rows = n, cols = m
        read_a(t + 1, i);
      }
    };
    const int nk2 = (nk + 1) & ~1;
    const int nk_full = CONV ? 0 : (int)(Kb / BKB);          // slices that lie completely inside K (CONV: every A slice is gathered: the checked form)
    int t = 0;
    for (; t + 4 < nk_full; t += 2) {              // issues slices t+3 and t+4: both complete
      step(std::true_type{}, fb0, fb1, t);
      step(std::true_type{}, fb1, fb0, t + 1);
    }
    for (; t < nk2; t += 2) {                      // K tail and the zero-filled padding slices
      step(std::false_type{}, fb0, fb1, t);
      step(std::false_type{}, fb1, fb0, t + 1);
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // retire the padding DMAs and the dead fragment reads
  __builtin_amdgcn_s_barrier();                                  // every wave is done with the ring: reuse it below
  if constexpr (SPLITK) {
    if (grp.mode == 1 && (wave >> (3 - grp.spread_shift)) != (int)(blockIdx.x & ((1 << grp.spread_shift) - 1))) return;   // merge launch: the other quarters belong to sibling blocks
    if (is_part) {   // partial sums of this K range (int32, or fp32 bits) -> workspace; the merge launch runs the epilogue
      int32_t* dst = grp.ws + ((int64_t)part_slot) * (32 * 512 * 4) + tid * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const i32x16 ai = __builtin_bit_cast(i32x16, acc[i][j]);      // (the whole register block at once: int32 sums, or the bits of fp32 sums)
            i32x4 v;
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = ai[4 * q + c];
            *reinterpret_cast<i32x4*>(dst + ((i * 2 + j) * 4 + q) * (512 * 4)) = v;
          }
      return;
    }
  }

#ifdef FDM_GEMM_STAMPS
  if (tid == 0) stamp[2] = __builtin_readcyclecounter();
#endif
  // ---- epilogue
  // MFMA layout (swapped operands): lane holds row m = wm*128 + i*32 + (lane&31) and columns
  // n = wn*64 + j*32 + 8*q + 4*(lane>>5) + (0..3) for register quad q = e>>2, i.e. 8-byte pieces [m][chunk j*4+q][half g].
  // Wave-private LDS tile: 128 rows x 128 B, 16-byte chunk c of row r stored at chunk c ^ (r & 7).
  uint8_t* wt = smem + wave * 16384;
  const bool full = (m0 + BM <= p.M) && (n0 + BNT <= p.N);
  auto epilogue = [&](auto epi_c) {
    constexpr int EPI = decltype(epi_c)::value;
    // pass 1: scale / bias / activation, round to bf16, write 8-byte pieces to the LDS tile
    float sam[4];
    int azpm[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int mc = min(m0 + wm * 128 + i * 32 + lrow, p.M - 1);
      sam[i] = 1.0f;
      azpm[i] = 0;
      if constexpr (DT != DT_BF16) sam[i] = p.sa[mc];
      if constexpr (DT == DT_INT8) {
        if (p.azp != nullptr) azpm[i] = p.azp[mc];
      }
    }
    // per-column operands, requested before any of them is used (inside the loop below every load sat in front of its first use:
    // 8 exposed L2 round trips per tile).  Kernels without a residual epilogue request all 8 column groups up front; the RES
    // instantiations (whose gate / residual state is live too) request them one 32-column half at a time -- with all 8 live beside
    // the 128 accumulator registers hipcc spilled 111 VGPRs (236 B of scratch per lane) in the fp8 kernel, 13 / 12 in int8 / bf16.
    f32x4 sbv[8];
    i32x2 biasv[8];
    i32x4 adjv[8];
    const bool has_bias = p.bias != nullptr;
    auto load_cols = [&](int jq0, int jq1) {
#pragma unroll
      for (int jq = jq0; jq < jq1; ++jq) {
        const int nc = min(n0 + wn * 64 + (jq >> 2) * 32 + 8 * (jq & 3) + 4 * g, p.N - 4);
        sbv[jq] = f32x4{1.f, 1.f, 1.f, 1.f};
        biasv[jq] = i32x2{0, 0};
        adjv[jq] = i32x4{0, 0, 0, 0};
        if constexpr (DT != DT_BF16) sbv[jq] = *reinterpret_cast<const f32x4*>(p.sb + nc);
      }
      if (has_bias) {
#pragma unroll
        for (int jq = jq0; jq < jq1; ++jq)
          biasv[jq] = *reinterpret_cast<const i32x2*>(p.bias + min(n0 + wn * 64 + (jq >> 2) * 32 + 8 * (jq & 3) + 4 * g, p.N - 4));
      }
      if constexpr (DT == DT_INT8) {
        if (p.azp_adj != nullptr) {
#pragma unroll
          for (int jq = jq0; jq < jq1; ++jq)
            adjv[jq] = *reinterpret_cast<const i32x4*>(p.azp_adj + min(n0 + wn * 64 + (jq >> 2) * 32 + 8 * (jq & 3) + 4 * g, p.N - 4));
        }
      }
    };
    if constexpr (!RES) load_cols(0, 8);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if constexpr (RES) load_cols(j * 4, j * 4 + 4);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float bias4[4];
        unpack4(biasv[j * 4 + q], bias4);
        const f32x4 sbq = sbv[j * 4 + q];
        const i32x4 adjq = adjv[j * 4 + q];
        float y[16];   // [i][c]
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int e = 4 * q + c;
            float yy;
            if constexpr (DT == DT_FP8) {
              yy = acc[i][j][e] * sam[i] * sbq[c] + bias4[c];
            } else if constexpr (DT == DT_INT8) {
              const int a = acc[i][j][e] - azpm[i] * adjq[c];
              yy = (float)a * (sam[i] * sbq[c]);
              if (has_bias) yy = rbf(yy) + bias4[c];
            } else {
              yy = acc[i][j][e] + bias4[c];
            }
            y[4 * i + c] = (EPI == FDM_EPI_GELU_TANH || EPI == FDM_EPI_GELU_ERF) ? rbf(yy) : yy;
          }
        }
        gelu16<EPI>(y);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          i32x2 o;
          o[0] = (int)pack2bf(y[4 * i], y[4 * i + 1]);
          o[1] = (int)pack2bf(y[4 * i + 2], y[4 * i + 3]);
          const int r = i * 32 + lrow;
          *reinterpret_cast<i32x2*>(wt + r * 128 + (((j * 4 + q) ^ (r & 7)) << 4) + g * 8) = o;
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private tile: no block barrier needed
    __builtin_amdgcn_sched_barrier(0);
    // pass 2: row-contiguous 16-byte pieces.  Instruction it: lane -> row 8*it + lane/8, chunk lane%8.
    const int rsub = lane >> 3, ch = lane & 7;
    const int ncol = n0 + wn * 64 + ch * 8;
    const bool n_ok = full || ncol < p.N;      // N % 8 == 0 is required by the host check for this path
    const int ncc = n_ok ? ncol : 0;
    // plane mode: the 8-column piece of this lane never straddles a plane (plane_cols % 8 == 0)
    const int64_t ccol = p.plane_cols > 0 ? (int64_t)(ncc / p.plane_cols) * p.plane_stride + (ncc % p.plane_cols) : (int64_t)ncc;
    auto lds_row = [&](int it) -> i32x4 {
      const int r = it * 8 + rsub;
      return *reinterpret_cast<const i32x4*>(wt + r * 128 + ((ch ^ (r & 7)) << 4));
    };
    auto store_row = [&](int it, const i32x4& v) {
      const int m = m0 + wm * 128 + it * 8 + rsub;
#ifdef FDM_DEBUG_BUILD
      if (grp.debug & 4) return;
#endif
      if (n_ok && (full || m < p.M)) *reinterpret_cast<i32x4*>(p.C + (int64_t)m * p.ldc + ccol) = v;
    };
    // out = res + gate * y.  bf16 gates (FLUX / Qwen-Image): the product is rounded to bf16 before the add, as the
    // reference's two bf16 ops do; fp32 gates (Wan: `(x.float() + y * gate).type_as(x)`, wan.py:102,113): one rounding.
    auto combine = [&](auto f32_c, const i32x4& yv, const float (&g8)[8], const i32x4& rv) -> i32x4 {
      constexpr bool F32 = decltype(f32_c)::value;
      i32x4 o;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const uint32_t yw = (uint32_t)yv[w], rw = (uint32_t)rv[w];
        float plo = g8[2 * w] * __builtin_bit_cast(float, yw << 16);
        float phi = g8[2 * w + 1] * __builtin_bit_cast(float, yw & 0xffff0000u);
        if constexpr (!F32) {
          plo = rbf(plo);
          phi = rbf(phi);
        }
        o[w] = (int)pack2bf(__builtin_bit_cast(float, rw << 16) + plo, __builtin_bit_cast(float, rw & 0xffff0000u) + phi);
      }
      return o;
    };
    if constexpr (EPI == FDM_EPI_GATE_RESIDUAL) {
      const int m_first = m0 + wm * 128, m_last = min(m0 + wm * 128 + 128, p.M) - 1;
      const bool one_batch = (m_first / p.rows_per_batch) == (m_last / p.rows_per_batch);
      auto res_row = [&](int it) -> i32x4 {
        const int mc = min(m0 + wm * 128 + it * 8 + rsub, p.M - 1);
        return *reinterpret_cast<const i32x4*>(p.res + (int64_t)mc * p.ldc + ncc);
      };
      auto load_gate = [&](int64_t batch, float (&g8)[8]) {
        if (p.gate_f32) {
          const float* gp = reinterpret_cast<const float*>(p.gate) + batch * p.gate_stride + ncc;
          const f32x4 a = *reinterpret_cast<const f32x4*>(gp), c = *reinterpret_cast<const f32x4*>(gp + 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            g8[i] = a[i];
            g8[4 + i] = c[i];
          }
        } else {
          const i32x4 gv = *reinterpret_cast<const i32x4*>(p.gate + batch * p.gate_stride + ncc);
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            g8[2 * w] = __builtin_bit_cast(float, (uint32_t)gv[w] << 16);
            g8[2 * w + 1] = __builtin_bit_cast(float, (uint32_t)gv[w] & 0xffff0000u);
          }
        }
      };
      bool fast = full;
#ifdef FDM_DEBUG_BUILD
      fast = fast && !(grp.debug & 4);
#endif
      bf16_t* cptr = p.C + (int64_t)(m0 + wm * 128 + rsub) * p.ldc + ccol;
      const int64_t step8 = 8 * p.ldc;
      const bf16_t* rptr = p.res + (int64_t)(m0 + wm * 128 + rsub) * p.ldc + ncc;
      auto pipelined = [&](auto f32_c) {
        // whole tiles only (unpredicated loads / stores off two base pointers).  Groups of 4 row-instructions: loads(g + 1) run one
        // group ahead of combine(g) + stores(g); every load of a group precedes the stores of the previous one in program order
        // (vmcnt retires in order).  The earlier two-halves form (8 + 8 rows, four 8-quad arrays and a 64-bit address per row live
        // beside them) is what hipcc spilled in this kernel: 111 VGPRs, 236 B of scratch per lane.
        float g8[8];
        load_gate(m_first / p.rows_per_batch, g8);
        i32x4 rr[2][4], oo[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) rr[0][it] = *reinterpret_cast<const i32x4*>(rptr + it * step8);
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {
          if (gi < 3) {
#pragma unroll
            for (int it = 0; it < 4; ++it) rr[(gi + 1) & 1][it] = *reinterpret_cast<const i32x4*>(rptr + ((gi + 1) * 4 + it) * step8);
          }
#pragma unroll
          for (int it = 0; it < 4; ++it) oo[it] = combine(f32_c, lds_row(gi * 4 + it), g8, rr[gi & 1][it]);
          asm volatile("" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int it = 0; it < 4; ++it) *reinterpret_cast<i32x4*>(cptr + (gi * 4 + it) * step8) = oo[it];
        }
      };
      if (one_batch && fast) {
        if (p.gate_f32) pipelined(std::true_type{});
        else pipelined(std::false_type{});
      } else {
        // edge tiles and waves whose 128 rows straddle a batch boundary: per-row gate vector, guarded stores, unpipelined
        // (every kernel of this file has a zero private segment: tests/test_abi.py)
        for (int it = 0; it < 16; ++it) {
          const int mc = min(m0 + wm * 128 + it * 8 + rsub, p.M - 1);
          float g8[8];
          load_gate(mc / p.rows_per_batch, g8);
          const i32x4 rr = res_row(it);
          store_row(it, p.gate_f32 ? combine(std::true_type{}, lds_row(it), g8, rr) : combine(std::false_type{}, lds_row(it), g8, rr));
        }
      }
    } else {
      bool fast = full;
#ifdef FDM_DEBUG_BUILD
      fast = fast && !(grp.debug & 4);
#endif
      if (fast) {
        // whole tile (the common case): all 16 LDS reads in flight, then 16 unpredicated stores off one base pointer.  The guarded
        // form below re-tests `full`, re-reads M / ldc / C from the kernel arguments and waits for its LDS read in every iteration.
        i32x4 v[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) v[it] = lds_row(it);
        bf16_t* cptr = p.C + (int64_t)(m0 + wm * 128 + rsub) * p.ldc + ccol;
        const int64_t step8 = 8 * p.ldc;
#pragma unroll
        for (int it = 0; it < 16; ++it) *reinterpret_cast<i32x4*>(cptr + it * step8) = v[it];
      } else {
#pragma unroll
        for (int it = 0; it < 16; ++it) store_row(it, lds_row(it));
      }
    }
  };
  if constexpr (RES) {
    if (p.epi == FDM_EPI_GATE_RESIDUAL) epilogue(std::integral_constant<int, FDM_EPI_GATE_RESIDUAL>{});
    else if (p.epi == FDM_EPI_GELU_TANH) epilogue(std::integral_constant<int, FDM_EPI_GELU_TANH>{});
    else if (p.epi == FDM_EPI_GELU_ERF) epilogue(std::integral_constant<int, FDM_EPI_GELU_ERF>{});
    else epilogue(std::integral_constant<int, FDM_EPI_NONE>{});
  } else {
    if (p.epi == FDM_EPI_GELU_TANH) epilogue(std::integral_constant<int, FDM_EPI_GELU_TANH>{});
    else if (p.epi == FDM_EPI_GELU_ERF) epilogue(std::integral_constant<int, FDM_EPI_GELU_ERF>{});
    else epilogue(std::integral_constant<int, FDM_EPI_NONE>{});
  }
#ifdef FDM_GEMM_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (tid == 0) stamp[3] = __builtin_readcyclecounter();
#endif
}


// =====================================================================================================================
// gemm_w4_kernel: the same 256 x 256 tile, laid out for the POWER limit of the chip (profiles/r03_experiments.md: with all
// 256 CUs in MFMA-dense code the clock falls to ~1.35 GHz, so wall time follows energy per flop, not cycles).
//   waves     4 (ONE per SIMD), 2 (M) x 2 (N), wave tile 128 x 128: a wave reads 128 + 128 operand rows per K byte instead of
//             128 + 64, i.e. 64 KB of LDS fragment reads per 64 bytes of K and CU instead of 96 KB
//   MFMA      16x16 shapes: v_mfma_f32_16x16x128_f8f6f4 (non-scaled encoding) / v_mfma_i32_16x16x64_i8 / v_mfma_f32_16x16x32_bf16 (the chip
//             holds a higher clock on them than on the 32x32 shapes)
//   K step    128 bytes, two 64 KiB stages (512 rows x 128 B): an LDS-DMA instruction moves 8 whole 128-byte lines.
//             16-byte chunk c of row r sits in slot ((c >> 1) ^ g(r)) * 2 + (c & 1), g(r) = bit 1 of r | bit 3 of r << 1: the
//             16-lane groups of ds_read_b128 (lanes {0-3, 12-15, 20-27}, ...) then cover all 64 banks for both fragment
//             shapes (fp8: lane (i, q) reads 32 bytes = chunk pair q of row i, odd q take the odd chunk first -- A and B
//             operands alike, so products still pair up; int8 / bf16: chunk 4 s + q).
//   pipeline  B fragments of step t + 1 (64 registers) are read during the first half of step t, A fragments stream one
//             16-row block ahead of their MFMAs.  ONE barrier per step, after the last A read of stage t: behind it the DMAs
//             of A(t + 2) go into A(t)'s buffer and those of B(t + 3) into B(t + 1)'s (already in registers); both have a
//             whole step to land (vmcnt(0) + the next barrier).
//   epilogue  same arithmetic as gemm_kernel; the wave's 128 x 128 bf16 sub-tile goes through 32 KiB of the dead ring
//             (8-byte pieces in, 16-byte row-contiguous pieces out: every global access is a full 256-byte row segment).
// Outputs: int8 bit-identical to gemm_kernel (integer sums); fp8 / bf16 within fp32 summation-order effects of the MFMA.
// The 256 accumulator registers of a wave (64 blocks of 16 x 16) live in a[0:255] and are touched by inline asm ONLY: hipcc, given
// 64 x f32x4 accumulators and a 256 + 256 register split, rotates them through copies (492 v_accvgpr moves per two K steps).
// The kernel must therefore never spill to AGPRs: tools/micro/audit_gemm_asm.sh checks the .s (no compiler v_accvgpr_*, no
// scratch).  Block IDX = mb * 8 + nb -> a[4 IDX : 4 IDX + 3].
#define FDM_A16(b) "a" #b
#define FDM_ACLOB16(x) FDM_A16(x##0), FDM_A16(x##1), FDM_A16(x##2), FDM_A16(x##3), FDM_A16(x##4), FDM_A16(x##5), FDM_A16(x##6), FDM_A16(x##7), FDM_A16(x##8), FDM_A16(x##9)
#define FDM_ACLOB_ALL                                                                                                               \
  "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", FDM_ACLOB16(1), FDM_ACLOB16(2), FDM_ACLOB16(3), FDM_ACLOB16(4), FDM_ACLOB16(5), \
      FDM_ACLOB16(6), FDM_ACLOB16(7), FDM_ACLOB16(8), FDM_ACLOB16(9), FDM_ACLOB16(10), FDM_ACLOB16(11), FDM_ACLOB16(12), FDM_ACLOB16(13),     \
      FDM_ACLOB16(14), FDM_ACLOB16(15), FDM_ACLOB16(16), FDM_ACLOB16(17), FDM_ACLOB16(18), FDM_ACLOB16(19), FDM_ACLOB16(20),                 \
      FDM_ACLOB16(21), FDM_ACLOB16(22), FDM_ACLOB16(23), FDM_ACLOB16(24), "a250", "a251", "a252", "a253", "a254", "a255"

template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

template <int DT>
struct Mma16;
template <>
struct Mma16<DT_FP8> {
  using Acc = f32x4;
  static constexpr int KS = 128;  // bytes of K per MFMA
  using Frag = i32x8;
  template <int IDX, bool ZERO = false>
  static __device__ __forceinline__ void run(const Frag& a, const Frag& b) {
    // the NON-scaled encoding (cbsz = blgp = 0: e4m3 x e4m3): bit-identical to v_mfma_scale_f32_16x16x128_f8f6f4 with unit E8M0 scales
    // (round 3's form; 0 of 80 M elements differ), half the instruction bytes and no scale register to read: 1-2 % in the A/B of round 4
    if constexpr (ZERO)   // a tile's first K step: C = 0 as an inline constant (the 256 accumulator registers are never zeroed)
      asm volatile("v_mfma_f32_16x16x128_f8f6f4 a[%c2:%c3], %0, %1, 0" ::"v"(a), "v"(b), "n"(4 * IDX), "n"(4 * IDX + 3));
    else
      asm volatile("v_mfma_f32_16x16x128_f8f6f4 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "n"(4 * IDX), "n"(4 * IDX + 3));
  }
};
template <>
struct Mma16<DT_INT8> {
  using Acc = i32x4;
  static constexpr int KS = 64;
  using Frag = i32x4;
  template <int IDX, bool ZERO = false>
  static __device__ __forceinline__ void run(const Frag& a, const Frag& b) {
    if constexpr (ZERO) asm volatile("v_mfma_i32_16x16x64_i8 a[%c2:%c3], %0, %1, 0" ::"v"(a), "v"(b), "n"(4 * IDX), "n"(4 * IDX + 3));
    else asm volatile("v_mfma_i32_16x16x64_i8 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "n"(4 * IDX), "n"(4 * IDX + 3));
  }
};
template <>
struct Mma16<DT_BF16> {
  using Acc = f32x4;
  static constexpr int KS = 64;
  using Frag = i32x4;
  template <int IDX, bool ZERO = false>
  static __device__ __forceinline__ void run(const Frag& a, const Frag& b) {
    if constexpr (ZERO) asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, 0" ::"v"(a), "v"(b), "n"(4 * IDX), "n"(4 * IDX + 3));
    else asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(a), "v"(b), "n"(4 * IDX), "n"(4 * IDX + 3));
  }
};
// accumulator block IDX as raw 32-bit words (float bits for fp8 / bf16, int32 for int8)
template <int IDX>
__device__ __forceinline__ i32x4 acc_get() {
  int a, b, c, d;
  asm volatile("v_accvgpr_read_b32 %0, a[%c4]\n\tv_accvgpr_read_b32 %1, a[%c5]\n\tv_accvgpr_read_b32 %2, a[%c6]\n\tv_accvgpr_read_b32 %3, a[%c7]"
               : "=v"(a), "=v"(b), "=v"(c), "=v"(d)
               : "n"(4 * IDX), "n"(4 * IDX + 1), "n"(4 * IDX + 2), "n"(4 * IDX + 3));
  return i32x4{a, b, c, d};
}
template <int IDX>
__device__ __forceinline__ void acc_set(const i32x4& v) {
  asm volatile("v_accvgpr_write_b32 a[%c4], %0\n\tv_accvgpr_write_b32 a[%c5], %1\n\tv_accvgpr_write_b32 a[%c6], %2\n\tv_accvgpr_write_b32 a[%c7], %3"
               :
               : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "n"(4 * IDX), "n"(4 * IDX + 1), "n"(4 * IDX + 2), "n"(4 * IDX + 3));
}

// 1: int8 / bf16 spread the fragment reads of a group behind its first MFMAs instead of issuing them in one burst (A/B switch)
#ifndef FDM_W4_SPREAD_READS
#define FDM_W4_SPREAD_READS 1
#endif
#ifndef FDM_W4_TOUCH_AHEAD
#define FDM_W4_TOUCH_AHEAD 3      // K steps between a B slice's cooperative touch and its LDS-DMAs (0: off; A/B switch)
#endif
constexpr int W4_THREADS = 256;
constexpr int W4_STAGE = (BM + BN) * 128;   // 64 KiB

// MB = 16-row activation blocks per wave: 8 = 256-row tiles; 7 = 224-row tiles (round 6, opt-in: fdm_set_gemm_tile_rows -- a launch whose tile
// count sits just above a whole number of rounds takes more, smaller tiles: FLUX's N = 3072 projections 408 -> 468; same MFMAs in the same K
// order per output element, so the results are bit-identical; what it is worth is measured beside `tile_rows_224` below).
template <int DT, bool RES, int MB = 8>
__global__ void __launch_bounds__(W4_THREADS, 1) gemm_w4_kernel(const GemmGroup grp) {
  static_assert(MB == 8 || MB == 7, "gemm_w4_kernel: 256- or 224-row tiles");
  constexpr int WROWS = 16 * MB, BMW = 2 * WROWS, NIT = WROWS / 4;    // rows of a wave's / the block's tile; 4-row store instructions per wave
  using M = Mma16<DT>;
  using Acc = typename M::Acc;
  using Frag = typename M::Frag;
  constexpr int NS = 128 / M::KS;  // MFMA k-slices per step (1 for fp8, 2 for int8 / bf16)
  __shared__ __attribute__((aligned(1024))) uint8_t smem[2 * W4_STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // ---- block -> (problem, tile): as gemm_kernel
  int prob = 0, m0, n0;
  int part = 0, part_slot = 0;
  bool is_part = false;
  {
    const int nwg = grp.total_tiles;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    if constexpr (DT == DT_INT8 && MB == 8) {
      if (grp.mode == 1) {
        t = grp.total_tiles - grp.split_tiles + (bid >> grp.spread_shift);
      } else if (grp.splits > 1) {
        const int tf = grp.total_tiles - grp.split_tiles, nb = grp.n_blocks;
        const int fq = tf / 8 + (xcd < tf % 8 ? 1 : 0), fbase = xcd * (tf / 8) + min(xcd, tf % 8);
        const int cbase = xcd * (nb / 8) + min(xcd, nb % 8), local = bid >> 3;
        if (local < fq) {
          t = fbase + local;
        } else {
          const int pi = (cbase - fbase) + (local - fq);
          t = tf + pi / grp.splits;
          part = pi % grp.splits;
          part_slot = pi;
          is_part = true;
        }
      }
    }
    const int t0 = grp.p[0].tiles_m * grp.p[0].tiles_n;
    if (t >= t0) {
      prob = 1;
      t -= t0;
    }
    const GemmProblem& pp = grp.p[prob];
    const int in_group = GROUP_M * pp.tiles_n;
    const int gid = t / in_group;
    const int first_m = gid * GROUP_M;
    const int gsz = min(pp.tiles_m - first_m, GROUP_M);
    const int rem = t - gid * in_group;
    m0 = (first_m + rem % gsz) * BMW;
    n0 = (rem / gsz) * BN;
  }
  const GemmProblem& p = grp.p[prob];
#ifdef FDM_GEMM_STAMPS
  unsigned long long* stamp = grp.stamps + (size_t)blockIdx.x * 8;
  if (tid == 0) {
    stamp[0] = __builtin_readcyclecounter();
    stamp[6] = __builtin_amdgcn_s_memrealtime();   // 100 MHz: in-kernel clock = (stamp[3] - stamp[0]) / (stamp[7] - stamp[6]) x 100 MHz
  }
#endif

  int64_t Kb = p.Kb, k_off = 0;
  if constexpr (DT == DT_INT8 && MB == 8) {
    if (is_part) {   // K range of this part in 64-byte units (the host's split arithmetic); the last part takes the K tail
      const int nk_all = (int)((p.Kb + BKB - 1) / BKB);
      const int kt0 = (int)((int64_t)part * nk_all / grp.splits), kt1 = (int)((int64_t)(part + 1) * nk_all / grp.splits);
      k_off = (int64_t)kt0 * BKB;
      Kb = min(p.Kb - k_off, (int64_t)(kt1 - kt0) * BKB);
    }
  }

  // ---- staging: round r (0..15) of a stage moves rows r * 32 + wave * 8 + lane / 8 (rounds 0-7: A, 8-15: B); lane -> slot lane % 8
  const int drow = wave * 8 + (lane >> 3);
  const int dg = ((drow >> 1) & 1) | (((drow >> 3) & 1) << 1);
  const int dchunk = ((((lane & 7) >> 1) ^ dg) << 1) | (lane & 1);     // source chunk of this lane's slot
  const uint8_t* bA = p.A + (int64_t)min(m0, p.M - 1) * p.lda + k_off;
  const uint8_t* bB = p.B + (int64_t)min(n0, p.N - 1) * p.ldb + k_off;
  uint32_t da[8], db[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    da[r] = (uint32_t)((min(m0 + r * 32 + drow, p.M - 1) - min(m0, p.M - 1)) * p.lda) + dchunk * 16;
    db[r] = (uint32_t)((min(n0 + r * 32 + drow, p.N - 1) - min(n0, p.N - 1)) * p.ldb) + dchunk * 16;
  }
  const uint8_t* zero_page = g_zero_page;
  // Stage kt: A rows in smem[(kt & 1) * 64 KiB ..], B rows 32 KiB behind them.  A and B are issued separately: B runs one
  // stage further ahead (its fragments are read a whole step before their MFMAs).
  auto stage_half = [&](auto inner_c, auto nr_c, int kt, const uint8_t* gbase, const uint32_t (&doff)[8], int lds_off) {
    constexpr int NR = decltype(nr_c)::value;     // rounds of 32 rows: MB for A, 8 for B
    uint8_t* sX = smem + (kt & 1) * W4_STAGE + lds_off + wave * 1024;
    const int64_t koff = (int64_t)kt * 128;
    if constexpr (decltype(inner_c)::value) {
      const uint32_t la = lds_addr_of(sX);
#pragma unroll
      for (int r = 0; r < NR; ++r) glds16_sv(gbase + koff, doff[r], la + r * 4096);
    } else {
      const bool ok = koff + dchunk * 16 < Kb;   // K % 16 == 0: a chunk lies inside K or outside
      // (the offsets are made opaque so that hipcc does not keep 16 loop-invariant 64-bit lane addresses alive for this path)
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        uint32_t o = doff[r];
        asm volatile("" : "+v"(o));
        glds16(ok ? gbase + koff + o : zero_page, sX + r * 4096);
      }
    }
  };
  auto stage_a = [&](auto inner_c, int kt) { stage_half(inner_c, std::integral_constant<int, MB>{}, kt, bA, da, 0); };
  auto stage_b = [&](auto inner_c, int kt) { stage_half(inner_c, std::integral_constant<int, 8>{}, kt, bB, db, BM * 128); };
  // DMA j (0..15) of the set issued behind step t's barrier: rounds 0..7 of A(t + 2), then rounds 0..7 of B(t + 3)
  auto dma_one = [&](auto inner_c, auto jc, int t) {
    constexpr int j = decltype(jc)::value, r = j & 7;
    if constexpr (j < 8 && r >= MB) return;      // (224-row tiles: A has 7 rounds)
    const int kt = j < 8 ? t + 2 : t + 3;
    const uint8_t* gbase = j < 8 ? bA : bB;
    const uint32_t off = j < 8 ? da[r] : db[r];
    uint8_t* sX = smem + (kt & 1) * W4_STAGE + (j < 8 ? 0 : BM * 128) + wave * 1024 + r * 4096;
    const int64_t koff = (int64_t)kt * 128;
    if constexpr (decltype(inner_c)::value) {
#ifndef FDM_W4_NODMA      // (timing ablation only)
      glds16_sv(gbase + koff, off, lds_addr_of(sX));
#endif
    } else {
      uint32_t o = off;
      asm volatile("" : "+v"(o));
      glds16(koff + dchunk * 16 < Kb ? gbase + koff + o : zero_page, sX);
    }
  };

  // ---- cooperative weight touch (round 6).  Inside the denoise step every launch streams weights nobody has read since the last step
  // (8.6 GB per step against 256 MiB of Infinity Cache).  The LDS-DMAs of a stage are issued 0.5-1 K step ahead of their readers: enough for an
  // L2 / Infinity-Cache hit, not for an HBM miss, so the FIRST round of tiles -- whoever asks for a B line first -- ran at up to half speed
  // (tools/micro/gemm_lib_ab.py, same launch warm / cold: K = 12288 gate + residual 235 / 302 us, K = 3072 82 / 103; hipBLASLt 239 / 267, 75 / 84).
  // While the first round computes, its blocks TOGETHER ask for one dword of every 128-byte line of B's K slice 3 + TOUCH_AHEAD steps ahead of
  // the DMAs: block b the rows b * rpb ... of each problem's B, one masked load per problem and step.  Each line is asked for ONCE chip-wide and
  // lands in the Infinity Cache, where every XCD finds it (every tile touching its own slices doubles the L1 fill traffic: 25 % slower, measured).
  // The dwords are never read; the touch sits right behind the step's barrier and the next step's vmcnt(0) retires it (a whole step to land);
  // the empty asm that "reads" the registers keeps them reserved until then.
  // The activations get the same treatment when they are large (K >= 8192: the [M, K] fp8 matrix the quantiser has just written is 100-134 MB
  // and the tail of its own traffic has pushed the head of it out of the Infinity Cache again).
  constexpr int TOUCH_AHEAD = FDM_W4_TOUCH_AHEAD;
  bool toucher = false, t_ok[4] = {false, false, false, false};     // B of problem 0 / 1, A of problem 0 / 1
  uint32_t t_off[4] = {0, 0, 0, 0}, touch_v[4] = {0, 0, 0, 0};
  if constexpr (TOUCH_AHEAD > 0) {
    const int g0 = min((int)gridDim.x, 256);
    toucher = (int)blockIdx.x < g0 && grp.mode == 0 && grp.splits <= 1;
    if (toucher) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (i < grp.n) {
          const int rpb = (grp.p[i].N + g0 - 1) / g0;
          const int row = (int)blockIdx.x * rpb + tid;
          t_ok[i] = tid < rpb && row < grp.p[i].N;
          t_off[i] = (uint32_t)((int64_t)row * grp.p[i].ldb);
          if (grp.p[i].Kb >= 8192) {
            const int rpa = (grp.p[i].M + g0 - 1) / g0;
            const int rowa = (int)blockIdx.x * rpa + tid;
            t_ok[2 + i] = tid < rpa && rowa < grp.p[i].M;
            t_off[2 + i] = (uint32_t)((int64_t)rowa * grp.p[i].lda);
          }
        }
      }
    }
  }
  auto touch = [&](int kt) {     // kt: the K slice asked for (wave-uniform)
    if constexpr (TOUCH_AHEAD > 0) {
      if (toucher) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (t_ok[i] && (int64_t)(kt + 1) * 128 <= grp.p[i & 1].Kb)
            asm volatile("global_load_dword %0, %1, %2" : "=&v"(touch_v[i]) : "v"(t_off[i]), "s"((i < 2 ? grp.p[i & 1].B : grp.p[i & 1].A) + (int64_t)kt * 128) : "memory");
      }
    }
  };
  auto touch_retired = [&]() {   // behind a vmcnt(0): the touch registers may be reused from here
    if constexpr (TOUCH_AHEAD > 0) asm volatile("" ::"v"(touch_v[0]), "v"(touch_v[1]), "v"(touch_v[2]), "v"(touch_v[3]));
  };

  // ---- fragment reads: lane (fi = lane % 16, fq = lane / 16) of block b reads row b * 16 + fi
  const int fi = lane & 15, fq = lane >> 4;
  const int fg = ((fi >> 1) & 1) | (((fi >> 3) & 1) << 1);
  const int a_row = (wm * WROWS + fi) * 128, b_row = BM * 128 + (wn * 128 + fi) * 128;
  // fp8: chunk pair fq, odd fq first take the odd chunk (bank spread; same rule for both operands)
#ifdef FDM_W4_NOSWAP
  const int f8_0 = ((fq ^ fg) << 1) * 16, f8_1 = f8_0 ^ 16;
#else
  const int f8_0 = (((fq ^ fg) << 1) | (fq & 1)) * 16, f8_1 = f8_0 ^ 16;
#endif
  // int8 / bf16 (two k-slices per step): k-slice s of block blk alone
  auto read_frag_slice = [&](const uint8_t* st_row, int blk, int s) {
    const int c = s * 4 + fq;
    return lds_frag<Frag>(st_row + blk * 16 * 128 + ((((c >> 1) ^ fg) << 1) | (c & 1)) * 16, nullptr);
  };
  constexpr bool SPREAD_READS = FDM_W4_SPREAD_READS && NS == 2;
  auto read_frag = [&](Frag (&f)[NS], const uint8_t* st_row, int blk) {
    const uint8_t* rp = st_row + blk * 16 * 128;
    if constexpr (M::KS == 128) {
      f[0] = lds_frag<Frag>(rp + f8_0, rp + f8_1);
    } else {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int c = s * 4 + fq;
        f[s] = lds_frag<Frag>(rp + ((((c >> 1) ^ fg) << 1) | (c & 1)) * 16, nullptr);
      }
    }
  };

  // accumulator block [mb][nb] = a[4 (8 mb + nb) ..]: lane (fi, fq) holds C[m = mb * 16 + fi][n = nb * 16 + fq * 4 + 0..3].
  // The accumulators are never zeroed: the MFMAs of a tile's first K step take C = 0 as an inline constant (256 v_accvgpr_write
  // in front of the first DMA were ~1 k cycles of every tile's prologue).  This empty statement, clobbering every AGPR, is what
  // makes the kernel descriptor allocate a[0:255].
  asm volatile("" ::: FDM_ACLOB_ALL);

  // int8 split-K tail, merge launch: accumulators = sum of the parts' int32 partials (exact), no K loop
  bool merged = false;
  if constexpr (DT == DT_INT8 && MB == 8) {
    if (grp.mode == 1) {
      merged = true;
      const int32_t* src = grp.ws + ((int64_t)(blockIdx.x >> grp.spread_shift) * grp.splits) * (64 * 256 * 4) + tid * 4;
      const int n_parts = (wave >> (2 - grp.spread_shift)) == (int)(blockIdx.x & ((1 << grp.spread_shift) - 1)) ? grp.splits : 0;
      // 16 accumulator quads at a time: the parts' 16 loads are in flight together (one quad with its own runtime loop over the
      // parts, as first written, is a chain of 64 x S dependent round trips)
      static_for<4>([&](auto cc) {
        constexpr int C0 = decltype(cc)::value * 16;
        i32x4 sum[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) sum[j] = i32x4{0, 0, 0, 0};
        for (int sp = 0; sp < n_parts; ++sp) {
          i32x4 v[16];
#pragma unroll
          for (int j = 0; j < 16; ++j) v[j] = *reinterpret_cast<const i32x4*>(src + (int64_t)sp * (64 * 256 * 4) + (C0 + j) * (256 * 4));
#pragma unroll
          for (int j = 0; j < 16; ++j) sum[j] += v[j];
        }
        static_for<16>([&](auto jc) { acc_set<C0 + decltype(jc)::value>(sum[decltype(jc)::value]); });
      });
    }
  }
  if (!merged) {
    Frag fbx[NS][8], fby[NS][8];   // B fragments of the current / next step, [k-slice][nb]
    Frag fa[2][NS];                // A fragments, one 16-row block ahead of their MFMAs
    auto read_b_blk = [&](Frag (&fb)[NS][8], int buf, int nb) {
      Frag f[NS];
      read_frag(f, smem + buf * W4_STAGE + b_row, nb);
#pragma unroll
      for (int s = 0; s < NS; ++s) fb[s][nb] = f[s];
    };
    const int nk = (int)((Kb + 127) / 128);
    const int nk_full = (int)(Kb / 128);
    if constexpr (TOUCH_AHEAD > 0) {   // the slices the K loop's touches do not reach: 3 .. 2 + TOUCH_AHEAD (0-2 are DMAed right here)
#pragma unroll
      for (int kt = 3; kt < 3 + TOUCH_AHEAD; ++kt) {
        touch(kt);
        touch_retired();     // (one register pair serves them all: none of the values is ever read)
      }
    }
    if (nk_full >= 3) {      // stages 0..2 lie inside K: the select-free DMA form (the checked one costs ~6 VALU per DMA)
      stage_a(std::true_type{}, 0);
      stage_b(std::true_type{}, 0);
      stage_a(std::true_type{}, 1);
      stage_b(std::true_type{}, 1);
    } else {
      stage_a(std::false_type{}, 0);
      stage_b(std::false_type{}, 0);
      stage_a(std::false_type{}, 1);
      stage_b(std::false_type{}, 1);
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MB + 8) : "memory");   // stage 0 (stage 1 lands while its B fragments are read)
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) read_b_blk(fbx, 0, nb);
    read_frag(fa[0], smem + a_row, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();          // stage 1 has landed (step 0 reads its B fragments); every wave holds B(0) in registers: its buffer takes B(2)
    if (nk_full >= 3) stage_b(std::true_type{}, 2);
    else stage_b(std::false_type{}, 2);
#ifdef FDM_GEMM_STAMPS
    if (tid == 0) stamp[1] = __builtin_readcyclecounter();
#endif
    // One step = 8 groups of 8 (NS x 8) MFMAs, group g on A block g.  Per group, in this order (pinned by sched_barriers):
    //   wait until A block g has landed -- it was issued a whole group ago, followed only by that group's 4 B reads: a COUNTED
    //   lgkmcnt(4) (LDS returns in order; at most 10 reads are ever outstanding, the counter holds 15);
    //   issue the reads of A block g + 1 (2) and, for g < 4, of B blocks 2 g, 2 g + 1 of step t + 1 (4);
    //   the MFMAs.
    // The block's one barrier per step sits in group 7 behind the wait: every read of stage t has returned and the DMAs issued
    // a step ago (A(t + 1), B(t + 2)) have landed; behind it A(t + 2) goes into A(t)'s buffer and B(t + 3) into B(t + 1)'s
    // (in registers since group 3), and A block 0 of stage t + 1 is read.
    // (MB = 7: an odd number of A blocks per step, so the two A fragment registers swap roles every step: PAR = t & 1 as a constant)
    auto step = [&](auto inner_c, auto first_c, auto par_c, Frag (&bc)[NS][8], Frag (&bn)[NS][8], int t) {
      constexpr int PAR = (MB & 1) ? decltype(par_c)::value : 0;
      const int cur = t & 1, nxt = cur ^ 1;
      const uint8_t* sa_cur = smem + cur * W4_STAGE + a_row;
      static_for<MB>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        if constexpr (g == MB - 1) {
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#ifndef FDM_W4_NOBAR      // (timing ablation only)
          __builtin_amdgcn_s_barrier();
#endif
          __builtin_amdgcn_s_waitcnt(0xC07F);     // lgkmcnt(0) again, for hipcc's own bookkeeping (nothing is outstanding)
          touch_retired();
          __builtin_amdgcn_sched_barrier(0);
          touch(t + 3 + TOUCH_AHEAD);               // (B(t + 3)'s DMAs follow in the next step)
          read_frag(fa[(MB + PAR) & 1], smem + nxt * W4_STAGE + a_row, 0);
        } else {
          constexpr int cnt = (g >= 1 && g <= 4) ? 4 : 0;
          __builtin_amdgcn_s_waitcnt(0xC07F | (cnt << 8));    // lgkmcnt(cnt) only
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (!SPREAD_READS) {
            read_frag(fa[(g + 1 + PAR) & 1], sa_cur, g + 1);
            if constexpr (g < 4) {
              read_b_blk(bn, nxt, 2 * g);
              read_b_blk(bn, nxt, 2 * g + 1);
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        static_for<NS * 8>([&](auto jc) {
          constexpr int s = decltype(jc)::value / 8, nb = decltype(jc)::value % 8;
          M::template run<g * 8 + nb, decltype(first_c)::value && s == 0>(bc[s][nb], fa[(g + PAR) & 1][s]);   // swapped operands: MFMA rows = n, columns = m
          if constexpr (SPREAD_READS && g < MB - 1) {
            // int8 / bf16: an MFMA lasts 16 cycles, the group's 2 + 4 fragment reads in one burst in front of it are 28 cycles of issue
            // with the matrix core idle for half of them.  One read behind each of the group's first MFMAs instead, in the same order
            // (A block g + 1 first, then the B blocks of step t + 1), so the counted lgkmcnt waits above still mean what they say.
            constexpr int q = decltype(jc)::value;
            if constexpr (q < NS) fa[(g + 1 + PAR) & 1][q] = read_frag_slice(sa_cur, g + 1, q);
            else if constexpr (g < 4 && q < NS + 2 * NS) {
              constexpr int k = q - NS, blk = 2 * g + k / NS, sl = k % NS;
              bn[sl][blk] = read_frag_slice(smem + nxt * W4_STAGE + b_row, blk, sl);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          // The 16 DMAs that follow step t's barrier are spread thinly: a burst from all four waves queues on the CU's one
          // load path and the waves then meet skewed at the next barrier (21 % of the loop, measured).  4 in group 7 of step t,
          // 3 in each of groups 0..3 of step t + 1 -- all land well before that step's barrier.
          constexpr int q = decltype(jc)::value, every = NS * 2;      // one DMA slot after every 2nd (NS = 1) / 4th MFMA
          if constexpr (q % every == every - 1) {
            constexpr int slot = q / every;                            // 0..3
            if constexpr (g == MB - 1) dma_one(inner_c, std::integral_constant<int, slot>{}, t);
            if constexpr (g < 4 && slot < 3 && !decltype(first_c)::value) dma_one(inner_c, std::integral_constant<int, 4 + 3 * g + slot>{}, t - 1);
          }
        });
        __builtin_amdgcn_sched_barrier(0);
      });
    };
    const int nk2 = (nk + 1) & ~1;
    // step 0 has no DMAs left over from a previous step (the prologue issued A(1), B(2) whole)
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    step(std::false_type{}, std::true_type{}, P0{}, fbx, fby, 0);
    step(std::false_type{}, std::false_type{}, P1{}, fby, fbx, 1);
    int t = 2;
    for (; t + 4 < nk_full; t += 2) {            // issues stages up to t + 4: all complete
      step(std::true_type{}, std::false_type{}, P0{}, fbx, fby, t);
      step(std::true_type{}, std::false_type{}, P1{}, fby, fbx, t + 1);
    }
    for (; t < nk2; t += 2) {                     // K tail and (odd nk) one zero-filled padding step
      step(std::false_type{}, std::false_type{}, P0{}, fbx, fby, t);
      step(std::false_type{}, std::false_type{}, P1{}, fby, fbx, t + 1);
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15" ::: "memory");  // retire the padding DMAs and the dead fragment reads; MFMA -> v_accvgpr_read distance
  __builtin_amdgcn_s_barrier();                                  // every wave is done with the ring: reuse it below
  if constexpr (DT == DT_INT8 && MB == 8) {
    if (grp.mode == 1 && (wave >> (2 - grp.spread_shift)) != (int)(blockIdx.x & ((1 << grp.spread_shift) - 1))) return;
    if (is_part) {   // int32 partial sums of this K range -> workspace ([tile * splits + part][64 blocks][256 threads] x 16 B)
      int32_t* dst = grp.ws + ((int64_t)part_slot) * (64 * 256 * 4) + tid * 4;
      static_for<64>([&](auto ic) {
        constexpr int IDX = decltype(ic)::value;
        *reinterpret_cast<i32x4*>(dst + IDX * (256 * 4)) = acc_get<IDX>();
      });
      return;
    }
  }
#ifdef FDM_GEMM_STAMPS
  if (tid == 0) stamp[2] = __builtin_readcyclecounter();
#endif

  // ---- epilogue.  Wave-private LDS tile: 128 rows x 256 B; 16-byte chunk c of row r stored at chunk c ^ (r & 15).
  uint8_t* wt = smem + wave * 32768;
  const bool full = (m0 + BMW <= p.M) && (n0 + BN <= p.N);
  auto epilogue = [&](auto epi_c) {
    constexpr int EPI = decltype(epi_c)::value;
    float sam[8];
    int azpm[8];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
      const int mc = min(m0 + wm * WROWS + mb * 16 + fi, p.M - 1);
      sam[mb] = 1.0f;
      azpm[mb] = 0;
      if constexpr (DT != DT_BF16) sam[mb] = p.sa[mc];
      if constexpr (DT == DT_INT8) {
        if (p.azp != nullptr) azpm[mb] = p.azp[mc];
      }
    }
    // per-column operands of all 8 column blocks, requested before any of them is used (as in gemm_kernel's epilogue)
    f32x4 sbv[8];
    i32x2 biasv[8];
    i32x4 adjv[8];
    const bool has_bias = p.bias != nullptr;
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) {
      sbv[nb] = f32x4{1.f, 1.f, 1.f, 1.f};
      biasv[nb] = i32x2{0, 0};
      adjv[nb] = i32x4{0, 0, 0, 0};
      if constexpr (DT != DT_BF16) sbv[nb] = *reinterpret_cast<const f32x4*>(p.sb + min(n0 + wn * 128 + nb * 16 + fq * 4, p.N - 4));
    }
    if (has_bias) {
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) biasv[nb] = *reinterpret_cast<const i32x2*>(p.bias + min(n0 + wn * 128 + nb * 16 + fq * 4, p.N - 4));
    }
    if constexpr (DT == DT_INT8) {
      if (p.azp_adj != nullptr) {
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) adjv[nb] = *reinterpret_cast<const i32x4*>(p.azp_adj + min(n0 + wn * 128 + nb * 16 + fq * 4, p.N - 4));
      }
    }
    // int8 rounds to bf16 before the bias is added (as the reference's two ops do), fp8 / bf16 do not: with `has_bias` tested per
    // element hipcc emitted both forms and a select for each of the 256 values per lane (pass 1: 13.5 k cycles per tile against 6.4 k for
    // fp8), so the flag is a compile-time parameter of the pass and tested once per tile.
    auto pass1 = [&](auto hb_c, auto half_c) {
    constexpr bool HB = decltype(hb_c)::value;
    constexpr int HALF = decltype(half_c)::value;   // 0 / 1: rows [64 HALF, 64 HALF + 64) of the wave tile only; -1: all 128 rows
    static_for<8>([&](auto nbc) {
      constexpr int nb = decltype(nbc)::value;
      float bias4[4];
      unpack4(biasv[nb], bias4);
      const f32x4 sbq = sbv[nb];
      const i32x4 adjq = adjv[nb];
      static_for<2>([&](auto hc) {
        constexpr int h = decltype(hc)::value;
        if constexpr (HALF == -1 || HALF == h) {
        float y[16] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // [mm][c]
        static_for<4>([&](auto mmc) {
          constexpr int mm = decltype(mmc)::value, mb = h * 4 + mm;
          if constexpr (mb < MB) {
          const i32x4 av = acc_get<mb * 8 + nb>();
          const int aw[4] = {av[0], av[1], av[2], av[3]};   // (a bit_cast applied to `av[c]` directly read element 0 for every c)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            float yy;
            if constexpr (DT == DT_FP8) {
              yy = __builtin_bit_cast(float, aw[c]) * sam[mb] * sbq[c] + bias4[c];
            } else if constexpr (DT == DT_INT8) {
              const int a = aw[c] - azpm[mb] * adjq[c];
              yy = (float)a * (sam[mb] * sbq[c]);
              if constexpr (HB) yy = rbf(yy) + bias4[c];
            } else {
              yy = __builtin_bit_cast(float, aw[c]) + bias4[c];
            }
            y[4 * mm + c] = (EPI == FDM_EPI_GELU_TANH || EPI == FDM_EPI_GELU_ERF) ? rbf(yy) : yy;
          }
          }
        });
        gelu16<EPI>(y);
#pragma unroll
        for (int mm = 0; mm < (h * 4 + 4 <= MB ? 4 : MB - h * 4); ++mm) {
          i32x2 o;
          o[0] = (int)pack2bf(y[4 * mm], y[4 * mm + 1]);
          o[1] = (int)pack2bf(y[4 * mm + 2], y[4 * mm + 3]);
          const int r = (h * 4 + mm) * 16 + fi;          // r & 15 == fi
          *reinterpret_cast<i32x2*>(wt + r * 256 + (((nb * 2 + (fq >> 1)) ^ fi) << 4) + (fq & 1) * 8) = o;
        }
        }
      });
    });
    };
    auto run_pass1 = [&](auto half_c) {
      if constexpr (DT == DT_INT8) {
        if (has_bias) pass1(std::true_type{}, half_c);
        else pass1(std::false_type{}, half_c);
      } else {
        pass1(std::false_type{}, half_c);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private tile: no block barrier needed
      __builtin_amdgcn_sched_barrier(0);
    };
    // pass 2: row-contiguous 16-byte pieces.  Instruction it (0..31): lane -> row 4 * it + lane / 16, chunk lane % 16.
    const int rsub = lane >> 4, ch = lane & 15;
    const int ncol = n0 + wn * 128 + ch * 8;
    const bool n_ok = full || ncol < p.N;      // N % 8 == 0 is required by the host check for this path
    const int ncc = n_ok ? ncol : 0;
    const int64_t ccol = p.plane_cols > 0 ? (int64_t)(ncc / p.plane_cols) * p.plane_stride + (ncc % p.plane_cols) : (int64_t)ncc;
    auto lds_row = [&](int it) -> i32x4 {
      const int r = it * 4 + rsub;
      return *reinterpret_cast<const i32x4*>(wt + r * 256 + ((ch ^ (r & 15)) << 4));
    };
    if constexpr (EPI != FDM_EPI_GATE_RESIDUAL) {
      if (full) {
        // whole tile, no residual: the two 64-row halves go out one after the other -- the row stores of the first half (queued,
        // asynchronous) drain through the CU's store path while the VALU scales and packs the second half (epilogue of a K = 3072
        // tile 12.6 k -> 11.8 k cycles)
        bf16_t* cptr = p.C + (int64_t)(m0 + wm * WROWS + rsub) * p.ldc + ccol;
        const int64_t step4 = 4 * p.ldc;
        static_for<2>([&](auto hfc) {
          constexpr int hf = decltype(hfc)::value;
          run_pass1(std::integral_constant<int, hf>{});
#ifdef FDM_GEMM_STAMPS
          if (hf == 1 && tid == 0) stamp[4] = __builtin_readcyclecounter();
#endif
          constexpr int NH = hf == 0 ? 16 : NIT - 16;
          i32x4 v[NH];
#pragma unroll
          for (int it = 0; it < NH; ++it) v[it] = lds_row(hf * 16 + it);
#pragma unroll
          for (int it = 0; it < NH; ++it) *reinterpret_cast<i32x4*>(cptr + (hf * 16 + it) * step4) = v[it];
        });
        return;
      }
    }
    run_pass1(std::integral_constant<int, -1>{});
#ifdef FDM_GEMM_STAMPS
    if (tid == 0) stamp[4] = __builtin_readcyclecounter();
#endif
    auto store_row = [&](int it, const i32x4& v) {
      const int m = m0 + wm * WROWS + it * 4 + rsub;
      if (n_ok && (full || m < p.M)) *reinterpret_cast<i32x4*>(p.C + (int64_t)m * p.ldc + ccol) = v;
    };
    auto combine = [&](auto f32_c, const i32x4& yv, const float (&g8)[8], const i32x4& rv) -> i32x4 {
      constexpr bool F32 = decltype(f32_c)::value;
      i32x4 o;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const uint32_t yw = (uint32_t)yv[w], rw = (uint32_t)rv[w];
        float plo = g8[2 * w] * __builtin_bit_cast(float, yw << 16);
        float phi = g8[2 * w + 1] * __builtin_bit_cast(float, yw & 0xffff0000u);
        if constexpr (!F32) {
          plo = rbf(plo);
          phi = rbf(phi);
        }
        o[w] = (int)pack2bf(__builtin_bit_cast(float, rw << 16) + plo, __builtin_bit_cast(float, rw & 0xffff0000u) + phi);
      }
      return o;
    };
    if constexpr (EPI == FDM_EPI_GATE_RESIDUAL) {
      const int m_first = m0 + wm * WROWS, m_last = min(m0 + wm * WROWS + WROWS, p.M) - 1;
      const bool one_batch = (m_first / p.rows_per_batch) == (m_last / p.rows_per_batch);
      const bool n_ok_all = n0 + BN <= p.N;     // (this kernel only takes problems of whole N tiles: always true)
      auto res_row = [&](int it) -> i32x4 {
        const int mc = min(m0 + wm * WROWS + it * 4 + rsub, p.M - 1);
        return *reinterpret_cast<const i32x4*>(p.res + (int64_t)mc * p.ldc + ncc);
      };
      auto load_gate = [&](int64_t batch, float (&g8)[8]) {
        if (p.gate_f32) {
          const float* gp = reinterpret_cast<const float*>(p.gate) + batch * p.gate_stride + ncc;
          const f32x4 a = *reinterpret_cast<const f32x4*>(gp), c = *reinterpret_cast<const f32x4*>(gp + 4);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            g8[i] = a[i];
            g8[4 + i] = c[i];
          }
        } else {
          const i32x4 gv = *reinterpret_cast<const i32x4*>(p.gate + batch * p.gate_stride + ncc);
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            g8[2 * w] = __builtin_bit_cast(float, (uint32_t)gv[w] << 16);
            g8[2 * w + 1] = __builtin_bit_cast(float, (uint32_t)gv[w] & 0xffff0000u);
          }
        }
      };
      bf16_t* cptr = p.C + (int64_t)(m0 + wm * WROWS + rsub) * p.ldc + ccol;
      const int64_t step4 = 4 * p.ldc;
      auto pipelined = [&](auto f32_c, auto fast_c) {
        // groups of 8 row-instructions: loads(g) run one group ahead of combine(g - 1) + stores(g - 1); every load of a
        // group precedes the stores of the previous one in program order (vmcnt retires in order).
        // FAST (whole tile): unpredicated stores off one base pointer.  !FAST (the last, ragged row of M tiles; residual rows are
        // clamped): the same pipeline, a lane stores its first `it_max` row-instructions only -- one compare against a per-lane
        // count per store.  (The row-at-a-time edge path below is one exposed gate + residual round trip per row-instruction:
        // +12 ... 19 us on a launch whose last tiles are ragged, Qwen-Image's M = 6889: profiles/r04_experiments.md.)
        constexpr bool FAST = decltype(fast_c)::value;
        constexpr int GS = NIT / 4;     // row-instructions per group: 8 (7 for 224-row tiles)
        int it_max = NIT;
        if constexpr (!FAST) {
          const int rows_valid = p.M - (m0 + wm * WROWS) - rsub;    // rows 4 it + rsub < M - tile row 0
          it_max = rows_valid <= 0 ? 0 : (rows_valid + 3) >> 2;
        }
        float g8[8];
        load_gate(m_first / p.rows_per_batch, g8);
        i32x4 rr[2][8], oo[8];
#pragma unroll
        for (int it = 0; it < GS; ++it) rr[0][it] = res_row(it);
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {
          if (gi < 3) {
#pragma unroll
            for (int it = 0; it < GS; ++it) rr[(gi + 1) & 1][it] = res_row((gi + 1) * GS + it);
          }
#pragma unroll
          for (int it = 0; it < GS; ++it) oo[it] = combine(f32_c, lds_row(gi * GS + it), g8, rr[gi & 1][it]);
          asm volatile("" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int it = 0; it < GS; ++it) {
            if (FAST || gi * GS + it < it_max) *reinterpret_cast<i32x4*>(cptr + (gi * GS + it) * step4) = oo[it];
          }
        }
      };
      if (one_batch && full) {
        if (p.gate_f32) pipelined(std::true_type{}, std::true_type{});
        else pipelined(std::false_type{}, std::true_type{});
      } else if (one_batch && n_ok_all) {
        if (p.gate_f32) pipelined(std::true_type{}, std::false_type{});
        else pipelined(std::false_type{}, std::false_type{});
      } else {
        // edge tiles and waves whose rows straddle a batch boundary (rare): per-row gate vector, guarded stores, unpipelined
        // (the double-buffered form with guarded stores ran out of registers: hipcc parked three of them in a0..a2)
        for (int it = 0; it < NIT; ++it) {
          const int mc = min(m0 + wm * WROWS + it * 4 + rsub, p.M - 1);
          float g8[8];
          load_gate(mc / p.rows_per_batch, g8);
          const i32x4 rv = res_row(it);
          store_row(it, p.gate_f32 ? combine(std::true_type{}, lds_row(it), g8, rv) : combine(std::false_type{}, lds_row(it), g8, rv));
        }
      }
    } else {
#pragma unroll
      for (int it = 0; it < NIT; ++it) store_row(it, lds_row(it));
    }
  };
  if constexpr (RES) {
    if (p.epi == FDM_EPI_GATE_RESIDUAL) epilogue(std::integral_constant<int, FDM_EPI_GATE_RESIDUAL>{});
    else if (p.epi == FDM_EPI_GELU_TANH) epilogue(std::integral_constant<int, FDM_EPI_GELU_TANH>{});
    else if (p.epi == FDM_EPI_GELU_ERF) epilogue(std::integral_constant<int, FDM_EPI_GELU_ERF>{});
    else epilogue(std::integral_constant<int, FDM_EPI_NONE>{});
  } else {
    if (p.epi == FDM_EPI_GELU_TANH) epilogue(std::integral_constant<int, FDM_EPI_GELU_TANH>{});
    else if (p.epi == FDM_EPI_GELU_ERF) epilogue(std::integral_constant<int, FDM_EPI_GELU_ERF>{});
    else epilogue(std::integral_constant<int, FDM_EPI_NONE>{});
  }
#ifdef FDM_GEMM_STAMPS
  if (tid == 0) stamp[5] = __builtin_readcyclecounter();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (tid == 0) {
    stamp[3] = __builtin_readcyclecounter();
    stamp[7] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

// =====================================================================================================================
// gemm_p4_kernel: the PERSISTENT form of gemm_w4_kernel for fp8 launches of whole tiles and more than one round -- one block per CU
// walks its tiles; one continuous K-step stream across tile seams (no per-tile prologue); tile i's accumulators are read out,
// scaled, rounded and stored under the first K step of tile i + 1.  The body is hand-written gfx950 assembly, generated by
// gen_gemm_p4.py (read its header): every one of the 512 vector registers is assigned by hand, which is what the C++ attempt of
// round 3 could not get from hipcc.  Same MFMAs in the same order and the same epilogue arithmetic as gemm_w4_kernel: outputs are
// bit-identical to it.
struct P4Prob {
  const void* A;
  const void* B;
  void* C;
  const float* sa;
  const float* sb;
  const void* bias;
  const void* gate;
  const void* res;
};
struct P4Args {           // offsets are part of the assembly's contract (gen_gemm_p4.py ARG_*)
  P4Prob p[2];            //   0, 64
  uint32_t tiles_m0, tiles_m1, tiles_n, t0_tiles;              // 128
  uint32_t total_tiles, nk, lda, ldb;                          // 144
  uint32_t ldc_bytes, plane_cols, plane_stride_bytes, blocks_per_xcd;  // 160 (grid = 8 * blocks_per_xcd)
  uint32_t pad[4];
};
static_assert(sizeof(P4Args) == 192, "P4Args layout is read by offset in gemm_p4_body_*.inc");
__device__ __attribute__((aligned(256))) uint8_t g_zero_bias[65536] = {0};   // bias == nullptr: 32 K columns of bf16 zeros
#ifndef FDM_P4_BODY_NONE
#define FDM_P4_BODY_NONE "gemm_p4_body_none.inc"     // (bring-up builds point this at a generator variant)
#endif
template <int EPI>
__global__ void __launch_bounds__(W4_THREADS, 1) gemm_p4_kernel(const P4Args args) {
  __shared__ __attribute__((aligned(1024))) uint8_t smem[160 * 1024];   // 128 KiB ring + 4 x 8 KiB epilogue strips
  const auto ka = __builtin_amdgcn_kernarg_segment_ptr();
  const uint32_t lds0 = lds_addr_of(smem);
  const uint8_t* zero_bias = g_zero_bias;      // what a null bias pointer is replaced by inside the assembly
  if constexpr (EPI == FDM_EPI_NONE) {
    asm volatile(
#include FDM_P4_BODY_NONE
        :
        : "s"(ka), "s"(blockIdx.x), "v"(threadIdx.x), "s"(lds0), "s"(zero_bias)
        :
#include "gemm_p4_clobbers.inc"
    );
  } else if constexpr (EPI == FDM_EPI_GELU_TANH) {
    asm volatile(
#include "gemm_p4_body_gelu_tanh.inc"
        :
        : "s"(ka), "s"(blockIdx.x), "v"(threadIdx.x), "s"(lds0), "s"(zero_bias)
        :
#include "gemm_p4_clobbers.inc"
    );
  } else {
    static_assert(EPI == FDM_EPI_GELU_ERF, "gemm_p4_kernel: plain, tanh-GELU and erf-GELU epilogues");
    asm volatile(
#include "gemm_p4_body_gelu_erf.inc"
        :
        : "s"(ka), "s"(blockIdx.x), "v"(threadIdx.x), "s"(lds0), "s"(zero_bias)
        :
#include "gemm_p4_clobbers.inc"
    );
  }
}

// Which tile kernel runs a launch.  Both compute the same 256 x 256 tiles (int8: bit-identical; fp8 / bf16: identical except for
// the fp32 summation order inside the MFMA shapes, <= 1-2 bf16 ulp on ~1e-6 of the elements).  Measured inside whole denoise steps
// on MI355X (bench.py A/B on one box, profiles/r03_experiments.md): gemm_w4_kernel everywhere it applies is 102.9-103.3 ms per
// FLUX fp8 step against 105.7-106.2 with gemm_kernel everywhere (int8: 107.7 vs 111.2 with the earlier mixed rule).  Before the
// epilogues of both kernels were rewritten (operands requested up front, whole-tile stores off one base pointer) the many-round
// K = 3072 launches were 2 % slower in the 4-wave kernel, whose one wave per SIMD has nothing to cover its epilogue with.
// FDM_GEMM_KERNEL = w4 | w8 | auto (default) overrides for A/B runs; read once.
#ifdef FDM_GEMM_AB
int g_gemm_use_w4 = 1;   // micro-benchmark builds only: both kernels in one process
#endif
// Dispatch policy knobs (A/B runs, parity tests): process-wide atomics set through fdm_set_gemm_dispatch() -- the library reads NO environment
// variable; fastdm_amd/_native.py translates FDM_GEMM_KERNEL / FDM_GEMM_P4 / FDM_GEMM_HALF_TILES on the host side when it loads
// the library.  kernel: 2 auto (default) | 1 the 4-wave kernel everywhere it runs | 0 the 8-wave kernel everywhere.
static std::atomic<int> g_dispatch_kernel{2}, g_dispatch_p4{1}, g_dispatch_half{-1}, g_dispatch_rows{-1};
static int gemm_kernel_mode() { return g_dispatch_kernel.load(std::memory_order_relaxed); }
template <int DT>
static bool use_w4(const GemmGroup& grp) {
#ifdef FDM_GEMM_AB
  return g_gemm_use_w4 != 0;
#else
  const int mode = gemm_kernel_mode();
  if (mode != 2) return mode == 1;
  // auto: the 4-wave kernel for every problem of whole N tiles and an even number of 128-byte K steps; a ragged M (edge tiles:
  // clamped source rows, masked stores) in launches of at least one full round -- the sub-round ragged launches (SDXL's 154-row
  // cross-attention projections) lose 1.4 % of the SDXL step there.  Measured per denoise step, one box each
  // (profiles/r03_experiments.md): Qwen-Image 1328^2, one forward of M = 7017 per prompt: 205.1 (whole tiles only) -> 199.3
  // (>= 2 rounds) -> 196.9 (>= 1 round, after the merge launch of the split-K tail got its loads batched); batch-2 forward
  // 188.4 -> 178.6; SDXL unchanged.
  // (ragged N / K in the 4-wave kernel: SD3.5-large N = K = 2432 56.9 -> 56.6 ms per step, SDXL 45.4 -> 44.5: within noise, not taken)
  const bool many_rounds = grp.total_tiles >= device_cu_count();
  // at most half a round of tiles: the 8-wave kernel's 256 x 128 half tiles (two blocks per CU) put twice as many CUs to work -- round 5,
  // tools/micro/gemm_bench: SDXL level 2 M = 2048, N = 1280, K = 5120 (40 tiles) 55.3 -> 42.4 us, M = 512 52 -> 42, N = K = 2560 35 -> 27;
  // above that (192 tiles: 44.1 vs 46-54) this kernel stays ahead
  if (grp.total_tiles * 2 <= device_cu_count()) return false;
  for (int i = 0; i < grp.n; ++i) {
    const GemmProblem& q = grp.p[i];
    if (q.Kb < 2048) return false;
    if (!many_rounds && (q.M % BM)) return false;
    // ragged N (a last column tile with masked stores) and an odd number of 128-byte K steps (one zero-filled padding step): SD3.5-large's
    // N = K = 2432.  Neutral in round 3 (56.9 -> 56.6 ms); with the cooperative weight touch of round 6 it pays: 55.05 / 55.12 -> 53.74 / 53.77 ms
    // per step on one box (profiles/r06af_*), so whole launches of >= 1 round take the 4-wave kernel whatever N and K are
    if (!many_rounds && ((q.N % BN) || (q.Kb % 256))) return false;
    if ((q.N % 8) || (q.Kb % 16)) return false;
    if ((q.N % BN) && (q.N % BN) < BN / 2) return false;     // a last column tile less than half full (Wan's N = 64 output projection) stays with the 8-wave kernel's half tiles
  }
  return true;
#endif
}
#ifdef FDM_GEMM_STAMPS
unsigned long long* g_gemm_stamps = nullptr;  // >= 4 * grid words, set by the micro-benchmark
#endif

// Does the persistent assembly kernel run this launch?  fp8, every problem made of whole 256 x 256 tiles and an even number (>= 6)
// of 128-byte K steps, more tiles than CUs (a one-round launch has no next tile to hide anything under), the epilogues the
// assembly implements, and -- for two problems in one launch -- the same shape parameters except M and the pointers (the FLUX
// image + text pairs are like that).  FDM_GEMM_KERNEL = p4 | w4 | w8 | auto; FDM_GEMM_P4 = 0 switches the persistent kernel off.
#ifdef FDM_GEMM_AB
int g_gemm_use_p4 = 0;   // micro-benchmark builds: 0 / 1
#endif
static bool use_p4(const GemmGroup& grp, P4Args& pa) {
#ifdef FDM_GEMM_AB
  if (!g_gemm_use_p4) return false;
#else
  // p4: 0 never; 1 (default) the epilogues it measured faster with (plain, tanh-GELU); 2 every epilogue it implements
  const int on = g_dispatch_p4.load(std::memory_order_relaxed);
  if (!on) return false;
  if (on == 1 && grp.p[0].epi == FDM_EPI_GELU_ERF) return false;   // erf-GELU drain is VALU-bound either way: 373-378 vs 366-372 us (profiles/r04_experiments.md)
#endif
#ifdef FDM_GEMM_AB
  if (grp.total_tiles < device_cu_count() || grp.splits > 1 || grp.mode != 0) return false;   // (bring-up: also the one-tile-per-block case)
#else
  if (grp.total_tiles <= device_cu_count() || grp.splits > 1 || grp.mode != 0) return false;
#endif
  const GemmProblem& a = grp.p[0];
  for (int i = 0; i < grp.n; ++i) {
    const GemmProblem& q = grp.p[i];
    if ((q.epi != FDM_EPI_NONE && q.epi != FDM_EPI_GELU_TANH && q.epi != FDM_EPI_GELU_ERF) || q.epi != a.epi) return false;
    if ((q.M % BM) || (q.N % BN) || (q.Kb % 256) || q.Kb < 768) return false;
    if (q.N != a.N || q.Kb != a.Kb || q.lda != a.lda || q.ldb != a.ldb || q.ldc != a.ldc || q.plane_cols != a.plane_cols ||
        q.plane_stride != a.plane_stride)
      return false;
    if ((int64_t)q.M * q.lda >= (1ll << 32) || (int64_t)q.N * q.ldb >= (1ll << 32) || q.ldc * 2 >= (1ll << 31) || q.N > 32768) return false;
    if (q.plane_cols > 0 && (q.plane_cols % BN)) return false;       // a tile never straddles two planes
    if (!q.sa || !q.sb) return false;
  }
  memset(&pa, 0, sizeof(pa));
  for (int i = 0; i < 2; ++i) {
    const GemmProblem& q = grp.p[i < grp.n ? i : 0];
    pa.p[i] = P4Prob{q.A, q.B, q.C, q.sa, q.sb, q.bias, q.gate, q.res};      // (a null bias becomes g_zero_bias inside the kernel)
  }
  pa.tiles_m0 = a.tiles_m;
  pa.tiles_m1 = grp.n > 1 ? grp.p[1].tiles_m : 0;
  pa.tiles_n = a.tiles_n;
  pa.t0_tiles = a.tiles_m * a.tiles_n;
  pa.total_tiles = grp.total_tiles;
  pa.nk = (uint32_t)(a.Kb / 128);
  pa.lda = (uint32_t)a.lda;
  pa.ldb = (uint32_t)a.ldb;
  pa.ldc_bytes = (uint32_t)(a.ldc * 2);
  pa.plane_cols = (uint32_t)a.plane_cols;
  if ((uint64_t)a.plane_stride * 2 >= (1ull << 32)) return false;
  pa.plane_stride_bytes = (uint32_t)(a.plane_stride * 2);
  pa.blocks_per_xcd = (uint32_t)(device_cu_count() / 8);
  if (pa.blocks_per_xcd == 0 || grp.total_tiles < (int)(8 * pa.blocks_per_xcd)) return false;
  return true;
}

// Do 256 x 128 half tiles (two 4-wave blocks per CU) beat 256 x 256 tiles (one 8-wave block per CU) for this launch of the 8-wave kernel?
// Unit = the time of one full tile on a CU.  Full tiles: whole rounds + one more for any remainder.  Half tiles move 1.5 x the LDS bytes per
// flop, so two of them side by side on a CU take h2 > 1 units and one alone h1 > 0.5 -- both grow with K as the fixed cost of a tile
// amortises (fitted to tools/micro/gemm_bench on MI355X, profiles/r05_experiments.md: h2 = 1.02 / 1.17 / 1.32 and h1 = 0.61 / 0.65 / 0.69 at
// K = 640 / 2432 / 9728 bytes).  They pay for sub-round launches (SDXL's 40 - 240 tile layers: -16 ... -40 %) and for a short K with a
// nearly empty last round (SD3.5 N = K = 2432: -9 %); not for long K or many rounds.  FDM_GEMM_HALF_TILES = 0 | 1 forces the answer (A/B).
static bool half_tiles_pay(const GemmGroup& grp) {
  const int forced = g_dispatch_half.load(std::memory_order_relaxed);     // -1 the cost model below (default), 0 / 1 forced
  if (grp.splits > 1 || grp.mode != 0) return false;
  if (forced >= 0) return forced != 0;
  const int cu = device_cu_count();
  int half = 0;
  int64_t kb = grp.p[0].Kb;
  for (int i = 0; i < grp.n; ++i) {
    half += grp.p[i].tiles_m * (int)((grp.p[i].N + BN / 2 - 1) / (BN / 2));
    kb = kb > grp.p[i].Kb ? kb : grp.p[i].Kb;
  }
  const float lk = log2f((float)(kb > 512 ? kb : 512) / 512.0f);
  const float h2 = 1.0f + 0.075f * lk, h1 = 0.6f + 0.02f * lk;
  const float t_full = (float)((grp.total_tiles + cu - 1) / cu);
  const int rem = half % (2 * cu);
  const float t_half = (float)(half / (2 * cu)) * h2 + (rem == 0 ? 0.f : (rem <= cu ? h1 : h2));
  return t_half < t_full - 0.05f;
}

// 224-row tiles for a launch of the 4-wave kernel (gemm_w4_kernel<.., 7>).  Unit = one 16-row block column of a tile: a launch costs rounds x rows / 16.
// FLUX's N = 3072 projections are 34 x 12 = 408 tiles of 256 rows = 2 rounds x 8, the second 59 % full; 39 x 12 = 468 tiles of 224 rows are 2 rounds x 7.
// Round 6 measurements (profiles/r06_experiments.md "224-row tiles"): back to back on random operands that is worth 9-14 % (370 -> 319 us, 259 -> 236);
// INSIDE the power-limited FLUX step it is worth nothing either way (312 -> 304, 263 -> 262, 93 -> 99 us; step 100.64 vs 100.66 ms on one box: the CUs a
// partial round leaves idle hand their share of the package power to the busy ones as clock); where the step is NOT at the power limit it pays
// (SD3.5-large with its N = K = 2432 launches on this kernel: 53.75 -> 52.68 ms).  Policy: 224 rows where the rounds arithmetic promises >= 10 %, never
// in place of the int8 split-K tail (Qwen-Image: the split gets closer to the ideal than 7/8).  fdm_set_gemm_tile_rows(256 | 224) forces the answer.
constexpr int BM7 = 224;
static GemmGroup with_tile_rows_224(const GemmGroup& grp) {
  GemmGroup g = grp;
  g.total_tiles = 0;
  for (int i = 0; i < g.n; ++i) {
    g.p[i].tiles_m = (g.p[i].M + BM7 - 1) / BM7;
    g.total_tiles += g.p[i].tiles_m * g.p[i].tiles_n;
  }
  g.n_blocks = g.total_tiles;
  return g;
}
static bool tile_rows_224(const GemmGroup& grp, bool forced_only = false) {
  if (grp.splits > 1 || grp.mode != 0) return false;
  const int forced = g_dispatch_rows.load(std::memory_order_relaxed);
  if (forced > 0 || forced_only) return forced == BM7;
  const int cu = device_cu_count();
  int t7 = 0;
  for (int i = 0; i < grp.n; ++i) t7 += ((grp.p[i].M + BM7 - 1) / BM7) * grp.p[i].tiles_n;
  const int c8 = 8 * ((grp.total_tiles + cu - 1) / cu), c7 = 7 * ((t7 + cu - 1) / cu);
  return 10 * c7 <= 9 * c8;
}

constexpr int64_t SPLIT_PART_BYTES = (int64_t)BM * BN * 4;   // one K part of one tile: 256 x 256 int32
static thread_local int g_last_gemm_kernel = -1;              // fdm_gemm_last_kernel()

template <int DT>
static int launch_gemm(const GemmGroup& grp, hipStream_t s, void* ws = nullptr, int64_t ws_bytes = 0) {
#ifdef FDM_GEMM_STAMPS
  const_cast<GemmGroup&>(grp).stamps = g_gemm_stamps;
#endif
  const int grid = grp.total_tiles;
#ifdef FDM_DEBUG_BUILD
  static int debug = -1;
  if (debug < 0) {
    const char* e = getenv("FDM_GEMM_DEBUG");
    debug = e ? atoi(e) : 0;
  }
  const_cast<GemmGroup&>(grp).debug = debug;
#endif
  const bool res = grp.p[0].epi == FDM_EPI_GATE_RESIDUAL || (grp.n > 1 && grp.p[1].epi == FDM_EPI_GATE_RESIDUAL);
  if constexpr (DT == DT_INT8) {
    // Split-K tail (int8 only: int32 partial sums add exactly, so the result does not depend on the split).  When the last
    // round of one-tile-per-CU work fills at most half of the chip, its `rem` tiles are cut into S K ranges (rem * S blocks
    // run side by side) and merged by a second launch.  Worth it only if a part still has >= 8 K steps and the int32
    // partials (256 KB per part, written and read once) cost less than the fraction of a round they save.
    const int cu = device_cu_count();
    const int rem = grp.total_tiles % cu;
    int64_t nk_min = (grp.p[0].Kb + BKB - 1) / BKB;
    if (grp.n > 1) nk_min = nk_min < (grp.p[1].Kb + BKB - 1) / BKB ? nk_min : (grp.p[1].Kb + BKB - 1) / BKB;
    int S = 1;
    if (ws != nullptr && rem > 0 && rem * 2 <= cu && grp.total_tiles > cu) {
      S = cu / rem;
      if (S > 8) S = 8;
      if (S > nk_min / 8) S = (int)(nk_min / 8);
      while (S > 1 && (int64_t)rem * S * SPLIT_PART_BYTES > ws_bytes) --S;
      if (rem * S < 8) S = 1;
      // Measured on MI355X (Qwen-Image's K = 12288 down-projection, 336 tiles: 346 -> 307 us): a part stores 256 KB through
      // one CU (~12 us) and the merge launch costs ~35-40 us (launch gap, S x 256 KB re-read, epilogue), so the split pays
      // only for long K loops: >= 160 steps of 64 bytes (a tile of >= ~120 us).  K = 3072 tiles (46 us) are never split.
      if (nk_min < 160) S = 1;
    }
    if (S > 1 && use_w4<DT>(grp) && tile_rows_224(grp, true)) S = 1;     // (A/B: the FORCED 224-row form has no split-K tail; the policy prefers the split)
    if (S > 1) {
      GemmGroup g2 = grp;
      g2.split_tiles = rem;
      g2.splits = S;
      g2.n_blocks = grp.total_tiles - rem + rem * S;
      g2.ws = reinterpret_cast<int32_t*>(ws);
      if (use_w4<DT>(grp)) {
        if (res) gemm_w4_kernel<DT, true><<<g2.n_blocks, W4_THREADS, 0, s>>>(g2);
        else gemm_w4_kernel<DT, false><<<g2.n_blocks, W4_THREADS, 0, s>>>(g2);
      } else {
        if (res) gemm_kernel<DT, true><<<g2.n_blocks, GEMM_THREADS, 0, s>>>(g2);
        else gemm_kernel<DT, false><<<g2.n_blocks, GEMM_THREADS, 0, s>>>(g2);
      }
      FDM_CHECK_LAUNCH();
      g_last_gemm_kernel = use_w4<DT>(grp) ? FDM_GEMM_KERNEL_W4 : FDM_GEMM_KERNEL_W8;
      g2.mode = 1;
      g2.spread_shift = rem * 4 <= cu ? 2 : (rem * 2 <= cu ? 1 : 0);   // as many merge blocks per tile as fit ONE round
      if (use_w4<DT>(grp)) {
        if (res) gemm_w4_kernel<DT, true><<<rem << g2.spread_shift, W4_THREADS, 0, s>>>(g2);
        else gemm_w4_kernel<DT, false><<<rem << g2.spread_shift, W4_THREADS, 0, s>>>(g2);
      } else {
        if (res) gemm_kernel<DT, true><<<rem << g2.spread_shift, GEMM_THREADS, 0, s>>>(g2);
        else gemm_kernel<DT, false><<<rem << g2.spread_shift, GEMM_THREADS, 0, s>>>(g2);
      }
      FDM_CHECK_LAUNCH();
      return FDM_OK;
    }
  }
  if constexpr (DT == DT_FP8) {
    P4Args pa;
    if (use_p4(grp, pa)) {
      const unsigned pgrid = 8 * pa.blocks_per_xcd;
      if (grp.p[0].epi == FDM_EPI_GELU_TANH) gemm_p4_kernel<FDM_EPI_GELU_TANH><<<pgrid, W4_THREADS, 0, s>>>(pa);
      else if (grp.p[0].epi == FDM_EPI_GELU_ERF) gemm_p4_kernel<FDM_EPI_GELU_ERF><<<pgrid, W4_THREADS, 0, s>>>(pa);
      else gemm_p4_kernel<FDM_EPI_NONE><<<pgrid, W4_THREADS, 0, s>>>(pa);
      g_last_gemm_kernel = FDM_GEMM_KERNEL_P4;
      FDM_CHECK_LAUNCH();
      return FDM_OK;
    }
  }
  if (use_w4<DT>(grp)) {
    if (tile_rows_224(grp)) {
      const GemmGroup g7 = with_tile_rows_224(grp);
      if (res) gemm_w4_kernel<DT, true, 7><<<g7.total_tiles, W4_THREADS, 0, s>>>(g7);
      else gemm_w4_kernel<DT, false, 7><<<g7.total_tiles, W4_THREADS, 0, s>>>(g7);
      g_last_gemm_kernel = FDM_GEMM_KERNEL_W4_224;
    } else {
      if (res) gemm_w4_kernel<DT, true><<<grid, W4_THREADS, 0, s>>>(grp);
      else gemm_w4_kernel<DT, false><<<grid, W4_THREADS, 0, s>>>(grp);
      g_last_gemm_kernel = FDM_GEMM_KERNEL_W4;
    }
  } else if (half_tiles_pay(grp)) {
    // 256 x 128 half tiles, two 4-wave blocks per CU (gemm_kernel<.., 2>): same wave code, twice the work items
    GemmGroup g2 = grp;
    g2.total_tiles = 0;
    for (int i = 0; i < g2.n; ++i) {
      g2.p[i].tiles_n = (int)((g2.p[i].N + BN / 2 - 1) / (BN / 2));
      g2.total_tiles += g2.p[i].tiles_m * g2.p[i].tiles_n;
    }
    g2.n_blocks = g2.total_tiles;
    if (res) gemm_kernel<DT, true, 2><<<g2.total_tiles, GEMM_THREADS / 2, 0, s>>>(g2);
    else gemm_kernel<DT, false, 2><<<g2.total_tiles, GEMM_THREADS / 2, 0, s>>>(g2);
    g_last_gemm_kernel = FDM_GEMM_KERNEL_W8_HALF;
  } else {
    if (res) gemm_kernel<DT, true><<<grid, GEMM_THREADS, 0, s>>>(grp);
    else gemm_kernel<DT, false><<<grid, GEMM_THREADS, 0, s>>>(grp);
    g_last_gemm_kernel = FDM_GEMM_KERNEL_W8;
  }
  FDM_CHECK_LAUNCH();
  return FDM_OK;
}

static int fill_problem(GemmProblem& p, void* out, const void* a, const void* b, int64_t M, int64_t N, int64_t K, int64_t lda,
                        int64_t ldb, int64_t ldc, int elem_bytes, int epi, const void* gate, int64_t gate_stride, const void* res,
                        int64_t rpb, int64_t plane_cols = 0) {
  if (!out || !a || !b || M <= 0 || N <= 0 || K <= 0) return FDM_ERR_INVALID;
  const int64_t min_ldc = plane_cols > 0 ? plane_cols : N;
  if (M > 0x7fffffff || N > 0x7fffffff) return FDM_ERR_UNSUPPORTED;
  if (epi < FDM_EPI_NONE || epi > FDM_EPI_GATE_RESIDUAL_F32) return FDM_ERR_UNSUPPORTED;
  const int gate_f32 = epi == FDM_EPI_GATE_RESIDUAL_F32;
  if (gate_f32) epi = FDM_EPI_GATE_RESIDUAL;
  if ((K * elem_bytes) % 16 || (lda * elem_bytes) % 16 || (ldb * elem_bytes) % 16 || lda < K || ldb < K || ldc < min_ldc)
    return FDM_ERR_INVALID;
  if ((reinterpret_cast<uintptr_t>(a) & 15) || (reinterpret_cast<uintptr_t>(b) & 15) || (reinterpret_cast<uintptr_t>(out) & 15) || (ldc & 7) || (N & 7))
    return FDM_ERR_INVALID;
  if (epi == FDM_EPI_GATE_RESIDUAL && (!gate || !res || rpb <= 0 || (reinterpret_cast<uintptr_t>(gate) & 15) ||
                                       (reinterpret_cast<uintptr_t>(res) & 15) || (gate_stride & (gate_f32 ? 3 : 7))))
    return FDM_ERR_INVALID;
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
  p.epi = epi;
  p.gate_f32 = gate_f32;
  p.sa = p.sb = nullptr;
  p.bias = nullptr;
  p.azp = p.azp_adj = nullptr;
  p.plane_cols = 0;
  p.plane_stride = 0;
  return FDM_OK;
}

// Plane-split output of a fused projection (q|k|v): validated separately because ldc is then the PLANE's row stride.
static int set_planes(GemmProblem& p, int64_t plane_cols, int64_t plane_stride) {
  if (plane_cols == 0) return FDM_OK;
  if (plane_cols < 8 || (plane_cols & 7) || plane_cols > p.N || p.ldc < plane_cols || plane_stride <= 0 || (plane_stride & 7) ||
      p.epi >= FDM_EPI_GATE_RESIDUAL)
    return FDM_ERR_INVALID;
  p.plane_cols = (int)plane_cols;
  p.plane_stride = plane_stride;
  return FDM_OK;
}

static void finish_group(GemmGroup& g, int n) {
  g.n = n;
  g.total_tiles = 0;
  for (int i = 0; i < n; ++i) g.total_tiles += g.p[i].tiles_m * g.p[i].tiles_n;
  g.split_tiles = 0;
  g.splits = 1;
  g.mode = 0;
  g.n_blocks = g.total_tiles;
  g.spread_shift = 0;
  g.ws = nullptr;
  if (n == 1) {
    g.p[1] = g.p[0];
    g.p[1].tiles_m = g.p[1].tiles_n = 0;
  }
}

}  // namespace fdm

using namespace fdm;

// ------------------------------------------------------------------------------------------ skinny-M bf16 linear
// The AdaLN modulation / timestep-embedding projections have M = batch (1..8) rows: 2*M*N*K flops against N*K*2 bytes
// of weights, i.e. HBM-bound GEMVs.  The weights are [K, N] with stride(0) == 1, so one output column is K contiguous
// bf16: a wave streams 4 columns at a time with 16-byte loads, dots them against the activation rows held in LDS and
// butterfly-reduces.  fp32 accumulate, out = bf16(acc + bias) like the tile kernel's EPI_NONE epilogue.  post_add (optional, [N]): a
// second bf16 addend applied AFTER that rounding where it is non-zero, out = bf16(out + post_add[n]) -- the AdaLN table writes
// bf16(1 + scale) for its scale columns this way (the reference's `1 + scale` tensor), so that fdm_ln_modulate need not.
template <int MR>
__global__ void __launch_bounds__(256) gemv_bf16_kernel(bf16_t* __restrict__ out, const bf16_t* __restrict__ a,
                                                        const bf16_t* __restrict__ w, const bf16_t* __restrict__ bias,
                                                        const bf16_t* __restrict__ post_add,
                                                        int M, int64_t N, int K, int64_t lda, int64_t ldb, int64_t ldc,
                                                        int passes) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gemv_smem[];
  bf16_t* sa = reinterpret_cast<bf16_t*>(gemv_smem);  // [MR][K], rows >= M zero
  constexpr int C = 4;
  const int kv = K >> 3;
  for (int i = threadIdx.x; i < MR * kv; i += 256) {
    const int m = i / kv, c = i - m * kv;
    i32x4 v = {0, 0, 0, 0};
    if (m < M) v = *reinterpret_cast<const i32x4*>(a + (int64_t)m * lda + c * 8);
    *reinterpret_cast<i32x4*>(sa + m * K + c * 8) = v;
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int ps = 0; ps < passes; ++ps) {
    const int64_t n0 = (((int64_t)blockIdx.x * passes + ps) * 4 + wave) * C;
    if (n0 >= N) return;
    float acc[MR][C];
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int c = 0; c < C; ++c) acc[m][c] = 0.f;
    const bf16_t* wp[C];
#pragma unroll
    for (int c = 0; c < C; ++c) wp[c] = w + (n0 + c < N ? n0 + c : N - 1) * ldb;
#pragma unroll 2
    for (int k = lane * 8; k < K; k += 512) {
      Vec8 wv[C];
#pragma unroll
      for (int c = 0; c < C; ++c) wv[c] = load8(wp[c] + k);
#pragma unroll
      for (int m = 0; m < MR; ++m) {
        const Vec8 av = load8(sa + m * K + k);
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[m][c] = __builtin_fmaf(av.v[j], wv[c].v[j], acc[m][c]);
      }
    }
    float mine = 0.f;
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
      for (int c = 0; c < C; ++c) {
        float v = acc[m][c];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == m * C + c) mine = v;
      }
    const int m = lane / C, c = lane % C;
    if (m < M && n0 + c < N) {
      const float bv = bias ? bf2f(bias[n0 + c]) : 0.f;
      bf16_t y = f2bf(mine + bv);
      if (post_add != nullptr) {
        const float pa = bf2f(post_add[n0 + c]);
        if (pa != 0.f) y = f2bf(bf2f(y) + pa);
      }
      out[(int64_t)m * ldc + n0 + c] = y;
    }
  }
}

static bool gemv_applicable(int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int epilogue, const void* a, const void* b) {
  return M >= 1 && M <= 8 && epilogue == FDM_EPI_NONE && N >= 1 && K >= 8 && (K & 7) == 0 && (lda & 7) == 0 && (ldb & 7) == 0 &&
         K <= 16384 && (M <= 1 ? 1 : M <= 2 ? 2 : M <= 4 ? 4 : 8) * K * 2 <= 65536 &&
         !(reinterpret_cast<uintptr_t>(a) & 15) && !(reinterpret_cast<uintptr_t>(b) & 15);
}

static int launch_gemv(void* out, const void* a, const void* b, const void* bias, const void* post_add, int64_t M, int64_t N, int64_t K,
                       int64_t lda, int64_t ldb, int64_t ldc, hipStream_t s) {
  const int MR = M <= 1 ? 1 : M <= 2 ? 2 : M <= 4 ? 4 : 8;
  const int64_t groups = (N + 15) / 16;  // 4 waves x 4 columns per pass
  int passes = (int)(groups / 2048);     // keep >= 2048 blocks in flight before a block loops
  passes = passes < 1 ? 1 : passes > 4 ? 4 : passes;
  const unsigned grid = (unsigned)((groups + passes - 1) / passes);
  const size_t smem = (size_t)MR * K * 2;
#define FDM_GEMV(R)                                                                                                     \
  gemv_bf16_kernel<R><<<grid, 256, smem, s>>>((bf16_t*)out, (const bf16_t*)a, (const bf16_t*)b, (const bf16_t*)bias,            \
                                              (const bf16_t*)post_add, (int)M, N, (int)K, lda, ldb, ldc, passes)
  if (MR == 1) FDM_GEMV(1);
  else if (MR == 2) FDM_GEMV(2);
  else if (MR == 4) FDM_GEMV(4);
  else FDM_GEMV(8);
#undef FDM_GEMV
  g_last_gemm_kernel = FDM_GEMM_KERNEL_GEMV;
  FDM_CHECK_LAUNCH();
  return FDM_OK;
}

extern "C" {

int fdm_gemm_last_kernel(void) { return g_last_gemm_kernel; }

int fdm_set_gemm_tile_rows(int rows) {
  if (rows != -1 && rows != 256 && rows != 224) return FDM_ERR_INVALID;     // -1 = the policy (tile_rows_224)
  fdm::g_dispatch_rows.store(rows, std::memory_order_relaxed);
  return FDM_OK;
}

int fdm_set_gemm_dispatch(int kernel, int p4, int half_tiles) {
  if (kernel < -1 || kernel > 2 || p4 < -1 || p4 > 2 || half_tiles < -1 || half_tiles > 1) return FDM_ERR_INVALID;
  // kernel: -1 auto | 0 w8 | 1 w4 | 2 p4 (= auto with every epilogue the persistent kernel implements); w8 / w4 switch the persistent kernel off
  int km = 2, pm = p4 < 0 ? 1 : p4;
  if (kernel == 0 || kernel == 1) {
    km = kernel;
    pm = 0;
  } else if (kernel == 2) {
    pm = 2;
  }
  fdm::g_dispatch_kernel.store(km, std::memory_order_relaxed);
  fdm::g_dispatch_p4.store(pm, std::memory_order_relaxed);
  fdm::g_dispatch_half.store(half_tiles, std::memory_order_relaxed);
  return FDM_OK;
}

int fdm_gemm_w8a8_fp8(void* out, const void* a, const void* b, const float* scale_a, const float* scale_b, const void* bias,
                      int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, int epilogue,
                      const void* epi_gate, int64_t epi_gate_stride, const void* epi_res, int64_t epi_rows_per_batch,
                      fdm_stream_t stream) {
  (void)hipGetLastError();  // drop any stale error of an earlier, unrelated runtime call: FDM_CHECK_LAUNCH must see only ours
  if (M == 0 && N > 0 && K > 0) return FDM_OK;
  GemmGroup g;
  int rc = fill_problem(g.p[0], out, a, b, M, N, K, lda, ldb, ldc, 1, epilogue, epi_gate, epi_gate_stride, epi_res, epi_rows_per_batch);
  if (rc != FDM_OK) return rc;
  if (!scale_a || !scale_b) return FDM_ERR_INVALID;
  g.p[0].sa = scale_a;
  g.p[0].sb = scale_b;
  g.p[0].bias = (const bf16_t*)bias;
  finish_group(g, 1);
  return launch_gemm<DT_FP8>(g, to_stream(stream));
}

int fdm_gemm_w8a8_fp8_x2(const fdm_gemm_fp8_desc* d0, const fdm_gemm_fp8_desc* d1, fdm_stream_t stream) {
  (void)hipGetLastError();  // drop any stale error of an earlier, unrelated runtime call: FDM_CHECK_LAUNCH must see only ours
  if (!d0 || !d1) return FDM_ERR_INVALID;
  // The smaller problem goes FIRST in the tile list (round 6).  The tile list is cut into XCD-contiguous ranges, so one problem's few tiles (a
  // FLUX text stream: 2 of 34 M tiles) all sit on one XCD; at the END of the list they ran last there, long after the launch's cooperative weight
  // touch had asked for their B lines and 200 MB of output had pushed those out of the Infinity Cache again (in-step: 280 us for the image +
  // text FF1 pair against 207 + 6 % for the image problem alone).  At the front they run while the touch streams their weights in.
  const bool swap = d1->M > 0 && d0->M > d1->M;
  const fdm_gemm_fp8_desc* ds[2] = {swap ? d1 : d0, swap ? d0 : d1};
  GemmGroup g;
  int n = 0;
  for (int i = 0; i < 2; ++i) {
    const fdm_gemm_fp8_desc* d = ds[i];
    if (d->M == 0 && d->N > 0 && d->K > 0) continue;
    int rc = fill_problem(g.p[n], d->out, d->a, d->b, d->M, d->N, d->K, d->lda, d->ldb, d->ldc, 1, (int)d->epilogue, d->epi_gate,
                          d->epi_gate_stride, d->epi_res, d->epi_rows_per_batch, d->out_plane_cols);
    if (rc == FDM_OK) rc = set_planes(g.p[n], d->out_plane_cols, d->out_plane_stride);
    if (rc != FDM_OK) return rc;
    if (!d->scale_a || !d->scale_b) return FDM_ERR_INVALID;
    g.p[n].sa = d->scale_a;
    g.p[n].sb = d->scale_b;
    g.p[n].bias = (const bf16_t*)d->bias;
    ++n;
  }
  if (n == 0) return FDM_OK;
  finish_group(g, n);
  return launch_gemm<DT_FP8>(g, to_stream(stream));
}

int fdm_gemm_w8a8_int8(void* out, const void* a, const void* b, const float* scale_a, const float* scale_b,
                       const int32_t* azp_adj, const int32_t* azp, const void* bias, int64_t M, int64_t N, int64_t K,
                       int64_t lda, int64_t ldb, int64_t ldc, int epilogue, const void* epi_gate, int64_t epi_gate_stride,
                       const void* epi_res, int64_t epi_rows_per_batch, fdm_stream_t stream) {
  (void)hipGetLastError();  // drop any stale error of an earlier, unrelated runtime call: FDM_CHECK_LAUNCH must see only ours
  if (M == 0 && N > 0 && K > 0) return FDM_OK;
  GemmGroup g;
  int rc = fill_problem(g.p[0], out, a, b, M, N, K, lda, ldb, ldc, 1, epilogue, epi_gate, epi_gate_stride, epi_res, epi_rows_per_batch);
  if (rc != FDM_OK) return rc;
  if (!scale_a || !scale_b || ((azp == nullptr) != (azp_adj == nullptr))) return FDM_ERR_INVALID;
  if (azp_adj && (reinterpret_cast<uintptr_t>(azp_adj) & 15)) return FDM_ERR_INVALID;
  g.p[0].sa = scale_a;
  g.p[0].sb = scale_b;
  g.p[0].bias = (const bf16_t*)bias;
  g.p[0].azp = azp;
  g.p[0].azp_adj = azp_adj;
  finish_group(g, 1);
  return launch_gemm<DT_INT8>(g, to_stream(stream));
}

int64_t fdm_gemm_int8_workspace_bytes(void) { return 256 * SPLIT_PART_BYTES; }   // rem * S <= the CUs of the device

int fdm_gemm_w8a8_int8_x2(const fdm_gemm_int8_desc* d0, const fdm_gemm_int8_desc* d1, fdm_stream_t stream) {
  if (!d0 || !d1) return FDM_ERR_INVALID;
  return fdm_gemm_w8a8_int8_x2_ws(d0, d1, nullptr, 0, stream);
}

int fdm_gemm_w8a8_int8_x2_ws(const fdm_gemm_int8_desc* d0, const fdm_gemm_int8_desc* d1, void* workspace, int64_t workspace_bytes,
                             fdm_stream_t stream) {
  (void)hipGetLastError();  // drop any stale error of an earlier, unrelated runtime call: FDM_CHECK_LAUNCH must see only ours
  if (!d0) return FDM_ERR_INVALID;
  if (workspace != nullptr && ((reinterpret_cast<uintptr_t>(workspace) & 15) || workspace_bytes < 0)) return FDM_ERR_INVALID;
  const bool swap = d1 != nullptr && d1->M > 0 && d0->M > d1->M;       // the smaller problem first: see fdm_gemm_w8a8_fp8_x2
  const fdm_gemm_int8_desc* ds[2] = {swap ? d1 : d0, swap ? d0 : d1};
  GemmGroup g;
  int n = 0;
  for (int i = 0; i < 2; ++i) {
    const fdm_gemm_int8_desc* d = ds[i];
    if (d == nullptr) continue;
    if (d->M == 0 && d->N > 0 && d->K > 0) continue;
    int rc = fill_problem(g.p[n], d->out, d->a, d->b, d->M, d->N, d->K, d->lda, d->ldb, d->ldc, 1, (int)d->epilogue, d->epi_gate,
                          d->epi_gate_stride, d->epi_res, d->epi_rows_per_batch, d->out_plane_cols);
    if (rc == FDM_OK) rc = set_planes(g.p[n], d->out_plane_cols, d->out_plane_stride);
    if (rc != FDM_OK) return rc;
    if (!d->scale_a || !d->scale_b || ((d->azp == nullptr) != (d->azp_adj == nullptr))) return FDM_ERR_INVALID;
    if (d->azp_adj && (reinterpret_cast<uintptr_t>(d->azp_adj) & 15)) return FDM_ERR_INVALID;
    g.p[n].sa = d->scale_a;
    g.p[n].sb = d->scale_b;
    g.p[n].bias = (const bf16_t*)d->bias;
    g.p[n].azp = d->azp;
    g.p[n].azp_adj = d->azp_adj;
    ++n;
  }
  if (n == 0) return FDM_OK;
  finish_group(g, n);
  return launch_gemm<DT_INT8>(g, to_stream(stream), workspace, workspace_bytes);
}

int fdm_linear_skinny_bf16(void* out, const void* a, const void* b, const void* bias, const void* post_add, int64_t M, int64_t N,
                           int64_t K, int64_t lda, int64_t ldb, int64_t ldc, fdm_stream_t stream) {
  (void)hipGetLastError();
  if (!out || !a || !b || M < 0 || N <= 0 || K <= 0 || ldc < N || lda < K || ldb < K) return FDM_ERR_INVALID;
  if (M == 0) return FDM_OK;
  if (!gemv_applicable(M, N, K, lda, ldb, FDM_EPI_NONE, a, b)) return FDM_ERR_UNSUPPORTED;
  return launch_gemv(out, a, b, bias, post_add, M, N, K, lda, ldb, ldc, to_stream(stream));
}

int fdm_gemm_bf16(void* out, const void* a, const void* b, const void* bias, int64_t M, int64_t N, int64_t K, int64_t lda,
                  int64_t ldb, int64_t ldc, int epilogue, const void* epi_gate, int64_t epi_gate_stride, const void* epi_res,
                  int64_t epi_rows_per_batch, fdm_stream_t stream) {
  (void)hipGetLastError();  // drop any stale error of an earlier, unrelated runtime call: FDM_CHECK_LAUNCH must see only ours
  if (M == 0 && N > 0 && K > 0) return FDM_OK;
  if (out && a && b && ldc >= N && lda >= K && ldb >= K && gemv_applicable(M, N, K, lda, ldb, epilogue, a, b))
    return launch_gemv(out, a, b, bias, nullptr, M, N, K, lda, ldb, ldc, to_stream(stream));
  GemmGroup g;
  int rc = fill_problem(g.p[0], out, a, b, M, N, K, lda, ldb, ldc, 2, epilogue, epi_gate, epi_gate_stride, epi_res, epi_rows_per_batch);
  if (rc != FDM_OK) return rc;
  g.p[0].bias = (const bf16_t*)bias;
  finish_group(g, 1);
  return launch_gemm<DT_BF16>(g, to_stream(stream));
}

int fdm_gemm_bf16_x2(const fdm_gemm_bf16_desc* d0, const fdm_gemm_bf16_desc* d1, fdm_stream_t stream) {
  (void)hipGetLastError();  // drop any stale error of an earlier, unrelated runtime call: FDM_CHECK_LAUNCH must see only ours
  if (!d0 || !d1) return FDM_ERR_INVALID;
  const bool swap = d1->M > 0 && d0->M > d1->M;       // the smaller problem first in the tile list (as fdm_gemm_w8a8_fp8_x2)
  const fdm_gemm_bf16_desc* ds[2] = {swap ? d1 : d0, swap ? d0 : d1};
  GemmGroup g;
  int n = 0;
  for (int i = 0; i < 2; ++i) {
    const fdm_gemm_bf16_desc* d = ds[i];
    if (d->M == 0 && d->N > 0 && d->K > 0) continue;
    int rc = fill_problem(g.p[n], d->out, d->a, d->b, d->M, d->N, d->K, d->lda, d->ldb, d->ldc, 2, (int)d->epilogue, d->epi_gate,
                          d->epi_gate_stride, d->epi_res, d->epi_rows_per_batch, d->out_plane_cols);
    if (rc == FDM_OK) rc = set_planes(g.p[n], d->out_plane_cols, d->out_plane_stride);
    if (rc != FDM_OK) return rc;
    g.p[n].bias = (const bf16_t*)d->bias;
    ++n;
  }
  if (n == 0) return FDM_OK;
  finish_group(g, n);
  return launch_gemm<DT_BF16>(g, to_stream(stream));
}

int fdm_conv2d_3x3_nhwc_bf16(void* out, const void* x, const void* w, const void* bias, const void* res, const void* res_gate,
                             int64_t batch, int64_t h_in, int64_t w_in, int64_t c_in, int64_t c_out, int stride, int upsample2x,
                             void* workspace, int64_t workspace_bytes, fdm_stream_t stream) {
  (void)hipGetLastError();  // drop any stale error of an earlier, unrelated runtime call: FDM_CHECK_LAUNCH must see only ours
  if (!out || !x || !w || batch < 0 || h_in <= 0 || w_in <= 0 || c_in <= 0 || c_out <= 0) return FDM_ERR_INVALID;
  if ((stride != 1 && stride != 2) || (upsample2x != 0 && upsample2x != 1) || (upsample2x && stride != 1)) return FDM_ERR_UNSUPPORTED;
  if ((c_in % 32) || (c_out % 8)) return FDM_ERR_UNSUPPORTED;      // a 64-byte K step is 32 channels of one tap; 16-byte output pieces
  if (batch == 0) return FDM_OK;
  const int64_t h_out = upsample2x ? 2 * h_in : (h_in - 1) / stride + 1, w_out = upsample2x ? 2 * w_in : (w_in - 1) / stride + 1;
  const int64_t M = batch * h_out * w_out, K = 9 * c_in;
  if (M > 0x7fffffff || batch * h_in * w_in > 0x7fffffff) return FDM_ERR_UNSUPPORTED;
  if ((res != nullptr) != (res_gate != nullptr)) return FDM_ERR_INVALID;
  GemmGroup g;
  // (lda is not used by the gathered A operand; K = ldb = 9 * c_in: the weights are [c_out][3][3][c_in])
  int rc = fill_problem(g.p[0], out, x, w, M, c_out, K, K, K, c_out, 2, res ? FDM_EPI_GATE_RESIDUAL : FDM_EPI_NONE, res_gate, 0, res, M);
  if (rc != FDM_OK) return rc;
  GemmProblem& p = g.p[0];
  p.bias = (const bf16_t*)bias;
  p.cv_hin = (int)h_in;
  p.cv_win = (int)w_in;
  p.cv_hout = (int)h_out;
  p.cv_wout = (int)w_out;
  p.cv_cin = (int)c_in;
  p.cv_stride = stride;
  p.cv_up = upsample2x;
  finish_group(g, 1);
  hipStream_t s = to_stream(stream);
  if (workspace != nullptr && ((reinterpret_cast<uintptr_t>(workspace) & 15) || workspace_bytes < 0)) return FDM_ERR_INVALID;
  {
    // Few pixels, long K (SDXL level 2: 2 x 32 x 32 pixels x 1280 channels = 40 tiles of 180-360 K steps on 256 CUs): every tile's K range is cut
    // into S parts on S blocks, fp32 partial accumulators go through the workspace and a merge launch sums them in order and runs the epilogue
    // (the int8 GEMMs' split-K machinery; 239 -> ~100 us where MIOpen takes 154)
    const int cu = device_cu_count(), tiles = g.total_tiles;
    const int64_t nk = (K * 2 + BKB - 1) / BKB;
    int S = 1;
    if (workspace != nullptr && tiles * 2 <= cu) {
      S = cu / tiles;
      if (S > 8) S = 8;
      if (S > nk / 16) S = (int)(nk / 16);
      while (S > 1 && (int64_t)tiles * S * SPLIT_PART_BYTES > workspace_bytes) --S;
    }
    if (S > 1) {
      GemmGroup g2 = g;
      g2.split_tiles = tiles;
      g2.splits = S;
      g2.n_blocks = tiles * S;
      g2.ws = reinterpret_cast<int32_t*>(workspace);
      if (res) gemm_kernel<DT_BF16, true, 4, true><<<g2.n_blocks, GEMM_THREADS, 0, s>>>(g2);
      else gemm_kernel<DT_BF16, false, 4, true><<<g2.n_blocks, GEMM_THREADS, 0, s>>>(g2);
      FDM_CHECK_LAUNCH();
      g2.mode = 1;
      g2.spread_shift = tiles * 4 <= cu ? 2 : (tiles * 2 <= cu ? 1 : 0);
      if (res) gemm_kernel<DT_BF16, true, 4, true><<<tiles << g2.spread_shift, GEMM_THREADS, 0, s>>>(g2);
      else gemm_kernel<DT_BF16, false, 4, true><<<tiles << g2.spread_shift, GEMM_THREADS, 0, s>>>(g2);
      FDM_CHECK_LAUNCH();
      g_last_gemm_kernel = FDM_GEMM_KERNEL_W8;
      return FDM_OK;
    }
  }
  if (half_tiles_pay(g)) {      // the 8-wave kernel's 256 x 128 half tiles where its cost model prefers them (C_out = 320, 640: 128-wide tiles waste less of N)
    GemmGroup g2 = g;
    g2.p[0].tiles_n = (int)((c_out + BN / 2 - 1) / (BN / 2));
    g2.total_tiles = g2.n_blocks = g2.p[0].tiles_m * g2.p[0].tiles_n;
    if (res) gemm_kernel<DT_BF16, true, 2, true><<<g2.total_tiles, GEMM_THREADS / 2, 0, s>>>(g2);
    else gemm_kernel<DT_BF16, false, 2, true><<<g2.total_tiles, GEMM_THREADS / 2, 0, s>>>(g2);
    g_last_gemm_kernel = FDM_GEMM_KERNEL_W8_HALF;
  } else {
    if (res) gemm_kernel<DT_BF16, true, 4, true><<<g.total_tiles, GEMM_THREADS, 0, s>>>(g);
    else gemm_kernel<DT_BF16, false, 4, true><<<g.total_tiles, GEMM_THREADS, 0, s>>>(g);
    g_last_gemm_kernel = FDM_GEMM_KERNEL_W8;
  }
  FDM_CHECK_LAUNCH();
  return FDM_OK;
}

}  // extern "C"
