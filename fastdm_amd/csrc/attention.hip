// Dense / block-sparse flash-attention forward for gfx950 (bf16, head_dim 64 or 128, non-causal).
//
//   block     dense: 512 threads = 8 waves, one 256-row Q block of one (batch, head); block-sparse: 256 threads, one
//             128-row Q block = one mask row.  Wave w owns rows [32w, 32w+32).
//   K/V       tiles of 128 keys (dense, head_dim 128) or 64 keys, staged global -> LDS by LDS-DMA (global_load_lds, 16 B
//             per lane, no staging VGPRs), double buffered: 2 x (K + V) = 128 KiB at D = 128.
//   QK^T      swapped: S^T[key][q] = K . Q^T on v_mfma_f32_32x32x16_bf16, so lane (q = lane&31, h = lane>>5) holds 16 of
//             each 32 keys of ITS query row: row max / sum are in-lane plus one cross-half exchange (permlane32_swap).
//   PV        O^T[d][q] += V^T[d][key] . P^T[key][q].  The S^T accumulator, converted pairwise to bf16, IS the B operand
//             (no lane movement); the matching (permuted-k) A operand comes from the row-major V tile with
//             ds_read_b64_tr_b16 (hardware transpose).
//   LDS       K rows: chunk c of key r at slot c ^ (r & 15) (D=128) -> conflict-free ds_read_b128;
//             V rows: chunk c of key r at slot c ^ ((r & 3) << 2)  -> conflict-free transposed reads.  LDS-DMA writes are
//             lane-linear, so both swizzles are applied to the per-lane SOURCE address.
//   softmax   fp32 running max / sum in the exp2 domain (scale * log2 e folded into one FMA), lazy rescale.
//   grid      block id -> (xcd = id % 8): every q-block of a (batch, head) runs on one XCD so that K/V (4.4 MB per head at
//             S = 8704) is fetched from HBM once and re-read from that L2.  Split-KV: the q-blocks of a short last round
//             are cut along KV into parts that run side by side and are merged by attn_combine_kernel.
//   mask      optional int8 block mask [B, H, ceil(Sq/bq), ceil(Sk/bk)] (1 = compute): the active tiles of the block's mask
//             row are compacted once into an LDS list; skipped tiles cost no loads and no barrier.
//   bound     LDS operand traffic (1 KB per MFMA at 32 query rows per wave) -- DESIGN.md 4.5.
#include <cstdlib>
#include <type_traits>
#include "common.h"

namespace fdm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// s_setprio(1) around the MFMA clusters: measured -2 % with 128-key tiles (it paid +1 % with 64-key tiles), so off
#ifndef FDM_ATT_PRIO_QK
#define FDM_ATT_PRIO_QK 0
#endif
#ifndef FDM_ATT_PRIO_PV
#define FDM_ATT_PRIO_PV 0
#endif
// 1: the dense head_dim-128 path runs attn_fwd64_kernel (64 query rows per wave, one wave per SIMD); 0: attn_fwd_kernel
#ifndef FDM_ATT_V2
#define FDM_ATT_V2 1
#endif
// 1 (default since round 4): Q is pre-scaled by a power of two only (exact in bf16) and the rest of scale * log2(e) (1.0202 at
// head_dim 128) is applied in fp32 to the scores in front of every exp2 -- the reference's arithmetic (scale on fp32 scores,
// fastdm/kernel/torch/attention.py:34-43) -- at one extra v_mul_f32 per score: +3.9 % kernel time (843.5 -> 876.5 us at C3, one
// process, profiles/r04_experiments.md; the same factor as v_pk_mul_f32 on register pairs: +10.5 %, packed fp32 beside MFMAs is
// an anti-lever).  0: Q * scale * log2(e) rounded to bf16 once more (rounds 2-3): the first FLUX block then sits at 1.67 x (fp8) /
// 1.42 x (bf16) of the two-implementation envelope instead of 1.32 x / 1.08 x.
#ifndef FDM_ATT_EXACT_Q
#define FDM_ATT_EXACT_Q 1
#endif
#ifndef FDM_ATT_THREADS
#define FDM_ATT_THREADS 512
#endif
constexpr int ATT_THREADS = FDM_ATT_THREADS;
constexpr int QB = ATT_THREADS / 2;  // query rows per block: 32 per wave
constexpr int KT_MASK = 64;   // keys per tile on the block-mask path (= the mask's key granularity)
constexpr int KT_DENSE = 128;

struct AttnParams {
  bf16_t* o;
  const bf16_t* q;
  const bf16_t* k;
  const bf16_t* v;
  const int8_t* mask;
  int64_t sq, sk;
  int64_t q_ts, k_ts, v_ts, o_ts;  // token strides (elements)
  int batch, heads, kv_heads;
  int n_qblocks;
  int mask_bq, mask_bk, mask_nq, mask_nk;
  int mask_stretch;   // 1: a wave pair computes only on its own stretch of the block's union list; 0 (FDM_ATT_STRETCH=0, A/B): on all of it
  float scale_log2e;
  float q_prescale;   // what Q is multiplied by before its bf16 rounding (= scale_log2e; FDM_ATT_EXACT_Q: the nearest power of two)
  float exp_r;        // FDM_ATT_EXACT_Q: scale_log2e / q_prescale, applied in fp32 in front of every exp2
  // split-KV tail balancing: block ids >= n_full are parts of tail q-blocks (S parts each, consecutive KV ranges)
  int n_full, split;
  float* ws_o;   // [tail part][256 rows][D] unnormalised fp32 O
  float* ws_ml;  // [tail part][256 rows][2] running max (exp2 domain), row sum
  // fp8 P.V (attn_fwd64_pv8_kernel): the e4m3 V^T image attn_vq_kernel wrote and its per-channel scales
  const uint8_t* vq;     // [batch][kv head][ceil(sk / 64)][128 d][64 slots]
  const float* vscale;   // [batch][kv head][128]
  const float* vmean;    // [batch][kv head][128]: the channel offset the V image was centred on, added back to the output
  // fp8 QK^T as well (attn_fwd64_qk8_kernel): the e4m3 K image attn_kq_kernel wrote and the power of two it was scaled by
  const uint8_t* kq;     // [batch][kv head][ceil(sk / 64)][64 keys][128 d], 16-byte chunk c of key r at c ^ ((r >> 1) & 7)
  const int* kexp;       // [batch][kv head]: K image = e4m3(k * 2^kexp)
};

template <int D>
struct Swz {
  static constexpr int CPR = D / 8;  // 16-byte chunks per row
  // K image: conflict-free ds_read_b128 for 16-lane groups of distinct rows
  static __device__ __forceinline__ int k(int row, int ch) {
    if constexpr (D == 128) return ch ^ (row & 15);
    else return ch ^ ((row >> 1) & 7);
  }
  // V image: conflict-free ds_read_b64_tr_b16 (4 consecutive keys x 64 B per 32-lane half)
  static __device__ __forceinline__ int v(int row, int ch) {
    if constexpr (D == 128) return ch ^ ((row & 3) << 2);
    else return ch ^ (((row >> 1) & 1) << 2);
  }
};

// max over the two 32-lane halves of a wave (the two halves of a query row's keys), identical in both halves afterwards.
// NOT `__builtin_amdgcn_permlane32_swap(x, x)` + fmaxf of the two results: hipcc 7.2 folds that to the first result alone (it takes the
// swap of a value with itself for the identity), i.e. every lane got the LOWER half's maximum -- rounds 1-4 shipped that: harmless for the
// bf16 kernels' results (a missed fix-up only lets P grow, bf16 has the exponent range) but it defeats the lazy row reference for half of
// the keys, and with e4m3 P (fp8 P.V) a missed fix-up is a NaN.  Found with the spiked-key check of tools/micro/attn_bench (round 5).
// s_nop: VALU-written VGPR -> v_permlane32_swap needs a wait state that nothing pads inside asm.
__device__ __forceinline__ float cross_half_max(float m) {
  float a = m, b = m;
  asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));     // a = (lower, lower), b = (upper, upper)
  return fmaxf(a, b);
}

__device__ __forceinline__ i32x4 tr_pair(const uint8_t* p0, const uint8_t* p1) {
  s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p0);
  s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p1);
  i32x2 lo = __builtin_bit_cast(i32x2, a), hi = __builtin_bit_cast(i32x2, b);
  i32x4 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = hi[0]; r[3] = hi[1];
  return r;
}

// NT threads = NT/64 waves of 32 query rows: 512 (256-row q-blocks, one block per CU: the dense path) or 256 (128-row
// q-blocks = exactly one block-mask row, two blocks per CU: the block-sparse path, where the two halves of a 256-row
// block would carry different masks and wait for each other).
// CAUSAL (round 6; `is_causal=True` of the operator API, which no model of the reference passes): key j contributes to query row i iff j <= i
// (torch's top-left alignment for Sq != Sk).  A block stops at the tile that holds its last row's diagonal; tiles that reach past a wave's
// first row get the element mask.  Dense 512-thread form only (no block mask, no split-KV).
template <int D, int NT, int KTT, bool CAUSAL = false>
__global__ void __launch_bounds__(NT, 2) attn_fwd_kernel(const AttnParams p) {
  constexpr int KT = KTT;        // keys per tile: 128 on the dense path (half the barriers and DMA waits), 64 with a block mask
  constexpr int NKT = KT / 32;   // 32-key S^T tiles per KV tile
  constexpr int QBT = NT / 2;   // query rows per block
  constexpr int CPR = D / 8;
  constexpr int TILE_BYTES = KT * D * 2;               // one K or V tile
  constexpr int LD_PER_THREAD = KT * CPR / NT;  // 16-B chunks per thread per tile (2 @128, 1 @64)
  constexpr int KSTEPS = D / 16;                        // QK^T k-steps
  constexpr int DT = D / 32;                            // 32-wide d tiles of O^T
  __shared__ __attribute__((aligned(1024))) uint8_t smem[4 * TILE_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lq = lane & 31, h = lane >> 5;

  // ---- block id -> (batch*head, q block), heads pinned to XCDs
  const int BH = p.batch * p.heads;
  int item = blockIdx.x, part = 0;
  const bool is_part = item >= p.n_full;
  if (is_part) {
    part = (item - p.n_full) % p.split;
    item = p.n_full + (item - p.n_full) / p.split;
  }
  const int xcd = item & 7, j = item >> 3;
  const int bh = xcd + 8 * (j / p.n_qblocks);
  if (bh >= BH) return;
  const int qb = j % p.n_qblocks;
  const int b = bh / p.heads, head = bh % p.heads;
  const int kvh = head / (p.heads / p.kv_heads);
  const int64_t q0 = (int64_t)qb * QBT;

  const bf16_t* qbase = p.q + ((int64_t)b * p.sq) * p.q_ts + (int64_t)head * D;
  const bf16_t* kbase = p.k + ((int64_t)b * p.sk) * p.k_ts + (int64_t)kvh * D;
  const bf16_t* vbase = p.v + ((int64_t)b * p.sk) * p.v_ts + (int64_t)kvh * D;

  // ---- Q fragments: B operand of the swapped QK^T: lane holds Q[q][16*ks + 8*h .. +8]
  i32x4 qf[KSTEPS];
  {
    int64_t qrow = q0 + wave * 32 + lq;
    if (qrow > p.sq - 1) qrow = p.sq - 1;
    const bf16_t* qp = qbase + qrow * p.q_ts + 8 * h;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) qf[ks] = *reinterpret_cast<const i32x4*>(qp + 16 * ks);
  }

  // ---- block-mask bookkeeping.  Only the 4-wave instantiation (128-row q-blocks = one mask row) reads a mask; the
  //      host never launches the 8-wave kernel with one.  The row's active KV tiles are compacted once into an LDS
  //      list (wave 0, ballot + popcount), so the tile loop never chases mask bytes through global memory.
  constexpr bool MASKED = (NT == 256);
  constexpr int MAXL = MASKED ? 4096 : 1;
  __shared__ uint16_t slist[MAXL];
  __shared__ int s_nlist;
  const int ntiles = (int)((p.sk + KT - 1) / KT);
  const int8_t* mrow = nullptr;
  bool use_list = false;
  if constexpr (MASKED) {
    if (p.mask != nullptr) {
      const int64_t mb = ((int64_t)b * p.heads + head) * p.mask_nq;
      mrow = p.mask + (mb + (int)(q0 / p.mask_bq)) * p.mask_nk;
      use_list = ntiles <= MAXL;
      if (use_list) {
        if (wave == 0) {
          int cnt = 0;
          for (int base = 0; base < ntiles; base += 64) {
            const int tt = base + lane;
            const bool act = tt < ntiles && mrow[(tt * KT) / p.mask_bk] != 0;
            const unsigned long long bal = __ballot(act);
            if (act) slist[cnt + __popcll(bal & ((1ull << lane) - 1ull))] = (uint16_t)tt;
            cnt += __popcll(bal);
          }
          if (lane == 0) s_nlist = cnt;
        }
        __syncthreads();
      }
    }
  }
  int lpos = 0;                                // position in the active list
  const int n_list = use_list ? s_nlist : 0;
  auto tile_active = [&](int t) -> bool {
    if constexpr (!MASKED) return true;
    return mrow == nullptr || mrow[(t * KT) / p.mask_bk] != 0;
  };
  // first active tile at or after t (list mode: the next list entry; the list is consumed in order)
  auto next_active = [&](int t) -> int {
    if constexpr (MASKED) {
      if (use_list) {
        while (lpos < n_list && (int)slist[lpos] < t) ++lpos;
        return lpos < n_list ? (int)slist[lpos] : ntiles;
      }
    }
    while (t < ntiles && !tile_active(t)) ++t;
    return t;
  };

  // ---- K/V staging by LDS-DMA (global_load_lds, 16 B per lane).  A wave-instruction writes 64 consecutive 16-byte
  //      slots, so the destination is linear: thread t, round i fills slot p = i*NT + t = (row p / CPR, position p % CPR);
  //      the swizzle is applied on the SOURCE side (the slot at position pos of row r holds chunk pos ^ swz(r); both
  //      swizzles are XOR involutions).  No staging VGPRs, no ds_write.
  int srow[LD_PER_THREAD];
  int64_t koff[LD_PER_THREAD], voff[LD_PER_THREAD];
#pragma unroll
  for (int i = 0; i < LD_PER_THREAD; ++i) {
    const int pslot = tid + i * NT;
    srow[i] = pslot / CPR;
    const int pos = pslot % CPR;
    koff[i] = (int64_t)Swz<D>::k(srow[i], pos) * 8;
    voff[i] = (int64_t)Swz<D>::v(srow[i], pos) * 8;
  }
  auto stage_tile = [&](int t, int buf) {
    uint8_t* sk_ = smem + buf * 2 * TILE_BYTES + wave * 1024;
    uint8_t* sv_ = sk_ + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < LD_PER_THREAD; ++i) {
      int64_t key = (int64_t)t * KT + srow[i];
      if (key > p.sk - 1) key = p.sk - 1;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kbase + key * p.k_ts + koff[i]),
                                       (__attribute__((address_space(3))) void*)(sk_ + i * (NT * 16)), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vbase + key * p.v_ts + voff[i]),
                                       (__attribute__((address_space(3))) void*)(sv_ + i * (NT * 16)), 16, 0, 0);
    }
  };

  // ---- accumulators
  f32x16 oacc[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int e = 0; e < 16; ++e) oacc[d][e] = 0.f;
  float m_run = -1.0e30f, l_run = 0.f;

  // ---- LDS read bases (byte offsets into smem; the buffer bit is toggled with one XOR per tile, every other
  //      term of an address is a compile-time immediate).  tr-read geometry: 16-lane group gi, lane li = 4*tq + tp.
  constexpr uint32_t ROWB = D * 2;
  const int gi = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
  // K: chunk (2ks + h) ^ Y = (h ^ Y) ^ 2ks, so one base and a compile-time XOR per k-step
  uint32_t kld0 = lq * ROWB + (uint32_t)Swz<D>::k(lq, h) * 16;
  uint32_t vld[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d)
    vld[d] = TILE_BYTES + (4 * h + tq) * ROWB + (uint32_t)Swz<D>::v(4 * h + tq, d * 4 + 2 * (gi & 1) + (tp >> 1)) * 16 + 8 * (tp & 1);
  constexpr float RESCALE_THR = 8.0f;  // lazy rescale: P <= 2^8 between rescales (exp2 domain)

  // One KV tile for this wave.  TAIL: keys >= sk are masked (only the last tile of a sequence can need it).
  auto process_tile = [&](auto tail_c, int t) {
    constexpr bool TAIL = decltype(tail_c)::value;
    // ---- S^T = K . Q^T  (two 32-key tiles)
    f32x16 sacc[NKT];
#if FDM_ATT_PRIO_QK
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int kt2 = 0; kt2 < NKT; ++kt2)
#pragma unroll
      for (int e = 0; e < 16; ++e) sacc[kt2][e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const i32x4 qfr = qf[ks];
#pragma unroll
      for (int kt2 = 0; kt2 < NKT; ++kt2) {
        const i32x4 kf = *reinterpret_cast<const i32x4*>(smem + (kld0 ^ (uint32_t)(ks << 5)) + kt2 * 32 * ROWB);
        sacc[kt2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qfr),
                                                            sacc[kt2], 0, 0, 0);
      }
    }
#if FDM_ATT_PRIO_QK
    __builtin_amdgcn_s_setprio(0);
#endif
    if constexpr (TAIL) {
      const int rem = (int)(p.sk - (int64_t)t * KT) - 4 * h;  // keys of this tile that exist, shifted by the lane-half offset
#pragma unroll
      for (int kt2 = 0; kt2 < NKT; ++kt2)
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (kt2 * 32 + (e & 3) + 8 * (e >> 2) >= rem) sacc[kt2][e] = -INFINITY;
    }
    if constexpr (CAUSAL) {
      if ((int64_t)(t + 1) * KT - 1 > q0 + wave * 32) {     // (wave-uniform) the tile reaches past this wave's first row
        const int lim = (int)(q0 + wave * 32 + lq - (int64_t)t * KT) - 4 * h;   // last local key index of this lane's row, shifted by the lane-half offset
#pragma unroll
        for (int kt2 = 0; kt2 < NKT; ++kt2)
#pragma unroll
          for (int e = 0; e < 16; ++e)
            if (kt2 * 32 + (e & 3) + 8 * (e >> 2) > lim) sacc[kt2][e] = -INFINITY;
      }
    }
    // ---- online softmax in the exp2 domain with lazy rescale: the running max only moves (and O, l are only
    //      rescaled) when some row's tile max exceeds it by more than RESCALE_THR; the decision is taken before
    //      this tile is exponentiated and after the previous tile's P.V has been issued (textbook order).
    // (max as med3(a, b, +inf): hipcc turns the chain into v_max3_f32 without the canonicalising v_max x,x it puts in
    // front of every fmaxf on an MFMA result -- 19 instead of 56 VALU instructions per tile)
    float mx = fmaxf(sacc[0][0], sacc[1][0]);
#pragma unroll
    for (int kt2 = 0; kt2 < NKT; ++kt2)
#pragma unroll
      for (int e = (kt2 < 2 ? 1 : 0); e < 16; ++e) mx = __builtin_amdgcn_fmed3f(mx, sacc[kt2][e], __builtin_inff());
    mx = cross_half_max(mx);   // both halves of the query row
    const float mt = mx * p.scale_log2e;
    if (__any(mt > m_run + RESCALE_THR)) {
      const float m_new = fmaxf(m_run, mt);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) oacc[d][e] *= alpha;
    }
    float psum = 0.f;
    i32x4 pf[NKT][2];
#pragma unroll
    for (int kt2 = 0; kt2 < NKT; ++kt2) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[kt2][e], p.scale_log2e, -m_run));
        sacc[kt2][e] = pv;
        psum += pv;
      }
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) pf[kt2][s][w2] = (int)pack2bf(sacc[kt2][8 * s + 2 * w2], sacc[kt2][8 * s + 2 * w2 + 1]);
    }
    l_run += psum;
    // ---- O^T += V^T . P^T
#if FDM_ATT_PRIO_PV
    __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
    for (int kt2 = 0; kt2 < NKT; ++kt2)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          const uint8_t* a0 = smem + vld[d] + (kt2 * 32 + 16 * s) * ROWB;  // keys 4h+tq .. (+8 for the second group)
          const i32x4 vf = tr_pair(a0, a0 + 8 * ROWB);
          oacc[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf), __builtin_bit_cast(bf16x8, pf[kt2][s]),
                                                            oacc[d], 0, 0, 0);
        }
      }
#if FDM_ATT_PRIO_PV
    __builtin_amdgcn_s_setprio(0);
#endif
  };

#ifdef FDM_ATT_STATIC_PRIO
  if (wave >= NT / 128) __builtin_amdgcn_s_setprio(1);   // the later-dispatched half loses every arbitration otherwise
#endif
  // KV range of this work item: everything, or the part-th of `split` consecutive ranges
  const int t_begin = is_part ? (int)((int64_t)part * ntiles / p.split) : 0;
  int t_hi = ntiles;
  if constexpr (CAUSAL) {      // the block's last (existing) query row bounds the keys anyone here may see
    const int64_t last_q = (q0 + QBT < p.sq ? q0 + QBT : p.sq) - 1;
    const int64_t t_diag = last_q / KT + 1;
    if (t_diag < t_hi) t_hi = (int)t_diag;
  }
  const int t_end = is_part ? (int)((int64_t)(part + 1) * ntiles / p.split) : t_hi;
  int t = next_active(t_begin);
  int cur = 0;
  if (t < t_end) stage_tile(t, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMAs (and its Q loads) have landed ...
  __syncthreads();                                    // ... and everybody's

  // Full tiles run the TAIL=false body in the loop; a partial last tile (sk % 64 != 0) is handled once, after
  // the loop, by the masked body (keeping both bodies inside the loop costs > 90 VGPRs on this compiler).
  const int n_main = (p.sk % KT == 0 || t_end < ntiles) ? t_end : ntiles - 1;
  while (t < n_main) {
    const int tn = next_active(t + 1);
    if (tn < t_end) stage_tile(tn, cur ^ 1);    // the other buffer was last read before the previous barrier

    process_tile(std::false_type{}, t);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
    kld0 ^= 2 * TILE_BYTES;
#pragma unroll
    for (int d = 0; d < DT; ++d) vld[d] ^= 2 * TILE_BYTES;
    t = tn;
  }
  if (t < t_end) process_tile(std::true_type{}, t);  // the partial tile is resident in buffer `cur`

  // ---- finalize: l = own half + other half; O = O^T / l
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (is_part) {  // unnormalised partial: O (fp32), m, l -> workspace; attn_combine_kernel merges the parts
    const int64_t slot = (int64_t)(blockIdx.x - p.n_full);
    const int row = wave * 32 + lq;
    float* wo = p.ws_o + (slot * QBT + row) * D;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        f32x4 w;
        w[0] = oacc[d][4 * g4]; w[1] = oacc[d][4 * g4 + 1]; w[2] = oacc[d][4 * g4 + 2]; w[3] = oacc[d][4 * g4 + 3];
        *reinterpret_cast<f32x4*>(wo + 32 * d + 8 * g4 + 4 * h) = w;
      }
    if (h == 0) {
      p.ws_ml[(slot * QBT + row) * 2] = m_run;
      p.ws_ml[(slot * QBT + row) * 2 + 1] = l_tot;
    }
    return;
  }
  const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
  const int64_t qrow = q0 + wave * 32 + lq;
  if (qrow < p.sq) {
    bf16_t* op = p.o + ((int64_t)b * p.sq + qrow) * p.o_ts + (int64_t)head * D;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        // registers 4*g4 .. 4*g4+3 hold d = 32*d + 8*g4 + 4*h + (0..3): 8 contiguous bytes
        i32x2 w;
        w[0] = (int)pack2bf(oacc[d][4 * g4] * inv, oacc[d][4 * g4 + 1] * inv);
        w[1] = (int)pack2bf(oacc[d][4 * g4 + 2] * inv, oacc[d][4 * g4 + 3] * inv);
        *reinterpret_cast<i32x2*>(op + 32 * d + 8 * g4 + 4 * h) = w;
      }
  }
}


// ================================================================================================================
// Dense path, head_dim 128: 64 query rows per wave, ONE wave per SIMD.
//
//   block     256 threads = 4 waves, one 256-row Q block of one (batch, head); wave w owns rows [64w, 64w+64) as two
//             32-row q-blocks.  __launch_bounds__(256, 1): the wave has the whole 512-entry register file:
//             accumulation registers  O (128) + pre-scaled Q fragments (64) + K / V^T fragments in flight,
//             vector registers        two S tiles (128) + row-reference tiles (32) + packed P (32) + addresses.
//   why       with 32 rows per wave every MFMA needs its own 1 KB LDS fragment and two waves share a SIMD, whose VALU
//             (softmax) and MFMA streams do not overlap (measured: 743 cycles for what one interleaved wave does in
//             551).  Here each K / V^T fragment feeds TWO MFMAs and one wave carries three strands -- QK^T of tile t+1,
//             exp2 / row sums of tile t, P.V of tile t -- interleaved MFMA gap by MFMA gap.
//   schedule  hipcc left alone moves whole tiles between the two register classes (2300 v_accvgpr moves), so the MFMAs
//             are inline asm with fixed classes and the issue order is pinned with sched_barrier(0) per MFMA gap:
//             gap = 1 MFMA + <= 1 v_exp_f32 + a few plain VALU + <= 2 LDS reads (or one LDS-DMA).
//   softmax   Q is pre-scaled once per block -- by the power of two nearest to scale*log2(e), exact in bf16; the remaining factor
//             r (1.0202 at head_dim 128) multiplies the fp32 scores in front of exp2 (FDM_ATT_EXACT_Q; rounds 2-3 rounded
//             Q*scale*log2(e) to bf16 instead and saved that multiply) -- and the S^T accumulators START at -m_ref[q] (the C
//             operand of the first MFMA of a tile is a register tile holding the row reference), so P = exp2(r S') needs no
//             per-element subtraction.  m_ref = (row max at the last fix-up) + 7: P <= 2 between fix-ups.  The row max of tile
//             t+1 is taken under the P.V MFMAs of tile t; only if some row's S' reaches 1 (its scores grew by 2^8 since
//             the last fix-up), on the first tile and on a masked tail tile does the wave take the fix-up path
//             (integer shift delta: S' -= delta, O and l scaled by 2^-delta -- exact -- and the reference tile rewritten).
//   K/V       64-key tiles in a 4-stage LDS ring (128 KiB), LDS-DMA three tiles ahead (wave-uniform base + 32-bit lane
//             offset: no VALU address math), ONE barrier per tile behind a counted vmcnt(8).
//   hazards   (the assembler pads nothing inside asm)  VALU-written P -> MFMA operand: s_nop 1 in the first MFMA of a
//             P.V group; MFMA result -> VALU: S tiles are read one phase after they were written and pass through an
//             ordering asm behind the P.V MFMAs; O is read by VALU only in the fix-up path / epilogue, behind >= 16
//             other instructions.
template <int I>
using ic = std::integral_constant<int, I>;
template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(ic<I>{});
    static_for<N, I + 1>(f);
  }
}
#define FDM_FENCE() __builtin_amdgcn_sched_barrier(0)

// 1: the softmax row sums are taken by 8 extra MFMAs per tile (ones x P^T, same B fragments as P.V) instead of 64 v_add_f32
#ifndef FDM_A64_LSUM_MFMA
#define FDM_A64_LSUM_MFMA 0
#endif
// 1: packed fp32 in the bf16 kernel's softmax: the exact-Q scale multiply and the row sums take two scores per instruction
// (v_pk_mul_f32 / v_pk_add_f32: 64 fewer of ~500 VALU instructions per tile step).  Measured in round 5 (profiles/r05_experiments.md):
// same results bit for bit and 12 % SLOWER (865 -> 975 us at S = 8704): beside a busy matrix core the packed fp32 forms cost more issue
// time than the two scalar instructions they replace.  Kept for the record.
#ifndef FDM_A64_PK
#define FDM_A64_PK 0
#endif
// 1: the VALU work sits in the odd MFMA gaps (no fragment read + wait there); 0: spread evenly
#ifndef FDM_A64_BAL
#define FDM_A64_BAL 0      // measured in one process (tools/micro/attn_ab): 762 vs 756 us, kept for the record
#endif
// timing-only ablations (tools/micro; wrong results): 1 no DMA in the loop, 2 no fragment reads, 4 no exp2, 8 no row sums,
// 16 no row maxima, 32 no barrier, 64 no packs
#ifndef FDM_A64_ABL
#define FDM_A64_ABL 0
#endif
#ifdef FDM_ATT_STAMPS
// diagnostic build only (tools/micro/attn_bench.hip): per-block cycle sums of the step's segments
__device__ unsigned long long g_att_stamps[4096 * 8];
#define FDM_STAMP(VAR)                                                                       \
  FDM_FENCE();                                                                               \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(VAR)::"memory");                \
  FDM_FENCE()
#else
#define FDM_STAMP(VAR)
#endif

// The kernel body lives in attn_fwd64_body.inc and is compiled twice:
//   attn_fwd64_kernel      bf16 P.V (v_mfma_f32_32x32x16_bf16): the default path, the reference torch backend's arithmetic
//   attn_fwd64_pv8_kernel  fp8 P.V  (v_mfma_f32_32x32x64_f8f6f4): opt-in (fdm_attn_fwd_pvfp8), the class of the reference's
//                          SageAttention path (/root/reference/fastdm/kernel/cuda/attention.py:35-75)
#define FDM_A64_PV8 0
#define FDM_A64_QK8 0
#define FDM_A64_NAME attn_fwd64_kernel
#include "attn_fwd64_body.inc"
#undef FDM_A64_QK8
#undef FDM_A64_PV8
#undef FDM_A64_NAME
#define FDM_A64_PV8 1
#define FDM_A64_QK8 0
#define FDM_A64_NAME attn_fwd64_pv8_kernel
#include "attn_fwd64_body.inc"
#undef FDM_A64_QK8
#undef FDM_A64_NAME
//   attn_fwd64_qk8_kernel  fp8 QK^T too (v_mfma_scale_f32_32x32x64_f8f6f4 with power-of-two block scales): mode 2 of fdm_attn_fwd_quant
#define FDM_A64_QK8 1
#define FDM_A64_NAME attn_fwd64_qk8_kernel
#include "attn_fwd64_body.inc"
#undef FDM_A64_QK8
#undef FDM_A64_PV8
#undef FDM_A64_NAME

// ---- pre-pass of the quantised modes: per-channel statistics, then the e4m3 images the fp8 kernels stream (layouts: attn_fwd64_body.inc).
// Both tensors are SMOOTHED the way SageAttention does it (/root/reference/fastdm/kernel/cuda/attention.py:35-75 calls sageattn with its
// defaults smooth_k / per-channel V): a per-channel offset c[d] (the mean over the tokens) is subtracted before quantisation --
//   K: softmax(q . (k - c)) = softmax(q . k - q . c) = softmax(q . k): a per-row constant, exact whatever c is;
//   V: sum_j p_j (v_j - c) + c = sum_j p_j v_j because the weights sum to 1 (the kernel divides by the sum of the SAME rounded weights):
//      exact whatever c is; c is added back in the epilogue.
// so e4m3's three mantissa bits are spent on the deviations, not on a common offset (K = -8 +- 0.25 has no information left otherwise).
// One statistics record per (batch, kv head): deterministic (the channel sums are order-free 64-bit fixed-point atomics, min / max atomics
// on monotone integer keys), zeroed by a memset of the call.  HBM-bound: each tensor is read twice, its image written once.
struct AttnChanStats {
  long long sum[128];      // sum over the tokens of round(x * 2^16)
  uint32_t maxk[128];      // max over the tokens of key(x)   (key: monotone float -> uint)
  uint32_t nminK[128];     // max over the tokens of key(-x)
};
__device__ __forceinline__ uint32_t fkey(float f) {
  const uint32_t b = __builtin_bit_cast(uint32_t, f);
  return b ^ ((b >> 31) ? 0xFFFFFFFFu : 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(uint32_t k) { return __builtin_bit_cast(float, (k & 0x80000000u) ? (k ^ 0x80000000u) : ~k); }

__global__ void __launch_bounds__(256) attn_chan_stats_kernel(const bf16_t* __restrict__ x, int64_t sk, int64_t ts, int kv_heads,
                                                               AttnChanStats* __restrict__ st) {
  __shared__ float rs[16][128], rmx[16][128], rmn[16][128];
  const int tid = threadIdx.x, dch = tid & 15, tl = tid >> 4;
  const int kvh = blockIdx.y, b = blockIdx.z;
  const int64_t t0 = (int64_t)blockIdx.x * 256;
  float sm[8], mx[8], mn[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { sm[j] = 0.f; mx[j] = -3.0e38f; mn[j] = 3.0e38f; }
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int64_t tok = t0 + i * 16 + tl;
    if (tok < sk) {
      const Vec8 v = load8(x + ((int64_t)b * sk + tok) * ts + (int64_t)kvh * 128 + dch * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float c = fminf(fmaxf(v.v[j], -1.0e6f), 1.0e6f);      // (NaN -> -1e6 by fmaxf / fminf: any finite offset is valid; the quantiser saturates)
        sm[j] += c; mx[j] = fmaxf(mx[j], c); mn[j] = fminf(mn[j], c);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { rs[tl][dch * 8 + j] = sm[j]; rmx[tl][dch * 8 + j] = mx[j]; rmn[tl][dch * 8 + j] = mn[j]; }
  __syncthreads();
  if (tid < 128) {
    float a = rs[0][tid], hi = rmx[0][tid], lo = rmn[0][tid];
#pragma unroll
    for (int i = 1; i < 16; ++i) { a += rs[i][tid]; hi = fmaxf(hi, rmx[i][tid]); lo = fminf(lo, rmn[i][tid]); }      // fixed order: deterministic
    AttnChanStats* r = st + ((int64_t)b * kv_heads + kvh);
    atomicAdd(reinterpret_cast<unsigned long long*>(&r->sum[tid]), (unsigned long long)(long long)__builtin_llrint((double)a * 65536.0));
    if (hi >= lo) {
      atomicMax(&r->maxk[tid], fkey(hi));
      atomicMax(&r->nminK[tid], fkey(-lo));
    }
  }
}
// channel offset and the largest |x - offset| of a channel, from its record
__device__ __forceinline__ void chan_center(const AttnChanStats& r, int d, int64_t sk, float& c, float& dev) {
  c = (float)((double)r.sum[d] * (1.0 / 65536.0) / (double)sk);
  const float hi = fkey_inv(r.maxk[d]), lo = -fkey_inv(r.nminK[d]);
  dev = fmaxf(hi - c, c - lo);
  if (!(dev > 0.f)) dev = 0.f;
}

__global__ void __launch_bounds__(256) attn_vq_kernel(const bf16_t* __restrict__ v, int64_t sk, int64_t v_ts, int kv_heads, int ntiles,
                                                       const AttnChanStats* __restrict__ st, uint8_t* __restrict__ vq,
                                                       float* __restrict__ vscale, float* __restrict__ vmean) {
  __shared__ __attribute__((aligned(16))) bf16_t tile[64][128 + 8];
  const int tid = threadIdx.x;
  const int t = blockIdx.x, kvh = blockIdx.y, b = blockIdx.z;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i, row = c >> 4, ch = c & 15;
    const int64_t tok = (int64_t)t * 64 + row;
    i32x4 x = {0, 0, 0, 0};
    if (tok < sk) x = *reinterpret_cast<const i32x4*>(v + ((int64_t)b * sk + tok) * v_ts + (int64_t)kvh * 128 + ch * 8);
    *reinterpret_cast<i32x4*>(&tile[row][ch * 8]) = x;
  }
  __syncthreads();
  const int d = tid >> 1, g = tid & 1;
  const int64_t bh = (int64_t)b * kv_heads + kvh;
  float cen, dev;
  chan_center(st[bh], d, sk, cen, dev);
  const float inv = dev > 0.f ? 448.0f / dev : 0.f;
  if (t == 0 && g == 0) {
    vscale[bh * 128 + d] = dev > 0.f ? dev / 448.0f : 1.0f;
    vmean[bh * 128 + d] = cen;
  }
  const int valid = (int)min((int64_t)64, sk - (int64_t)t * 64);      // keys past the end: zeros (their P is 0)
  uint32_t w[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {          // register r = 4 kb + m holds slots 4 r .. 4 r + 3 of this lane group: keys 32 kb + 8 m + 4 g + c
    const int kb = r >> 2, m = r & 3, key = 32 * kb + 8 * m + 4 * g;
    float x[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) x[c] = key + c < valid ? clamp448((bf2f(tile[key + c][d]) - cen) * inv) : 0.f;
    w[r] = cvt2_e4m3<false>(x[0], x[1]) | (cvt2_e4m3<false>(x[2], x[3]) << 16);
  }
  uint8_t* row = vq + ((bh * ntiles + t) * 128 + d) * 64;
  const int f = (d >> 2) & 3;
  *reinterpret_cast<i32x4*>(row + (((2 * g) ^ f) << 4)) = i32x4{(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
  *reinterpret_cast<i32x4*>(row + (((2 * g + 1) ^ f) << 4)) = i32x4{(int)w[4], (int)w[5], (int)w[6], (int)w[7]};
}

// K image: e4m3((k - c) * 2^kexp), one power of two per (batch, kv head) with max |k - c| * 2^kexp in [128, 256) (exact; undone by the MFMA's
// E8M0 block scale).  Row-major per key, no transpose.
__global__ void __launch_bounds__(256) attn_kq_kernel(const bf16_t* __restrict__ k, int64_t sk, int64_t k_ts, int kv_heads, int ntiles,
                                                       const AttnChanStats* __restrict__ st, uint8_t* __restrict__ kq, int* __restrict__ kexp) {
  __shared__ float cen_s[128];
  __shared__ float red[4];
  const int tid = threadIdx.x;
  const int t = blockIdx.x, kvh = blockIdx.y, b = blockIdx.z;
  const int64_t bh = (int64_t)b * kv_heads + kvh;
  float dev = 0.f;
  if (tid < 128) {
    float c;
    chan_center(st[bh], tid, sk, c, dev);
    cen_s[tid] = c;
  }
  const float amax = block_reduce<256>(dev, OpMax{}, red);      // (also the barrier that publishes cen_s)
  int e = 0;
  if (amax > 0.f) {
    e = 7 - (int)(((__builtin_bit_cast(uint32_t, amax) >> 23) & 0xff) - 127);      // amax in [2^x, 2^(x+1)) -> amax * 2^e in [128, 256)
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
  }
  if (t == 0 && tid == 0) kexp[bh] = e;
  const float mul = __builtin_bit_cast(float, (uint32_t)(127 + e) << 23);
  const int key = tid >> 2, quarter = tid & 3;       // 32 d per thread: chunks 2 * quarter, 2 * quarter + 1
  const int64_t tok = (int64_t)t * 64 + key;
  uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};          // keys past the end: zeros
  if (tok < sk) {
    const bf16_t* src = k + ((int64_t)b * sk + tok) * k_ts + (int64_t)kvh * 128 + quarter * 32;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const Vec8 x = load8(src + 8 * c);
      float y[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) y[j] = clamp448((x.v[j] - cen_s[quarter * 32 + 8 * c + j]) * mul);
      w[2 * c] = cvt2_e4m3<false>(y[0], y[1]) | (cvt2_e4m3<false>(y[2], y[3]) << 16);
      w[2 * c + 1] = cvt2_e4m3<false>(y[4], y[5]) | (cvt2_e4m3<false>(y[6], y[7]) << 16);
    }
  }
  uint8_t* row = kq + ((bh * ntiles + t) * 64 + key) * 128;
  const int f = (key >> 1) & 7;
  *reinterpret_cast<i32x4*>(row + (((2 * quarter) ^ f) << 4)) = i32x4{(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
  *reinterpret_cast<i32x4*>(row + (((2 * quarter + 1) ^ f) << 4)) = i32x4{(int)w[4], (int)w[5], (int)w[6], (int)w[7]};
}

// Merge the split-KV parts of the tail q-blocks: O = sum_p 2^(m_p - M) O_p / sum_p 2^(m_p - M) l_p, M = max_p m_p.
// One thread per (row, 8 consecutive d).
template <int D>
__global__ void __launch_bounds__(256) attn_combine_kernel(const AttnParams p, int n_tail) {
  constexpr int CPR = D / 8;
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (int64_t)n_tail * QB * CPR) return;
  const int ch = (int)(gid % CPR);
  const int row = (int)((gid / CPR) % QB);
  const int tail = (int)(gid / ((int64_t)CPR * QB));
  const int item = p.n_full + tail;
  const int xcd = item & 7, j = item >> 3;
  const int bh = xcd + 8 * (j / p.n_qblocks);
  const int qb = j % p.n_qblocks;
  const int b = bh / p.heads, head = bh % p.heads;
  const int64_t qrow = (int64_t)qb * QB + row;
  if (bh >= p.batch * p.heads || qrow >= p.sq) return;
  float M = -3.0e38f;
  for (int s = 0; s < p.split; ++s) M = fmaxf(M, p.ws_ml[(((int64_t)tail * p.split + s) * QB + row) * 2]);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float L = 0.f;
  for (int s = 0; s < p.split; ++s) {
    const int64_t slot = (int64_t)tail * p.split + s;
    const float w = __builtin_amdgcn_exp2f(p.ws_ml[(slot * QB + row) * 2] - M);
    L += w * p.ws_ml[(slot * QB + row) * 2 + 1];
    const f32x4 a = *reinterpret_cast<const f32x4*>(p.ws_o + (slot * QB + row) * D + ch * 8);
    const f32x4 c = *reinterpret_cast<const f32x4*>(p.ws_o + (slot * QB + row) * D + ch * 8 + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[i] += w * a[i];
      acc[4 + i] += w * c[i];
    }
  }
  const float inv = L > 0.f ? 1.0f / L : 0.f;
  float off[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (p.vmean != nullptr && L > 0.f) {      // quantised modes: the V image was centred per channel; the weights sum to 1
    const float* vm = p.vmean + (int64_t)(b * p.kv_heads + head / (p.heads / p.kv_heads)) * D + ch * 8;
#pragma unroll
    for (int i = 0; i < 8; ++i) off[i] = vm[i];
  }
  i32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = (int)pack2bf(acc[2 * i] * inv + off[2 * i], acc[2 * i + 1] * inv + off[2 * i + 1]);
  *reinterpret_cast<i32x4*>(p.o + ((int64_t)b * p.sq + qrow) * p.o_ts + (int64_t)head * D + ch * 8) = o;
}

}  // namespace fdm

using namespace fdm;

extern "C" int64_t fdm_attn_fwd_workspace_bytes(int head_dim) {
  // 1024 part slots x 256 rows x (D + 2) floats (136 MB at D = 128); fdm_attn_fwd only picks splits that fit
  return (int64_t)1024 * QB * (head_dim + 2) * 4;
}

extern "C" int64_t fdm_attn_pvfp8_workspace_bytes(int64_t batch, int64_t sk, int kv_heads, int head_dim) {
  if (batch <= 0 || sk <= 0 || kv_heads <= 0 || head_dim != 128) return 0;
  const int64_t ntiles = (sk + 63) / 64;
  return batch * kv_heads * (ntiles * 64 * 128 + (int64_t)sizeof(AttnChanStats) + 2 * 128 * 4);      // e4m3 V^T image + channel statistics + scales + offsets
}
extern "C" int64_t fdm_attn_fp8_workspace_bytes(int64_t batch, int64_t sk, int kv_heads, int head_dim) {
  if (batch <= 0 || sk <= 0 || kv_heads <= 0 || head_dim != 128) return 0;
  const int64_t ntiles = (sk + 63) / 64;
  // the P.V part, then the e4m3 K image + its channel statistics + per-head exponents (16-byte aligned)
  return fdm_attn_pvfp8_workspace_bytes(batch, sk, kv_heads, head_dim) + batch * kv_heads * (ntiles * 64 * 128 + (int64_t)sizeof(AttnChanStats)) +
         ((batch * kv_heads * 4 + 15) & ~(int64_t)15);
}

static int attn_fwd_impl(void* o, const void* q, const void* k, const void* v, int64_t batch, int64_t sq, int64_t sk,
                         int q_heads, int kv_heads, int head_dim, int64_t q_tok_stride, int64_t k_tok_stride,
                         int64_t v_tok_stride, int64_t o_tok_stride, float scale, const int8_t* block_mask, int block_q,
                         int block_k, void* workspace, int64_t workspace_bytes, fdm_stream_t stream, int qmode, void* vq_ws,
                         int64_t vq_bytes, bool causal = false) {
  const bool pv8 = qmode != 0;      // qmode: 0 bf16, 1 fp8 P.V, 2 fp8 QK^T + fp8 P.V
  if (causal && (pv8 || block_mask != nullptr)) return FDM_ERR_UNSUPPORTED;
  (void)hipGetLastError();  // drop any stale error of an earlier, unrelated runtime call: FDM_CHECK_LAUNCH must see only ours
  if (!o || !q || !k || !v || batch <= 0 || sq <= 0 || sk <= 0 || q_heads <= 0 || kv_heads <= 0 || q_heads % kv_heads)
    return FDM_ERR_INVALID;
  if (head_dim != 64 && head_dim != 128) return FDM_ERR_UNSUPPORTED;
  if (pv8) {
    if (head_dim != 128) return FDM_ERR_UNSUPPORTED;                       // quantised modes: the head_dim-128 64-row kernels only
    if (block_mask != nullptr && (sk + 63) / 64 > 4096) return FDM_ERR_UNSUPPORTED;
    const int64_t need = qmode == 2 ? fdm_attn_fp8_workspace_bytes(batch, sk, kv_heads, head_dim) : fdm_attn_pvfp8_workspace_bytes(batch, sk, kv_heads, head_dim);
    if (!vq_ws || (reinterpret_cast<uintptr_t>(vq_ws) & 15) || vq_bytes < need) return FDM_ERR_INVALID;
  }
  if ((q_tok_stride & 7) || (k_tok_stride & 7) || (v_tok_stride & 7) || (o_tok_stride & 7)) return FDM_ERR_INVALID;
  if ((reinterpret_cast<uintptr_t>(q) & 15) || (reinterpret_cast<uintptr_t>(k) & 15) || (reinterpret_cast<uintptr_t>(v) & 15) ||
      (reinterpret_cast<uintptr_t>(o) & 15))
    return FDM_ERR_INVALID;
  AttnParams p;
  p.o = (bf16_t*)o;
  p.q = (const bf16_t*)q;
  p.k = (const bf16_t*)k;
  p.v = (const bf16_t*)v;
  p.mask = block_mask;
  p.sq = sq;
  p.sk = sk;
  p.q_ts = q_tok_stride;
  p.k_ts = k_tok_stride;
  p.v_ts = v_tok_stride;
  p.o_ts = o_tok_stride;
  p.batch = (int)batch;
  p.heads = q_heads;
  p.kv_heads = kv_heads;
  const bool mask_v2 = FDM_ATT_V2 && block_mask != nullptr && head_dim == 128 && (sk + 63) / 64 <= 4096;   // 64-row-wave kernel, union of the block's mask rows
  const int qb_rows = (block_mask != nullptr && !mask_v2) ? 128 : QB;   // old block-sparse path: 4-wave blocks, one mask row each
  p.n_qblocks = (int)((sq + qb_rows - 1) / qb_rows);
  p.mask_bq = p.mask_bk = 1;
  p.mask_nq = p.mask_nk = 0;
  {
#ifdef FDM_ATT_ENV_KNOBS      // A/B builds only: the product library reads no environment variable
    static const int stretch = [] { const char* e = getenv("FDM_ATT_STRETCH"); return (e != nullptr && e[0] == '0') ? 0 : 1; }();
    p.mask_stretch = stretch;
#else
    p.mask_stretch = 1;
#endif
  }
  if (block_mask != nullptr) {
    if (block_q <= 0 || block_k <= 0 || block_q % 128 || block_k % 64) return FDM_ERR_UNSUPPORTED;
    p.mask_bq = block_q;
    p.mask_bk = block_k;
    p.mask_nq = (int)((sq + block_q - 1) / block_q);
    p.mask_nk = (int)((sk + block_k - 1) / block_k);
  }
  p.scale_log2e = scale * 1.4426950408889634f;
#if FDM_ATT_EXACT_Q
  p.q_prescale = pv8 ? p.scale_log2e : exp2f(rintf(log2f(p.scale_log2e)));     // fp8 P.V folds the whole factor into Q (attn_fwd64_body.inc)
  p.exp_r = p.scale_log2e / p.q_prescale;
#else
  p.q_prescale = p.scale_log2e;
  p.exp_r = 1.0f;
#endif
  const int64_t bh = batch * q_heads;
  int64_t grid = 8 * ((bh + 7) / 8) * p.n_qblocks;
  hipStream_t s = to_stream(stream);
  // Split-KV tail balancing: q-blocks are equal-sized work items, one per CU at a time; when the last round would
  // fill less than half of the chip (FLUX 1024x2048: 816 items on 256 CUs = 3.19 rounds), the items of that round
  // are split along KV into S parts that run side by side and are merged by a small combine kernel.
  p.n_full = (int)grid;
  p.split = 1;
  p.ws_o = nullptr;
  p.ws_ml = nullptr;
  int n_tail = 0;
  // a workspace that cannot be used is an error, not a silent fall-back to the (20 % slower) unsplit schedule
  if (workspace != nullptr && ((reinterpret_cast<uintptr_t>(workspace) & 15) || workspace_bytes < 0)) return FDM_ERR_INVALID;
  if (block_mask == nullptr && workspace != nullptr && (bh % 8) == 0 && !causal) {
    const int num_cu = device_cu_count() * (512 / ATT_THREADS);  // concurrent work-item slots on the chip (per device)
    const int rem = (int)(grid % num_cu);
    const int kt_dense = head_dim == 128 ? KT_DENSE : KT_MASK;
    const int ntiles = (int)((sk + kt_dense - 1) / kt_dense);
    if (rem > 0) {
      // time of the last round in units of one whole item: ceil(rem*S / CUs) rounds of 1/S-sized parts, plus a
      // per-part overhead (Q load, prologue, fp32 partial write + merge) priced at 2% of an item
      // (a split must save >= 0.3 item-times: measured at 432 items, a predicted 0.17 saving was a 5% loss -- a
      // partly filled last round runs faster per item than the model assumes)
      int best = 1;
      float best_cost = 0.75f;
      for (int S = 2; S <= 8 && S * 8 <= ntiles; ++S) {
        const int64_t need = (int64_t)rem * S * QB * (head_dim + 2) * 4;
        if (need > workspace_bytes) break;
        const float cost = (float)(((int64_t)rem * S + num_cu - 1) / num_cu) / (float)S + 0.02f * (float)S;
        if (cost < best_cost - 0.05f) {
          best = S;
          best_cost = cost;
        }
      }
      if (best > 1) {
        n_tail = rem;
        p.split = best;
        p.n_full = (int)(grid - rem);
        p.ws_o = reinterpret_cast<float*>(workspace);
        p.ws_ml = p.ws_o + (int64_t)rem * best * QB * head_dim;
        grid = p.n_full + (int64_t)rem * best;
      }
    }
  }
  p.vq = nullptr;
  p.vscale = nullptr;
  p.vmean = nullptr;
  p.kq = nullptr;
  p.kexp = nullptr;
  if (pv8) {
    const int ntl = (int)((sk + 63) / 64);
    uint8_t* img = reinterpret_cast<uint8_t*>(vq_ws);
    AttnChanStats* vst = reinterpret_cast<AttnChanStats*>(img + batch * kv_heads * (int64_t)ntl * 8192);
    float* vsc = reinterpret_cast<float*>(vst + batch * kv_heads);
    float* vmn = vsc + batch * kv_heads * 128;
    if (hipMemsetAsync(vst, 0, (size_t)batch * kv_heads * sizeof(AttnChanStats), s) != hipSuccess) return FDM_ERR_LAUNCH;
    attn_chan_stats_kernel<<<dim3((unsigned)((sk + 255) / 256), (unsigned)kv_heads, (unsigned)batch), 256, 0, s>>>(p.v, sk, v_tok_stride, kv_heads, vst);
    FDM_CHECK_LAUNCH();
    attn_vq_kernel<<<dim3((unsigned)ntl, (unsigned)kv_heads, (unsigned)batch), 256, 0, s>>>(p.v, sk, v_tok_stride, kv_heads, ntl, vst, img, vsc, vmn);
    FDM_CHECK_LAUNCH();
    p.vq = img;
    p.vscale = vsc;
    p.vmean = vmn;
    if (qmode == 2) {
      uint8_t* kimg = reinterpret_cast<uint8_t*>(vq_ws) + fdm_attn_pvfp8_workspace_bytes(batch, sk, kv_heads, head_dim);
      AttnChanStats* kst = reinterpret_cast<AttnChanStats*>(kimg + batch * kv_heads * (int64_t)ntl * 8192);
      int* kex = reinterpret_cast<int*>(kst + batch * kv_heads);
      if (hipMemsetAsync(kst, 0, (size_t)batch * kv_heads * sizeof(AttnChanStats), s) != hipSuccess) return FDM_ERR_LAUNCH;
      attn_chan_stats_kernel<<<dim3((unsigned)((sk + 255) / 256), (unsigned)kv_heads, (unsigned)batch), 256, 0, s>>>(p.k, sk, k_tok_stride, kv_heads, kst);
      FDM_CHECK_LAUNCH();
      attn_kq_kernel<<<dim3((unsigned)ntl, (unsigned)kv_heads, (unsigned)batch), 256, 0, s>>>(p.k, sk, k_tok_stride, kv_heads, ntl, kst, kimg, kex);
      FDM_CHECK_LAUNCH();
      p.kq = kimg;
      p.kexp = kex;
      if (mask_v2) attn_fwd64_qk8_kernel<128, true><<<(unsigned)grid, 256, 0, s>>>(p);
      else attn_fwd64_qk8_kernel<128, false><<<(unsigned)grid, 256, 0, s>>>(p);
    } else if (mask_v2) {
      attn_fwd64_pv8_kernel<128, true><<<(unsigned)grid, 256, 0, s>>>(p);
    } else {
      attn_fwd64_pv8_kernel<128, false><<<(unsigned)grid, 256, 0, s>>>(p);
    }
  } else if (causal) {
    if (head_dim == 128) attn_fwd_kernel<128, 512, KT_MASK, true><<<(unsigned)grid, 512, 0, s>>>(p);
    else attn_fwd_kernel<64, 512, KT_MASK, true><<<(unsigned)grid, 512, 0, s>>>(p);
  } else if (mask_v2) {
    attn_fwd64_kernel<128, true><<<(unsigned)grid, 256, 0, s>>>(p);
  } else if (block_mask != nullptr) {
    if (head_dim == 128) attn_fwd_kernel<128, 256, KT_MASK><<<(unsigned)grid, 256, 0, s>>>(p);
    else attn_fwd_kernel<64, 256, KT_MASK><<<(unsigned)grid, 256, 0, s>>>(p);
  } else {
#if FDM_ATT_V2
    if (head_dim == 128) attn_fwd64_kernel<128, false><<<(unsigned)grid, 256, 0, s>>>(p);
#else
    if (head_dim == 128) attn_fwd_kernel<128, 512, KT_DENSE><<<(unsigned)grid, 512, 0, s>>>(p);
#endif
    else attn_fwd_kernel<64, 512, KT_MASK><<<(unsigned)grid, 512, 0, s>>>(p);    // head_dim 64: 64-key tiles measured faster
  }
  FDM_CHECK_LAUNCH();
  if (n_tail > 0) {
    const int64_t threads = (int64_t)n_tail * QB * (head_dim / 8);
    const unsigned cgrid = (unsigned)((threads + 255) / 256);
    if (head_dim == 128) attn_combine_kernel<128><<<cgrid, 256, 0, s>>>(p, n_tail);
    else attn_combine_kernel<64><<<cgrid, 256, 0, s>>>(p, n_tail);
    FDM_CHECK_LAUNCH();
  }
  return FDM_OK;
}

extern "C" int fdm_attn_fwd(void* o, const void* q, const void* k, const void* v, int64_t batch, int64_t sq, int64_t sk,
                            int q_heads, int kv_heads, int head_dim, int64_t q_tok_stride, int64_t k_tok_stride,
                            int64_t v_tok_stride, int64_t o_tok_stride, float scale, const int8_t* block_mask, int block_q,
                            int block_k, void* workspace, int64_t workspace_bytes, fdm_stream_t stream) {
  return attn_fwd_impl(o, q, k, v, batch, sq, sk, q_heads, kv_heads, head_dim, q_tok_stride, k_tok_stride, v_tok_stride, o_tok_stride,
                       scale, block_mask, block_q, block_k, workspace, workspace_bytes, stream, 0, nullptr, 0);
}

extern "C" int fdm_attn_fwd_causal(void* o, const void* q, const void* k, const void* v, int64_t batch, int64_t sq, int64_t sk,
                                   int q_heads, int kv_heads, int head_dim, int64_t q_tok_stride, int64_t k_tok_stride,
                                   int64_t v_tok_stride, int64_t o_tok_stride, float scale, fdm_stream_t stream) {
  return attn_fwd_impl(o, q, k, v, batch, sq, sk, q_heads, kv_heads, head_dim, q_tok_stride, k_tok_stride, v_tok_stride, o_tok_stride,
                       scale, nullptr, 0, 0, nullptr, 0, stream, 0, nullptr, 0, true);
}

extern "C" int fdm_attn_fwd_pvfp8(void* o, const void* q, const void* k, const void* v, int64_t batch, int64_t sq, int64_t sk,
                                  int q_heads, int kv_heads, int head_dim, int64_t q_tok_stride, int64_t k_tok_stride,
                                  int64_t v_tok_stride, int64_t o_tok_stride, float scale, void* workspace, int64_t workspace_bytes,
                                  void* vq_workspace, int64_t vq_workspace_bytes, fdm_stream_t stream) {
  return attn_fwd_impl(o, q, k, v, batch, sq, sk, q_heads, kv_heads, head_dim, q_tok_stride, k_tok_stride, v_tok_stride, o_tok_stride,
                       scale, nullptr, 0, 0, workspace, workspace_bytes, stream, 1, vq_workspace, vq_workspace_bytes);
}

extern "C" int fdm_attn_fwd_quant_masked(void* o, const void* q, const void* k, const void* v, int64_t batch, int64_t sq, int64_t sk,
                                         int q_heads, int kv_heads, int head_dim, int64_t q_tok_stride, int64_t k_tok_stride,
                                         int64_t v_tok_stride, int64_t o_tok_stride, float scale, const int8_t* block_mask, int block_q,
                                         int block_k, int mode, void* fp8_workspace, int64_t fp8_workspace_bytes, fdm_stream_t stream) {
  if (mode != 1 && mode != 2) return FDM_ERR_INVALID;
  if (block_mask == nullptr) return FDM_ERR_INVALID;
  return attn_fwd_impl(o, q, k, v, batch, sq, sk, q_heads, kv_heads, head_dim, q_tok_stride, k_tok_stride, v_tok_stride, o_tok_stride,
                       scale, block_mask, block_q, block_k, nullptr, 0, stream, mode, fp8_workspace, fp8_workspace_bytes);
}

extern "C" int fdm_attn_fwd_fp8(void* o, const void* q, const void* k, const void* v, int64_t batch, int64_t sq, int64_t sk,
                                int q_heads, int kv_heads, int head_dim, int64_t q_tok_stride, int64_t k_tok_stride,
                                int64_t v_tok_stride, int64_t o_tok_stride, float scale, void* workspace, int64_t workspace_bytes,
                                void* fp8_workspace, int64_t fp8_workspace_bytes, fdm_stream_t stream) {
  return attn_fwd_impl(o, q, k, v, batch, sq, sk, q_heads, kv_heads, head_dim, q_tok_stride, k_tok_stride, v_tok_stride, o_tok_stride,
                       scale, nullptr, 0, 0, workspace, workspace_bytes, stream, 2, fp8_workspace, fp8_workspace_bytes);
}
