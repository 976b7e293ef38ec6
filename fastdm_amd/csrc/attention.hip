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
template <int D, int NT, int KTT>
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
    {
      const int mi = __builtin_bit_cast(int, mx);
      const auto sw = __builtin_amdgcn_permlane32_swap(mi, mi, false, false);  // both halves of the query row
      mx = fmaxf(__builtin_bit_cast(float, sw[0]), __builtin_bit_cast(float, sw[1]));
    }
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
  const int t_end = is_part ? (int)((int64_t)(part + 1) * ntiles / p.split) : ntiles;
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

template <int D, bool MASKED>
__global__ void __launch_bounds__(256, 1) attn_fwd64_kernel(const AttnParams p) {
  static_assert(D == 128, "64-row kernel: head_dim 128");
  constexpr int KT = 64, NST = 4, CPR = D / 8;
  constexpr uint32_t ROWB = D * 2;
  constexpr int TILE_BYTES = KT * D * 2;   // 16 KiB: one K or V tile
  constexpr int V_RING = NST * TILE_BYTES; // smem = [K ring: 4 tiles][V ring: 4 tiles]
  constexpr int KSTEPS = D / 16, DT = D / 32;
  constexpr int QBT = 256;
  __shared__ __attribute__((aligned(1024))) uint8_t smem[2 * NST * TILE_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lq = lane & 31, h = lane >> 5;

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
  const int qb_idx = j % p.n_qblocks;
  const int b = bh / p.heads, head = bh % p.heads;
  const int kvh = head / (p.heads / p.kv_heads);
  const int64_t q0 = (int64_t)qb_idx * QBT;

  const bf16_t* qbase = p.q + ((int64_t)b * p.sq) * p.q_ts + (int64_t)head * D;
  const uint8_t* kbase = reinterpret_cast<const uint8_t*>(p.k + ((int64_t)b * p.sk) * p.k_ts + (int64_t)kvh * D);
  const uint8_t* vbase = reinterpret_cast<const uint8_t*>(p.v + ((int64_t)b * p.sk) * p.v_ts + (int64_t)kvh * D);

  const int ntiles = (int)((p.sk + KT - 1) / KT);
  const int t_tail = (p.sk % KT != 0) ? ntiles - 1 : -1;   // the one tile whose keys need masking

  // ---- block mask (MASKED): a 256-row block spans the mask rows of its four waves (two rows of a 128 x 64 block mask: waves 0-1 and
  //      waves 2-3).  The block walks the UNION of their active key tiles -- one ring, one barrier per tile -- but a wave computes only
  //      on the stretch of the list its own pair needs: the list is ordered [kept by pair 0 only | kept by both | kept by pair 1 only]
  //      (the online softmax does not care about key order), pair 0 works on positions [0, nA + nAB), pair 1 on [nA rounded down to a
  //      multiple of 4 (the ring phase), end); outside its stretch a wave only does its quarter of the tile DMAs and the barrier
  //      (`idle_rt`).  entry = tile index | (wave w's row keeps it) << (12 + w): inside its stretch a wave whose own row drops a tile
  //      (the rounded-down head of pair 1, or mask rows finer than a wave pair) still runs its MFMAs and overwrites the scores with
  //      -inf.  Compacted once into LDS by wave 0 (ballot + popcount).  The loop runs over list positions instead of tile indices.
  constexpr int MAXL = MASKED ? 4096 : 4, LPAD = 16;
  __shared__ __attribute__((aligned(16))) uint16_t slist[MAXL + LPAD];
  __shared__ int s_nlist, s_na, s_nab;
  if constexpr (MASKED) {
    if (wave == 0) {
      const int8_t* mrow[4];
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        int64_t r = (q0 + w * 64) / p.mask_bq;
        if (r > p.mask_nq - 1) r = p.mask_nq - 1;
        mrow[w] = p.mask + (((int64_t)b * p.heads + head) * p.mask_nq + r) * p.mask_nk;
      }
      auto tile_bits = [&](int tt) -> unsigned {
        unsigned bits = 0;
        if (tt < ntiles) {
          const int c = (int)(((int64_t)tt * KT) / p.mask_bk);
#pragma unroll
          for (int w = 0; w < 4; ++w) bits |= (mrow[w][c] != 0 ? 1u : 0u) << w;
        }
        return bits;
      };
      int na = 0, nab = 0, nb = 0, last_t = 0;
      for (int base = 0; base < ntiles; base += 64) {   // pass 1: class sizes
        const unsigned bits = tile_bits(base + lane);
        const bool a = (bits & 3u) != 0, bb = (bits & 12u) != 0;
        na += __popcll(__ballot(a && !bb));
        nab += __popcll(__ballot(a && bb));
        const unsigned long long bal = __ballot(bits != 0);
        nb += __popcll(__ballot(!a && bb));
        if (bal != 0) last_t = base + 63 - __builtin_clzll(bal);
      }
      int oa = 0, oab = na, ob = na + nab;
      const unsigned long long below = (1ull << lane) - 1ull;
      for (int base = 0; base < ntiles; base += 64) {   // pass 2: placement
        const int tt = base + lane;
        const unsigned bits = tile_bits(tt);
        const bool a = (bits & 3u) != 0, bb = (bits & 12u) != 0;
        const unsigned long long ba = __ballot(a && !bb), bab = __ballot(a && bb), bo = __ballot(!a && bb);
        const uint16_t ent = (uint16_t)(tt | (bits << 12));
        if (a && !bb) slist[oa + __popcll(ba & below)] = ent;
        if (a && bb) slist[oab + __popcll(bab & below)] = ent;
        if (!a && bb) slist[ob + __popcll(bo & below)] = ent;
        oa += __popcll(ba);
        oab += __popcll(bab);
        ob += __popcll(bo);
      }
      const int cnt = na + nab + nb;
      if (lane < LPAD) slist[cnt + lane] = (uint16_t)last_t;   // run-ahead DMAs past the end re-read the last tile (no wave keeps it)
      if (lane == 0) {
        s_nlist = cnt;
        s_na = na;
        s_nab = nab;
      }
    }
    __syncthreads();
  }
  int n_union = 0, t_begin_m = 0, t_end_m = 0;   // MASKED: list length, this wave's stretch of it
  if constexpr (MASKED) {
    n_union = __builtin_amdgcn_readfirstlane(s_nlist);
    const int na = __builtin_amdgcn_readfirstlane(s_na), nab = __builtin_amdgcn_readfirstlane(s_nab);
    if (!p.mask_stretch) {
      t_end_m = n_union;
    } else if (wave < 2) {
      t_end_m = na + nab;
    } else if (n_union > na) {
      t_begin_m = na & ~3;
      t_end_m = n_union;
    }
  }
  const int t_begin = MASKED ? t_begin_m : (is_part ? (int)((int64_t)part * ntiles / p.split) : 0);
  const int t_end = MASKED ? t_end_m : (is_part ? (int)((int64_t)(part + 1) * ntiles / p.split) : ntiles);
  int le[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // MASKED: list entries of positions g .. g+7 (g = the 4-step group's first position), wave-uniform
  if constexpr (MASKED) {
#pragma unroll
    for (int i = 0; i < 8; ++i) le[i] = __builtin_amdgcn_readfirstlane((int)slist[i]);
  }
  const uint32_t list_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)slist;

  // ---- K/V staging by LDS-DMA: lane-linear destination, swizzle on the source side (as in attn_fwd_kernel).  The source
  //      is a wave-uniform tile base (SALU) plus a 32-bit per-lane offset that never changes; the rows of a partial last
  //      tile are clamped through a second offset set (rows past the end re-read the last key: masked later).
  //      Ring slot of tile t = (t - t_begin) % 4: a compile-time constant in the 4x unrolled loop.
  uint32_t kofs[4], vofs[4], kofs_t[4], vofs_t[4];
  {
    const int last_rows = (int)(p.sk - (int64_t)(ntiles - 1) * KT);   // 1..64 valid rows in the last tile
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int pslot = tid + i * 256;
      const int r = pslot / CPR, pos = pslot % CPR;
      const int rc = r < last_rows ? r : last_rows - 1;
      kofs[i] = (uint32_t)(r * p.k_ts * 2 + Swz<D>::k(r, pos) * 16);
      vofs[i] = (uint32_t)(r * p.v_ts * 2 + Swz<D>::v(r, pos) * 16);
      kofs_t[i] = (uint32_t)(rc * p.k_ts * 2 + Swz<D>::k(r, pos) * 16);
      vofs_t[i] = (uint32_t)(rc * p.v_ts * 2 + Swz<D>::v(r, pos) * 16);
    }
  }
  // DMA piece i (0..3) of a K (V) tile into ring slot `slot`; src = the tile's base (tile_src, once per phase), last = it is
  // the sequence's last tile (clamped row offsets)
  auto tile_src = [&](const uint8_t* base, int64_t ts, int t, bool& last) -> const uint8_t* {
    const int tc = t < ntiles - 1 ? t : ntiles - 1;
    last = tc == ntiles - 1;
    return base + (int64_t)tc * KT * ts * 2;
  };
  auto dma_k = [&](const uint8_t* src, bool last, int slot, auto idx) {
    constexpr int I = decltype(idx)::value;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (last ? kofs_t[I] : kofs[I])),
                                     (__attribute__((address_space(3))) void*)(smem + slot * TILE_BYTES + wave * 1024 + I * 4096), 16, 0, 0);
  };
  auto dma_v = [&](const uint8_t* src, bool last, int slot, auto idx) {
    constexpr int I = decltype(idx)::value;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (last ? vofs_t[I] : vofs[I])),
                                     (__attribute__((address_space(3))) void*)(smem + V_RING + slot * TILE_BYTES + wave * 1024 + I * 4096), 16, 0, 0);
  };
  // Q first: its 16 raw loads go out back to back and in front of the 28 K / V DMAs of the prologue (loads return in order), so the
  // conversion below waits for Q alone while the tiles are still landing.  (Loading inside the conversion loop, as first written,
  // became 16 load -> s_waitcnt vmcnt(0) -> convert round trips per work item, the first of them behind all 28 DMAs.)
  i32x4 qraw[2][KSTEPS];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    int64_t qrow = q0 + wave * 64 + qb * 32 + lq;
    if (qrow > p.sq - 1) qrow = p.sq - 1;
    const bf16_t* qp = qbase + qrow * p.q_ts + 8 * h;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) qraw[qb][ks] = *reinterpret_cast<const i32x4*>(qp + 16 * ks);
  }
  static_for<4>([&](auto s_) {
    bool last;
    auto& le_ = le;
    const uint8_t* src = tile_src(kbase, p.k_ts, MASKED ? (le_[decltype(s_)::value] & 0xfff) : t_begin + decltype(s_)::value, last);
    static_for<4>([&](auto i) { dma_k(src, last, decltype(s_)::value, i); });
  });
  static_for<3>([&](auto s_) {
    bool last;
    auto& le_ = le;
    const uint8_t* src = tile_src(vbase, p.v_ts, MASKED ? (le_[decltype(s_)::value] & 0xfff) : t_begin + decltype(s_)::value, last);
    static_for<4>([&](auto i) { dma_v(src, last, decltype(s_)::value, i); });
  });

  // ---- Q fragments, pre-scaled by p.q_prescale (a power of two: exact in bf16; FDM_ATT_EXACT_Q = 0: scale * log2(e), one extra
  //      bf16 rounding of q), parked in accumulation registers
  i32x4 qa[2][KSTEPS];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      i32x4 v = qraw[qb][ks];
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const uint32_t u = (uint32_t)v[w];
        v[w] = (int)pack2bf(__builtin_bit_cast(float, u << 16) * p.q_prescale, __builtin_bit_cast(float, u & 0xffff0000u) * p.q_prescale);
      }
      asm volatile("" : "=a"(qa[qb][ks]) : "0"(v));
    }
  }

  // ---- state
  f32x16 oacc[2][DT];   // accumulation registers
  {
    f32x16 z;
#pragma unroll
    for (int e = 0; e < 16; ++e) z[e] = 0.f;
#pragma unroll
    for (int qb = 0; qb < 2; ++qb)
#pragma unroll
      for (int d = 0; d < DT; ++d) asm volatile("" : "=a"(oacc[qb][d]) : "0"(z));
  }
  f32x16 mneg[2];       // -m_ref of this lane's query row, replicated over 16 vector registers (C operand of a tile's first MFMA)
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
#pragma unroll
    for (int e = 0; e < 16; ++e) mneg[qb][e] = 0.f;
    asm volatile("" : "+v"(mneg[qb]));
  }
  float mref[2] = {0.f, 0.f}, l_run[2] = {0.f, 0.f};
#if FDM_A64_LSUM_MFMA
  f32x16 lacc[2];       // accumulation registers: every row of ones[32 x 16] . P^T[16 x 32] is the row-sum vector
  {
    f32x16 z;
#pragma unroll
    for (int e = 0; e < 16; ++e) z[e] = 0.f;
    asm volatile("" : "=a"(lacc[0]) : "0"(z));
    asm volatile("" : "=a"(lacc[1]) : "0"(z));
  }
  i32x4 ones = {0x3f803f80, 0x3f803f80, 0x3f803f80, 0x3f803f80};   // bf16 1.0 x 8
  asm volatile("" : "+v"(ones));
#endif
#ifdef FDM_ATT_STAMPS
  unsigned long long st_a = 0, st_w = 0, st_b = 0, st_d = 0, st_n = 0, st_t0;
#endif

  // ---- LDS read addresses: per-lane bases (8 for the K k-steps, 4 for the V^T d-blocks); ring slot, key block and key
  //      group are compile-time immediates of the DS instructions (no address VALU in the loop)
  const int gi = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
  const uint32_t smem_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
  uint32_t kaddr[KSTEPS], vaddr[DT];
  {
    const uint32_t kld0 = lq * ROWB + (uint32_t)Swz<D>::k(lq, h) * 16;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) kaddr[ks] = smem_base + (kld0 ^ (uint32_t)(ks << 5));
#pragma unroll
    for (int d = 0; d < DT; ++d)
      vaddr[d] = smem_base + V_RING + (4 * h + tq) * ROWB + (uint32_t)Swz<D>::v(4 * h + tq, d * 4 + 2 * (gi & 1) + (tp >> 1)) * 16 + 8 * (tp & 1);
  }
  // Fragment reads are inline asm: their lgkmcnt waits are COUNTED by hand below (the compiler's own waits drain to
  // lgkmcnt(0) in front of every MFMA), so every LDS read of the loop must be one of these two.
  i32x4 kf[4];   // K fragment f lives in kf[f % 4]: three fragments in flight
#define FDM_READ_K(SLOT, F)                                                                                       \
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(kf[(F) & 3]) : "v"(kaddr[(F) & 7]),                          \
               "i"((SLOT) * TILE_BYTES + ((F) >> 3) * 32 * (int)ROWB) : FDM_VREGS)
#define FDM_LGKM(N) asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N) : "memory", FDM_VREGS)
#define FDM_MFMA_S_FIRST(S, KF, QA, C) \
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(S) : "a"(KF), "a"(QA), "v"(C) : FDM_VREGS)
#define FDM_MFMA_S_ACC(S, KF, QA) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(S) : "a"(KF), "a"(QA) : FDM_VREGS)
#define FDM_MFMA_O(O, FSET, PF) \
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, a[%c2:%c3], %1, %0" : "+a"(O) : "v"(PF), "i"(240 + 4 * (FSET)), "i"(243 + 4 * (FSET)) : FDM_VREGS)
#define FDM_MFMA_O_PAD(O, FSET, PF) \
  asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, a[%c2:%c3], %1, %0" : "+a"(O) : "v"(PF), "i"(240 + 4 * (FSET)), "i"(243 + 4 * (FSET)) : FDM_VREGS)

  // V^T fragments: a 128-bit MFMA operand filled by TWO 64-bit transposed reads.  Inline asm cannot name the halves of a
  // tuple operand (and two 64-bit outputs are glued together with 8 v_accvgpr moves per fragment), so the four fragment
  // sets live in accumulation registers a[240:255] that are owned by the asm statements of this kernel: every asm
  // statement of the loop lists them as clobbers, which keeps the register allocator from parking anything there (audit:
  // no instruction outside ASMSTART/ASMEND may touch a240..a255 -- tools/micro/audit_attn_asm.sh).
#undef FDM_READ_V
#define FDM_VREGS "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"
#define FDM_READ_V(SLOT, F)                                                                                              \
  asm volatile("ds_read_b64_tr_b16 a[%c1:%c2], %0 offset:%5\n\tds_read_b64_tr_b16 a[%c3:%c4], %0 offset:%6"              \
               :: "v"(vaddr[(F) & 3]), "i"(240 + 4 * ((F) & 3)), "i"(241 + 4 * ((F) & 3)), "i"(242 + 4 * ((F) & 3)),       \
                  "i"(243 + 4 * ((F) & 3)),                                                                               \
                  "i"((SLOT) * TILE_BYTES + (((F) >> 3) * 32 + 16 * (((F) >> 2) & 1)) * (int)ROWB),                      \
                  "i"((SLOT) * TILE_BYTES + (((F) >> 3) * 32 + 16 * (((F) >> 2) & 1) + 8) * (int)ROWB)                   \
               : FDM_VREGS)

  // softmax pieces of a decided S tile, element n = 0..63 in the order the P.V groups consume them:
  // n -> group g = n>>4 (kb = g>>1, sx = g&1), qb = (n>>3)&1, r = n&7 -> register e = 8*sx + r of S[kb][qb]
  auto sm_exp = [&](f32x16 (&S)[2][2], auto nidx) {
    constexpr int N = decltype(nidx)::value, g = N >> 4, kb = g >> 1, sx = g & 1, qb = (N >> 3) & 1, r = N & 7, e = 8 * sx + r;
    if constexpr (FDM_A64_ABL & 4) asm volatile("" : "+v"(S[kb][qb][e]));
    else if constexpr (FDM_ATT_EXACT_Q) S[kb][qb][e] = __builtin_amdgcn_exp2f(S[kb][qb][e] * p.exp_r);
    else S[kb][qb][e] = __builtin_amdgcn_exp2f(S[kb][qb][e]);
  };
  auto sm_add = [&](f32x16 (&S)[2][2], auto nidx) {   // one gap after sm_exp(n): the transcendental's latency is not exposed
    constexpr int N = decltype(nidx)::value, g = N >> 4, kb = g >> 1, sx = g & 1, qb = (N >> 3) & 1, r = N & 7, e = 8 * sx + r;
#if FDM_A64_LSUM_MFMA || (FDM_A64_ABL & 8)
    (void)S; (void)e;
#else
    l_run[qb] += S[kb][qb][e];
    asm volatile("" : "+v"(l_run[qb]));   // pins the add inside this gap (the optimiser otherwise sinks the whole sum chain below the tile)
#endif
  };
  auto sm_pack = [&](f32x16 (&S)[2][2], i32x4 (&pf)[2][2][2], auto nidx) {   // n odd: elements n-1, n
    constexpr int N = decltype(nidx)::value, g = N >> 4, kb = g >> 1, sx = g & 1, qb = (N >> 3) & 1, r = N & 7, e = 8 * sx + r;
    if constexpr (FDM_A64_ABL & 64) pf[kb][qb][sx][r >> 1] = __builtin_bit_cast(int, S[kb][qb][e]);
    else pf[kb][qb][sx][r >> 1] = (int)pack2bf(S[kb][qb][e - 1], S[kb][qb][e]);
    asm volatile("" : "+v"(pf[kb][qb][sx][r >> 1]));
  };
  // softmax slot q of a step (q = 0..63): exp of element q, add of element q-1, pack of the pair ending at q-2
  auto sm_slot = [&](f32x16 (&S)[2][2], i32x4 (&pf)[2][2][2], auto qidx) {
    constexpr int Q = decltype(qidx)::value;
    if constexpr (Q < 64) sm_exp(S, ic<(Q < 64 ? Q : 0)>{});
    if constexpr (Q >= 1 && Q <= 64) sm_add(S, ic<(Q >= 1 && Q <= 64 ? Q - 1 : 0)>{});
    if constexpr (Q >= 3 && Q <= 65 && ((Q - 2) & 1)) sm_pack(S, pf, ic<(Q >= 3 && Q <= 65 ? Q - 2 : 1)>{});
  };

  // Fix-up path (rare): move the row references.  first: the references are not set yet (O = l = 0).
  auto fixup = [&](f32x16 (&S)[2][2], const float (&mx)[2], bool first) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      float delta = __builtin_ceilf(mx[qb] + 7.0f);              // integer: every rescale below is a power of two (exact)
      if (!first) delta = fmaxf(delta, 0.f);
      if (!(mx[qb] > -3.0e38f)) delta = 0.f;                     // a fully masked row keeps its reference
      const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(FDM_ATT_EXACT_Q ? -delta * p.exp_r : -delta);
      mref[qb] += delta;
      l_run[qb] *= alpha;
#if FDM_A64_LSUM_MFMA
      if (!first) {
        f32x16 v;
        asm volatile("s_nop 7\n\ts_nop 7" : "=v"(v) : "0"(lacc[qb]));
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] *= alpha;
        asm volatile("" : "=a"(lacc[qb]) : "0"(v));
      }
#endif
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int e = 0; e < 16; ++e) S[kb][qb][e] -= delta;
      if (!first) {
#pragma unroll
        for (int d = 0; d < DT; ++d) {
          f32x16 v;
          asm volatile("s_nop 7\n\ts_nop 7" : "=v"(v) : "0"(oacc[qb][d]));   // MFMA result -> VALU read: pad (nothing inside asm is padded)
#pragma unroll
          for (int e = 0; e < 16; ++e) v[e] *= alpha;
          asm volatile("" : "=a"(oacc[qb][d]) : "0"(v));
        }
      }
      f32x16 mn;
#pragma unroll
      for (int e = 0; e < 16; ++e) mn[e] = -mref[qb];
      asm volatile("" : "=v"(mneg[qb]) : "0"(mn));               // keep 16 live registers: an MFMA's C operand cannot broadcast one
    }
  };
  // MASKED: a wave whose mask row drops the first tiles of the block's union list has no row reference yet (it stays 0); its first
  // LIVE tile must set it the way the first tile of a dense row does (delta may be negative), or rows whose kept logits all
  // lie below ~ -126 (log2 domain) underflow to P = 0 and an all-zero output.  Wave-uniform.
  bool ref_set = !MASKED;
  // decision for tile t: tail masking and the wave-wide vote on the row maxima mx (of S' = scores - reference)
  auto decide = [&](f32x16 (&S)[2][2], float (&mx)[2], int t, bool first, bool have_mx, bool dead) {
    if (dead) {   // MASKED: this wave's mask row drops the tile
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
          for (int e = 0; e < 16; ++e) S[kb][qb][e] = -INFINITY;
      mx[0] = mx[1] = -INFINITY;
      if (first) fixup(S, mx, true);   // a row without any key yet keeps reference 0
      return;
    }
    const bool tail = (t == ntiles - 1) && (p.sk % KT != 0);
    if (tail) {
      const int rem = (int)(p.sk - (int64_t)t * KT) - 4 * h;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
          for (int e = 0; e < 16; ++e)
            if (kb * 32 + (e & 3) + 8 * (e >> 2) >= rem) S[kb][qb][e] = -INFINITY;
    }
    if (tail || !have_mx) {
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        float m = fmaxf(S[0][qb][0], S[1][qb][0]);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int e = 1; e < 16; ++e) m = __builtin_amdgcn_fmed3f(m, S[kb][qb][e], __builtin_inff());
        const int mi = __builtin_bit_cast(int, m);   // the other half of the query row
        const auto sw = __builtin_amdgcn_permlane32_swap(mi, mi, false, false);
        mx[qb] = fmaxf(__builtin_bit_cast(float, sw[0]), __builtin_bit_cast(float, sw[1]));
      }
    }
    if constexpr (MASKED) {
      const bool f = first || !ref_set;
      if (f || tail || __any(fmaxf(mx[0], mx[1]) >= 1.0f)) fixup(S, mx, f);
      ref_set = true;
    } else {
      if (first || tail || __any(fmaxf(mx[0], mx[1]) >= 1.0f)) fixup(S, mx, first);
    }
  };

  // One pipeline step for tile t (relative index I = (t - t_begin) % 4 fixes every ring slot at compile time):
  //   phase A   32 QK^T MFMAs of tile t+1 (-> Sn)  | exp2 / sums / packs 0..41 of tile t (Sc) | K DMAs of tile t+4
  //   barrier   (one per tile) behind a counted vmcnt: K(t+2), V(t+1) and older have landed; newer DMAs stay in flight
  //   phase B   32 P.V MFMAs of tile t             | softmax 42..63 | row maxima of Sn | V DMAs of tile t+3
  // Fragment reads run three fragments ahead and cross the phase boundaries (V^T 0..2 are read under the last QK^T
  // MFMAs, the next tile's K 0..2 under the last P.V MFMAs), so no phase starts on an exposed LDS latency.
  auto step = [&](f32x16 (&Sc)[2][2], f32x16 (&Sn)[2][2], int t, auto iidx) {
    constexpr int I = decltype(iidx)::value;
    constexpr int KS1 = (I + 1) & 3, KS2 = (I + 2) & 3, VS0 = I & 3;   // ring slots: K(t+1), K(t+2), V(t)
    i32x4 pf[2][2][2];   // [kb][qb][sx]
#ifdef FDM_ATT_STAMPS
    unsigned long long ts0, ts1, ts2, ts3, ts4;
#endif
    FDM_STAMP(ts0);
    bool klast, vlast;
    int tk = t + 4, tv = t + 3, tn = t + 1;   // tiles: K DMA (position t+4), V DMA (t+3), the S tile of phase A (t+1)
    bool dead = false;
    if constexpr (MASKED) {
      auto& le_ = le;
      tk = le_[I + 4] & 0xfff;
      tv = le_[I + 3] & 0xfff;
      tn = le_[I + 1] & 0xfff;
      dead = ((le_[I + 1] >> (12 + wave)) & 1) == 0;
    }
    const uint8_t* ksrc = tile_src(kbase, p.k_ts, tk, klast);
    const uint8_t* vsrc = tile_src(vbase, p.v_ts, tv, vlast);
    FDM_FENCE();
    // ---- phase A
    static_for<32>([&](auto g_) {
      constexpr int G = decltype(g_)::value, F = G >> 1, kb = G >> 4, ks = (G >> 1) & 7, qb = G & 1;
      auto& qa_ = qa;
      auto& mneg_ = mneg;
      if constexpr ((G & 1) == 0) {
        if constexpr (FDM_A64_ABL & 2) {
        } else if constexpr (F + 3 < 16) FDM_READ_K(KS1, (F + 3 < 16 ? F + 3 : 0));
        else FDM_READ_V(VS0, (F + 3 - 16 >= 0 ? F + 3 - 16 : 0));
        // reads issued after fragment F's: K fragments count 1, V^T fragments 2
        // (MASKED, first step of a group: the list read issued in front of the step sits between K fragments 2 and 3)
        constexpr int newer = (F <= 12 ? 3 : (F == 13 ? 4 : (F == 14 ? 5 : 6))) + ((MASKED && I == 0 && F < 3) ? 1 : 0);
        if constexpr (!(FDM_A64_ABL & (2 | 128))) FDM_LGKM(newer);
      }
      if constexpr (ks == 0) FDM_MFMA_S_FIRST(Sn[kb][qb], kf[F & 3], qa_[qb][ks], mneg_[qb]);
      else FDM_MFMA_S_ACC(Sn[kb][qb], kf[F & 3], qa_[qb][ks]);
#if FDM_A64_BAL
      // softmax slots 0..41: one in every even gap (which also carries a fragment read + wait), 26 in the 16 odd gaps
      constexpr int PR = G >> 1;                               // gap pair
      constexpr int o0 = (PR * 26) / 16, o1 = ((PR + 1) * 26) / 16;   // odd-gap slots of this pair: 1 or 2
      constexpr int base_ = PR + o0;                           // slots before this pair
      if constexpr ((G & 1) == 0) sm_slot(Sc, pf, ic<base_>{});
      else {
        sm_slot(Sc, pf, ic<base_ + 1>{});
        if constexpr (o1 - o0 == 2) sm_slot(Sc, pf, ic<base_ + 2>{});
      }
      if constexpr ((G & 7) == 5 && !(FDM_A64_ABL & 1)) dma_k(ksrc, klast, I & 3, ic<(G >> 3)>{});
#else
      // softmax slots 0..41 over the 32 gaps (21 per 16 gaps)
      constexpr int q0_ = (G * 21) / 16, q1_ = ((G + 1) * 21) / 16;
      sm_slot(Sc, pf, ic<q0_>{});
      if constexpr (q1_ - q0_ == 2) sm_slot(Sc, pf, ic<q0_ + 1>{});
      if constexpr ((G & 7) == 4 && !(FDM_A64_ABL & 1)) dma_k(ksrc, klast, I & 3, ic<(G >> 3)>{});
#endif
      FDM_FENCE();
    });
    FDM_STAMP(ts1);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // K(t+2), V(t+1) and older have landed; K(t+3), V(t+2), K(t+4) stay in flight
    if constexpr (!(FDM_A64_ABL & 32)) __builtin_amdgcn_s_barrier();
    FDM_FENCE();
    FDM_STAMP(ts2);
    // ---- phase B
    float mxa[2][2], mx[2];
    unsigned long long vote = 0;
    static_for<32>([&](auto g_) {
      constexpr int G = decltype(g_)::value, F = G >> 1, grp = G >> 3, kb = grp >> 1, sx = grp & 1, d = (G >> 1) & 3, qb = G & 1;
      auto& oacc_ = oacc;
      if constexpr ((G & 1) == 0) {
        if constexpr (FDM_A64_ABL & 2) {
        } else if constexpr (F + 3 < 16) FDM_READ_V(VS0, (F + 3 < 16 ? F + 3 : 0));
        else FDM_READ_K(KS2, (F + 3 - 16 >= 0 ? F + 3 - 16 : 0));
        constexpr int newer = F <= 12 ? 6 : (F == 13 ? 5 : (F == 14 ? 4 : 3));
        if constexpr (!(FDM_A64_ABL & (2 | 128))) FDM_LGKM(newer);
      }
      if constexpr ((G & 7) == 0) FDM_MFMA_O_PAD(oacc_[qb][d], F & 3, pf[kb][qb][sx]);
      else FDM_MFMA_O(oacc_[qb][d], F & 3, pf[kb][qb][sx]);
#if FDM_A64_LSUM_MFMA
      if constexpr ((G & 7) == 7) {   // the group's P fragments once more: row sums
        auto& lacc_ = lacc;
        FDM_FENCE();
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(lacc_[0]) : "v"(ones), "v"(pf[kb][0][sx]) : FDM_VREGS);
        FDM_FENCE();
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(lacc_[1]) : "v"(ones), "v"(pf[kb][1][sx]) : FDM_VREGS);
      }
#endif
      if constexpr (G < 24) sm_slot(Sc, pf, ic<(G < 24 ? 42 + G : 0)>{});   // slots 42..65 (the last two only add / pack): group 3's last pack is
                                                                             // out at gap 23, its first P.V MFMA issues at gap 24
#if FDM_A64_BAL
      if constexpr (G < 16 && (G & 1) && !(FDM_A64_ABL & 16)) {
        // row maxima of Sn (written in phase A) in the odd gaps: one chain per (q block qq, key block c), 2 gaps of 8
        // elements each; med3(a, b, +inf) = max(a, b) pairs fold into v_max3_f32
        constexpr int qq = (G >> 3) & 1, c = (G >> 2) & 1, q2 = (G >> 1) & 1;
        float m = q2 == 0 ? Sn[c][qq][0] : mxa[qq][c];
#pragma unroll
        for (int e = (q2 == 0 ? 1 : 8); e < 8 * q2 + 8; ++e) m = __builtin_amdgcn_fmed3f(m, Sn[c][qq][e], __builtin_inff());
        mxa[qq][c] = m;
        asm volatile("" : "+v"(mxa[qq][c]));   // pins the max chain inside this gap
      }
#else
      if constexpr (G < 16 && !(FDM_A64_ABL & 16)) {
        // row maxima of Sn (written in phase A): one chain per (q block qq, key block c), 4 gaps of 4 elements each;
        // med3(a, b, +inf) = max(a, b) pairs fold into v_max3_f32
        constexpr int qq = (G >> 3) & 1, c = (G >> 2) & 1, q4 = G & 3;
        float m = q4 == 0 ? Sn[c][qq][0] : mxa[qq][c];
#pragma unroll
        for (int e = (q4 == 0 ? 1 : 4 * q4); e < 4 * q4 + 4; ++e) m = __builtin_amdgcn_fmed3f(m, Sn[c][qq][e], __builtin_inff());
        mxa[qq][c] = m;
        asm volatile("" : "+v"(mxa[qq][c]));   // pins the max chain inside this gap
      }
#endif
      if constexpr (G == 16 && (FDM_A64_ABL & 16)) mx[0] = mx[1] = 0.f;
      if constexpr (G == 16 && !(FDM_A64_ABL & 16)) {
        mx[0] = fmaxf(mxa[0][0], mxa[0][1]);
        mx[1] = fmaxf(mxa[1][0], mxa[1][1]);
        asm volatile("" : "+v"(mx[0]), "+v"(mx[1]));
      }
      if constexpr (G == 17 || G == 18) {   // the other half of the query row
        constexpr int qq = G - 17;
        const int mi = __builtin_bit_cast(int, mx[qq]);
        const auto sw = __builtin_amdgcn_permlane32_swap(mi, mi, false, false);
        mx[qq] = fmaxf(__builtin_bit_cast(float, sw[0]), __builtin_bit_cast(float, sw[1]));
        asm volatile("" : "+v"(mx[qq]));
      }
      if constexpr (G == 20) {
        vote = __builtin_amdgcn_ballot_w64(fmaxf(mx[0], mx[1]) >= 1.0f);
        asm volatile("" : "+s"(vote));
      }
      if constexpr (G >= 24 && (G & 1) == FDM_A64_BAL && !(FDM_A64_ABL & 1)) dma_v(vsrc, vlast, (I + 3) & 3, ic<((G - 24) >> 1 < 4 && G >= 24 ? (G - 24) >> 1 : 0)>{});
      FDM_FENCE();
    });
    FDM_STAMP(ts3);
    // MFMA result -> VALU read of Sn: the ordering asm keeps every reader of Sn below the P.V MFMAs above
    asm volatile("" : "+v"(Sn[0][0]), "+v"(Sn[0][1]), "+v"(Sn[1][0]), "+v"(Sn[1][1]));
    // fix-up path only if some row's S' reached 1 (vote), or tile t+1 is the sequence's masked last tile
    if (t + 1 < t_end && (vote != 0 || tn == t_tail || dead || (MASKED && !ref_set))) decide(Sn, mx, tn, false, true, dead);
    FDM_STAMP(ts4);
#ifdef FDM_ATT_STAMPS
    st_a += ts1 - ts0; st_w += ts2 - ts1; st_b += ts3 - ts2; st_d += ts4 - ts3; st_n += 1;
#endif
  };

  // The same step for a wave that has nothing to compute at list position `pos` (dense ablation: tile pos): its quarter of the K / V
  // DMAs and the tile's barrier, nothing else.  The ring protocol -- 4 K DMAs of position pos + 4 in front of the counted wait, 4 V DMAs
  // of position pos + 3 behind the barrier, slot = position % 4 -- is the one `step` keeps; the slot is a run-time value here.
  auto idle_k = [&](int pos) {
    bool klast;
    int tk = pos + 4;
    if constexpr (MASKED) tk = __builtin_amdgcn_readfirstlane((int)slist[pos + 4]) & 0xfff;
    const uint8_t* ksrc = tile_src(kbase, p.k_ts, tk, klast);
    static_for<4>([&](auto i) { dma_k(ksrc, klast, pos & 3, i); });
  };
  auto idle_v = [&](int pos) {
    bool vlast;
    int tv = pos + 3;
    if constexpr (MASKED) tv = __builtin_amdgcn_readfirstlane((int)slist[pos + 3]) & 0xfff;
    const uint8_t* vsrc = tile_src(vbase, p.v_ts, tv, vlast);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    FDM_FENCE();
    static_for<4>([&](auto i) { dma_v(vsrc, vlast, (pos + 3) & 3, i); });
    FDM_FENCE();
  };

#ifdef FDM_ATT_STAMPS
  FDM_STAMP(st_t0);
  const unsigned long long st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  f32x16 Sa[2][2], Sb[2][2];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // Q and the prologue tiles have landed
  __builtin_amdgcn_s_barrier();
  FDM_FENCE();
  // MASKED, a wave whose stretch starts late (t_begin = 4k > 0): ring duty for positions 0 .. t_begin - 2, then position t_begin - 1
  // split around this wave's FIRST S tile -- K(t_begin) has landed since the barrier of position t_begin - 2 and its slot (0) is not
  // refilled before position t_begin's own K DMAs, so the first S tile is computed in front of position t_begin - 1's barrier (while the
  // other pair runs that position's phase A), the next tile's first K fragments behind it (K(t_begin + 1) lands with that barrier).
  const bool late = MASKED && t_begin > 0;
  if (late) {
    __builtin_amdgcn_s_barrier();   // the other pair's "first K tile read" barrier below
    for (int pos = 0; pos < t_begin - 1; ++pos) {
      idle_k(pos);
      idle_v(pos);
    }
    idle_k(t_begin - 1);
#pragma unroll
    for (int i = 0; i < 8; ++i) le[i] = __builtin_amdgcn_readfirstlane((int)slist[t_begin + i]);
    FDM_FENCE();
  }
  if (t_begin < t_end) {
    // S tile of the first KV tile (ring slot 0), plain fragment reads
    static_for<16>([&](auto f_) {
      constexpr int F = decltype(f_)::value, kb = F >> 3, ks = F & 7;
      auto& qa_ = qa;
      auto& mneg_ = mneg;
      FDM_READ_K(0, F);
      FDM_LGKM(0);
      if constexpr (ks == 0) {
        FDM_MFMA_S_FIRST(Sa[kb][0], kf[F & 3], qa_[0][ks], mneg_[0]);
        FDM_MFMA_S_FIRST(Sa[kb][1], kf[F & 3], qa_[1][ks], mneg_[1]);
      } else {
        FDM_MFMA_S_ACC(Sa[kb][0], kf[F & 3], qa_[0][ks]);
        FDM_MFMA_S_ACC(Sa[kb][1], kf[F & 3], qa_[1][ks]);
      }
      FDM_FENCE();
    });
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(Sa[0][0]), "+v"(Sa[0][1]), "+v"(Sa[1][0]), "+v"(Sa[1][1]));
    float mx[2];
    if constexpr (MASKED) decide(Sa, mx, le[0] & 0xfff, true, false, ((le[0] >> (12 + wave)) & 1) == 0);
    else decide(Sa, mx, t_begin, true, false, false);
    FDM_FENCE();
  }
  if (late) idle_v(t_begin - 1);
  if (t_begin < t_end) {   // the next tile's first three K fragments
    FDM_READ_K(1, 0);
    FDM_READ_K(1, 1);
    FDM_READ_K(1, 2);
    FDM_FENCE();
  }
  if (!late) __builtin_amdgcn_s_barrier();   // every wave has read the first K tile: step 0 refills its ring slot
  FDM_FENCE();
  if ((FDM_A64_ABL & 256) && wave >= 2) {   // ablation: half of the block's waves only keep the ring going
    for (int t = t_begin; t < t_end; ++t) {
      idle_k(t);
      idle_v(t);
    }
  } else
  for (int t = t_begin; t < t_end; t += 4) {
    i32x2 nxt;
    if constexpr (MASKED) {   // list entries of the NEXT group's positions t+8 .. t+11, consumed behind step 3 (dozens of counted waits later)
      asm volatile("ds_read_b64 %0, %1" : "=v"(nxt) : "v"(list_base + 2u * (uint32_t)(t + 8)) : FDM_VREGS);
      FDM_FENCE();
    }
    step(Sa, Sb, t, ic<0>{});
    if (t + 1 < t_end) step(Sb, Sa, t + 1, ic<1>{});
    if (t + 2 < t_end) step(Sa, Sb, t + 2, ic<2>{});
    if (t + 3 < t_end) step(Sb, Sa, t + 3, ic<3>{});
    if constexpr (MASKED) {
      int s0, s1;
      asm volatile("v_readfirstlane_b32 %0, %2\n\tv_readfirstlane_b32 %1, %3" : "=s"(s0), "=s"(s1) : "v"(nxt[0]), "v"(nxt[1]) : FDM_VREGS);
      le[0] = le[4]; le[1] = le[5]; le[2] = le[6]; le[3] = le[7];
      le[4] = s0 & 0xffff; le[5] = (int)((unsigned)s0 >> 16); le[6] = s1 & 0xffff; le[7] = (int)((unsigned)s1 >> 16);
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // run-ahead DMAs / fragment reads must not outlive the block
#ifdef FDM_ATT_STAMPS
  {
    unsigned long long st_t1;
    FDM_STAMP(st_t1);
    if (tid == 0 && blockIdx.x < 4096) {
      unsigned long long* o_ = g_att_stamps + (size_t)blockIdx.x * 8;
      o_[0] = st_a; o_[1] = st_w; o_[2] = st_b; o_[3] = st_d; o_[4] = st_n; o_[5] = st_t1 - st_t0;
      o_[6] = __builtin_amdgcn_s_memrealtime() - st_r0;   // 100 MHz ticks
    }
  }
#endif

  // ---- finalize
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
#if FDM_A64_LSUM_MFMA
    float l_tot;
    {
      f32x16 lv;
      asm volatile("s_nop 7\n\ts_nop 7" : "=v"(lv) : "0"(lacc[qb]));
      l_tot = lv[0];   // every row of the tile is the row-sum vector: this lane's column = its query row, both halves included
    }
#else
    const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
#endif
    const int row = wave * 64 + qb * 32 + lq;
    f32x16 o[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) asm volatile("s_nop 7\n\ts_nop 7" : "=v"(o[d]) : "0"(oacc[qb][d]));
    if (is_part) {
      const int64_t slot = (int64_t)(blockIdx.x - p.n_full);
      float* wo = p.ws_o + (slot * QBT + row) * D;
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          f32x4 w;
          w[0] = o[d][4 * g4]; w[1] = o[d][4 * g4 + 1]; w[2] = o[d][4 * g4 + 2]; w[3] = o[d][4 * g4 + 3];
          *reinterpret_cast<f32x4*>(wo + 32 * d + 8 * g4 + 4 * h) = w;
        }
      if (h == 0) {
        p.ws_ml[(slot * QBT + row) * 2] = t_begin < t_end ? (FDM_ATT_EXACT_Q ? mref[qb] * p.exp_r : mref[qb]) : -1.0e30f;
        p.ws_ml[(slot * QBT + row) * 2 + 1] = l_tot;
      }
      continue;
    }
    const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
    const int64_t qrow = q0 + row;
    if (qrow < p.sq) {
      bf16_t* op = p.o + ((int64_t)b * p.sq + qrow) * p.o_ts + (int64_t)head * D;
#pragma unroll
      for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          i32x2 w;
          w[0] = (int)pack2bf(o[d][4 * g4] * inv, o[d][4 * g4 + 1] * inv);
          w[1] = (int)pack2bf(o[d][4 * g4 + 2] * inv, o[d][4 * g4 + 3] * inv);
          *reinterpret_cast<i32x2*>(op + 32 * d + 8 * g4 + 4 * h) = w;
        }
    }
  }
  if constexpr (MASKED) {
    // a wave whose stretch ended early has written its rows; the ring duty for the rest of the list remains
    if (t_end < n_union) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the output stores are out: the counted waits of idle_v see DMAs only
      for (int pos = t_end; pos < n_union; ++pos) {
        idle_k(pos);
        idle_v(pos);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
#undef FDM_READ_K
#undef FDM_READ_V
#undef FDM_LGKM
#undef FDM_MFMA_S_FIRST
#undef FDM_MFMA_S_ACC
#undef FDM_MFMA_O
#undef FDM_MFMA_O_PAD
#undef FDM_VREGS
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
  i32x4 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) o[i] = (int)pack2bf(acc[2 * i] * inv, acc[2 * i + 1] * inv);
  *reinterpret_cast<i32x4*>(p.o + ((int64_t)b * p.sq + qrow) * p.o_ts + (int64_t)head * D + ch * 8) = o;
}

}  // namespace fdm

using namespace fdm;

extern "C" int64_t fdm_attn_fwd_workspace_bytes(int head_dim) {
  // 1024 part slots x 256 rows x (D + 2) floats (136 MB at D = 128); fdm_attn_fwd only picks splits that fit
  return (int64_t)1024 * QB * (head_dim + 2) * 4;
}

extern "C" int fdm_attn_fwd(void* o, const void* q, const void* k, const void* v, int64_t batch, int64_t sq, int64_t sk,
                            int q_heads, int kv_heads, int head_dim, int64_t q_tok_stride, int64_t k_tok_stride,
                            int64_t v_tok_stride, int64_t o_tok_stride, float scale, const int8_t* block_mask, int block_q,
                            int block_k, void* workspace, int64_t workspace_bytes, fdm_stream_t stream) {
  (void)hipGetLastError();  // drop any stale error of an earlier, unrelated runtime call: FDM_CHECK_LAUNCH must see only ours
  if (!o || !q || !k || !v || batch <= 0 || sq <= 0 || sk <= 0 || q_heads <= 0 || kv_heads <= 0 || q_heads % kv_heads)
    return FDM_ERR_INVALID;
  if (head_dim != 64 && head_dim != 128) return FDM_ERR_UNSUPPORTED;
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
    static const int stretch = [] { const char* e = getenv("FDM_ATT_STRETCH"); return (e != nullptr && e[0] == '0') ? 0 : 1; }();
    p.mask_stretch = stretch;
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
  p.q_prescale = exp2f(rintf(log2f(p.scale_log2e)));
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
  if (block_mask == nullptr && workspace != nullptr && (bh % 8) == 0) {
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
  if (mask_v2) {
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
