#!/usr/bin/env python3
"""Generator of fdm::gemm_p4_kernel's body: the PERSISTENT fp8 W8A8 tile loop of gemm.hip, written as gfx950 assembly.

    python3 gen_gemm_p4.py            # writes gemm_p4_body.inc next to this file (build.sh runs it before hipcc)

Why assembly (profiles/r03_experiments.md "Persistent gemm_w4_kernel"): gemm_w4_kernel spends 25 % of a K = 3072 tile
outside its K loop (prologue 5-6 k, accumulators -> bf16 -> LDS 6-9 k, row stores 2-5 k, drain 1-1.5 k of 74 k cycles) and a lone
wave per SIMD has nobody to cover it.  A persistent block that drains tile i under tile i + 1's first K step needs ~250 live vector
registers with exact live ranges; the C++ attempt of round 3 lost to hipcc's spills.  Here every register is assigned by hand.

Same tile, same MFMAs, same K order, same epilogue arithmetic as gemm_w4_kernel => outputs are BIT-IDENTICAL to it (tools/micro/
gemm_bench.hip -DFDM_GEMM_AB, tests/test_ops_gpu.py::test_gemm_p4_*).

Structure of one block (256 threads = 4 waves, ONE per SIMD, wave tile 128 x 128, 16x16x128 fp8 MFMAs in a[0:255]):
  * tiles t = first, first + 32, ... of the block's XCD range (the XCD-contiguous raster of gemm_kernel);
  * ONE continuous K-step stream across tile seams: the LDS ring (2 stages x (A 32 KiB + B 32 KiB)) never drains; the DMA source
    pointers switch to the next tile 2 (A) / 3 (B) steps before the seam -- no per-tile prologue after the block's first tile;
  * a tile's first K step issues its MFMAs with C = 0; the previous tile's accumulators are read out row block by row block
    (16 rows x 128 columns: 32 values per lane) just ahead of the MFMA group that overwrites them: acc -> * sa * sb + bias ->
    (GELU) -> bf16 -> 8-byte pieces into an 8 KiB per-wave LDS strip beside the ring -> 16-byte row-contiguous pieces -> global
    stores, all interleaved MFMA by MFMA into the step (drain of row block 0 rides on the last K step of its own tile);
  * per-tile epilogue operands (sa rows, sb / bias columns) are requested in the tile's last K step and sit in 72 dedicated VGPRs.

The emitted text is the body of ONE `asm volatile` statement of gemm.hip (operands: %0 kernel-argument pointer, %1 blockIdx.x,
%2 threadIdx.x, %3 LDS base of the block's 160 KiB, %4 address of a zero bias vector); it clobbers v8-v255, a0-a255, s8-s99 without s32, vcc.
"""
import os
import sys

EPI_NONE, EPI_GELU_TANH, EPI_GELU_ERF = 0, 1, 2

# ------------------------------------------------------------------------------------------------ kernel-argument layout (P4Args)
ARG_P0, ARG_P1, ARG_CFG = 0, 64, 128            # two problems' pointers (A B C sa | sb bias gate res), then 12 dwords of shape
STAGE = 65536
B_HALF = 32768
STRIP0 = 2 * STAGE                               # 32 KiB of per-wave epilogue strips behind the ring


class Gen:
    def __init__(self, epi):
        self.epi = epi
        self.out = []
        self.s_next = 8
        self.s_used = set()
        self.sreg = {}
        self.label_n = 0
        self.packed_bias = epi != EPI_NONE        # GELU variants keep the bias packed (2 bf16 per register): their temporaries need the room
        # weight touch: the plain body (every launch of the models since the GELUs moved into the row quantiser); P4_TOUCH=0 builds it without (A/B)
        self.touch = epi == EPI_NONE and not (os.environ.get("P4_OUT") and (os.environ.get("P4_SCALED") or os.environ.get("P4_TOUCH") == "0"))

    # ---- text
    def e(self, line):
        self.out.append(line)

    def c(self, text):
        self.out.append("; " + text)

    def label(self, stem):
        self.label_n += 1
        return f".Lp4_{stem}_{self.label_n}_%="

    # ---- SGPR allocation: first fit in s8 .. s99 without s32.  hipcc holds s32 (the stack pointer of its calling convention) and
    # s100 / s101 (the kernel uses VCC, which the hardware places behind the last allocated SGPR: 102 - 2) as RESERVED and warns that an asm
    # statement clobbering them "may lead to undefined behaviour" -- the body names none of them, so the clobber list does not either.
    S_FIRST, S_END, S_AVOID = 8, 100, (32,)

    def s(self, name, n=1, align=None):
        if name not in self.sreg:
            al = align or (4 if n >= 4 else n)
            b = (self.S_FIRST + al - 1) // al * al
            while any((r in self.s_used or r in self.S_AVOID) for r in range(b, b + n)):
                b += al
            assert b + n <= self.S_END, f"out of SGPRs at {name}"
            self.sreg[name] = (b, n)
            self.s_used.update(range(b, b + n))
            self.s_next = max(self.s_next, b + n)
        b, n_ = self.sreg[name]
        return f"s{b}" if n_ == 1 else f"s[{b}:{b + n_ - 1}]"

    def s1(self, name, i):        # element i of a tuple
        b, _ = self.sreg[name]
        return f"s{b + i}"

    def s2(self, name, i=0):      # 64-bit pair starting at element i
        b, _ = self.sreg[name]
        return f"s[{b + i}:{b + i + 1}]"


# ------------------------------------------------------------------------------------------------ VGPR map (fixed; 256 of 256)
V_T = [0, 1, 2, 3, 4, 5, 6]           # temporaries
V_ONE = 7                              # 0x7f7f7f7f: unit E8M0 scales of the MFMA
V_OFFA, V_OFFB = 8, 9                  # DMA lane offsets
V_ALO = [10, 12]                       # A fragment read bases, stage 0 / 1 (lo half; +1 = hi half)
V_BLO = [14, 16]
V_STRIPW, V_STRIPR, V_STOREOFF = 18, 19, 20
V_SAOFF, V_SBOFF, V_BIASOFF = 21, 22, 23
# cooperative weight touch (plain body only, round 6 -- see touch_loads): v7 is V_ONE's slot (the non-scaled MFMA encoding does not read it);
# v23 is V_BIASOFF's (= V_SBOFF >> 1, recomputed into a temporary where the plain body needs it).  The loaded dwords go to v0 / v1, never read.
V_TOUCHOFF, V_TOUCHOK = 7, 23
TOUCH_AHEAD = 3
V_SAM = 24                             # 8: sa of this lane's row in each of the 8 row blocks
V_TRAIL = 32                           # 16: a row block read back from the strip as row-contiguous pieces
V_FA = [48, 56]                        # A fragments (2 x 8)
V_FB = [64, 128]                       # B fragments of the even / odd steps (2 x 64)
V_SB = 192                             # 32: sb of this lane's 4 columns in each of the 8 column blocks
V_BIAS = 224                           # 32 (fp32, plain epilogue) or 16 (packed bf16 pairs, activation epilogues: v240..v255 are their temporaries)


def vr(b, n=1):
    return f"v{b}" if n == 1 else f"v[{b}:{b + n - 1}]"


def build(epi):
    g = Gen(epi)
    e, c, s = g.e, g.c, g.s
    T = V_T
    trail_regs = [V_TRAIL + i for i in range(16)]
    XT = [240 + i for i in range(16)] if g.packed_bias else []
    SAM = [V_SAM + i for i in range(8)]
    ONE = V_ONE

    # ================================================================================================ setup
    c("---- inputs -> fixed registers")
    KARG = s("karg", 2)
    BID, WAVE, WM, WN, LDSW = s("bid"), s("wave"), s("wm"), s("wn"), s("ldsw")
    e(f"s_mov_b64 {KARG}, %0")
    e(f"s_mov_b32 {BID}, %1")
    e(f"v_and_b32 v{T[0]}, 0xff, %2")               # tid (256 threads; the upper bits of the work-item register are not ours)
    LDS0 = s("lds0")
    e(f"s_mov_b32 {LDS0}, %3")
    P0, P1, CFG = s("p0", 12, 4), s("p1", 12, 4), s("cfg", 12, 4)
    for name, off in (("p0", ARG_P0), ("p1", ARG_P1)):
        b, _ = g.sreg[name]
        e(f"s_load_dwordx8 s[{b}:{b + 7}], {KARG}, {off}")
        e(f"s_load_dwordx4 s[{b + 8}:{b + 11}], {KARG}, {off + 32}")
    b, _ = g.sreg["cfg"]
    e(f"s_load_dwordx8 s[{b}:{b + 7}], {KARG}, {ARG_CFG}")
    e(f"s_load_dwordx4 s[{b + 8}:{b + 11}], {KARG}, {ARG_CFG + 32}")
    # problem fields: 0-1 A, 2-3 B, 4-5 C, 6-7 sa, 8-9 sb, 10-11 bias
    TM0, TM1, TN, T0T, TOTAL, NK, LDA, LDB, LDCB, PCOLS, PSTRB, STRIDE = (g.s1("cfg", i) for i in range(12))   # STRIDE = blocks per XCD = grid / 8
    e("s_waitcnt lgkmcnt(0)")
    # bias == nullptr: 32 K columns of bf16 zeros in the library's own device memory (%4)
    for nm in ("p0", "p1"):
        e(f"s_cmp_eq_u64 {g.s2(nm, 10)}, 0")
        e(f"s_cselect_b64 {g.s2(nm, 10)}, %4, {g.s2(nm, 10)}")

    c("---- lane decomposition")
    lane, fi, fq = T[1], T[2], T[3]
    e(f"v_and_b32 v{lane}, 63, v{T[0]}")
    e(f"v_lshrrev_b32 v{T[4]}, 6, v{T[0]}")
    # gfx940 / gfx950 software hazards (no compiler pads this text): a VALU write of a VGPR needs 1 wait state before v_readfirstlane /
    # v_readlane reads it (measured on MI355X: without the pad the wave id read back as the register's stale content), a VALU
    # write of an SGPR 2 before a VALU and 5 before a VMEM instruction reads it.  Other hazards this file must respect and does:
    # s_mov m0 -> LDS-DMA 1; transcendental result -> VALU 1; MFMA result -> v_accvgpr_read >= 19 (here: a whole MFMA group).
    e("s_nop 1")
    e(f"v_readfirstlane_b32 {WAVE}, v{T[4]}")
    e("s_nop 4")
    e(f"s_lshr_b32 {WM}, {WAVE}, 1")
    e(f"s_and_b32 {WN}, {WAVE}, 1")
    e(f"s_lshl_b32 {LDSW}, {WAVE}, 10")
    e(f"s_add_u32 {LDSW}, {LDSW}, {LDS0}")
    e(f"v_and_b32 v{fi}, 15, v{lane}")
    e(f"v_lshrrev_b32 v{fq}, 4, v{lane}")
    c("fragment read bases: row (w?*128 + fi) * 128 + chunk-pair swizzle; fg = bit1(fi) | bit3(fi) << 1")
    fg, f80 = T[4], T[5]
    e(f"v_bfe_u32 v{fg}, v{fi}, 1, 1")
    e(f"v_bfe_u32 v{T[6]}, v{fi}, 3, 1")
    e(f"v_lshl_or_b32 v{fg}, v{T[6]}, 1, v{fg}")
    e(f"v_xor_b32 v{f80}, v{fq}, v{fg}")
    e(f"v_lshlrev_b32 v{f80}, 1, v{f80}")
    e(f"v_and_b32 v{T[6]}, 1, v{fq}")
    e(f"v_or_b32 v{f80}, v{f80}, v{T[6]}")
    e(f"v_lshlrev_b32 v{f80}, 4, v{f80}")                       # f8_0 (bytes)
    X0, X1, X2, X3 = s("x0"), s("x1"), s("x2"), s("x3")
    # A: (wm*128 + fi)*128 + f8_0 + lds0 (+ 65536 for stage 1); hi half = lo ^ 16
    e(f"s_lshl_b32 {X0}, {WM}, 14")
    e(f"s_add_u32 {X0}, {X0}, {LDS0}")
    e(f"v_lshlrev_b32 v{T[6]}, 7, v{fi}")
    e(f"v_add_u32 v{T[6]}, v{T[6]}, v{f80}")
    e(f"v_add_u32 v{V_ALO[0]}, {X0}, v{T[6]}")
    e(f"v_xor_b32 v{V_ALO[0] + 1}, 16, v{V_ALO[0]}")
    e(f"v_add_u32 v{V_ALO[1]}, 0x10000, v{V_ALO[0]}")
    e(f"v_add_u32 v{V_ALO[1] + 1}, 0x10000, v{V_ALO[0] + 1}")
    e(f"s_lshl_b32 {X0}, {WN}, 14")
    e(f"s_add_u32 {X0}, {X0}, {LDS0}")
    e(f"s_add_u32 {X0}, {X0}, {B_HALF}")
    e(f"v_add_u32 v{V_BLO[0]}, {X0}, v{T[6]}")
    e(f"v_xor_b32 v{V_BLO[0] + 1}, 16, v{V_BLO[0]}")
    e(f"v_add_u32 v{V_BLO[1]}, 0x10000, v{V_BLO[0]}")
    e(f"v_add_u32 v{V_BLO[1] + 1}, 0x10000, v{V_BLO[0] + 1}")
    c("DMA lane offsets: row drow = wave*8 + lane/8, source chunk dchunk of slot lane%8")
    drow, dg, dch = T[4], T[5], T[6]
    e(f"v_lshrrev_b32 v{drow}, 3, v{lane}")
    e(f"s_lshl_b32 {X0}, {WAVE}, 3")
    e(f"v_add_u32 v{drow}, {X0}, v{drow}")
    e(f"v_bfe_u32 v{dg}, v{drow}, 1, 1")
    e(f"v_bfe_u32 v{dch}, v{drow}, 3, 1")
    e(f"v_lshl_or_b32 v{dg}, v{dch}, 1, v{dg}")
    e(f"v_bfe_u32 v{dch}, v{lane}, 1, 2")
    e(f"v_xor_b32 v{dch}, v{dch}, v{dg}")
    e(f"v_lshlrev_b32 v{dch}, 1, v{dch}")
    e(f"v_and_b32 v{dg}, 1, v{lane}")
    e(f"v_or_b32 v{dch}, v{dch}, v{dg}")
    e(f"v_lshlrev_b32 v{dch}, 4, v{dch}")
    e(f"v_mul_lo_u32 v{V_OFFA}, v{drow}, {LDA}")
    e(f"v_add_u32 v{V_OFFA}, v{V_OFFA}, v{dch}")
    e(f"v_mul_lo_u32 v{V_OFFB}, v{drow}, {LDB}")
    e(f"v_add_u32 v{V_OFFB}, v{V_OFFB}, v{dch}")
    c("epilogue lane constants")
    e(f"s_lshl_b32 {X0}, {WAVE}, 13")
    e(f"s_add_u32 {X0}, {X0}, {LDS0}")
    e(f"s_add_u32 {X0}, {X0}, {STRIP0}")                          # this wave's strip
    # write: strip + fi*256 + ((fi ^ (fq >> 1)) << 4) + (fq & 1) * 8
    e(f"v_lshrrev_b32 v{T[4]}, 1, v{fq}")
    e(f"v_xor_b32 v{T[4]}, v{T[4]}, v{fi}")
    e(f"v_lshlrev_b32 v{T[4]}, 4, v{T[4]}")
    e(f"v_lshl_add_u32 v{T[4]}, v{fi}, 8, v{T[4]}")
    e(f"v_and_b32 v{T[5]}, 1, v{fq}")
    e(f"v_lshl_add_u32 v{T[4]}, v{T[5]}, 3, v{T[4]}")
    e(f"v_add_u32 v{V_STRIPW}, {X0}, v{T[4]}")
    # read: strip + rsub*256 + ((ch ^ rsub) << 4), rsub = lane >> 4 (= fq), ch = lane & 15 (= fi)
    e(f"v_xor_b32 v{T[4]}, v{fi}, v{fq}")
    e(f"v_lshlrev_b32 v{T[4]}, 4, v{T[4]}")
    e(f"v_lshl_add_u32 v{T[4]}, v{fq}, 8, v{T[4]}")
    e(f"v_add_u32 v{V_STRIPR}, {X0}, v{T[4]}")
    e(f"v_mul_lo_u32 v{V_STOREOFF}, v{fq}, {LDCB}")
    e(f"v_lshl_add_u32 v{V_STOREOFF}, v{fi}, 4, v{V_STOREOFF}")
    e(f"v_lshlrev_b32 v{V_SAOFF}, 2, v{fi}")
    e(f"v_lshlrev_b32 v{V_SBOFF}, 4, v{fq}")
    if g.touch:
        c("weight touch: this block asks for rows bid * rpb + tid (tid < rpb, row < N) of every B slice; rpb = ceil(N / grid) <= 256")
        # grid = 8 * STRIDE; N = TN * 256; rpb = (N + grid - 1) / grid by repeated subtraction (SALU has no divide; <= a few hundred rounds, once)
        e(f"s_lshl_b32 {X0}, {TN}, 8")                      # N
        e(f"s_lshl_b32 {X1}, {STRIDE}, 3")                  # grid
        e(f"s_add_u32 {X2}, {X0}, {X1}")
        e(f"s_sub_u32 {X2}, {X2}, 1")                       # N + grid - 1
        e(f"s_mov_b32 {X3}, 0")
        lp, dn = g.label("rl"), g.label("rd")
        e(f"{lp}:")
        e(f"s_cmp_lt_u32 {X2}, {X1}")
        e(f"s_cbranch_scc1 {dn}")
        e(f"s_sub_u32 {X2}, {X2}, {X1}")
        e(f"s_add_u32 {X3}, {X3}, 1")
        e(f"s_branch {lp}")
        e(f"{dn}:")                                          # X3 = rpb
        e(f"s_mul_i32 {X2}, {BID}, {X3}")                    # first row of this block
        e(f"v_add_u32 v{T[4]}, {X2}, v{T[0]}")               # row = bid * rpb + tid   (T[0] still holds tid)
        e(f"v_cmp_gt_u32 vcc, {X3}, v{T[0]}")                # tid < rpb
        e(f"v_cndmask_b32 v{V_TOUCHOK}, 0, 1, vcc")
        e(f"v_cmp_gt_u32 vcc, {X0}, v{T[4]}")                # row < N
        e(f"v_cndmask_b32 v{V_TOUCHOK}, 0, v{V_TOUCHOK}, vcc")
        e(f"v_mul_lo_u32 v{V_TOUCHOFF}, v{T[4]}, {LDB}")
    else:
        e(f"v_lshlrev_b32 v{V_BIASOFF}, 3, v{fq}")
    if os.environ.get("P4_OUT") and os.environ.get("P4_SCALED"):      # unit E8M0 scales: only the scaled MFMA encoding (bring-up builds) reads them
        e(f"v_mov_b32 v{ONE}, 0x7f7f7f7f")

    A32, B32, LDC4 = s("a32"), s("b32"), s("ldc4")
    e(f"s_lshl_b32 {A32}, {LDA}, 5")
    e(f"s_lshl_b32 {B32}, {LDB}, 5")
    e(f"s_lshl_b32 {LDC4}, {LDCB}, 2")

    c("---- this block's tiles: XCD-contiguous ranges of the tile list (as gemm_kernel), tiles first, first + grid / 8, ...")
    TCUR, TEND = s("tcur"), s("tend")
    XCD, Q, R = X0, X1, X2
    e(f"s_and_b32 {XCD}, {BID}, 7")
    e(f"s_lshr_b32 {Q}, {TOTAL}, 3")
    e(f"s_and_b32 {R}, {TOTAL}, 7")
    # xs = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q  ==  xcd * q + min(xcd, r) ; count = q + (xcd < r)
    e(f"s_mul_i32 {TCUR}, {XCD}, {Q}")
    e(f"s_min_u32 {X3}, {XCD}, {R}")
    e(f"s_add_u32 {TCUR}, {TCUR}, {X3}")
    e(f"s_cmp_lt_u32 {XCD}, {R}")
    e(f"s_addc_u32 {TEND}, {TCUR}, {Q}")                         # xs + q + (xcd < r)
    e(f"s_lshr_b32 {X3}, {BID}, 3")
    e(f"s_add_u32 {TCUR}, {TCUR}, {X3}")

    # ---- tile -> (problem, m tile, n tile)
    def coords(t_in, prob, mt, nt):
        tt, tm, ig, gid, rem, gsz, mg = X0, X1, X2, X3, s("x4"), s("x5"), s("x6")
        e(f"s_cmp_ge_u32 {t_in}, {T0T}")
        e(f"s_cselect_b32 {prob}, 1, 0")
        e(f"s_cselect_b32 {tm}, {TM1}, {TM0}")
        e(f"s_cselect_b32 {tt}, {T0T}, 0")
        e(f"s_sub_u32 {tt}, {t_in}, {tt}")
        e(f"s_lshl_b32 {ig}, {TN}, 2")
        e(f"s_mov_b32 {gid}, 0")
        e(f"s_mov_b32 {rem}, {tt}")
        lp, dn = g.label("cl"), g.label("cd")
        e(f"{lp}:")
        e(f"s_cmp_ge_u32 {rem}, {ig}")
        e(f"s_cbranch_scc0 {dn}")
        e(f"s_sub_u32 {rem}, {rem}, {ig}")
        e(f"s_add_u32 {gid}, {gid}, 1")
        e(f"s_branch {lp}")
        e(f"{dn}:")
        e(f"s_lshl_b32 {gid}, {gid}, 2")                          # first_m
        e(f"s_sub_u32 {gsz}, {tm}, {gid}")
        e(f"s_min_u32 {gsz}, {gsz}, 4")
        e(f"s_mov_b32 {mg}, 0x4000")
        e(f"s_cmp_eq_u32 {gsz}, 3")
        e(f"s_cselect_b32 {mg}, 0x5556, {mg}")
        e(f"s_cmp_eq_u32 {gsz}, 2")
        e(f"s_cselect_b32 {mg}, 0x8000, {mg}")
        e(f"s_cmp_eq_u32 {gsz}, 1")
        e(f"s_cselect_b32 {mg}, 0x10000, {mg}")
        e(f"s_mul_i32 {nt}, {rem}, {mg}")
        e(f"s_lshr_b32 {nt}, {nt}, 16")
        e(f"s_mul_i32 {tt}, {nt}, {gsz}")
        e(f"s_sub_u32 {tt}, {rem}, {tt}")
        e(f"s_add_u32 {mt}, {gid}, {tt}")

    def ab_bases(prob, mt, nt, pa, pb):
        """pa / pb (SGPR pairs) = A + mt*256*lda, B + nt*256*ldb of problem `prob`"""
        e(f"s_cmp_eq_u32 {prob}, 1")
        e(f"s_cselect_b64 {pa}, {g.s2('p1', 0)}, {g.s2('p0', 0)}")
        e(f"s_cselect_b64 {pb}, {g.s2('p1', 2)}, {g.s2('p0', 2)}")
        lo_a, hi_a = pa[2:-1].split(":")
        lo_b, hi_b = pb[2:-1].split(":")
        e(f"s_mul_i32 {X0}, {mt}, {LDA}")
        e(f"s_lshl_b32 {X0}, {X0}, 8")
        e(f"s_add_u32 s{lo_a}, s{lo_a}, {X0}")
        e(f"s_addc_u32 s{hi_a}, s{hi_a}, 0")
        e(f"s_mul_i32 {X0}, {nt}, {LDB}")
        e(f"s_lshl_b32 {X0}, {X0}, 8")
        e(f"s_add_u32 s{lo_b}, s{lo_b}, {X0}")
        e(f"s_addc_u32 s{hi_b}, s{hi_b}, 0")

    def pair(name):
        b, _ = g.sreg[name]
        return f"s{b}", f"s{b + 1}"

    PROB, MT, NT = s("prob"), s("mt"), s("nt")
    NPROB, NMT, NNT, HASNEXT, HAVEPREV = s("nprob"), s("nmt"), s("nnt"), s("hasnext"), s("haveprev")
    PTRA, PTRB, NEXTA, NEXTB, STA, STB = s("ptra", 2), s("ptrb", 2), s("nexta", 2), s("nextb", 2), s("sta", 2), s("stb", 2)
    POSA, POSB, LOOP = s("posa"), s("posb"), s("loop")
    SAP, SBP, BIASP, CRUN, CWAVE = s("sap", 2), s("sbp", 2), s("biasp", 2), s("crun", 2), s("cwave", 2)
    if g.touch:
        # the kernel-argument pointer, block id, wave id and LDS base are read for the last time in the setup above: their registers carry
        # the two touch pointers (aligned pairs) and their K position from here on
        assert g.sreg["karg"][0] % 2 == 0 and g.sreg["bid"][0] % 2 == 0 and g.sreg["wave"][0] == g.sreg["bid"][0] + 1
        g.sreg["tp0"] = (g.sreg["karg"][0], 2)
        g.sreg["tp1"] = (g.sreg["bid"][0], 2)
        TP0, TP1, TPOS = g.s2("tp0"), g.s2("tp1"), LDS0
        c(f"touch pointers: K slice {3 + TOUCH_AHEAD} of each problem's B (slices 0-2 are DMAed by the prologue; 3 .. {2 + TOUCH_AHEAD} are asked for right here)")
        e(f"s_mov_b64 {TP0}, {g.s2('p0', 2)}")
        e(f"s_mov_b64 {TP1}, {g.s2('p1', 2)}")
        e(f"v_cmp_ne_u32 vcc, 0, v{V_TOUCHOK}")
        e("s_mov_b64 exec, vcc")
        for k in range(3, 3 + TOUCH_AHEAD):
            e(f"global_load_dword v{T[0]}, v{V_TOUCHOFF}, {TP0} offset:{k * 128}")
            e(f"global_load_dword v{T[1]}, v{V_TOUCHOFF}, {TP1} offset:{k * 128}")
        e("s_mov_b64 exec, -1")
        for nm in ("tp0", "tp1"):
            e(f"s_add_u32 {g.s1(nm, 0)}, {g.s1(nm, 0)}, {(3 + TOUCH_AHEAD) * 128}")
            e(f"s_addc_u32 {g.s1(nm, 1)}, {g.s1(nm, 1)}, 0")
        e(f"s_mov_b32 {TPOS}, {3 + TOUCH_AHEAD}")

    def epi_ptrs(prob, mt, nt):
        """per-tile epilogue pointers of the CURRENT tile: sa rows, sb / bias columns of this wave, and the wave's first output row"""
        m0w, n0w = X0, X1                                   # m0 + wm*128, n0 + wn*128
        e(f"s_lshl_b32 {m0w}, {mt}, 8")
        e(f"s_lshl_b32 {X2}, {WM}, 7")
        e(f"s_add_u32 {m0w}, {m0w}, {X2}")
        e(f"s_lshl_b32 {n0w}, {nt}, 8")
        e(f"s_lshl_b32 {X2}, {WN}, 7")
        e(f"s_add_u32 {n0w}, {n0w}, {X2}")
        e(f"s_cmp_eq_u32 {prob}, 1")
        e(f"s_cselect_b64 {SAP}, {g.s2('p1', 6)}, {g.s2('p0', 6)}")
        e(f"s_cselect_b64 {SBP}, {g.s2('p1', 8)}, {g.s2('p0', 8)}")
        e(f"s_cselect_b64 {BIASP}, {g.s2('p1', 10)}, {g.s2('p0', 10)}")
        e(f"s_cselect_b64 {CWAVE}, {g.s2('p1', 4)}, {g.s2('p0', 4)}")
        lo, hi = pair("sap")
        e(f"s_lshl_b32 {X2}, {m0w}, 2")
        e(f"s_add_u32 {lo}, {lo}, {X2}")
        e(f"s_addc_u32 {hi}, {hi}, 0")
        lo, hi = pair("sbp")
        e(f"s_lshl_b32 {X2}, {n0w}, 2")
        e(f"s_add_u32 {lo}, {lo}, {X2}")
        e(f"s_addc_u32 {hi}, {hi}, 0")
        lo, hi = pair("biasp")
        e(f"s_lshl_b32 {X2}, {n0w}, 1")
        e(f"s_add_u32 {lo}, {lo}, {X2}")
        e(f"s_addc_u32 {hi}, {hi}, 0")
        # C: plane = n0w / plane_cols (plane_cols > 0), column inside the plane; + m0w * ldcB
        lo, hi = pair("cwave")
        ncol = X3
        e(f"s_mov_b32 {ncol}, {n0w}")
        skip, lp = g.label("np"), g.label("pl")
        e(f"s_cmp_eq_u32 {PCOLS}, 0")
        e(f"s_cbranch_scc1 {skip}")
        e(f"{lp}:")
        e(f"s_cmp_lt_u32 {ncol}, {PCOLS}")
        e(f"s_cbranch_scc1 {skip}")
        e(f"s_sub_u32 {ncol}, {ncol}, {PCOLS}")
        e(f"s_add_u32 {lo}, {lo}, {PSTRB}")
        e(f"s_addc_u32 {hi}, {hi}, 0")
        e(f"s_branch {lp}")
        e(f"{skip}:")
        e(f"s_lshl_b32 {ncol}, {ncol}, 1")
        e(f"s_add_u32 {lo}, {lo}, {ncol}")
        e(f"s_addc_u32 {hi}, {hi}, 0")
        e(f"s_mul_hi_u32 {X2}, {m0w}, {LDCB}")
        e(f"s_mul_i32 {ncol}, {m0w}, {LDCB}")
        e(f"s_add_u32 {lo}, {lo}, {ncol}")
        e(f"s_addc_u32 {hi}, {hi}, {X2}")

    def next_tile():
        """(NPROB, NMT, NNT, HASNEXT, NEXTA, NEXTB) for tile TCUR + 32; no such tile: the current tile again (its DMAs land in stages
        nobody reads) and HASNEXT = 0"""
        TN_ = LOOP        # (the K loop's counter is dead between a tile's last step and the next tile's first: one register for both)
        no, done = g.label("nn"), g.label("nd")
        e(f"s_add_u32 {TN_}, {TCUR}, {STRIDE}")
        e(f"s_cmp_lt_u32 {TN_}, {TEND}")
        e(f"s_cbranch_scc0 {no}")
        e(f"s_mov_b32 {HASNEXT}, 1")
        coords(TN_, NPROB, NMT, NNT)
        e(f"s_branch {done}")
        e(f"{no}:")
        e(f"s_mov_b32 {HASNEXT}, 0")
        e(f"s_mov_b32 {NPROB}, {PROB}")
        e(f"s_mov_b32 {NMT}, {MT}")
        e(f"s_mov_b32 {NNT}, {NT}")
        e(f"{done}:")
        ab_bases(NPROB, NMT, NNT, NEXTA, NEXTB)

    coords(TCUR, PROB, MT, NT)
    ab_bases(PROB, MT, NT, PTRA, PTRB)
    epi_ptrs(PROB, MT, NT)

    # ================================================================================================ building blocks
    def mfma(idx, bfrag, afrag, zero):
        cc = "0" if zero else f"a[{4 * idx}:{4 * idx + 3}]"
        # the NON-scaled encoding of the same fp8 MFMA (cbsz = blgp = 0: e4m3 x e4m3): 8 bytes instead of 16, no scale operands to read.
        # Bit-identical to v_mfma_scale_... with unit E8M0 scales (0 of 80 M elements differ) and 1-2 % faster in the A/B
        # (profiles/r04_experiments.md); P4_SCALED=1 (bring-up builds) emits the scaled form.
        if os.environ.get("P4_OUT") and os.environ.get("P4_SCALED"):
            return f"v_mfma_scale_f32_16x16x128_f8f6f4 a[{4 * idx}:{4 * idx + 3}], {vr(bfrag, 8)}, {vr(afrag, 8)}, {cc}, v{ONE}, v{ONE} op_sel_hi:[0,0,0]"
        return f"v_mfma_f32_16x16x128_f8f6f4 a[{4 * idx}:{4 * idx + 3}], {vr(bfrag, 8)}, {vr(afrag, 8)}, {cc}"

    def read_frag(dst, base_lo, blk):
        return [f"ds_read_b128 {vr(dst, 4)}, v{base_lo} offset:{blk * 2048}",
                f"ds_read_b128 {vr(dst + 4, 4)}, v{base_lo + 1} offset:{blk * 2048}"]

    def dma(kind, stage, rnd):
        """one LDS-DMA of 8 rows x 128 B: kind 'a' / 'b', LDS stage, round 0..7; advances the running source pointer by 32 rows"""
        lds_off = stage * STAGE + (0 if kind == "a" else B_HALF) + rnd * 4096
        st = STA if kind == "a" else STB
        lo, hi = pair("sta" if kind == "a" else "stb")
        step = A32 if kind == "a" else B32
        voff = V_OFFA if kind == "a" else V_OFFB
        nt = " nt" if kind in os.environ.get("P4_NT", "") and os.environ.get("P4_OUT") else ""      # experiment: streaming hint on one operand
        return [f"s_add_u32 m0, {LDSW}, {lds_off}",
                "s_nop 0",
                f"global_load_lds_dwordx4 v{voff}, {st}{nt}",
                f"s_add_u32 {lo}, {lo}, {step}",
                f"s_addc_u32 {hi}, {hi}, 0"]

    def new_set():
        """group 7, in front of the first DMA of a set: running pointers <- stream pointers; stream pointers advance one K step and
        wrap into the next tile at the seam"""
        out = [f"s_mov_b64 {STA}, {PTRA}", f"s_mov_b64 {STB}, {PTRB}"]
        for ptr, pos, nxt, nm in ((PTRA, POSA, NEXTA, "ptra"), (PTRB, POSB, NEXTB, "ptrb")):
            lo, hi = pair(nm)
            out += [f"s_add_u32 {lo}, {lo}, 128", f"s_addc_u32 {hi}, {hi}, 0", f"s_add_u32 {pos}, {pos}, 1",
                    f"s_cmp_eq_u32 {pos}, {NK}", f"s_cselect_b64 {ptr}, {nxt}, {ptr}", f"s_cselect_b32 {pos}, 0, {pos}"]
        return out

    def touch_loads():
        """Cooperative weight touch, a block's FIRST tile only.  Inside the denoise step every launch streams weights that nobody has read
        since the last step (8.6 GB per step against 256 MiB of Infinity Cache), and the LDS-DMAs of a stage are issued 0.5-1 K step ahead
        of their readers: enough for an L2 / Infinity-Cache hit, not for an HBM miss -- the first tile of every CU ran at half speed
        (tools/micro/gemm_lib_ab.py, same launch warm / cold: 168 / 202 us at K = 3072 N = 9216; hipBLASLt 194 / 200).  So while the first
        tile computes, the grid's blocks TOGETHER ask for one dword of every 128-byte line of B's K slice 3 + TOUCH_AHEAD steps ahead of
        the DMAs: block b the rows b * rpb ... of both problems' B, one masked load per problem and step (a line is asked for ONCE chip-wide:
        it lands in the Infinity Cache, where every XCD finds it; per-tile touches of the tile's own slices double the L1 fill traffic and
        were 25 % SLOWER).  The dwords go to v0 / v1 and are never read; the next step's vmcnt wait retires the loads (a whole step to land)."""
        skip = g.label("ts")
        return [f"s_cmp_ge_u32 {TPOS}, {NK}", f"s_cbranch_scc1 {skip}",
                f"v_cmp_ne_u32 vcc, 0, v{V_TOUCHOK}", "s_mov_b64 exec, vcc",
                f"global_load_dword v{T[0]}, v{V_TOUCHOFF}, {TP0}", f"global_load_dword v{T[1]}, v{V_TOUCHOFF}, {TP1}",
                "s_mov_b64 exec, -1",
                f"s_add_u32 {g.s1('tp0', 0)}, {g.s1('tp0', 0)}, 128", f"s_addc_u32 {g.s1('tp0', 1)}, {g.s1('tp0', 1)}, 0",
                f"s_add_u32 {g.s1('tp1', 0)}, {g.s1('tp1', 0)}, 128", f"s_addc_u32 {g.s1('tp1', 1)}, {g.s1('tp1', 1)}, 0",
                f"s_add_u32 {TPOS}, {TPOS}, 1", f"{skip}:"]

    # ---- epilogue pieces ------------------------------------------------------------------------------------------------------
    def bias_regs(nb):
        return [V_BIAS + nb * 4 + k for k in range(4)] if not g.packed_bias else None

    def act(regs, tmps):
        """activation on 4 fp32 values in `regs` (already rounded to bf16 precision), in place -- the op order of gemm.hip's gelu16<EPI>
        (bit-identical results).  tmps: 12 temporaries.  No instruction consumes a transcendental's result in the slot right behind it
        (gfx950: 1 wait state)."""
        o = []
        if g.epi == EPI_GELU_TANH:
            # 0.5 x (1 + tanh(u)) = x - x / (exp2(x (c0 + c1 x^2)) + 1)
            ec = tmps[:4]
            o += [f"v_mul_f32 v{ec[k]}, v{regs[k]}, v{regs[k]}" for k in range(4)]
            o += [f"v_fma_f32 v{ec[k]}, v{ec[k]}, {K['c1']}, v{KV['c0']}" for k in range(4)]
            o += [f"v_mul_f32 v{ec[k]}, v{regs[k]}, v{ec[k]}" for k in range(4)]
            o += [f"v_exp_f32 v{ec[k]}, v{ec[k]}" for k in range(4)]
            o += [f"v_add_f32 v{ec[k]}, 1.0, v{ec[k]}" for k in range(4)]
            o += [f"v_rcp_f32 v{ec[k]}, v{ec[k]}" for k in range(4)]
            o += [f"v_fma_f32 v{regs[k]}, -v{regs[k]}, v{ec[k]}, v{regs[k]}" for k in range(4)]
        elif g.epi == EPI_GELU_ERF:
            # Abramowitz-Stegun 7.1.26: erf(z) = 1 - poly(t) exp(-z^2), t = 1 / (1 + p z)
            z, t, pl = tmps[0:4], tmps[4:8], tmps[8:12]
            o += [f"v_mul_f32 v{z[k]}, |v{regs[k]}|, {K['sqrt1_2']}" for k in range(4)]
            o += [f"v_fma_f32 v{t[k]}, {K['p']}, v{z[k]}, 1.0" for k in range(4)]
            o += [f"v_rcp_f32 v{t[k]}, v{t[k]}" for k in range(4)]
            o += [f"v_fma_f32 v{pl[k]}, v{t[k]}, {K['a1']}, v{KV['a2']}" for k in range(4)]
            o += [f"v_fma_f32 v{pl[k]}, v{t[k]}, v{pl[k]}, {K['a3']}" for k in range(4)]
            o += [f"v_fma_f32 v{pl[k]}, v{t[k]}, v{pl[k]}, v{KV['a4']}" for k in range(4)]
            o += [f"v_fma_f32 v{pl[k]}, v{t[k]}, v{pl[k]}, v{KV['a5']}" for k in range(4)]
            o += [f"v_mul_f32 v{pl[k]}, v{pl[k]}, v{t[k]}" for k in range(4)]
            o += [f"v_mul_f32 v{z[k]}, -v{z[k]}, v{z[k]}" for k in range(4)]
            o += [f"v_mul_f32 v{z[k]}, {K['log2e']}, v{z[k]}" for k in range(4)]
            o += [f"v_exp_f32 v{z[k]}, v{z[k]}" for k in range(4)]
            o += [f"v_and_b32 v{t[k]}, 0x80000000, v{regs[k]}" for k in range(4)]          # sign of x (t is dead)
            o += [f"v_fma_f32 v{pl[k]}, -v{pl[k]}, v{z[k]}, 1.0" for k in range(4)]      # er
            o += [f"v_xor_b32 v{pl[k]}, v{pl[k]}, v{t[k]}" for k in range(4)]             # x < 0 ? -er : er
            o += [f"v_add_f32 v{pl[k]}, 1.0, v{pl[k]}" for k in range(4)]
            o += [f"v_mul_f32 v{regs[k]}, 0.5, v{regs[k]}" for k in range(4)]
            o += [f"v_mul_f32 v{regs[k]}, v{regs[k]}, v{pl[k]}" for k in range(4)]
        return o

    def drain(mb):
        """accumulator row block mb (16 rows x 128 columns of the wave tile) -> scaled, biased, (activated), bf16 -> LDS strip half mb & 1"""
        o = []
        h = mb & 1
        for nb in range(8):
            idx = mb * 8 + nb
            t = T[0:4]
            o += [f"v_accvgpr_read_b32 v{t[k]}, a{4 * idx + k}" for k in range(4)]
            o += [f"v_mul_f32 v{t[k]}, v{t[k]}, v{SAM[mb]}" for k in range(4)]
            o += [f"v_mul_f32 v{t[k]}, v{t[k]}, v{V_SB + nb * 4 + k}" for k in range(4)]
            if g.packed_bias:
                bp = V_BIAS + nb * 2
                o += [f"v_lshlrev_b32 v{T[4]}, 16, v{bp}", f"v_and_b32 v{T[5]}, 0xffff0000, v{bp}",
                      f"v_add_f32 v{t[0]}, v{t[0]}, v{T[4]}", f"v_add_f32 v{t[1]}, v{t[1]}, v{T[5]}",
                      f"v_lshlrev_b32 v{T[4]}, 16, v{bp + 1}", f"v_and_b32 v{T[5]}, 0xffff0000, v{bp + 1}",
                      f"v_add_f32 v{t[2]}, v{t[2]}, v{T[4]}", f"v_add_f32 v{t[3]}, v{t[3]}, v{T[5]}"]
            else:
                o += [f"v_add_f32 v{t[k]}, v{t[k]}, v{V_BIAS + nb * 4 + k}" for k in range(4)]
            if g.epi != EPI_NONE:
                # the activation's input is the bf16-rounded pre-activation (gemm.hip: y = rbf(yy))
                o += [f"v_cvt_pk_bf16_f32 v{T[4]}, v{t[0]}, v{t[1]}", f"v_cvt_pk_bf16_f32 v{T[5]}, v{t[2]}, v{t[3]}",
                      f"v_lshlrev_b32 v{t[0]}, 16, v{T[4]}", f"v_and_b32 v{t[1]}, 0xffff0000, v{T[4]}",
                      f"v_lshlrev_b32 v{t[2]}, 16, v{T[5]}", f"v_and_b32 v{t[3]}, 0xffff0000, v{T[5]}"]
                o += act(t, XT)
            o += [f"v_cvt_pk_bf16_f32 v{t[0]}, v{t[0]}, v{t[1]}", f"v_cvt_pk_bf16_f32 v{t[1]}, v{t[2]}, v{t[3]}"]
            if nb == 0:
                o += [f"ds_write_b64 v{V_STRIPW}, {vr(t[0], 2)} offset:{h * 4096}"]
            else:
                o += [f"v_xor_b32 v{T[6]}, {nb * 32}, v{V_STRIPW}", f"ds_write_b64 v{T[6]}, {vr(t[0], 2)} offset:{h * 4096}"]
        return o

    def trail_reads(mb):
        """the 16 rows of strip half mb & 1 as row-contiguous 16-byte pieces: row-instruction j -> rows 4 j + lane / 16"""
        o = []
        h = mb & 1
        for j in range(4):
            if j == 0:
                o += [f"ds_read_b128 {vr(trail_regs[0], 4)}, v{V_STRIPR} offset:{h * 4096}"]
            else:
                o += [f"v_xor_b32 v{T[6]}, {j * 64}, v{V_STRIPR}",
                      f"ds_read_b128 {vr(trail_regs[4 * j], 4)}, v{T[6]} offset:{h * 4096 + j * 1024}"]
        return o

    def trail_stores():
        o = []
        lo, hi = pair("crun")
        for j in range(4):
            o += [f"global_store_dwordx4 v{V_STOREOFF}, {vr(trail_regs[4 * j], 4)}, {CRUN}",
                  f"s_add_u32 {lo}, {lo}, {LDC4}", f"s_addc_u32 {hi}, {hi}, 0"]
        return o

    def operand_loads():
        """sa rows, sb / bias columns of the CURRENT tile (24 loads, L2 hits) -> their dedicated registers"""
        o = []
        for mb in range(8):
            o += [f"global_load_dword v{SAM[mb]}, v{V_SAOFF}, {SAP} offset:{mb * 64}"]
        for nb in range(8):
            o += [f"global_load_dwordx4 {vr(V_SB + nb * 4, 4)}, v{V_SBOFF}, {SBP} offset:{nb * 64}"]
        boff = V_BIASOFF
        if g.touch:      # V_BIASOFF's register holds the touch flag: fq * 8 = V_SBOFF >> 1 into a temporary (nothing else uses v6 in this group)
            boff = T[6]
            o += [f"v_lshrrev_b32 v{boff}, 1, v{V_SBOFF}"]
        for nb in range(8):
            dst = V_BIAS + (nb * 2 if g.packed_bias else nb * 4)
            o += [f"global_load_dwordx2 {vr(dst, 2)}, v{boff}, {BIASP} offset:{nb * 32}"]
        return o

    def bias_unpack():
        if g.packed_bias:
            return []
        o = []
        for nb in range(8):
            b = V_BIAS + nb * 4
            o += [f"v_and_b32 v{b + 3}, 0xffff0000, v{b + 1}", f"v_lshlrev_b32 v{b + 2}, 16, v{b + 1}",
                  f"v_and_b32 v{b + 1}, 0xffff0000, v{b}", f"v_lshlrev_b32 v{b}, 16, v{b}"]
        return o

    def interleave(mfmas, fillers, after=None):
        """one MFMA, then its share of the filler instructions (after[q]: instructions pinned directly behind MFMA q)"""
        o = []
        n = len(mfmas)
        k = 0
        for q, m in enumerate(mfmas):
            o.append(m)
            if after and q in after:
                o += after[q]
            take = (len(fillers) * (q + 1)) // n - k
            o += fillers[k:k + take]
            k += take
        return o

    # ---- one K step -----------------------------------------------------------------------------------------------------------
    def step(par, kind):
        """par: stage parity of the step; kind: 'reg' | 'first' (C = 0) | 'first_drain' (C = 0 + the previous tile's row blocks 1..7) |
        'last' (requests the tile's epilogue operands, drains its row block 0)"""
        zero = kind in ("first", "first_drain")
        cur, nxt = par, par ^ 1
        bc, bn = V_FB[par], V_FB[par ^ 1]
        c(f"==== K step, parity {par}, {kind}")
        for grp in range(8):
            fa_cur, fa_nxt = V_FA[grp & 1], V_FA[(grp + 1) & 1]
            pre = []
            vm_after = 0
            if grp == 7:
                if kind == "first_drain":
                    pre += ["s_waitcnt vmcnt(12) lgkmcnt(0)"]   # the 12 stores of row blocks 2..4 (groups 4..6) are younger than the step's last DMA
                else:
                    pre += ["s_waitcnt vmcnt(0) lgkmcnt(0)"]
                pre += ["s_barrier"]
                if kind == "first_drain":
                    pre += trail_stores()                        # row block 5 (read back in group 6)
                pre += read_frag(V_FA[0], V_ALO[nxt], 0)
            else:
                if kind == "first_drain":
                    pre += ["s_waitcnt lgkmcnt(0)"]
                else:
                    pre += [f"s_waitcnt lgkmcnt({4 if 1 <= grp <= 4 else 0})"]
                if kind == "first_drain" and grp >= 2:
                    pre += trail_stores()                        # row block grp - 2 (read back in the previous group)
                pre += read_frag(fa_nxt, V_ALO[cur], grp + 1)
                if grp < 4:
                    pre += read_frag(bn + 8 * (2 * grp), V_BLO[nxt], 2 * grp)
                    pre += read_frag(bn + 8 * (2 * grp + 1), V_BLO[nxt], 2 * grp + 1)
            if kind == "first_drain" and grp >= 1:
                pre += trail_reads(grp - 1)                      # row block grp - 1 was written to the strip in the previous group
            for ln in pre:
                e(ln)
            mf = [mfma(grp * 8 + nb, bc + 8 * nb, fa_cur, zero) for nb in range(8)]
            after = {}
            fill = []
            if grp == 7:
                ns = new_set()
                # (the scalar pointer arithmetic of the new set sits in front of the first MFMA: SALU, a few cycles)
                for ln in ns:
                    e(ln)
                for slot in range(4):
                    after[2 * slot + 1] = dma("a", par, slot)                      # A(s + 2) rounds 0..3 -> stage (s + 2) & 1 = par
                if g.touch and kind == "reg":
                    after[0] = touch_loads()     # not in a tile's first / last step: v0 / v1 are the drain's temporaries there
            elif grp < 4:
                for slot in range(3):
                    j = 4 + 3 * grp + slot
                    after[2 * slot + 1] = dma("a", nxt, j) if j < 8 else dma("b", par, j - 8)   # A(s + 1) -> stage nxt, B(s + 2) -> stage par
            if kind == "first_drain" and grp <= 6:
                fill = drain(grp + 1)
            if kind == "last" and grp == 4:
                fill = operand_loads()
            if kind == "last" and grp == 7:
                fill = bias_unpack() + drain(0)
            for ln in interleave(mf, fill, after):
                e(ln)
        if kind == "first_drain":
            # row blocks 6 and 7: read-backs were issued in group 7 (block 6) -> stores; block 7 -> read back and store
            e("s_waitcnt lgkmcnt(0)")
            for ln in trail_stores():
                e(ln)
            for ln in trail_reads(7):
                e(ln)
            e("s_waitcnt lgkmcnt(0)")
            for ln in trail_stores():
                e(ln)

    # activation constants: float32 values exactly as hipcc folds them in gemm.hip; one SGPR operand per VALU instruction on gfx9,
    # so a second constant of an instruction sits in a VGPR.  The SGPRs are setup registers that are dead from here on.
    import struct
    f32 = lambda x: struct.unpack("<f", struct.pack("<f", x))[0]
    bits = lambda x: struct.unpack("<I", struct.pack("<f", x))[0]
    K, KV = {}, {}
    dead_s = [g.s1("karg", 0), g.s1("karg", 1), BID, LDS0, WAVE]
    if epi == EPI_GELU_TANH:
        c0 = f32(f32(0.79788456080286535588) * f32(2.885390081777927))
        c1 = f32(c0 * f32(0.044715))
        K["c1"] = dead_s[0]
        KV["c0"] = XT[15]
        e(f"s_mov_b32 {K['c1']}, 0x{bits(c1):08x}")
        e(f"v_mov_b32 v{KV['c0']}, 0x{bits(c0):08x}")
    elif epi == EPI_GELU_ERF:
        vals = {"sqrt1_2": 0.70710678118654752440, "p": 0.3275911, "a1": 1.061405429, "log2e": 1.4426950408889634, "a3": 1.421413741}
        for (nm, v), r in zip(vals.items(), dead_s):
            K[nm] = r
            e(f"s_mov_b32 {r}, 0x{bits(f32(v)):08x}")
        for nm, v, r in (("a2", -1.453152027, XT[12]), ("a4", -0.284496736, XT[13]), ("a5", 0.254829592, XT[14])):
            KV[nm] = r
            e(f"v_mov_b32 v{r}, 0x{bits(f32(v)):08x}")

    if os.environ.get("P4_OUT") and int(os.environ.get("P4_DEBUG", "0")) & 8:
        # bring-up: dump the setup state of every wave to p0.C ([block * 4 + wave][slot][64 lanes] dwords) and stop
        names = ["karg", "bid", "wave", "wm", "wn", "ldsw", "lds0", "p0", "p1", "cfg", "a32", "b32", "ldc4", "tcur", "tend", "prob", "mt", "nt",
                 "ptra", "ptrb", "sap", "sbp", "biasp", "cwave"]
        slots = []
        for nm in names:
            b, n = g.sreg[nm]
            slots += [("s", b + i) for i in range(n)]
        slots += [("v", r) for r in (0, 1, 2, 3, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23)]
        g.debug_slots = [(nm, g.sreg[nm]) for nm in names]
        e(f"s_lshl_b32 {X0}, {BID}, 2")
        e(f"s_add_u32 {X0}, {X0}, {WAVE}")
        e(f"s_mul_i32 {X0}, {X0}, {len(slots) * 256}")
        e(f"v_lshlrev_b32 v{T[4]}, 2, v{lane}")
        e(f"v_add_u32 v{T[4]}, {X0}, v{T[4]}")
        for k, (kind, r) in enumerate(slots):
            if kind == "s":
                e(f"v_mov_b32 v{T[5]}, s{r}")
                e(f"global_store_dword v{T[4]}, v{T[5]}, {g.s2('p0', 4)} offset:{k * 256}" if k * 256 < 4096 else f"v_add_u32 v{T[6]}, {k * 256}, v{T[4]}")
                if k * 256 >= 4096:
                    e(f"global_store_dword v{T[6]}, v{T[5]}, {g.s2('p0', 4)}")
            else:
                e(f"v_add_u32 v{T[6]}, {k * 256}, v{T[4]}")
                e(f"global_store_dword v{T[6]}, v{r}, {g.s2('p0', 4)}")
        e("s_waitcnt vmcnt(0)")
        e("s_endpgm")
        print("debug dump slots:", len(slots), [(nm, g.sreg[nm]) for nm in names])

    # ================================================================================================ prologue of the block's first tile
    c("---- prologue: A(0), B(0), A(1) rounds 0-3, B(1); the rest of 'set -1' (A(1) rounds 4-7, B(2)) rides on step 0")
    e(f"s_mov_b64 {STA}, {PTRA}")
    e(f"s_mov_b64 {STB}, {PTRB}")
    for r in range(8):
        for ln in dma("a", 0, r):
            e(ln)
    for r in range(8):
        for ln in dma("b", 0, r):
            e(ln)
    lo, hi = pair("ptra")
    e(f"s_add_u32 {lo}, {lo}, 128")
    e(f"s_addc_u32 {hi}, {hi}, 0")
    lo, hi = pair("ptrb")
    e(f"s_add_u32 {lo}, {lo}, 128")
    e(f"s_addc_u32 {hi}, {hi}, 0")
    e(f"s_mov_b64 {STA}, {PTRA}")
    e(f"s_mov_b64 {STB}, {PTRB}")
    for r in range(4):
        for ln in dma("a", 1, r):
            e(ln)
    for r in range(8):
        for ln in dma("b", 1, r):
            e(ln)
    # STA now points at A(1) round 4; STB must point at B(2) round 0; stream pointers: A at step 2, B at step 3
    lo, hi = pair("ptra")
    e(f"s_add_u32 {lo}, {lo}, 128")
    e(f"s_addc_u32 {hi}, {hi}, 0")
    lo, hi = pair("ptrb")
    e(f"s_add_u32 {lo}, {lo}, 128")
    e(f"s_addc_u32 {hi}, {hi}, 0")
    e(f"s_mov_b64 {STB}, {PTRB}")
    e(f"s_add_u32 {lo}, {lo}, 128")
    e(f"s_addc_u32 {hi}, {hi}, 0")
    e(f"s_mov_b32 {POSA}, 2")
    e(f"s_mov_b32 {POSB}, 3")
    next_tile()
    e(f"s_mov_b32 {HAVEPREV}, 0")
    e("s_waitcnt vmcnt(12)")                                   # A(0), B(0) landed (A(1) r0-3 and B(1) may still fly)
    e("s_barrier")
    for nb in range(8):
        for ln in read_frag(V_FB[0] + 8 * nb, V_BLO[0], nb):
            e(ln)
    for ln in read_frag(V_FA[0], V_ALO[0], 0):
        e(ln)
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    e("s_barrier")

    # ================================================================================================ tile loop
    TILE, NODRAIN, AFTER0, KLOOP, KDONE, FINAL = (g.label(x) for x in ("tile", "nodrain", "after0", "kloop", "kdone", "final"))
    e(f"{TILE}:")
    e(f"s_cmp_eq_u32 {HAVEPREV}, 0")
    e(f"s_cbranch_scc1 {NODRAIN}")
    step(0, "first_drain")
    e(f"s_branch {AFTER0}")
    e(f"{NODRAIN}:")
    step(0, "first")
    e(f"{AFTER0}:")
    step(1, "reg")
    e(f"s_sub_u32 {LOOP}, {NK}, 4")
    e(f"s_lshr_b32 {LOOP}, {LOOP}, 1")
    e(f"s_cmp_eq_u32 {LOOP}, 0")
    e(f"s_cbranch_scc1 {KDONE}")
    e(f"{KLOOP}:")
    step(0, "reg")
    step(1, "reg")
    e(f"s_sub_u32 {LOOP}, {LOOP}, 1")
    e(f"s_cmp_lg_u32 {LOOP}, 0")
    e(f"s_cbranch_scc1 {KLOOP}")
    e(f"{KDONE}:")
    step(0, "reg")
    step(1, "last")
    c("---- tile switch: the finished tile's output pointer becomes the drain pointer; current <- next")
    e(f"s_mov_b64 {CRUN}, {CWAVE}")
    e(f"s_cmp_eq_u32 {HASNEXT}, 0")
    e(f"s_cbranch_scc1 {FINAL}")
    e(f"s_add_u32 {TCUR}, {TCUR}, {STRIDE}")
    e(f"s_mov_b32 {PROB}, {NPROB}")
    e(f"s_mov_b32 {MT}, {NMT}")
    e(f"s_mov_b32 {NT}, {NNT}")
    epi_ptrs(PROB, MT, NT)
    next_tile()
    e(f"s_mov_b32 {HAVEPREV}, 1")
    e(f"s_branch {TILE}")

    # ================================================================================================ the block's last tile: drain without cover
    e(f"{FINAL}:")
    for mb in range(1, 8):
        for ln in drain(mb):
            e(ln)
        for ln in trail_reads(mb - 1):
            e(ln)
        e("s_waitcnt lgkmcnt(0)")
        for ln in trail_stores():
            e(ln)
    for ln in trail_reads(7):
        e(ln)
    e("s_waitcnt lgkmcnt(0)")
    for ln in trail_stores():
        e(ln)
    e("s_waitcnt vmcnt(0)")
    return g


def render(g):
    lines = []
    # bring-up only, and only when writing to another directory (P4_OUT): 1 drop global stores, 2 drop operand loads, 4 drop LDS-DMAs
    dbg = int(os.environ.get("P4_DEBUG", "0")) if os.environ.get("P4_OUT") else 0
    for ln in g.out:
        if ln.startswith(";"):
            continue
        if (dbg & 1) and ln.startswith("global_store"):
            continue
        if (dbg & 2) and ln.startswith("global_load_dword"):
            continue
        if (dbg & 4) and ln.startswith("global_load_lds"):
            continue
        lines.append('"' + ln + '\\n\\t"')
    return "\n".join(lines) + "\n"


def outputs():
    """file name -> text of every generated file."""
    files = {}
    for epi, name in ((EPI_NONE, "none"), (EPI_GELU_TANH, "gelu_tanh"), (EPI_GELU_ERF, "gelu_erf")):
        g = build(epi)
        files[f"gemm_p4_body_{name}.inc"] = "// generated by gen_gemm_p4.py -- do not edit\n" + render(g)
        n_ins = sum(1 for l in g.out if not l.startswith(";") and not l.rstrip().endswith(":"))
        print(f"gemm_p4_body_{name}.inc: {n_ins} instructions, {g.s_next} SGPRs", file=sys.stderr if "--check" in sys.argv else sys.stdout)
    # the statement's clobber list: everything the body names (inputs are therefore allocated in v0-v7 / s0-s7)
    regs = ([f"v{i}" for i in range(8, 256)] + [f"a{i}" for i in range(256)] + [f"s{i}" for i in range(Gen.S_FIRST, Gen.S_END) if i not in Gen.S_AVOID]
            + ["vcc", "memory"])
    files["gemm_p4_clobbers.inc"] = ("// generated by gen_gemm_p4.py -- do not edit\n"
                                     + ",\n".join(", ".join(f'"{r}"' for r in regs[i:i + 16]) for i in range(0, len(regs), 16)) + "\n")
    return files


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    files = outputs()
    if "--check" in sys.argv:      # build.sh: the committed files must be exactly what this generator emits (nothing is written)
        stale = [n for n, text in files.items() if not os.path.exists(os.path.join(here, n)) or open(os.path.join(here, n)).read() != text]
        if stale:
            print("stale generated files: " + ", ".join(stale), file=sys.stderr)
            sys.exit(1)
        sys.exit(0)
    for n, text in files.items():
        with open(os.path.join(os.environ.get("P4_OUT", here), n), "w") as f:
            f.write(text)
