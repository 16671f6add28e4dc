#!/usr/bin/env python3
"""Generator of the 16-bit causal-attention FORWARD for gfx950 as ONE hand-placed instruction stream
(kfunca_amd/csrc/device/attn_fwd_w4.inc, included by attention.hip; replaces the hot loop of the reference's
CausalAttentionForwardFN, src/device/utils/causal_attention.h:66-258, for D = 128).

Structure (cdna_hip_programming.md, "4-wave, one-wave-per-SIMD, persistent structure"):
  * a workgroup = 4 waves = one 256-row query block; a wave = 64 query rows = two 32-row blocks b0, b1 and the WHOLE 512-register file:
      a[0:127]   O^T accumulators  [b][db]            (32 d x 32 queries each)
      a[128:191] Q fragments       [b][kk]            (B operand of S^T = K Q^T)
      a[192:255] K fragments       [sub][kk]          (A operand; one 64-key tile, re-read per tile)
      v[0:63]    S^T accumulators  [b][sub] -> exponentiated in place
      v[64:95]   P (16-bit pairs)  [b][ks]            (B operand of O^T += V^T P^T)
      v[96:159]  V^T fragments     [ks][db]           (A operand, ds_read_b64_tr_b16)
      v[160:..]  addresses, running maxima, row sums, temporaries
  * per 64-key tile a wave issues 64 MFMAs in four slots of 16:  A: S(b0)   B: PV(b1, previous tile)   C: S(b1)   D: PV(b0).
    The two query blocks run HALF A TILE APART, so each block's softmax (16 max3, 32 fma, 32 exp, 32 add, 16 cvt_pk) has the 32
    MFMA gaps between its S slot and its PV slot to itself: one v_exp_f32 per gap, everywhere. A rescale of O (deferred running
    maximum) happens at the decision point of a block, when that block's previous P V has long finished: the textbook order.
  * K / V tiles arrive by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB pieces, 8 per wave and tile) into two-slot rings, K two tiles
    ahead, V one; ONE barrier per tile (end of slot A) behind an s_waitcnt vmcnt(0) that the DMA issued a whole tile earlier has
    long satisfied. LDS image: 8-row x 32-column subtiles with the chunk XOR inside (cdna_hip_programming.md T10 image (a)): two
    base registers for the row reads, two for the transposed reads, everything else immediates.
  * every filler (VALU, LDS read, DMA piece, wait) is ASSIGNED to an MFMA gap by the tables below; `check()` walks the emitted
    stream and enforces the distances the hardware does not interlock (MFMA result -> VALU, VALU -> MFMA operand, LDS read -> use).

The file written is a C++ header with one string literal per element type (bf16 / f16) and the clobber list."""
import argparse
import os
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
OUT = ROOT / "kfunca_amd" / "csrc" / "device" / "attn_fwd_w4.inc"

# ------------------------------------------------------------------ register map
def S(b, n): return 32 * b + n                  # score / probability n (0..31 = sub * 16 + register) of block b
def P(b, ks): return 64 + 16 * b + 4 * ks        # packed 16-bit P fragment of k-step ks (4 VGPRs)
# (V^T / Q / K fragments and the O accumulators depend on the head size: Gen.VF / QF / KF / OA)
KB = (160, 161)          # K row-read bases (even / odd k-step)
VB = (162, 163)          # V transposed-read bases (first / second read of a fragment)
DMA = (164, 165)         # LDS-DMA source offsets (even / odd row group)
MC = (166, 167)          # running maximum in use, exponent units (score * scale * log2 e), per block
LA, LB = (168, 170), (169, 171)   # row sums, two partial chains per block
MXA, MXB = (172, 174), (173, 175)  # per-lane tile maxima (two chains)
NEGINF, RM, DV = 176, 177, 178
T = list(range(180, 196))  # temporaries (slow path, prologue, epilogue)
NMC = (200, 216)           # sixteen copies of -MC per block: the C operand of a score chain's first MFMA (S^T arrives as score - maximum in use)

# scalar registers (all clobbered; inputs are copied in)
# (s32 .. s35 are the ABI's stack / frame registers: not ours to clobber)
K_SRD, V_SRD, Q_SRD, O_SRD = 36, 40, 44, 48
S_C, S_DEFER, S_T, S_DT, S_IT, S_KVSR, S_TSTEP, S_RB0 = 52, 53, 54, 55, 56, 57, 58, 59
S_KOFF0, S_KOFF1, S_VOFF0, S_VOFF1, S_M0K, S_M0V = 60, 61, 62, 63, 64, 65
S_TMP, S_TMP2, S_QSR, S_OSR, S_WID, S_TM1 = 66, 67, 68, 69, 70, 71
S_LSE = 72  # pair
S_LDS, S_STAGE, S_X0, S_X1, S_X2 = 74, 75, 76, 77, 78
S_MUT = 79  # mutation build: the tile whose probabilities are dropped (-1: none)
N_VGPR = 232  # v0 .. v231 are the stream's; the rest of the arch file stays the compiler's (it has nowhere else to keep a scalar it cannot hold in SGPRs)

# ---- placement tables (gap = position behind MFMA number `gap` of the 64 of a tile iteration; slot A 0..15, B 16..31, C 32..47, D 48..63)
# Measured (tools/attn_fwd_w4_timeline.py, cycles per steady iteration, slots A / B / C / D): all 8 pieces in slot B and the V reads two per gap
# over slot C: 606 / 824 / 742 / 703 = 2875; the 4 V pieces moved to slot D's tail and the V reads spread over gaps 22 .. 47 (into the DMA slot):
# 612 / 891 / 683 / 754 = 2940 - LDS reads beside DMA pieces cost more than they relieve. (Exact-score stream, k-step 0's eight reads in gaps 18 .. 21 and
# the rest two per gap in slot C: 611 / 878 / 674 / 665 = 2827 against 608 / 814 / 695 / 662 = 2779: slot B is the full one. The pieces on every second gap
# 17 .. 31 with the bookkeeping between them: 613 / 832 / 712 / 703 = 2861.)
DMA_G = [int(os.environ.get("KF_GEN_F128_DMA0", 23)) + i for i in range(4)] + [int(os.environ.get("KF_GEN_F128_VDMA0", 27)) + i for i in range(4)]   # 4 K + 4 V pieces behind the barrier, one per gap (K first: it is needed a slot earlier)
BOOK_K_G, BOOK_V_G, BOOK_VB_G = 31, 31, 31  # ring toggles + next source offsets, each behind the last use of the old slot
# (round 5: four per gap instead of three, and the K reads of slot D two per gap: both sets are back eight gaps before the s_waitcnt lgkmcnt(0)
#  that drains the queue for them instead of five / four - same-box 1.178 / 1.195 / 1.186 ms against 1.203 / 1.223 / 1.208, tools/scratch/ab_fwd128.sh)
VREAD_G = [32 + i // int(os.environ.get("KF_GEN_F128_V", 4)) for i in range(32)]  # slot C, four per gap: done eight gaps before slot D's wait

KSLOT = 16384
VSLOT0 = 32768
STAGE0 = 65536
STAGE_ROW = 272
LDS_BYTES = STAGE0 + 4 * 64 * STAGE_ROW


def vr(i, n=1): return f"v{i}" if n == 1 else f"v[{i}:{i + n - 1}]"
def ar(i, n=1): return f"a{i}" if n == 1 else f"a[{i}:{i + n - 1}]"
def sr(i, n=1): return f"s{i}" if n == 1 else f"s[{i}:{i + n - 1}]"


class Ins:
    """One instruction with what the checker needs to know about it."""
    def __init__(self, text, kind, reads=(), writes=(), tag=""):
        self.text, self.kind, self.reads, self.writes, self.tag = text, kind, tuple(reads), tuple(writes), tag


def V(i, n=1): return [("v", j) for j in range(i, i + n)]
def A(i, n=1): return [("a", j) for j in range(i, i + n)]


class Gen:
    def __init__(self, f16=False, mutant=False, ablate=(), stamps=False, scaled=False, D=128):
        # D = 64 (round 5; the reference's second fast head size, causal_attention_kernel.cu:40-52): half the k-steps and half
        # the column blocks - a 64-key tile is 32 MFMAs in four slots of 8 - with the SAME softmax work per tile, so the tile is VALU-issue bound
        # (three scores per gap in the exponent chain). LDS images keep their 256-byte rows with the first 128 bytes used: every address formula
        # is the D = 128 one, a row group of a tile is ONE 1-KiB DMA piece instead of two. Exact scores only.
        assert D in (64, 128) and not (scaled and D == 64)
        self.D, self.NKK, self.NDB = D, D // 16, D // 32
        self.SL = 2 * self.NKK               # MFMAs per slot (S of a block: 2 sub-tiles x NKK k-steps; P V: 4 k-steps x NDB column blocks)
        self.NG = 4 * self.SL                # gaps per tile iteration
        self.VF = lambda ks, db: 96 + 4 * self.NDB * ks + 4 * db        # V^T fragment (4 VGPRs)
        self.OA = lambda b, db: 16 * self.NDB * b + 16 * db             # AGPRs
        self.QF = lambda b, kk: 128 + 4 * self.NKK * b + 4 * kk
        self.KF = lambda sub, kk: 192 + 4 * self.NKK * sub + 4 * kk
        self.mutant = mutant
        # scaled: the query is multiplied by scale log2(e) and ROUNDED to the element type once per pass, the score chains start from -max and
        # deliver exponents (no multiply per score: -64 VALU per tile, ~4 % of the kernel) - at the price of a score error of eps scale sum|q k|,
        # which grows with the logits (KF_ATTN_SCALED_OPERANDS). Default: exact f32 scores, one fma per score.
        self.scaled = scaled
        self.stamps = stamps        # diagnostic build (tools/attn_fwd_w4_timeline.py): s_memtime at the slot boundaries, eight sums per wave and block
        self.ablate = set(ablate)   # timing experiments only (tools/scratch/fwd_w4_ablate.sh): parts of the tile body left out - WRONG results
        self.f16 = f16
        self.mfma = "v_mfma_f32_32x32x16_f16" if f16 else "v_mfma_f32_32x32x16_bf16"
        self.cvt = "v_cvt_pk_f16_f32" if f16 else "v_cvt_pk_bf16_f32"
        self.out = []   # list of Ins (and labels / comments as kind "raw")
        self.uid = 0

    # -------------------------------------------------------------- emit helpers
    def raw(self, text): self.out.append(Ins(text, "raw"))
    def label(self, name): self.out.append(Ins(f"{name}:", "label"))
    def salu(self, text): self.out.append(Ins(text, "salu"))

    def valu(self, text, reads=(), writes=(), trans=False):
        self.out.append(Ins(text, "trans" if trans else "valu", reads, writes))

    def qk(self, b, sub, kk, first=False):
        """One MFMA of S^T = K (c Q)^T. The chain starts from -MC (sixteen equal registers per lane: a lane holds one query), so the
        accumulator IS the exponent; only a pass's first tile, which has no maximum yet, starts from 0."""
        d = S(b, 16 * sub)
        if kk:
            c, cr = vr(d, 16), V(d, 16)
        elif first or not self.scaled:
            c, cr = "0", []
        else:
            c, cr = vr(NMC[b], 16), V(NMC[b], 16)
        self.out.append(Ins(f"{self.mfma} {vr(d, 16)}, {ar(self.KF(sub, kk), 4)}, {ar(self.QF(b, kk), 4)}, {c}", "mfma",
                            A(self.KF(sub, kk), 4) + A(self.QF(b, kk), 4) + cr, V(d, 16), tag=f"qk b{b} sub{sub} kk{kk}"))

    def pv(self, b, ks, db):
        o = self.OA(b, db)
        self.out.append(Ins(f"{self.mfma} {ar(o, 16)}, {vr(self.VF(ks, db), 4)}, {vr(P(b, ks), 4)}, {ar(o, 16)}", "mfma",
                            V(self.VF(ks, db), 4) + V(P(b, ks), 4) + A(o, 16), A(o, 16), tag=f"pv b{b} ks{ks} db{db}"))

    def lds_k(self, sub, kk):   # K fragment (sub, kk) of the tile at the K bases
        if "lds" in self.ablate: return
        imm = 8192 * sub + 512 * (kk >> 1)
        self.out.append(Ins(f"ds_read_b128 {ar(self.KF(sub, kk), 4)}, {vr(KB[kk & 1])} offset:{imm}", "lds", V(KB[kk & 1]), A(self.KF(sub, kk), 4)))

    def lds_v(self, ks, db, second):  # half of V^T fragment (ks, db): keys 16 ks + 4 h + {0..3} (+ 8 for the second half)
        sub, s = ks >> 1, ks & 1
        if "lds" in self.ablate: return
        imm = 2048 * (4 * sub + 2 * s + second) + 512 * db
        self.out.append(Ins(f"ds_read_b64_tr_b16 {vr(self.VF(ks, db) + 2 * second, 2)}, {vr(VB[second])} offset:{imm}", "lds",
                            V(VB[second]), V(self.VF(ks, db) + 2 * second, 2)))

    def wait(self, vm=None, lgkm=None):
        parts = ([f"vmcnt({vm})"] if vm is not None else []) + ([f"lgkmcnt({lgkm})"] if lgkm is not None else [])
        self.out.append(Ins("s_waitcnt " + " ".join(parts), "wait", tag=f"{'vm' if vm is not None else ''}{'lgkm' if lgkm is not None else ''}"))

    def barrier(self):
        if "barrier" in self.ablate and getattr(self, "in_loop", False): return
        self.out.append(Ins("s_barrier", "barrier"))

    def dma_m0(self, m0_from, m0_add):
        if "dma" in self.ablate and getattr(self, "in_loop", False): return
        self.salu(f"s_add_u32 m0, {sr(m0_from)}, {m0_add}" if m0_add else f"s_mov_b32 m0, {sr(m0_from)}")

    def dma_load(self, srd, voff, soff, inst_off):
        if "dma" in self.ablate and getattr(self, "in_loop", False): return
        if self.out[-1].text.startswith(("s_add_u32 m0", "s_mov_b32 m0")):
            self.salu("s_nop 0")   # SALU write of M0 -> LDS-DMA: one wait state (in the tile loop a gap's other fillers stand between the two)
        o = f" offset:{inst_off}" if inst_off else ""
        self.out.append(Ins(f"buffer_load_dwordx4 {vr(voff)}, {sr(srd, 4)}, {sr(soff)} offen{o} lds", "dma", V(voff)))

    def dma(self, srd, voff, soff, m0_from, m0_add, inst_off):
        self.dma_m0(m0_from, m0_add)
        self.dma_load(srd, voff, soff, inst_off)

    def stamp(self, bucket):
        """Diagnostic build only: the cycles since the previous stamp go to accumulator `bucket` (0 prologue, 1..4 the steady tile's slots
        A..D, 5 the other variants' iterations, 6 epilogue; s91 counts steady iterations). The SMEM round trip of each stamp (~100 cycles)
        lands in the bucket that FOLLOWS it: tools/attn_fwd_w4_timeline.py calibrates it with two stamps back to back (bucket 7)."""
        if not self.stamps: return
        self.salu("s_memtime s[80:81]")
        self.salu("s_waitcnt lgkmcnt(0)")
        self.salu("s_sub_u32 s83, s80, s82")
        self.salu(f"s_add_u32 s{84 + bucket}, s{84 + bucket}, s83")
        self.salu("s_mov_b32 s82, s80")

    # -------------------------------------------------------------- the filler streams of one iteration
    def softmax_ops(self, b, cur, masked, drop, first=False):
        """All VALU ops of block b's softmax of one tile as (gap, order, emit) tuples; gaps are relative to the block's S slot
        (0 = behind the slot's first MFMA) and run to 57; the caller shifts block 1 by 32 and folds modulo 64."""
        ops = []
        s_c = sr(S_C)
        def add(g, fn, key=None): ops.append((g, len(ops) if key is None else key, fn))
        # ---- masking of the diagonal tile (this wave's own diagonal): key (e & 3) + 8 (e >> 2) + 4 h > query r
        # (this variant runs once per wave and block: its gaps may be as full as they need to be; the chain's timing stays the steady one)
        if masked:
            sub = 0 if b == 0 else 1   # b0: sub 0 is the diagonal, sub 1 lies wholly above it; b1: sub 0 is visible, sub 1 is the diagonal
            d64 = self.D == 64
            if b == 0:
                for e in range(16):  # sub 1 of block 0: all of it (its 8 MFMAs are not issued)
                    add((1 + e // 4) if d64 else (2 + e // 3), lambda e=e: self.valu(f"v_mov_b32 {vr(S(0, 16 + e))}, {vr(NEGINF)}", V(NEGINF), V(S(0, 16 + e))))
                gn = self.NKK        # behind the diagonal sub-tile's last MFMA
                add(gn, lambda: self.valu(f"v_mov_b32 {vr(MXB[0])}, {vr(NEGINF)}", V(NEGINF), V(MXB[0])))
                add(gn, lambda: self.salu("s_nop 15"))   # no MFMAs follow the diagonal sub-tile's chain in this slot: give its last one its time
                add(gn, lambda: self.salu("s_nop 7"))
            for e in range(16):
                kc = (e & 3) + 8 * (e >> 2)
                g = ((5 if sub == 0 else 9) if d64 else (9 if sub == 0 else 17)) + e // 8
                def m(e=e, kc=kc, sub=sub):
                    self.valu(f"v_cmp_gt_i32 vcc, {kc}, {vr(RM)}", V(RM), [("vcc", 0)])
                    self.valu(f"v_cndmask_b32 {vr(S(b, 16 * sub + e))}, {vr(S(b, 16 * sub + e))}, {vr(NEGINF)}, vcc",
                              V(S(b, 16 * sub + e)) + V(NEGINF) + [("vcc", 0)], V(S(b, 16 * sub + e)))
                add(g, m)
        # ---- tile maximum per lane: two chains (sub 0 -> MXA, sub 1 -> MXB), v_max3 takes two new values per instruction
        # sub 1's sixteen values (ready only at gap 17, the decision at 21) go as TWO interleaved chains of four: no max3 waits for the one in front of it
        MXC = T[3]
        def chain(vals, mx, gaps):
            seq = [(vals[0], vals[1], vals[2])] + [(mx, vals[3 + 2 * j], vals[min(4 + 2 * j, len(vals) - 1)]) for j in range((len(vals) - 2) // 2)]
            assert len(seq) == len(gaps), (len(seq), len(gaps))
            for (x, y, z), g in zip(seq, gaps):
                add(g, lambda x=x, y=y, z=z, mx=mx: self.valu(f"v_max3_f32 {vr(mx)}, {vr(x)}, {vr(y)}, {vr(z)}", V(x) + V(y) + V(z), V(mx)))
        s0, s1 = [S(b, e) for e in range(16)], [S(b, 16 + e) for e in range(16)]
        if self.D == 64:             # sub 0's scores are readable from gap 5 (its chain ends at MFMA 3), sub 1's from gap 9
            if masked and b == 0:
                chain(s0, MXA[0], [7, 7, 8, 8, 9, 9, 10, 10])
                add(11, lambda: self.valu(f"v_mov_b32 {vr(MXC)}, {vr(NEGINF)}", V(NEGINF), V(MXC)))
            else:
                chain(s0, MXA[b], [5, 5, 6, 6, 7, 7, 8, 8])
                g1 = [11, 11, 12, 12] if masked else [9, 9, 10, 10]
                chain(s1[:8], MXB[b], g1)
                chain(s1[8:], MXC, g1)
            gd = 13
        elif masked and b == 0:      # sub 1 lies wholly above the diagonal (MXB = -inf above); sub 0 behind its mask ops
            chain(s0, MXA[0], [11, 11, 12, 12, 13, 13, 14, 14])
            add(15, lambda: self.valu(f"v_mov_b32 {vr(MXC)}, {vr(NEGINF)}", V(NEGINF), V(MXC)))
            gd = 21
        else:
            chain(s0, MXA[b], [9, 9, 10, 11, 12, 13, 14, 14])
            g1 = [19, 19, 20, 20] if masked else [17, 18, 19, 20]
            chain(s1[:8], MXB[b], g1)
            chain(s1[8:], MXC, g1)
            gd = 21
        # ---- decision: does any query of the wave exceed the maximum in use by more than `defer` exponent units?
        add(gd, lambda: self.valu(f"v_max3_f32 {vr(MXA[b])}, {vr(MXA[b])}, {vr(MXB[b])}, {vr(MXC)}", V(MXA[b]) + V(MXB[b]) + V(MXC), V(MXA[b])))
        if not self.scaled:
            # exact scores: the excess of the tile's maximum over the one in use, in exponent units
            add(gd, lambda: self.valu(f"v_fma_f32 {vr(DV)}, {vr(MXA[b])}, {s_c}, -{vr(MC[b])}", V(MXA[b]) + V(MC[b]), V(DV)))
            add(gd + 1, lambda: self.valu(f"v_cmp_lt_f32 vcc, {sr(S_DEFER)}, {vr(DV)}", V(DV), [("vcc", 0)]))
            add(gd + 1, lambda: self.rescale(b))
        elif first:
            # (scaled query: the scores arrive as exponents relative to the maximum in use, the tile maximum IS the excess)
            add(gd + 1, lambda: self.adopt_first(b))
        else:
            add(gd + 1, lambda: self.valu(f"v_cmp_lt_f32 vcc, {sr(S_DEFER)}, {vr(MXA[b])}", V(MXA[b]), [("vcc", 0)]))
            add(gd + 1, lambda: self.rescale(b))
        # ---- the exponent chain, one value per gap: [fma: scale, subtract the maximum |] exp2 | row sum | pack pairs
        # (order inside a gap: exp | pack | row sum | fma - the consumer of an exp stands at least two instructions behind it)
        g0 = gd + 2
        per = 3 if self.D == 64 else 1   # scores per gap: the block's P V slot starts 24 (D = 64) or 48 gaps behind its S slot
        for n in range(32):
            x = S(b, n)
            gn = g0 + n // per
            if drop:  # mutation build: this tile's probabilities are dropped (p = exp2(-inf) = 0)
                add(gn, lambda x=x: self.valu(f"v_mov_b32 {vr(x)}, {vr(NEGINF)}", V(NEGINF), V(x)), key=1003)
            elif not self.scaled:
                add(gn, lambda x=x: self.valu(f"v_fma_f32 {vr(x)}, {vr(x)}, {s_c}, -{vr(MC[b])}", V(x) + V(MC[b]), V(x)), key=1003)
            add(gn + 1, lambda x=x: self.valu(f"v_exp_f32 {vr(x)}, {vr(x)}", V(x), V(x), trans=True), key=1000)
            l = (LA if n % 2 == 0 else LB)[b]
            add(gn + 2, lambda x=x, l=l: self.valu(f"v_add_f32 {vr(l)}, {vr(l)}, {vr(x)}", V(l) + V(x), V(l)), key=1002)
            if n % 2 == 1:
                d = P(b, n // 8) + (n % 8) // 2
                add(gn + 3, lambda x=x, d=d: self.valu(f"{self.cvt} {vr(d)}, {vr(x - 1)}, {vr(x)}", V(x - 1) + V(x), V(d)), key=1001)
        return ops

    def adopt_first(self, b):
        """A pass's first tile: no maximum is in use yet (its score chains started from 0), so the tile's own maximum is adopted
        unconditionally - MC, the sixteen copies of -MC, and the tile's 32 exponents shifted by it. O and the row sums are still 0."""
        t0, t1 = T[0], T[1]
        self.valu(f"v_mov_b32 {vr(t0)}, {vr(MXA[b])}", V(MXA[b]), V(t0))
        self.valu(f"v_mov_b32 {vr(t1)}, {vr(MXA[b])}", V(MXA[b]), V(t1))
        self.salu("s_nop 1")
        self.valu(f"v_permlane32_swap_b32 {vr(t0)}, {vr(t1)}", V(t0) + V(t1), V(t0) + V(t1))
        self.valu(f"v_max_f32 {vr(MC[b])}, {vr(t0)}, {vr(t1)}", V(t0) + V(t1), V(MC[b]))       # the same for the two lanes of a query
        self.valu(f"v_xor_b32 {vr(t1)}, 0x80000000, {vr(MC[b])}", V(MC[b]), V(t1))
        for i in range(16):
            self.valu(f"v_mov_b32 {vr(NMC[b] + i)}, {vr(t1)}", V(t1), V(NMC[b] + i))
        for n in range(32):
            self.valu(f"v_add_f32 {vr(S(b, n))}, {vr(S(b, n))}, {vr(t1)}", V(S(b, n)) + V(t1), V(S(b, n)))

    def rescale(self, b):
        """The rare path, inline behind the decision: adopt the new maximum (the same for the two lanes of a query), scale the row
        sums and the O accumulators of block b. The block's last P V finished at least 20 MFMAs ago."""
        self.uid += 1
        skip = f"L_norescale_{self.uid}_%="
        self.salu(f"s_cbranch_vccz {skip}")
        n_before = len(self.out)
        t0, t1, t2 = T[0], T[1], T[2]
        self.valu(f"v_mov_b32 {vr(t0)}, {vr(MXA[b])}", V(MXA[b]), V(t0))
        self.valu(f"v_mov_b32 {vr(t1)}, {vr(MXA[b])}", V(MXA[b]), V(t1))
        self.salu("s_nop 1")
        self.valu(f"v_permlane32_swap_b32 {vr(t0)}, {vr(t1)}", V(t0) + V(t1), V(t0) + V(t1))
        self.valu(f"v_max_f32 {vr(t0)}, {vr(t0)}, {vr(t1)}", V(t0) + V(t1), V(t0))
        if self.scaled:
            self.valu(f"v_max_f32 {vr(t0)}, 0, {vr(t0)}", V(t0), V(t0))                                    # how far the maximum in use moves up (0: this query stays)
            self.valu(f"v_add_f32 {vr(MC[b])}, {vr(MC[b])}, {vr(t0)}", V(t0) + V(MC[b]), V(MC[b]))       # new maximum
            self.valu(f"v_xor_b32 {vr(t2)}, 0x80000000, {vr(MC[b])}", V(MC[b]), V(t2))
            for i in range(16):
                self.valu(f"v_mov_b32 {vr(NMC[b] + i)}, {vr(t2)}", V(t2), V(NMC[b] + i))
            for n in range(32):                                                                            # this tile's exponents were formed against the old one
                self.valu(f"v_sub_f32 {vr(S(b, n))}, {vr(S(b, n))}, {vr(t0)}", V(S(b, n)) + V(t0), V(S(b, n)))
            self.valu(f"v_exp_f32 {vr(t1)}, -{vr(t0)}", V(t0), V(t1), trans=True)                          # alpha = 2^(old - new)
        else:
            self.valu(f"v_mul_f32 {vr(t0)}, {sr(S_C)}, {vr(t0)}", V(t0), V(t0))
            self.valu(f"v_max_f32 {vr(t0)}, {vr(t0)}, {vr(MC[b])}", V(t0) + V(MC[b]), V(t0))           # new maximum
            self.valu(f"v_sub_f32 {vr(t1)}, {vr(MC[b])}, {vr(t0)}", V(t0) + V(MC[b]), V(t1))
            self.valu(f"v_mov_b32 {vr(MC[b])}, {vr(t0)}", V(t0), V(MC[b]))
            self.valu(f"v_exp_f32 {vr(t1)}, {vr(t1)}", V(t1), V(t1), trans=True)                           # alpha = 2^(old - new), 0 at the first tile
        self.salu("s_nop 0")
        self.valu(f"v_mul_f32 {vr(LA[b])}, {vr(LA[b])}, {vr(t1)}", V(LA[b]) + V(t1), V(LA[b]))
        self.valu(f"v_mul_f32 {vr(LB[b])}, {vr(LB[b])}, {vr(t1)}", V(LB[b]) + V(t1), V(LB[b]))
        for base in range(self.OA(b, 0), self.OA(b, 0) + 16 * self.NDB, 8):
            for j in range(8):
                self.valu(f"v_accvgpr_read_b32 {vr(T[4 + j])}, {ar(base + j)}", A(base + j), V(T[4 + j]))
            for j in range(8):
                self.valu(f"v_mul_f32 {vr(T[4 + j])}, {vr(T[4 + j])}, {vr(t1)}", V(T[4 + j]) + V(t1), V(T[4 + j]))
            for j in range(8):
                self.valu(f"v_accvgpr_write_b32 {ar(base + j)}, {vr(T[4 + j])}", V(T[4 + j]), A(base + j))
        for x in self.out[n_before:]:
            x.tag = "rare"
        self.label(skip)

    # -------------------------------------------------------------- one iteration (tile `it`) of one variant
    def iteration(self, name, has_prev, has_cur, masked=False, drop=False):
        """has_cur: this wave computes tile `it` (S of both blocks, block 0's softmax and P V, the head of block 1's softmax);
        has_prev: it owes tile it - 1 its second half (block 1's softmax tail and P V). Neither: only the DMA and the barrier."""
        SL, NG, NKK, NDB = self.SL, self.NG, self.NKK, self.NDB
        G = [[] for _ in range(NG)]   # fillers per gap, (order key, emit function)
        def put(g, key, fn): G[g % NG].append((key, fn))
        self.in_loop = True
        first = has_cur and not has_prev   # a pass's first tile: no maximum in use yet
        if "valu" in self.ablate:
            has_cur_sm = has_prev_sm = False
        else:
            has_cur_sm, has_prev_sm = has_cur, has_prev
        if has_cur_sm:
            for (g, k, fn) in self.softmax_ops(0, True, masked, drop, first):
                put(g, (0, k), fn)
            for (g, k, fn) in self.softmax_ops(1, True, masked, drop, first):
                if 2 * SL + g < NG:
                    put(2 * SL + g, (1, k), fn)
        if has_prev_sm:
            for (g, k, fn) in self.softmax_ops(1, False, False, False):   # the tail of the PREVIOUS tile's block 1 (never the diagonal tile's mask: that sits in the head)
                if 2 * SL + g >= NG:
                    put(2 * SL + g - NG, (1, k), fn)
        # V^T fragments of tile it: read ONCE, in slot C, in consumption order, 2 per gap; they serve slot D (block 0) and slot B of
        # the NEXT iteration (block 1's P V of the same tile): the 64 registers are rewritten only by slot C of that iteration
        # A fragment (ks, db) may be rewritten once slot B's MFMA 16 + 4 ks + db has read it, and V(it) is in LDS since the barrier at gap
        # 16: the 32 reads spread over gaps 22 .. 47 (VREAD_G) instead of crowding slot C.
        if has_cur:
            seq = [(ks, db, sec) for ks in range(4) for db in range(NDB) for sec in (0, 1)]
            for i, (ks, db, sec) in enumerate(seq):
                # (D = 64: 16 reads, two per gap, each right behind slot B's last use of its registers (gaps 10 .. 17): the wait in front of slot D
                #  drains the LDS queue, so a read issued just before it would stand there for its whole latency - first form, reads in 16 .. 23: 1.95 -> 1.71 ms only)
                g = VREAD_G[i] if self.D == 128 else max(SL + 2 + i // 2, int(os.environ.get("KF_GEN_F64_V", SL + 2)) + i // 2)
                assert g >= SL + NDB * ks + db + 2
                put(g, (2, i), lambda ks=ks, db=db, sec=sec: self.lds_v(ks, db, sec))
        # K fragments of tile it + 1 (slot D; its own tiles only)
        if has_cur and not masked:
            seq = [(sub, kk) for sub in range(2) for kk in range(NKK)]
            for i, (sub, kk) in enumerate(seq):
                put((48 + i // int(os.environ.get("KF_GEN_F128_K", 2))) if self.D == 128 else (3 * SL + i // int(os.environ.get("KF_GEN_F64_K", 2))), (2, i), lambda sub=sub, kk=kk: self.lds_k(sub, kk))   # (D = 64: done four gaps before the next tile's wait)
        # LDS-DMA of K(it + 2) and V(it + 1): this wave's 4 + 4 pieces, behind the barrier, one per gap from gap 23
        pieces = [(K_SRD, DMA[0], S_KOFF0, S_M0K, 0, 0), (K_SRD, DMA[0], S_KOFF0, S_M0K, 896, 128),
                  (K_SRD, DMA[1], S_KOFF1, S_M0K, 2048, 0), (K_SRD, DMA[1], S_KOFF1, S_M0K, 2048 + 896, 128),
                  (V_SRD, DMA[0], S_VOFF0, S_M0V, 0, 0), (V_SRD, DMA[0], S_VOFF0, S_M0V, 896, 128),
                  (V_SRD, DMA[1], S_VOFF1, S_M0V, 2048, 0), (V_SRD, DMA[1], S_VOFF1, S_M0V, 2048 + 896, 128)]
        dma_g = DMA_G
        if self.D == 64:     # a row group's 8 x 128 B are ONE piece (the first half of the image): 2 K + 2 V pieces per wave, behind the barrier at gap 8
            pieces = [p for p in pieces if p[5] == 0]
            dma_g = [11, 12, 13, 14]
        for i, p in enumerate(pieces):
            put(dma_g[i], (-1, i), lambda p=p: self.dma_m0(p[3], p[4]))      # M0 first in the gap, the load last
            put(dma_g[i], (3, i), lambda p=p: self.dma_load(p[0], p[1], p[2], p[5]))
        # loop bookkeeping (slot B, behind the DMA): ring toggles, next source offsets
        # each toggle sits between the last use of the old slot and the first use of the new one
        def book_vb():   # V read bases -> the slot of V(it): behind the barrier, before the first V read
            for r in VB:
                self.valu(f"v_xor_b32 {vr(r)}, {KSLOT}, {vr(r)}", V(r), V(r))
        def book_k():    # behind the K pieces, before slot D's K reads: K read bases -> K(it + 1)'s slot; DMA side -> K(it + 3)
            for r in KB:
                self.valu(f"v_xor_b32 {vr(r)}, {KSLOT}, {vr(r)}", V(r), V(r))
            if salu_gap is None: book_k_salu()
        kb_gap = int(os.environ.get("KF_GEN_F128_KBG", 31)) if self.D == 128 else None      # (experiments: the two read-base toggles elsewhere)
        vb_gap = int(os.environ.get("KF_GEN_F128_VBG", 31)) if self.D == 128 else None
        def book_k_salu():
            self.salu(f"s_xor_b32 {sr(S_M0K)}, {sr(S_M0K)}, {KSLOT}")
            self.salu(f"s_add_u32 {sr(S_KOFF0)}, {sr(S_KOFF0)}, {sr(S_TSTEP)}")   # one tile further, saturating at the last tile (a clamped piece is fetched again, never read)
            self.salu(f"s_min_u32 {sr(S_KOFF0)}, {sr(S_KOFF0)}, {sr(S_X2)}")
            self.salu(f"s_add_u32 {sr(S_KOFF1)}, {sr(S_KOFF0)}, {sr(S_TMP2)}")
        # round 5: the SCALAR half of the bookkeeping (DMA destinations and source offsets of the next tile: 9 s_* instructions) leaves the full slot B for slot
        # C (gap 44: behind this tile's pieces, ahead of the next tile's) - same-box 1.170 / 1.147 / 1.166 ms against 1.175 / 1.203 / 1.177; gap 60: no difference
        salu_gap = int(os.environ.get("KF_GEN_F128_BOOKS", 44)) if self.D == 128 else None
        def book_v():    # behind the V pieces: DMA side -> V(it + 2)
            self.salu(f"s_xor_b32 {sr(S_M0V)}, {sr(S_M0V)}, {KSLOT}")
            self.salu(f"s_add_u32 {sr(S_VOFF0)}, {sr(S_VOFF0)}, {sr(S_TSTEP)}")
            self.salu(f"s_min_u32 {sr(S_VOFF0)}, {sr(S_VOFF0)}, {sr(S_X2)}")
            self.salu(f"s_add_u32 {sr(S_VOFF1)}, {sr(S_VOFF0)}, {sr(S_TMP2)}")
        gb = 31 if self.D == 128 else 15      # (BOOK_*_G: the last gap of slot B)
        put(vb_gap if self.D == 128 else SL + 1, (4, 0), book_vb)   # (D = 64: the V reads start at gap SL + 2)
        put(kb_gap if self.D == 128 else gb, (4, 1), book_k)
        if salu_gap is None:
            put(gb, (4, 2), book_v)
        else:
            put(salu_gap, (4, 1), book_k_salu)
            put(salu_gap, (4, 2), book_v)

        self.label(f"L_{name}_%=")
        for g in range(NG):
            slot, j = g // SL, g % SL
            # ---- waits in front of the slot's first MFMA
            if g == 0 and has_cur:
                self.wait(lgkm=0)          # K fragments (read in slot D of the previous iteration / the prologue)
            if name == "steady" and self.D == 128 and g in (16, 32, 48):
                self.stamp(g // 16)        # slot A / B / C ends here
            if g == SL and has_cur and not has_prev:
                self.salu("s_nop 15")   # no MFMAs in this slot of the first tile: the S chain that has just been issued gets its time
                self.salu("s_nop 7")
            if g == SL:
                self.wait(vm=0)            # this wave's DMA pieces of the previous iteration have landed
                self.barrier()             # everyone's: K(it + 1), V(it) are in LDS, V(it - 1) and K(it) are no longer read
            if g == 3 * SL and has_cur:
                self.wait(lgkm=0)          # V^T fragments (read in slot C)
            # ---- the MFMA of this gap
            if slot == 0 and has_cur and not (masked and j >= NKK):
                self.qk(0, j // NKK, j % NKK, first)
            elif slot == 1 and has_prev:
                self.pv(1, j // NDB, j % NDB)
            elif slot == 2 and has_cur:
                self.qk(1, j // NKK, j % NKK, first)
            elif slot == 3 and has_cur:
                self.pv(0, j // NDB, j % NDB)
            else:
                self.out.append(Ins("", "nomfma"))
            for _, fn in sorted(G[g], key=lambda t: t[0]):
                fn()
        self.in_loop = False
        self.stamp(4 if name == "steady" else 5)
        if self.stamps and name == "steady":
            self.salu("s_add_u32 s91, s91, 1")

    # -------------------------------------------------------------- whole pass of one query block
    def prologue(self):
        e = self
        if self.stamps:
            for i in range(84, 92):
                e.salu(f"s_mov_b32 s{i}, 0")
            e.salu("s_memtime s[80:81]")
            e.salu("s_waitcnt lgkmcnt(0)")
            e.salu("s_mov_b32 s82, s80")
            self.stamp(7)   # calibration: two stamps back to back
        e.raw("; ---- inputs into fixed registers")
        e.salu(f"s_mov_b64 {sr(K_SRD, 2)}, %[kp]")
        e.salu(f"s_mov_b64 {sr(V_SRD, 2)}, %[vp]")
        e.salu(f"s_mov_b64 {sr(Q_SRD, 2)}, %[qp]")
        e.salu(f"s_mov_b64 {sr(O_SRD, 2)}, %[op]")
        # num_records = the bytes of the tensor that lie behind the base (round 6: ragged sequence lengths). gfx950 range-checks voffset +
        # soffset + the instruction offset against it (tools/scratch/buffer_bounds.hip, profiles/r06_buffer_bounds.txt): a K / V / Q row
        # beyond the tensor's last one arrives in LDS as ZEROS (LDS-DMA included), an O row beyond it is not stored - a last query block of
        # fewer than 256 rows and a last key tile of fewer than 64 keys cost no instruction (zero keys lie above every real query's diagonal)
        for srd, n in ((K_SRD, "kvn"), (V_SRD, "kvn"), (Q_SRD, "qn"), (O_SRD, "on")):
            e.salu(f"s_mov_b32 {sr(srd + 2)}, %[{n}]")
            e.salu(f"s_mov_b32 {sr(srd + 3)}, 0x00020000")
        e.salu(f"s_mov_b64 {sr(S_LSE, 2)}, %[lsep]")
        for dst, src in ((S_C, "c"), (S_DEFER, "defer"), (S_T, "T"), (S_KVSR, "kvsr"), (S_QSR, "qsr"), (S_OSR, "osr"), (S_WID, "wid"), (S_LDS, "lds"), (S_MUT, "mut")):
            e.salu(f"s_mov_b32 {sr(dst)}, %[{src}]")
        e.salu(f"s_sub_u32 {sr(S_TM1)}, {sr(S_T)}, 1")
        e.salu(f"s_lshl_b32 {sr(S_X2)}, {sr(S_KVSR)}, 6")
        e.salu(f"s_mul_i32 {sr(S_X2)}, {sr(S_X2)}, {sr(S_TM1)}")
        e.salu(f"s_sub_u32 {sr(S_DT)}, {sr(S_T)}, 4")
        e.salu(f"s_add_u32 {sr(S_DT)}, {sr(S_DT)}, {sr(S_WID)}")            # this wave's diagonal tile
        e.salu(f"s_lshl_b32 {sr(S_TSTEP)}, {sr(S_KVSR)}, 6")                  # bytes per 64-key tile
        e.salu(f"s_lshl_b32 {sr(S_TMP)}, {sr(S_WID)}, 4")
        e.salu(f"s_mul_i32 {sr(S_RB0)}, {sr(S_TMP)}, {sr(S_KVSR)}")          # rows 16 w .. of a tile: this wave's two row groups
        e.salu(f"s_add_u32 {sr(S_X2)}, {sr(S_X2)}, {sr(S_RB0)}")             # ... of the LAST tile: where the source offsets saturate
        e.salu(f"s_lshl_b32 {sr(S_TMP)}, {sr(S_WID)}, 12")
        e.salu(f"s_add_u32 {sr(S_M0K)}, {sr(S_LDS)}, {sr(S_TMP)}")           # DMA destination of this wave inside a K slot ...
        e.salu(f"s_add_u32 {sr(S_M0V)}, {sr(S_M0K)}, {VSLOT0}")              # ... and a V slot
        e.salu(f"s_mul_i32 {sr(S_TMP)}, {sr(S_WID)}, {64 * STAGE_ROW}")
        e.salu(f"s_add_u32 {sr(S_STAGE)}, {sr(S_LDS)}, {STAGE0}")
        e.salu(f"s_add_u32 {sr(S_STAGE)}, {sr(S_STAGE)}, {sr(S_TMP)}")
        # ---- lane constants
        lane, r, h, t0, t1, t2, t3 = T[0], T[1], T[2], T[3], T[4], T[5], T[6]
        e.raw("; ---- lane constants")
        e.valu(f"v_mbcnt_lo_u32_b32 {vr(lane)}, -1, 0")
        e.valu(f"v_mbcnt_hi_u32_b32 {vr(lane)}, -1, {vr(lane)}")
        e.valu(f"v_and_b32 {vr(r)}, 31, {vr(lane)}")
        e.valu(f"v_lshrrev_b32 {vr(h)}, 5, {vr(lane)}")
        e.valu(f"v_mov_b32 {vr(NEGINF)}, 0xff800000")
        e.valu(f"v_lshlrev_b32 {vr(t0)}, 2, {vr(h)}")
        e.valu(f"v_sub_u32 {vr(RM)}, {vr(r)}, {vr(t0)}")                     # r - 4 h: key register e is masked where (e & 3) + 8 (e >> 2) > r - 4 h
        # K row-read bases: 2048 (r >> 3) + 64 (r & 7) + 16 (h ^ ((r >> 2) & 3)), slot 0
        e.valu(f"v_lshrrev_b32 {vr(t0)}, 3, {vr(r)}")
        e.valu(f"v_lshlrev_b32 {vr(t0)}, 11, {vr(t0)}")
        e.valu(f"v_and_b32 {vr(t1)}, 7, {vr(r)}")
        e.valu(f"v_lshlrev_b32 {vr(t1)}, 6, {vr(t1)}")
        e.valu(f"v_bfe_u32 {vr(t2)}, {vr(r)}, 2, 2")
        e.valu(f"v_xor_b32 {vr(t2)}, {vr(t2)}, {vr(h)}")
        e.valu(f"v_lshlrev_b32 {vr(t2)}, 4, {vr(t2)}")
        e.valu(f"v_add3_u32 {vr(KB[0])}, {vr(t0)}, {vr(t1)}, {vr(t2)}")
        e.valu(f"v_add_u32 {vr(KB[0])}, {sr(S_LDS)}, {vr(KB[0])}")
        e.valu(f"v_xor_b32 {vr(KB[1])}, 32, {vr(KB[0])}")
        # V transposed-read bases: i = lane & 15, q = i >> 2, p = i & 3, g = lane >> 4:  64 (4 h + q) + 16 ((2 (g & 1) + (p >> 1)) ^ h) + 8 (p & 1), V slot 1
        e.valu(f"v_bfe_u32 {vr(t0)}, {vr(lane)}, 2, 2")                      # q
        e.valu(f"v_lshl_add_u32 {vr(t0)}, {vr(h)}, 2, {vr(t0)}")             # 4 h + q
        e.valu(f"v_lshlrev_b32 {vr(t0)}, 6, {vr(t0)}")
        e.valu(f"v_bfe_u32 {vr(t1)}, {vr(lane)}, 4, 1")                      # g & 1
        e.valu(f"v_bfe_u32 {vr(t2)}, {vr(lane)}, 1, 1")                      # p >> 1
        e.valu(f"v_lshl_add_u32 {vr(t1)}, {vr(t1)}, 1, {vr(t2)}")
        e.valu(f"v_xor_b32 {vr(t1)}, {vr(t1)}, {vr(h)}")
        e.valu(f"v_lshlrev_b32 {vr(t1)}, 4, {vr(t1)}")
        e.valu(f"v_and_b32 {vr(t2)}, 1, {vr(lane)}")
        e.valu(f"v_lshlrev_b32 {vr(t2)}, 3, {vr(t2)}")
        e.valu(f"v_add3_u32 {vr(VB[0])}, {vr(t0)}, {vr(t1)}, {vr(t2)}")
        e.valu(f"v_add_u32 {vr(VB[0])}, {sr(S_LDS)}, {vr(VB[0])}")
        e.valu(f"v_add_u32 {vr(VB[0])}, {VSLOT0 + KSLOT}, {vr(VB[0])}")
        e.valu(f"v_xor_b32 {vr(VB[1])}, 32, {vr(VB[0])}")
        # LDS-DMA source offsets: row7 = (lane >> 2) & 7, sub32 = lane >> 5, slot = lane & 3, b4 = (lane >> 4) & 1
        #   even row group: row7 sr + 64 sub32 + 16 (slot ^ b4); odd: ... 16 (slot ^ (2 | b4))
        e.valu(f"v_bfe_u32 {vr(t0)}, {vr(lane)}, 2, 3")
        e.valu(f"v_mul_lo_u32 {vr(t0)}, {vr(t0)}, {sr(S_KVSR)}")
        e.valu(f"v_lshl_add_u32 {vr(t0)}, {vr(h)}, 6, {vr(t0)}")
        e.valu(f"v_and_b32 {vr(t1)}, 3, {vr(lane)}")
        e.valu(f"v_bfe_u32 {vr(t2)}, {vr(lane)}, 4, 1")
        e.valu(f"v_xor_b32 {vr(t1)}, {vr(t1)}, {vr(t2)}")
        e.valu(f"v_lshl_add_u32 {vr(DMA[0])}, {vr(t1)}, 4, {vr(t0)}")
        e.valu(f"v_xor_b32 {vr(t1)}, 2, {vr(t1)}")
        e.valu(f"v_lshl_add_u32 {vr(DMA[1])}, {vr(t1)}, 4, {vr(t0)}")
        # ---- Q first (the first MFMA needs it and K(0)): this wave's 64 query rows = two 32-row tiles by LDS-DMA into its quarter of the
        #      staging slab (its own slab: idle from its epilogue's last read to its next one), 16 pieces of 1 KiB; the fragments (query 32 b + r, k = 16 kk + 8 h ..) are then row
        #      reads of those tiles. (First form: 16 buffer_load_dwordx4 per wave straight into registers - a lane's 16 bytes of a 256-byte
        #      row each, 64 separate requests per instruction and 4096 per workgroup through the CU's one address unit.)
        qe, qo = T[8], T[9]
        e.valu(f"v_bfe_u32 {vr(t0)}, {vr(lane)}, 2, 3")
        e.valu(f"v_mul_lo_u32 {vr(t0)}, {vr(t0)}, {sr(S_QSR)}")
        e.valu(f"v_lshl_add_u32 {vr(t0)}, {vr(h)}, 6, {vr(t0)}")
        e.valu(f"v_and_b32 {vr(t1)}, 3, {vr(lane)}")
        e.valu(f"v_bfe_u32 {vr(t2)}, {vr(lane)}, 4, 1")
        e.valu(f"v_xor_b32 {vr(t1)}, {vr(t1)}, {vr(t2)}")
        e.valu(f"v_lshl_add_u32 {vr(qe)}, {vr(t1)}, 4, {vr(t0)}")
        e.valu(f"v_xor_b32 {vr(t1)}, 2, {vr(t1)}")
        e.valu(f"v_lshl_add_u32 {vr(qo)}, {vr(t1)}, 4, {vr(t0)}")
        e.salu(f"s_lshl_b32 {sr(S_TMP)}, {sr(S_WID)}, 6")
        e.salu(f"s_mul_i32 {sr(S_X0)}, {sr(S_TMP)}, {sr(S_QSR)}")                # this wave's first query row, bytes
        e.salu(f"s_lshl_b32 {sr(S_X1)}, {sr(S_QSR)}, 3")                         # 8 rows further
        for g in range(8):                                                         # (into the wave's OWN slab, S_STAGE: a wave that is early must not write where a late one still reads its O rows)
            for half in range(2 if self.D == 128 else 1):
                e.salu(f"s_add_u32 m0, {sr(S_STAGE)}, {2048 * g + 1024 * half - 128 * half}")
                e.salu("s_nop 0")
                o = " offset:128" if half else ""
                e.out.append(Ins(f"buffer_load_dwordx4 {vr(qo if g & 1 else qe)}, {sr(Q_SRD, 4)}, {sr(S_X0)} offen{o} lds", "dma", V(qo if g & 1 else qe)))
            e.salu(f"s_add_u32 {sr(S_X0)}, {sr(S_X0)}, {sr(S_X1)}")
        # ---- tile 0 of K into slot 0
        e.raw("; ---- K(0) -> slot 0; then K(1) -> slot 1 and V(0) -> slot 0")
        e.salu(f"s_mov_b32 {sr(S_KOFF0)}, {sr(S_RB0)}")
        e.salu(f"s_lshl_b32 {sr(S_TMP2)}, {sr(S_KVSR)}, 3")
        e.salu(f"s_add_u32 {sr(S_KOFF1)}, {sr(S_KOFF0)}, {sr(S_TMP2)}")
        for i in (range(4) if self.D == 128 else (0, 2)):
            self.dma(K_SRD, DMA[i >> 1], S_KOFF0 + (i >> 1), S_M0K, 2048 * (i >> 1) + 896 * (i & 1), 128 * (i & 1))
        # K(1) (clamped) and V(0)
        e.salu(f"s_min_u32 {sr(S_TMP)}, 1, {sr(S_TM1)}")
        e.salu(f"s_mul_i32 {sr(S_TMP)}, {sr(S_TMP)}, {sr(S_TSTEP)}")
        e.salu(f"s_add_u32 {sr(S_KOFF0)}, {sr(S_TMP)}, {sr(S_RB0)}")
        e.salu(f"s_add_u32 {sr(S_KOFF1)}, {sr(S_KOFF0)}, {sr(S_TMP2)}")
        e.salu(f"s_mov_b32 {sr(S_VOFF0)}, {sr(S_RB0)}")
        e.salu(f"s_add_u32 {sr(S_VOFF1)}, {sr(S_VOFF0)}, {sr(S_TMP2)}")
        e.salu(f"s_xor_b32 {sr(S_M0K)}, {sr(S_M0K)}, {KSLOT}")
        for i in (range(4) if self.D == 128 else (0, 2)):
            self.dma(K_SRD, DMA[i >> 1], S_KOFF0 + (i >> 1), S_M0K, 2048 * (i >> 1) + 896 * (i & 1), 128 * (i & 1))
        for i in (range(4) if self.D == 128 else (0, 2)):
            self.dma(V_SRD, DMA[i >> 1], S_VOFF0 + (i >> 1), S_M0V, 2048 * (i >> 1) + 896 * (i & 1), 128 * (i & 1))
        e.salu(f"s_xor_b32 {sr(S_M0K)}, {sr(S_M0K)}, {KSLOT}")                # iteration 0 stages K(2) into slot 0 ...
        e.salu(f"s_xor_b32 {sr(S_M0V)}, {sr(S_M0V)}, {KSLOT}")                # ... and V(1) into slot 1
        # offsets for iteration 0's DMA: K(2), V(1), clamped
        for (dst0, dst1, ahead) in ((S_KOFF0, S_KOFF1, 2), (S_VOFF0, S_VOFF1, 1)):
            e.salu(f"s_min_u32 {sr(S_TMP)}, {ahead}, {sr(S_TM1)}")
            e.salu(f"s_mul_i32 {sr(S_TMP)}, {sr(S_TMP)}, {sr(S_TSTEP)}")
            e.salu(f"s_add_u32 {sr(dst0)}, {sr(S_TMP)}, {sr(S_RB0)}")
            e.salu(f"s_add_u32 {sr(dst1)}, {sr(dst0)}, {sr(S_TMP2)}")
        e.raw("; ---- O = 0, running maximum = -inf, row sums = 0")
        for i in range(32 * self.NDB):
            e.valu(f"v_accvgpr_write_b32 {ar(i)}, 0", (), A(i))
        for b in range(2):
            e.valu(f"v_mov_b32 {vr(MC[b])}, {vr(NEGINF)}")
            e.valu(f"v_mov_b32 {vr(LA[b])}, 0")
            e.valu(f"v_mov_b32 {vr(LB[b])}, 0")
        # ---- the Q fragments out of the slab: straight into a[128:191] (exact scores), or through the score registers where they are multiplied by
        #      scale log2(e) and rounded to the element type once (scaled query: the score MFMAs then deliver exponents)
        e.wait(vm=12 if self.D == 128 else 6)     # in-order counter: this wave's 16 (8) Q pieces are older than its 12 (6) K / V pieces; the quarter is its own: no barrier
        qb = (T[10], T[11])
        e.salu(f"s_sub_u32 {sr(S_TMP)}, {sr(S_STAGE)}, {sr(S_LDS)}")           # (S_TMP2 holds 8 key rows' bytes from here to the end of the loop)
        for i in range(2):
            e.valu(f"v_add_u32 {vr(qb[i])}, {sr(S_TMP)}, {vr(KB[i])}")           # (the K bases sit at slot 0 here: S_LDS + the lane's row / chunk)
        for b in range(2):
            for kk in range(self.NKK):
                if self.scaled:
                    e.out.append(Ins(f"ds_read_b128 {vr(self.QF(b, kk) - 128, 4)}, {vr(qb[kk & 1])} offset:{8192 * b + 512 * (kk >> 1)}", "lds", V(qb[kk & 1]), V(self.QF(b, kk) - 128, 4)))
                else:
                    e.out.append(Ins(f"ds_read_b128 {ar(self.QF(b, kk), 4)}, {vr(qb[kk & 1])} offset:{8192 * b + 512 * (kk >> 1)}", "lds", V(qb[kk & 1]), A(self.QF(b, kk), 4)))
        e.wait(lgkm=0)
        if not self.scaled:
            pass
        elif self.f16:
            c2 = T[8]
            e.valu(f"v_cvt_f16_f32 {vr(c2)}, {sr(S_C)}")
            e.valu(f"v_pack_b32_f16 {vr(c2)}, {vr(c2)}, {vr(c2)}")
            for i in range(64):
                e.valu(f"v_pk_mul_f16 {vr(i)}, {vr(i)}, {vr(c2)}")
                e.valu(f"v_accvgpr_write_b32 {ar(128 + i)}, {vr(i)}", V(i), A(128 + i))
        else:
            for i in range(64):
                hi, lo = T[8 + 2 * (i & 3)], T[9 + 2 * (i & 3)]
                e.valu(f"v_and_b32 {vr(hi)}, 0xffff0000, {vr(i)}")
                e.valu(f"v_lshlrev_b32 {vr(lo)}, 16, {vr(i)}")
                e.valu(f"v_mul_f32 {vr(hi)}, {sr(S_C)}, {vr(hi)}")
                e.valu(f"v_mul_f32 {vr(lo)}, {sr(S_C)}, {vr(lo)}")
                e.valu(f"{self.cvt} {vr(i)}, {vr(lo)}, {vr(hi)}")
                e.valu(f"v_accvgpr_write_b32 {ar(128 + i)}, {vr(i)}", V(i), A(128 + i))
        # ---- K(0) has landed for everyone: its fragments; then the ring bases move to where iteration 0 expects them
        e.wait(vm=8 if self.D == 128 else 4)      # in-order counter: the Q loads and the K(0) pieces are older than the 4 + 4 (2 + 2) pieces of K(1) and V(0)
        e.barrier()
        for sub in range(2):
            for kk in range(self.NKK):
                self.lds_k(sub, kk)
        e.salu(f"s_mov_b32 {sr(S_IT)}, 0")
        self.stamp(0)

    def dispatch(self):
        """Head of every iteration: which variant does this wave run at tile S_IT?  it > dt + 1: idle; == dt + 1: drain; == dt: masked;
        == 0: first; else steady. The loop ends after iteration T (the last wave's drain)."""
        e = self
        e.salu("s_branch L_loop_%=")
        e.label("L_tail_%=")      # it >= dt: the diagonal tile, the drain, then idle iterations
        e.salu(f"s_cmp_eq_u32 {sr(S_IT)}, {sr(S_DT)}")
        e.salu("s_cbranch_scc1 L_diag_%=")
        e.salu(f"s_add_u32 {sr(S_TMP)}, {sr(S_DT)}, 1")
        e.salu(f"s_cmp_eq_u32 {sr(S_IT)}, {sr(S_TMP)}")
        e.salu("s_cbranch_scc1 L_drain_%=")
        e.salu("s_branch L_idle_%=")
        e.label("L_diag_%=")
        e.salu(f"s_cmp_eq_u32 {sr(S_IT)}, 0")
        e.salu("s_cbranch_scc1 L_firstmasked_%=")
        e.salu("s_branch L_masked_%=")
        for _ in range(getattr(self, "pad", 0)):
            e.salu("s_nop 0")     # (code-placement experiment: shifts the steady body by 4 bytes per pad)
        e.label("L_loop_%=")      # the common case first: 0 < it < dt falls through into the steady body
        e.salu(f"s_cmp_ge_u32 {sr(S_IT)}, {sr(S_DT)}")
        e.salu("s_cbranch_scc1 L_tail_%=")
        e.salu(f"s_cmp_eq_u32 {sr(S_IT)}, 0")
        e.salu("s_cbranch_scc1 L_first_%=")
        if self.mutant:
            e.salu(f"s_cmp_eq_u32 {sr(S_IT)}, {sr(S_MUT)}")
            e.salu("s_cbranch_scc1 L_steadydrop_%=")

    def next_iter(self, steady=False):
        e = self
        e.salu(f"s_add_u32 {sr(S_IT)}, {sr(S_IT)}, 1")
        if steady and not self.mutant and not os.environ.get("KF_GEN_NO_STEADY_LOOP"):
            # round 6: behind a steady tile comes a steady tile until the wave's diagonal one (it == dt <= T - 1: the general dispatch's tail) - three
            # scalar instructions instead of seven per tile
            e.salu(f"s_cmp_lt_u32 {sr(S_IT)}, {sr(S_DT)}")
            e.salu("s_cbranch_scc1 L_steady_%=")
            e.salu("s_branch L_tail_%=")
            return
        e.salu(f"s_cmp_le_u32 {sr(S_IT)}, {sr(S_T)}")
        e.salu("s_cbranch_scc1 L_loop_%=")
        e.salu("s_branch L_epilogue_%=")

    def epilogue(self):
        e = self
        lane, r, h = T[0], T[1], T[2]
        e.label("L_epilogue_%=")
        # (round 6: no vmcnt here. What is still in flight - the pieces of the tiles fetched past the block's end - lands in this wave's OWN quarters of
        #  the ring, which nothing reads any more, in order ahead of whatever the next pass requests into them; the epilogue works in registers and in
        #  the wave's slab. KF_GEN_EPI_VMWAIT=1 restores the wait: same-box A/B.)
        if os.environ.get("KF_GEN_EPI_VMWAIT"):
            e.wait(vm=0, lgkm=0)
        else:
            e.wait(lgkm=0)
        e.salu("s_nop 15")
        e.valu(f"v_mbcnt_lo_u32_b32 {vr(lane)}, -1, 0")
        e.valu(f"v_mbcnt_hi_u32_b32 {vr(lane)}, -1, {vr(lane)}")
        e.valu(f"v_and_b32 {vr(r)}, 31, {vr(lane)}")
        e.valu(f"v_lshrrev_b32 {vr(h)}, 5, {vr(lane)}")
        inv = (T[3], T[4])
        for b in range(2):
            l, t = T[5], T[6]
            e.valu(f"v_add_f32 {vr(l)}, {vr(LA[b])}, {vr(LB[b])}")
            e.valu(f"v_mov_b32 {vr(t)}, {vr(l)}")
            e.salu("s_nop 1")
            e.valu(f"v_permlane32_swap_b32 {vr(l)}, {vr(t)}")
            e.valu(f"v_add_f32 {vr(l)}, {vr(l)}, {vr(t)}")                   # the row sum of the query: both key halves
            e.valu(f"v_rcp_f32 {vr(inv[b])}, {vr(l)}")
            e.valu(f"v_log_f32 {vr(l)}, {vr(l)}")
            e.salu("s_nop 0")
            e.valu(f"v_add_f32 {vr(l)}, {vr(l)}, {vr(MC[b])}")
            e.valu(f"v_mul_f32 {vr(l)}, 0x3f317218, {vr(l)}")                 # ln 2: LSE in natural-log units
            dbg = __import__("os").environ.get("KF_GEN_DBG", "")
            if dbg:  # debugging aid: raw state instead of the LSE (l: row sum; mc; la / lb: the lane's partial sums; *_hi: the upper lane half's)
                src = {"mc": MC[b], "la": LA[b], "lb": LB[b], "nmc": NMC[b], "s0": S(b, 0), "s16": S(b, 16), "s31": S(b, 31)}.get(dbg.replace("_hi", ""))
                if dbg == "l":
                    e.valu(f"v_rcp_f32 {vr(l)}, {vr(inv[b])}")
                    e.salu("s_nop 0")
                else:
                    e.valu(f"v_mov_b32 {vr(l)}, {vr(src)}")
                    if dbg.endswith("_hi"):
                        e.valu(f"v_mov_b32 {vr(t)}, {vr(src)}")
                        e.salu("s_nop 1")
                        e.valu(f"v_permlane32_swap_b32 {vr(l)}, {vr(t)}")
                        e.valu(f"v_mov_b32 {vr(l)}, {vr(t)}")
            # lanes 0..31 store the LSE of query 64 w + 32 b + r
            e.valu(f"v_lshlrev_b32 {vr(t)}, 2, {vr(r)}")
            e.salu(f"s_lshl_b32 {sr(S_TMP)}, {sr(S_WID)}, 8")
            e.salu(f"s_add_u32 {sr(S_TMP)}, {sr(S_TMP)}, {128 * b}")
            e.valu(f"v_add_u32 {vr(t)}, {sr(S_TMP)}, {vr(t)}")
            e.salu(f"s_cmp_eq_u64 {sr(S_LSE, 2)}, 0")
            e.salu(f"s_cbranch_scc1 L_nolse{b}_%=")
            if b == 0:   # through a descriptor of the block's valid rows (the K descriptor's registers: K is done with): rows beyond Sq are not stored
                e.salu(f"s_mov_b64 {sr(K_SRD, 2)}, {sr(S_LSE, 2)}")
                e.salu(f"s_mov_b32 {sr(K_SRD + 2)}, %[lsen]")
            e.salu("s_mov_b32 exec_hi, 0")
            e.out.append(Ins(f"buffer_store_dword {vr(l)}, {vr(t)}, {sr(K_SRD, 4)}, 0 offen", "vmem", V(t) + V(l)))
            e.salu("s_mov_b64 exec, -1")
            e.label(f"L_nolse{b}_%=")
        # O^T accumulators -> 16-bit rows of this wave's staging slab: lane (r, h) writes 4 consecutive d of query 32 b + r
        st = T[7]
        e.valu(f"v_mul_u32_u24 {vr(st)}, {STAGE_ROW}, {vr(r)}")
        e.valu(f"v_lshl_add_u32 {vr(st)}, {vr(h)}, 3, {vr(st)}")
        e.valu(f"v_add_u32 {vr(st)}, {sr(S_STAGE)}, {vr(st)}")
        for b in range(2):
            for db in range(self.NDB):
                for gq in range(4):
                    a0 = self.OA(b, db) + 4 * gq
                    x = T[8:12]
                    for j in range(4):
                        e.valu(f"v_accvgpr_read_b32 {vr(x[j])}, {ar(a0 + j)}")
                    for j in range(4):
                        e.valu(f"v_mul_f32 {vr(x[j])}, {vr(x[j])}, {vr(inv[b])}")
                    e.valu(f"{self.cvt} {vr(x[0])}, {vr(x[0])}, {vr(x[1])}")
                    e.valu(f"{self.cvt} {vr(x[1])}, {vr(x[2])}, {vr(x[3])}")
                    e.out.append(Ins(f"ds_write_b64 {vr(st)}, {vr(x[0], 2)} offset:{32 * b * STAGE_ROW + 64 * db + 16 * gq}", "lds"))
        e.wait(lgkm=0)   # the same wave reads back what it wrote: LDS operations of one wave complete in order
        # rows out: 4 rows per instruction (16 lanes x 16 B each)
        rd, oo = T[12], T[13]
        e.valu(f"v_lshrrev_b32 {vr(rd)}, {4 if self.D == 128 else 3}, {vr(lane)}")      # 16 lanes per 256-byte row | 8 lanes per 128-byte row
        e.valu(f"v_and_b32 {vr(oo)}, {15 if self.D == 128 else 7}, {vr(lane)}")
        e.valu(f"v_lshlrev_b32 {vr(oo)}, 4, {vr(oo)}")
        e.valu(f"v_mul_lo_u32 {vr(T[14])}, {vr(rd)}, {sr(S_OSR)}")
        e.valu(f"v_mul_u32_u24 {vr(rd)}, {STAGE_ROW}, {vr(rd)}")
        e.valu(f"v_add3_u32 {vr(rd)}, {vr(rd)}, {vr(oo)}, {sr(S_STAGE)}")
        e.valu(f"v_add_u32 {vr(oo)}, {vr(oo)}, {vr(T[14])}")
        e.salu(f"s_lshl_b32 {sr(S_TMP)}, {sr(S_WID)}, 6")
        e.salu(f"s_mul_i32 {sr(S_X0)}, {sr(S_TMP)}, {sr(S_OSR)}")
        rows_per = 4 if self.D == 128 else 8                                        # rows one store instruction covers
        e.salu(f"s_lshl_b32 {sr(S_X1)}, {sr(S_OSR)}, {2 if self.D == 128 else 3}")
        for j in range(64 // rows_per):
            d = 100 + 4 * (j % 8)   # v[100..131]: the fragment registers are free now
            e.out.append(Ins(f"ds_read_b128 {vr(d, 4)}, {vr(rd)} offset:{rows_per * j * STAGE_ROW}", "lds"))
            if j % 8 == 7:
                e.wait(lgkm=0)
                for i in range(8):
                    e.out.append(Ins(f"buffer_store_dwordx4 {vr(100 + 4 * i, 4)}, {vr(oo)}, {sr(O_SRD, 4)}, {sr(S_X0)} offen", "vmem"))
                    e.salu(f"s_add_u32 {sr(S_X0)}, {sr(S_X0)}, {sr(S_X1)}")
        if self.stamps:
            self.stamp(6)
            e.salu(f"s_lshl_b32 {sr(S_TMP)}, {sr(S_WID)}, 5")                  # 8 dwords per wave
            e.salu(f"s_add_u32 {sr(S_TMP)}, {sr(S_TMP)}, %[dbgoff]")
            e.valu(f"v_mov_b32 {vr(T[0])}, {sr(S_TMP)}")
            for i in range(8):
                e.valu(f"v_mov_b32 {vr(100 + i)}, s{84 + i}")
            e.salu("s_mov_b64 exec, 1")
            e.out.append(Ins(f"global_store_dwordx4 {vr(T[0])}, {vr(100, 4)}, %[dbg]", "vmem"))
            e.out.append(Ins(f"global_store_dwordx4 {vr(T[0])}, {vr(104, 4)}, %[dbg] offset:16", "vmem"))
            e.salu("s_mov_b64 exec, -1")
        # no wait and no barrier here: nobody reads the ring after the last iteration's barrier, a wave's DMA pieces always land in its
        # own quarter of a slot (in order behind its older ones), the slab is the wave's own, and the stores drain under the next block's prologue

    def build(self):
        self.prologue()
        self.dispatch()
        for name, kw in (("steady", dict(has_prev=True, has_cur=True)), ("first", dict(has_prev=False, has_cur=True)),
                         ("masked", dict(has_prev=True, has_cur=True, masked=True)), ("firstmasked", dict(has_prev=False, has_cur=True, masked=True)),
                         ("drain", dict(has_prev=True, has_cur=False)), ("idle", dict(has_prev=False, has_cur=False))):
            self.iteration(name, **kw)
            self.next_iter(steady=(name == "steady"))
        if self.mutant:
            self.iteration("steadydrop", has_prev=True, has_cur=True, drop=True)
            self.next_iter()
        self.epilogue()
        self.fix_trans_use()
        return self

    def fix_trans_use(self):
        """gfx950 does not interlock a transcendental result against the VERY NEXT instruction reading it: the lanes with
        (lane & 4) == 0 get the old register value (tools/scratch/trans_hazard.hip, profiles/r04_trans_hazard.txt). One wait state
        is what the hardware needs (LLVM's hasTransForwardingHazard; hipcc inserts it for compiled code, never inside inline asm).
        Behind an MFMA the stream never has the two as neighbours; where gaps hold no MFMA (drain, a first tile's empty slot,
        prologue, epilogue, the rare rescale) they can be: `s_nop 1` (one more than needed) goes between them, and an `s_nop 0`
        that is all that separates them is widened."""
        out, prev, sep = [], None, 0
        for x in self.out:
            if x.kind in ("raw", "label", "nomfma"):
                out.append(x)
                if x.kind == "label": prev = None
                continue
            if prev is not None and x.kind == "salu" and x.text.startswith("s_nop") and sep == 0:
                if int(x.text.split()[1]) < 1:
                    x = Ins("s_nop 1", "salu")
                out.append(x)
                sep = 2
                continue
            if prev is not None and sep == 0 and x.kind in ("valu", "trans") and set(prev.writes) & set(x.reads):
                out.append(Ins("s_nop 1", "salu"))
            out.append(x)
            prev, sep = (x, 0) if x.kind == "trans" else (None, 0)
        self.out = out


# ------------------------------------------------------------------ static checks on the emitted stream
def check(ins):
    """Walks each variant's stream twice (an iteration behind itself) and enforces:
       * an MFMA result is read or overwritten by a non-MFMA instruction only >= 2 MFMAs later (and by an MFMA only as its own
         accumulate chain or >= 2 MFMAs later);
       * a VALU result feeds an MFMA's A / B / C only with >= 4 instructions in between;
       * an LDS read's destination is used only behind an s_waitcnt lgkmcnt(0) that follows the read."""
    problems = []
    # split into variants
    cur, variants = None, {}
    for i in ins:
        if i.kind == "label" and re.match(r"L_(steady|first|masked|firstmasked|drain|idle|steadydrop)_%=:", i.text):
            cur = i.text[2:-4]
            variants[cur] = []
        elif i.kind == "label" and i.text.startswith("L_epilogue"):
            cur = None
        elif cur is not None:
            variants[cur].append(i)
    for name, body in variants.items():
        seq = [x for x in body if x.kind not in ("raw", "label", "nomfma")] * 2
        last_mfma_write, last_valu_write, pending_lds = {}, {}, {}
        n_mfma, n_ins = 0, 0
        prev, tws = None, 0
        for x in seq:
            if x.kind == "salu" and x.text.startswith("s_nop") and prev is not None:
                tws += int(x.text.split()[1]) + 1
            elif prev is not None and x.kind in ("valu", "trans") and set(prev.writes) & set(x.reads) and tws < 2:
                problems.append(f"{name}: '{x.text}' reads the result of '{prev.text}' {tws} wait state(s) behind it (a transcendental needs 2)")
            if not (x.kind == "salu" and x.text.startswith("s_nop")):
                prev, tws = (x, 0) if x.kind == "trans" else (None, 0)
            n_ins += 1
            if x.kind == "mfma":
                n_mfma += 1
            if x.kind == "salu" and x.text.startswith("s_nop"):
                n_ins += int(x.text.split()[1])
            if x.kind == "wait" and "lgkm" in x.tag:
                pending_lds.clear()
            for reg in x.reads:
                if reg in pending_lds:
                    problems.append(f"{name}: '{x.text}' uses {reg} of an LDS read not yet waited for")
                if reg in last_mfma_write:
                    m, tag, at = last_mfma_write[reg]
                    chain = x.kind == "mfma" and reg in x.writes and n_mfma - m == 1
                    if not chain and n_mfma - m < 2 and n_ins - at < 22:
                        problems.append(f"{name}: '{x.text}' reads {reg} {n_mfma - m} MFMA(s) / {n_ins - at} wait states after '{tag}' wrote it")
                if x.kind == "mfma" and reg in last_valu_write and n_ins - last_valu_write[reg] < 4:
                    problems.append(f"{name}: '{x.text}' reads {reg} {n_ins - last_valu_write[reg]} instruction(s) after a VALU wrote it")
            for reg in x.writes:
                if x.kind != "mfma" and reg in last_mfma_write and n_mfma - last_mfma_write[reg][0] < 2 and n_ins - last_mfma_write[reg][2] < 22:
                    problems.append(f"{name}: '{x.text}' overwrites {reg} right behind '{last_mfma_write[reg][1]}'")
                if x.kind == "mfma":
                    last_mfma_write[reg] = (n_mfma, x.tag, n_ins)
                    last_valu_write.pop(reg, None)
                elif x.kind == "lds":
                    pending_lds[reg] = n_ins
                    last_mfma_write.pop(reg, None)
                else:
                    last_valu_write[reg] = n_ins
                    last_mfma_write.pop(reg, None)
    return problems


def gap_table(ins, name="steady"):
    """Fillers per MFMA gap of one variant: the issue-cost estimate the placement is judged by (MI355X_MICROARCH.md: MFMA 8, v_exp 8,
    other VALU 4, LDS read ~2, DMA piece ~60, SALU / wait 4)."""
    cost = {"valu": 4, "trans": 8, "lds": 2, "dma": 60, "salu": 4, "wait": 4, "barrier": 4, "vmem": 8}
    rows, cur, on = [], None, False
    for i in ins:
        if i.kind == "label":
            if re.match(r"L_(steady|first|masked|firstmasked|drain|idle|steadydrop|epilogue|loop|diag)_%=:", i.text):
                on = i.text == f"L_{name}_%=:"
            continue
        if not on or i.tag == "rare":
            continue
        if i.kind in ("mfma", "nomfma"):
            cur = {"mfma": i.tag, "n": 0, "cycles": 8 if i.kind == "mfma" else 0}
            rows.append(cur)
        elif cur is not None and i.kind in cost:
            cur["n"] += 1
            cur["cycles"] += cost[i.kind]
    return rows


# ------------------------------------------------------------------ address-map self test (pure Python model of the LDS image)
def selftest():
    """The three views of one tile image must agree: what the DMA pieces write, what the K row reads deliver as the A operand of
    v_mfma_32x32x16 (lane (r, h): A[row r][k = 8 h + j]), and what the transposed reads deliver as the V^T operand in the k order of
    a score accumulator used as the B operand (element j of half h <-> key 16 ks + 8 (j >> 2) + 4 h + (j & 3))."""
    sr_b = 256 + 64   # a row stride that is not the tile's own
    lds = {}
    for w in range(4):
        for rgi, rg in enumerate((2 * w, 2 * w + 1)):
            for hp in range(2):
                base = 4096 * w + 2048 * rgi + 1024 * hp
                for L in range(64):
                    row7, sub32, slot, b4 = (L >> 2) & 7, L >> 5, L & 3, (L >> 4) & 1
                    x = b4 if rg % 2 == 0 else (2 | b4)
                    src = 8 * rg * sr_b + row7 * sr_b + 64 * sub32 + 16 * (slot ^ x) + 128 * hp   # soffset + voffset + inst offset
                    row, colbyte = src // sr_b, src % sr_b
                    for byte in range(0, 16, 2):
                        lds[base + 16 * L + byte] = (row, (colbyte + byte) // 2)
    assert len(lds) == 64 * 128
    # K row reads
    for sub in range(2):
        for kk in range(8):
            for lane in range(64):
                r, h = lane & 31, lane >> 5
                b0 = 2048 * (r >> 3) + 64 * (r & 7) + 16 * (h ^ ((r >> 2) & 3))
                addr = (b0 ^ (32 if kk & 1 else 0)) + 8192 * sub + 512 * (kk >> 1)
                for j in range(8):
                    assert lds[addr + 2 * j] == (32 * sub + r, 16 * kk + 8 * h + j), ("K", sub, kk, lane, j, lds[addr + 2 * j])
    # V transposed reads
    for ks in range(4):
        for db in range(4):
            for sec in range(2):
                sub, s = ks >> 1, ks & 1
                imm = 2048 * (4 * sub + 2 * s + sec) + 512 * db
                got = {}
                for grp in range(4):
                    blk = {}
                    for i in range(16):
                        lane = 16 * grp + i
                        h, q, p, g1 = lane >> 5, i >> 2, i & 3, grp & 1
                        t0 = 64 * (4 * h + q) + 16 * ((2 * g1 + (p >> 1)) ^ h) + 8 * (p & 1)
                        addr = (t0 ^ (32 if sec else 0)) + imm
                        for c in range(4):
                            blk[(q, 4 * p + c)] = lds[addr + 2 * c]
                    for i in range(16):
                        got[16 * grp + i] = [blk[(qq, i)] for qq in range(4)]
                for lane in range(64):
                    r, h = lane & 31, lane >> 5
                    for e in range(4):
                        j = 4 * sec + e
                        key = 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3)
                        assert got[lane][e] == (key, 32 * db + r), ("V", ks, db, sec, lane, e, got[lane][e], (key, 32 * db + r))
    return True


# m0 is written by every DMA piece; exec is all ones on entry (a full 256-thread block, no divergence in the wrapper) and is restored to -1
# wherever the stream narrows it (lse store): the block assumes and leaves exec == -1
CLOBBERS = (["memory", "vcc", "scc", "m0"] + [f"s{i}" for i in range(36, 80)] + [f"v{i}" for i in range(N_VGPR)] + [f"a{i}" for i in range(256)])


def render(ins):
    lines = []
    for i in ins:
        if i.kind == "nomfma":
            continue
        if i.kind == "raw" and i.text.startswith("#"):
            lines.append(i.text)
        elif i.kind == "raw":
            lines.append(f'    "{i.text}\\n"')
        elif i.kind == "label":
            lines.append(f'    "{i.text}\\n"')
        else:
            lines.append(f'    "\\t{i.text}\\n"')
    return "\n".join(lines)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check-only", action="store_true")
    ap.add_argument("--gaps", default="", help="print the gap table of a variant")
    ap.add_argument("--out", default=str(OUT))
    ap.add_argument("--pad", type=int, default=0, help="s_nop 0 x N in front of the loop head (code-placement experiment)")
    ap.add_argument("--stamps", action="store_true", help="diagnostic build: s_memtime stamps at the slot boundaries (needs -DKF_FWD_W4_STAMPS)")
    ap.add_argument("--ablate", default="", help="comma list of dma, valu, lds, barrier: leave that part of the tile body out (timing experiments; wrong results)")
    ap.add_argument("--scaled", action="store_true", help="--gaps / --check-only look at the scaled-query stream (the file always holds both)")
    args = ap.parse_args()
    abl = tuple(x for x in args.ablate.split(",") if x)
    assert selftest()
    Gen.pad = args.pad
    g = Gen(False, ablate=abl, stamps=args.stamps, scaled=args.scaled).build()
    probs = check(g.out) + ([] if args.scaled else check(Gen(False, ablate=abl, stamps=args.stamps, scaled=True).build().out))
    if args.gaps:
        tot = 0
        for k, row in enumerate(gap_table(g.out, args.gaps)):
            tot += max(32, row["cycles"]) if row["mfma"] else row["cycles"]
            print(f"{k:3d} {row['mfma']:20s} fillers {row['n']:2d}  est. cycles {row['cycles']:4d}")
        print("estimated cycles per iteration:", tot)
    for p in probs[:40]:
        print("HAZARD:", p, file=sys.stderr)
    if probs and not abl:   # (an ablated stream is a timing experiment: its results are wrong anyway)
        return 1
    if args.check_only:
        return 0
    texts = {}
    for f16 in (False, True):
        for mut in (False, True):
            for sq in (False, True):
                gg = Gen(f16, mut, ablate=abl, stamps=args.stamps, scaled=sq).build()
                assert abl or not check(gg.out), check(gg.out)[:5]
                texts[(f16, mut, sq)] = render(gg.out).replace(chr(10), " " + chr(92) + chr(10))
    texts64 = {}
    for f16 in (False, True):
        for mut in (False, True):
            gg = Gen(f16, mut, ablate=abl, D=64).build()
            assert abl or not check(gg.out), check(gg.out)[:5]
            texts64[(f16, mut)] = render(gg.out).replace(chr(10), " " + chr(92) + chr(10))
    n_ins = sum(1 for i in g.out if i.kind not in ("raw", "label", "nomfma"))
    def four(mut):
        return "\n".join([f"#define KF_FWD_W4_ASM_{'F16' if f16 else 'BF16'}{'_SQ' if sq else ''} \\\n{texts[(f16, mut, sq)]}" for f16 in (False, True) for sq in (False, True)] +
                         [f"#define KF_FWD_W4_D64_ASM_{'F16' if f16 else 'BF16'} \\\n{texts64[(f16, mut)]}" for f16 in (False, True)])
    text = f"""// GENERATED by tools/gen_attn_fwd.py - do not edit; edit the generator and run it again.
// The 16-bit causal-attention forward of one 256-row query block as ONE instruction stream per element type ({n_ins} instructions):
// 4 waves x 64 query rows, one wave per SIMD, all 512 registers asm-owned; see the generator's header for the structure.
// Two forms: exact f32 scores (default), and _SQ = the query scaled and rounded once per pass (KF_ATTN_SCALED_OPERANDS: faster, less exact).
// _D64_ = head size 64 (exact form only): a 64-key tile is 32 MFMAs, the exponent chain runs three scores per gap.
#pragma once
#define KF_FWD_W4_LDS_BYTES {LDS_BYTES}
#define KF_FWD_W4_CLOBBERS {", ".join('"' + c + '"' for c in CLOBBERS + ([f"s{i}" for i in range(80, 92)] if args.stamps else []))}
#ifdef KF_MUTANT  // + one more variant of the tile body: the probabilities of tile %[mut] are dropped (tests/test_gpu_attention_mutants.py)
{four(True)}
#else
{four(False)}
#endif
"""
    Path(args.out).write_text(text)
    print(f"wrote {args.out} ({n_ins} instructions per element type)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
