#!/usr/bin/env python3
"""Generator of the hand-placed instruction stream of attn_fwd16_p5 (csrc/attn_fwd16_p5.h): forward attention for
128 < D <= 256 with 16-bit Q/K/V.

A workgroup is four waves x 64 query rows (one wave per SIMD, the whole 512-entry register file); it walks the keys in
32-key steps.  Per step j (parity = j & 1) and wave, 64 matrix instructions (32x32x16) in two phases, plus the extra k-step
of the FOLD streams:

    A(j): S(j)^T = K(j) Q'^T         | exp2 / row sum / 16-bit pack (in place) of step j-1
    B(j): O^T += V(j-1)^T P(j-1)^T   | row maximum of S(j), deferred-rescale decision, (exact streams: s * scale2 - m),
                                       the seam: own LDS-DMA pieces of step j+1 landed, barrier, first K(j+1) fragments

Every K and V^T fragment read from LDS feeds TWO matrix instructions (row blocks rb = 0, 1 of the wave).  The online
softmax is that of tools/p4gen.py (deferred rescale with threshold THR, +Softmax.swift:290-301; FOLD streams subtract the
running maximum inside the matrix pipe through an extra k-step).

Register map:

    a[0:255]    O^T accumulators (rb, db) -> 16 (8 rb + db)          lane = row, registers = head-dimension rows
    v[30:31]    m + THR (exact streams) / rescale temporaries
    v[32:35]    FOLD: A operand of the extra k-step (-1.0 pattern, 0, 0, 0)
    v[36:39]    FOLD: (m pair rb0, 0, m pair rb1, 0): B operands of the extra k-step are v[36:39] and v[38:41]
    v[40:167]   Q' fragments (rb, ks) -> 40 + 4 (16 rb + ks)
    v[168:231]  score blocks (parity, rb) -> 168 + 16 (2 parity + rb); P^T fragments (rb, u) packed in place: + 4 u
    v[232:247]  ring of four A-operand fragments read from LDS (K rows, V^T), fragment i in slot i % 4
    v[248:255]  softmax temporaries: new row maximum, half-swap copies / mask limits, correction factors, second row sums
    v[0:29]     left to hipcc (operands of the statement)

LDS: ring of four stages {K tile | V tile}, each tile [D/32][32 keys][32 elements], 16-byte chunks of a 64-byte row
XOR-swizzled by (key >> 2) & 3; filled by LDS-DMA two steps ahead.  One barrier per step.

Executed by tools/f256sim.py on the lane-exact model of tools/p4sim.py: tests/test_f256_stream.py.
Usage: python tools/f256gen.py   (rewrites metal_flash_attention_amd/csrc/attn_fwd16_p5_stream.inc)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from p4gen import A, F, I, M0, SN, V, VCC, VN, Stream as _P4Stream, render  # noqa: E402

T_THR = T_RS = 30
ONES, MF = 32, (36, 38)
QF, SB, AF = 40, 168, 232
T_MN, T_SW, T_CORR, T_LB = 248, 250, 252, 254
T_TL = T_SW                      # mask limits: only inside the mask section, which runs before the maxima
FIRST_OWNED_VGPR = 30
MASK_VALUE = -(0.875 / 1.44269504089) * 3.402823466e+38   # +Softmax.swift:242-243
NKS = 16                         # k-steps the Q' fragment map is laid out for (D = 256)
STAGE, VIMG, RING = 32768, 16384, 4

INOUT_V = ["m0", "m1", "l0", "l1", "koff0", "koff1", "koff2", "koff3", "voff0", "voff1", "voff2", "voff3", "ka0", "ka1", "ta0", "ta1"]
TMP_S = ["j", "stg", "delta", "deltav", "wr", "pend", "t0", "t1", "pa", "pb", "plast"]
TMP_S64 = ["sv", "ptime"]
IN_V = ["lim0", "lim1", "onesw", "qback"]
IN_S = ["kres", "vres", "nt", "wnt", "scale2", "kinc", "vinc", "wr0", "ringend", "maskfrom"]


class Cfg:
    def __init__(self, dtype="bf16", thr=8.0, fold=0, prof=0, abl=(), D=256, xbp=0, tr=0):
        """D: head-dimension bucket of the code object (160, 192 or 256): D / 16 k-steps per score block, D / 32 blocks of O^T
        per row block; the register map keeps its D = 256 positions (smaller buckets leave the tail of the Q' and O ranges unused)"""
        self.dtype, self.thr, self.fold, self.prof, self.abl = dtype, float(thr), fold, prof, frozenset(abl)
        self.D, self.nks, self.ndb = D, D // 16, D // 32
        self.pw = -(-D // 64)            # LDS-DMA pieces (1 KiB) per wave and operand tile: 32 keys x D x 2 bytes over four waves
        # FOLD streams of the buckets up to 192: the Q' map leaves sixteen registers per row block unused -- they hold -m of the
        # lane's row and START the accumulation of every score block (no extra k-step; m an exact fp32 value), as in
        # tools/p4gen.py.  xbp: pairs of scores per row block pair whose exp2 already runs in phase B (behind the decision)
        self.cinit = bool(fold) and self.nks <= 12
        self.xbp = xbp
        # tr (model-verified, NOT yet behind a kernel -- DESIGN.md 10.4): K and V stored TRANSPOSED ([D][keys]).  A step's tile in
        # the source orientation is [D elements][32 keys x 2 bytes] = the same [D/32][32][64 bytes] image with the roles of its
        # two read recipes exchanged: K row fragments by transposing reads (ka0 / ka1 = rows + 0 / + 8 of a 16-element step; the
        # kernel parks the Q' fragments in the element order they return), V^T fragments as two 8-byte reads of the lane's
        # element row, chunks 2 u and 2 u + 1 at 8 hi (four addresses ta0..ta3).  Whole steps only (C % 32 == 0).
        # tr is a bit mask like tools/p4gen.py's: bit 0 = K, bit 1 = V transposed -- the two halves of a step's fragment list are
        # independent, an operand that is NOT transposed keeps the row-major recipe (and its two addresses).
        self.tr, self.kt, self.vt = tr, bool(tr & 1), bool(tr & 2)
        assert D % 32 == 0 and (self.nks + 2 * self.ndb) % 4 == 0


def q_frag(rb, ks):
    return V(QF + 4 * (16 * rb + ks), 4)


def s_blk(par, rb):
    return V(SB + 16 * (2 * par + rb), 16)


def s_elem(par, rb, r):
    return V(SB + 16 * (2 * par + rb) + r)


def p_frag(par, rb, u):
    return V(SB + 16 * (2 * par + rb) + 4 * u, 4)


def af(i):
    return V(AF + 4 * (i % 4), 4)


def af_half(i, h):
    return V(AF + 4 * (i % 4) + 2 * h, 2)


def cm_blk(rb):
    return V(QF + 64 * rb + 48, 16)     # the tail of row block rb's Q' range (free when D <= 192)


def o_acc(rb, db):
    return A(16 * (8 * rb + db), 16)    # (stride 8 blocks per rb whatever the bucket)


class Stream(_P4Stream):
    def __init__(self, cfg):
        _P4Stream.__init__(self, cfg)
        self.frag_rid = {}

    # ---- fragment i of a step: 0..nks-1 K rows (k-step i) of the step's own tile, then 2 ndb V^T fragments (u, db) of the PREVIOUS tile
    def frag_read(self, i):
        nks, ndb = self.cfg.nks, self.cfg.ndb
        if i < nks and self.cfg.kt:
            off = (i >> 1) * 2048 + (i & 1) * 1024
            self.lds_read("ds_read_b64_tr_b16", af_half(i, 0), VN("ka0"), off, note="K^T rows ks%d" % i)
            self.frag_rid[i] = self.lds_read("ds_read_b64_tr_b16", af_half(i, 1), VN("ka1"), off)
            return
        if i >= nks and self.cfg.vt:
            u, db = divmod(i - nks, ndb)
            off = VIMG + db * 2048
            self.lds_read("ds_read_b64", af_half(i, 0), VN("ta%d" % (2 * u)), off, note="V u%d db%d" % (u, db))
            self.frag_rid[i] = self.lds_read("ds_read_b64", af_half(i, 1), VN("ta%d" % (2 * u + 1)), off)
            return
        if i < nks:
            self.frag_rid[i] = self.lds_read("ds_read_b128", af(i), VN("ka%d" % (i & 1)), (i >> 1) * 2048, note="K rows ks%d" % i)
        else:
            u, db = divmod(i - nks, ndb)
            off = VIMG + db * 2048 + u * 1024
            self.lds_read("ds_read_b64_tr_b16", af_half(i, 0), VN("ta0"), off, note="V^T u%d db%d" % (u, db))
            self.frag_rid[i] = self.lds_read("ds_read_b64_tr_b16", af_half(i, 1), VN("ta1"), off)

    def dma_piece(self, i):
        """piece i of this wave: 0..pw-1 of the K tile, pw..2pw-1 of the V tile"""
        pw = self.cfg.pw
        name, res, base = (("koff%d" % i, "kres", 0) if i < pw else ("voff%d" % (i - pw), "vres", VIMG))
        self.emit("s_add_u32", M0, [SN("wr"), I(base + (i % pw) * 1024)])
        self.emit("buffer_load_dwordx4_lds", None, [VN(name), SN(res, 4)])

    def dma_advance(self, i):
        pw = self.cfg.pw
        name, inc = ("koff%d" % i, "kinc") if i < pw else ("voff%d" % (i - pw), "vinc")
        self.emit("v_add_u32_e64", VN(name), [VN(name), SN(inc)], clamp=1)

    def wr_advance(self):
        self.emit("s_add_u32", SN("wr"), [SN("wr"), I(STAGE)])
        self.emit("s_cmp_ge_u32", None, [SN("wr"), SN("ringend")])
        self.emit("s_cselect_b32", SN("t1"), [I(RING * STAGE), I(0)])
        self.emit("s_sub_u32", SN("wr"), [SN("wr"), SN("t1")])

    def stage_delta(self):
        """delta = the step of the K read addresses to the next stage; the V^T addresses follow one step behind (deltav)"""
        self.emit("s_mov_b32", SN("deltav"), [SN("delta")])
        self.emit("s_add_u32", SN("stg"), [SN("stg"), I(1)])
        self.emit("s_and_b32", SN("stg"), [SN("stg"), I(RING - 1)])
        self.emit("s_cmp_eq_u32", None, [SN("stg"), I(0)])
        self.emit("s_cselect_b32", SN("t1"), [I(RING * STAGE), I(0)])
        self.emit("s_sub_u32", SN("delta"), [I(STAGE), SN("t1")])

    # ------------------------------------------------------------ phase A
    def phase_a(self, par, mfma, softmax, zero_o=False, dma=True):
        """A(j), par = j & 1: S[par] = K(j) Q'^T | finish-softmax of S[par ^ 1]"""
        cfg = self.cfg
        prev = par ^ 1
        mm = []
        if cfg.fold and not cfg.cinit:
            for rb in range(2):
                mm.append((s_blk(par, rb), V(ONES, 4), V(MF[rb], 4), I(0), None))
        nks, ndb, npc = cfg.nks, cfg.ndb, 2 * cfg.pw
        for ks in range(nks):
            for rb in range(2):
                c = s_blk(par, rb)
                if ks == 0 and (cfg.cinit or not cfg.fold):
                    c = cm_blk(rb) if cfg.cinit else I(0)
                mm.append((s_blk(par, rb), af(ks), q_frag(rb, ks), c, ks))
        ng = len(mm)
        f0 = ng - 2 * nks                  # matrix instructions in front of the first fragment's
        fill = [[] for _ in range(ng)]

        def at(g, fn):
            fill[g].append(fn)

        if mfma:
            for i in range(nks - 4):       # K fragments 4.. (0..3 were requested behind the previous seam)
                at(f0 + 2 * i + 1, lambda i=i: self.frag_read(i + 4))
            if softmax:                    # the first four V^T fragments of the previous tile, for phase B
                for i in range(nks - 4, nks):
                    at(f0 + 2 * i + 1, lambda i=i: self.frag_read(i + 4))
            if dma and "dma" not in cfg.abl:
                for i in range(npc):       # LDS-DMA of tile j+2
                    at(f0 + 2 * i, lambda i=i: self.dma_piece(i))
                    at(f0 + 2 * npc + 2 * (i // 2), lambda i=i: self.dma_advance(i))
                at(f0 + 3 * npc, lambda: self.wr_advance())
            if zero_o:
                regs = [16 * (8 * rb + db) + r for rb in range(2) for db in range(ndb) for r in range(16)]
                for n, r in enumerate(regs):
                    at(f0 + (n * 2 * nks) // len(regs), lambda r=r: self.emit("v_accvgpr_write_b32", A(r), [I(0)]))
        if softmax:
            # element pair (r, r + 1) of block rb: exp2 in one gap, row sums and pack in the next (in place: word 4 u + w)
            pairs = [(rb, r) for rb in range(2) for r in range(0, 16, 2)]
            for n, (rb, r) in enumerate(pairs):
                g = f0 + (n * (2 * nks - 1)) // 16 if nks < 16 else f0 + 2 * n
                if n >= cfg.xbp:       # (the first xbp pairs were exponentiated in phase B of their own step)
                    at(g, lambda rb=rb, r=r: self.exp_pair(prev, rb, r))
                at(g + 1, lambda rb=rb, r=r: self.sum_pack(prev, rb, r))
        for g in range(ng):
            d, a_, b_, c_, fr = mm[g]
            if mfma:
                if fr is not None:
                    self.lds_need(self.frag_rid[fr])
                self.mfma(d, a_, b_, c_)
            for fn in fill[g]:
                fn()

    def exp_pair(self, prev, rb, r):
        for x in (s_elem(prev, rb, r), s_elem(prev, rb, r + 1)):
            self.emit("v_exp_f32", x, [x])

    def sum_pack(self, prev, rb, r):
        x0, x1 = s_elem(prev, rb, r), s_elem(prev, rb, r + 1)
        self.emit("v_add_f32", VN("l%d" % rb), [x0, VN("l%d" % rb)])
        self.emit("v_add_f32", V(T_LB + rb), [x1, V(T_LB + rb)])
        self.emit("v_cvt_pk_%s_f32" % self.cfg.dtype, V(SB + 16 * (2 * prev + rb) + r // 2), [x0, x1])

    # ------------------------------------------------------------ phase B
    def phase_b(self, par, mfma, softmax, seam=True):
        """B(j): O += V^T(j-1) P^T(j-1) | start-softmax of S[par]; the seam to step j+1"""
        cfg = self.cfg
        prev = par ^ 1
        if not mfma and softmax:
            self.emit("s_nop", None, [I(15)], note="S(0) is still leaving the matrix pipe")
        nks, ndb = cfg.nks, cfg.ndb
        nb = 4 * ndb                       # matrix instructions of the phase
        seam_gap = nb - 8                  # behind the products of the fourth-last V^T fragment: the last four slots free up after it
        fill = [[] for _ in range(nb)]

        def at(g, fn):
            fill[g].append(fn)

        if mfma:
            for j in range(2 * ndb - 4):   # V^T fragments 4.. of the tile (the first four were requested in phase A)
                at(2 * j + 1, lambda j=j: self.frag_read(nks + j + 4))
        if softmax:
            mask_lbl, mask_back = self.newlabel("MASK"), self.newlabel("MASKBACK")

            def mask_branch():
                self.emit("s_cmp_ge_i32", None, [SN("j"), SN("maskfrom")])
                self.emit("s_cbranch_scc1", None, [], target=mask_lbl)
                self.label(mask_back)
            at(2, mask_branch)
            self.outofline.append(("mask", mask_lbl, mask_back, par, False))
            for i in range(16):            # row maxima: rb0 in gaps 3..6, rb1 in 5..8 (its last k-step ended phase A)
                rb, st = divmod(i, 8)
                at(3 + 2 * rb + st // 2, lambda rb=rb, st=st: self.max_op(par, rb, st))
            at(9, lambda: self.decide_1())
            at(10, lambda: self.decide_2())
            dec_lbl = self.newlabel("DEC")
            first = not mfma
            if cfg.fold:
                at(11, lambda: self.decide_4_fold(dec_lbl, first))
                pairs = [(rb, r) for rb in range(2) for r in range(0, 16, 2)]
                for n in range(cfg.xbp):       # exp2 of the first pairs right here: S' needs no further arithmetic
                    at(12 + (n * (nb - 12)) // cfg.xbp, lambda n=n: self.exp_pair(par, pairs[n][0], pairs[n][1]))
            else:
                at(11, lambda: self.decide_3())
                at(12, lambda: self.decide_4(dec_lbl))
                for e in range(32):        # s * scale2 - m from gap 13 on (D = 256: three per gap up to gap 23)
                    at(13 + (e // 3 if nb == 32 else (e * (nb - 13)) // 32), lambda e=e: self.fma_op(par, e // 16, e % 16))
        if seam:
            def do_seam():
                self.emit("s_waitcnt", None, [], vmcnt=2 * cfg.pw if "dma" not in cfg.abl else 0)
                self.emit("s_barrier")
                for n in ("ka0", "ka1"):
                    self.emit("v_add_u32", VN(n), [SN("delta"), VN(n)])
            at(seam_gap - 2, lambda: self.stage_delta())
            at(seam_gap, do_seam)
            for i in range(4):
                at(seam_gap + 1 + 2 * i, lambda i=i: self.frag_read(i))
            # the V^T addresses move once the last fragment of this step is requested (gap seam_gap - 1)
            at(seam_gap + 2, lambda: [self.emit("v_add_u32", VN(n), [SN("deltav"), VN(n)])
                                      for n in (("ta0", "ta1", "ta2", "ta3") if cfg.vt else ("ta0", "ta1"))])
        for g in range(nb):
            if mfma:
                u, db, rb = g // (2 * ndb), (g % (2 * ndb)) // 2, g % 2
                fr = nks + ndb * u + db
                self.lds_need(self.frag_rid[fr])
                self.mfma(o_acc(rb, db), af(fr), p_frag(prev, rb, u), o_acc(rb, db))
            for fn in fill[g]:
                fn()
        if softmax:
            resc, back = self.newlabel("RESC"), self.newlabel("RESCBACK")
            self.emit("s_cmp_eq_u32", None, [SN("pend"), I(0)])
            self.emit("s_cbranch_scc0", None, [], target=resc)
            self.label(back)
            self.outofline.append(("resc", resc, back, par, False))
            self.outofline.append(("dec", dec_lbl, dec_lbl + "_BACK", par, not mfma))

    def max_op(self, par, rb, st):
        mx = V(T_MN + rb)
        if st == 0:
            self.emit("v_max3_f32", mx, [s_elem(par, rb, 0), s_elem(par, rb, 1), s_elem(par, rb, 2)])
        elif st < 7:
            self.emit("v_max3_f32", mx, [mx, s_elem(par, rb, 2 * st + 1), s_elem(par, rb, 2 * st + 2)])
        else:
            self.emit("v_max_f32", mx, [mx, s_elem(par, rb, 15)])

    def decide_1(self):
        for rb in range(2):
            self.emit("v_mov_b32", V(T_SW + rb), [V(T_MN + rb)])

    def decide_2(self):   # lanes l and l ^ 32 hold the two halves of a row's keys
        self.emit("s_nop", None, [I(1)], note="VALU write -> permlane read")
        for rb in range(2):
            self.emit("v_permlane32_swap_b32", V(T_SW + rb), [V(T_MN + rb)], swap=1)
        for rb in range(2):
            self.emit("v_max_f32", V(T_MN + rb), [V(T_SW + rb), V(T_MN + rb)])

    def decide_3(self):
        for rb in range(2):
            self.emit("v_mul_f32", V(T_MN + rb), [SN("scale2"), V(T_MN + rb)])
        for rb in range(2):
            self.emit("v_add_f32", V(T_THR + rb), [F(self.cfg.thr), VN("m%d" % rb)])

    def decide_4(self, lbl):
        self.emit("v_cmp_gt_f32", VCC, [V(T_MN), V(T_THR)])
        self.emit("s_mov_b64", SN("sv", 2), [VCC])
        self.emit("v_cmp_gt_f32", VCC, [V(T_MN + 1), V(T_THR + 1)])
        self.emit("s_or_b64", VCC, [VCC, SN("sv", 2)])
        self.emit("s_cbranch_vccnz", None, [], target=lbl)
        self.label(lbl + "_BACK")

    def decide_4_fold(self, lbl, first):
        if first:
            self.emit("s_branch", None, [], target=lbl)
        else:
            self.emit("v_cmp_lt_f32", VCC, [F(self.cfg.thr), V(T_MN)])
            self.emit("s_mov_b64", SN("sv", 2), [VCC])
            self.emit("v_cmp_lt_f32", VCC, [F(self.cfg.thr), V(T_MN + 1)])
            self.emit("s_or_b64", VCC, [VCC, SN("sv", 2)])
            self.emit("s_cbranch_vccnz", None, [], target=lbl)
        self.label(lbl + "_BACK")

    def fma_op(self, par, rb, r):
        x = s_elem(par, rb, r)
        self.emit("v_fma_f32", x, [x, SN("scale2"), VN("m%d" % rb)], neg2=1)

    def to16_f32(self, d, x):
        self.emit("v_and_b32", d, [I(0xFFFF0000 if self.cfg.dtype == "bf16" else 0xFFFFE000), x])

    # ------------------------------------------------------------ out-of-line sections
    def emit_outofline(self):
        cfg = self.cfg
        for kind, lbl, back, par, first in self.outofline:
            self.label(lbl)
            if kind == "mask":     # key c of row r is visible iff c <= lim[r] (lim = min(C - 1, causal limit) - 4 hi)
                self.emit("s_lshl_b32", SN("t0"), [SN("j"), I(5)])
                for rb in range(2):
                    self.emit("v_subrev_u32", V(T_TL + rb), [SN("t0"), VN("lim%d" % rb)])
                self.emit("v_mov_b32", V(T_THR), [F(MASK_VALUE)])     # (free until the decision)
                for rb in range(2):
                    for r in range(16):
                        x = s_elem(par, rb, r)
                        self.emit("v_cmp_gt_i32", VCC, [I((r & 3) + 8 * (r >> 2)), V(T_TL + rb)])
                        self.emit("v_cndmask_b32", x, [x, V(T_THR), VCC])
                self.emit("s_branch", None, [], target=back)
            elif kind == "dec" and cfg.cinit:
                ta, tb = V(T_SW), V(T_SW + 1)
                for rb in range(2):
                    mn, m = V(T_MN + rb), VN("m%d" % rb)
                    if not first:
                        self.emit("v_max_f32", mn, [I(0), mn])
                    self.emit("v_add_f32", ta, [m, mn])       # m_up
                    self.emit("v_sub_f32", tb, [ta, m])       # shift
                    self.emit("v_mov_b32", m, [ta])
                    self.emit("v_exp_f32", V(T_CORR + rb), [tb], neg0=1)
                    for r in range(16):
                        x = s_elem(par, rb, r)
                        self.emit("v_sub_f32", x, [x, tb])
                    for r in range(16):
                        self.emit("v_sub_f32", V(QF + 64 * rb + 48 + r), [I(0), ta])    # the start block of the following steps
                self.emit("s_mov_b32", SN("pend"), [I(0 if first else 1)])
                self.emit("s_branch", None, [], target=back)
            elif kind == "dec" and cfg.fold:
                ta, tb = V(T_SW), V(T_SW + 1)
                for rb in range(2):
                    mn, m = V(T_MN + rb), VN("m%d" % rb)
                    if not first:
                        self.emit("v_max_f32", mn, [I(0), mn])
                    self.emit("v_add_f32", ta, [m, mn])
                    self.to16_f32(tb, ta)
                    self.emit("v_sub_f32", ta, [ta, tb])
                    self.to16_f32(ta, ta)
                    self.emit("v_cvt_pk_%s_f32" % cfg.dtype, V(MF[rb]), [tb, ta])
                    self.emit("v_add_f32", ta, [tb, ta])
                    self.emit("v_sub_f32", tb, [ta, m])
                    self.emit("v_mov_b32", m, [ta])
                    self.emit("v_exp_f32", V(T_CORR + rb), [tb], neg0=1)
                    for r in range(16):
                        x = s_elem(par, rb, r)
                        self.emit("v_sub_f32", x, [x, tb])
                self.emit("s_mov_b32", SN("pend"), [I(0 if first else 1)])
                self.emit("s_branch", None, [], target=back)
            elif kind == "dec":
                for rb in range(2):
                    self.emit("v_max_f32", V(T_THR + rb), [VN("m%d" % rb), V(T_MN + rb)])
                for rb in range(2):
                    self.emit("v_sub_f32", V(T_CORR + rb), [VN("m%d" % rb), V(T_THR + rb)])
                for rb in range(2):
                    self.emit("v_mov_b32", VN("m%d" % rb), [V(T_THR + rb)])
                for rb in range(2):
                    self.emit("v_exp_f32", V(T_CORR + rb), [V(T_CORR + rb)])
                self.emit("s_mov_b32", SN("pend"), [I(0 if first else 1)])
                self.emit("s_branch", None, [], target=back)
            else:     # O, l *= corr once every matrix instruction that accumulates P(j-1) has been issued
                self.emit("s_nop", None, [I(15)])
                self.emit("s_nop", None, [I(7)])
                for rb in range(2):
                    for i0 in range(0, 16 * cfg.ndb, 2):
                        for t in range(2):
                            self.emit("v_accvgpr_read_b32", V(T_RS + t), [A(128 * rb + i0 + t)])
                        for t in range(2):
                            self.emit("v_mul_f32", V(T_RS + t), [V(T_CORR + rb), V(T_RS + t)])
                        for t in range(2):
                            self.emit("v_accvgpr_write_b32", A(128 * rb + i0 + t), [V(T_RS + t)])
                    self.emit("v_mul_f32", VN("l%d" % rb), [V(T_CORR + rb), VN("l%d" % rb)])
                    self.emit("v_mul_f32", V(T_LB + rb), [V(T_CORR + rb), V(T_LB + rb)])
                self.emit("s_mov_b32", SN("pend"), [I(0)])
                self.emit("s_nop", None, [I(4)], note="accvgpr write -> MFMA SrcC")
                self.emit("s_branch", None, [], target=back)

    # ------------------------------------------------------------ whole traversal
    def build(self):
        cfg = self.cfg
        self.outofline = []
        # ---- Q' fragments: attn_fwd16_p5.h parks them in LDS, lane-linear, 32 x 1 KiB per wave at `qback`
        self.emit("s_waitcnt", None, [], lgkmcnt=0)
        for rb in range(2):
            for ks in range(cfg.nks):
                self.lds_read("ds_read_b128", q_frag(rb, ks), VN("qback"), (cfg.nks * rb + ks) * 1024)
        for rb in range(2):
            self.emit("v_mov_b32", V(T_LB + rb), [I(0)])
            self.emit("v_mov_b32", V(T_CORR + rb), [F(1.0)])
        if cfg.cinit:
            for rb in range(2):
                for r in range(16):
                    self.emit("v_mov_b32", V(QF + 64 * rb + 48 + r), [I(0)])
        elif cfg.fold:
            self.emit("v_mov_b32", V(ONES), [VN("onesw")])
            for r in (ONES + 1, ONES + 2, ONES + 3, MF[0], MF[0] + 1, MF[1], MF[1] + 1):
                self.emit("v_mov_b32", V(r), [I(0)])
        self.lds_flush()
        self.emit("s_barrier")                                # every wave has its fragments: the ring may be written
        self.emit("s_mov_b32", SN("wr"), [SN("wr0")])
        npc = 2 * cfg.pw
        for t in range(2):                                   # tiles 0 and 1
            for i in range(npc):
                self.dma_piece(i)
            for i in range(npc):
                self.dma_advance(i)
            self.wr_advance()
        self.emit("s_waitcnt", None, [], vmcnt=npc)
        self.emit("s_barrier")
        self.emit("s_mov_b32", SN("pend"), [I(0)])
        self.emit("s_mov_b32", SN("j"), [I(0)])
        self.emit("s_mov_b32", SN("stg"), [I(0)])
        self.emit("s_mov_b32", SN("delta"), [I(-(RING - 1) * STAGE)])    # the V^T addresses start one stage behind: at the ring's end
        for acc in ("pa", "pb"):
            self.emit("s_mov_b32", SN(acc), [I(0)])
        for i in range(4):
            self.frag_read(i)
        self.phase_a(0, mfma=True, softmax=False, zero_o=True)
        self.phase_b(0, mfma=False, softmax=True)
        self.emit("s_mov_b32", SN("j"), [I(1)])
        if cfg.prof:
            self.emit("s_memtime", SN("ptime", 2))
            self.emit("s_waitcnt", None, [], lgkmcnt=0)
            self.lds_done = self.lds_issued
            self.emit("s_mov_b64", VCC, [SN("ptime", 2)])
            self.emit("s_mov_b32", SN("plast"), [("vcc_lo",)])
        loop, end_even, end_odd, skip, done, fin = (self.newlabel(x) for x in ("LOOP", "ENDEVEN", "ENDODD", "SKIP", "DONE", "FIN"))
        self.label(loop)
        head_state = (self.lds_issued - self.lds_done, self.lds_issued - self.frag_rid[0])
        for par, endl in ((1, end_even), (0, end_odd)):
            self.emit("s_cmp_ge_i32", None, [SN("j"), SN("wnt")])
            self.emit("s_cbranch_scc1", None, [], target=endl)
            self.phase_a(par, mfma=True, softmax=True)
            self.stamp("pa")
            self.phase_b(par, mfma=True, softmax=True)
            self.stamp("pb")
            self.emit("s_add_u32", SN("j"), [SN("j"), I(1)])
            assert (self.lds_issued - self.lds_done <= head_state[0]) and self.lds_issued - self.frag_rid[0] == head_state[1], "LDS queue state"
        self.emit("s_branch", None, [], target=loop)
        # tails: finish step wnt-1 (its scores are in S[last parity]): exp2 / sums / pack, then its V^T P^T products
        for lastpar, lbl in ((0, end_even), (1, end_odd)):
            self.label(lbl)
            self.lds_done = self.lds_issued - head_state[0]
            self.lds_flush()                                  # the K fragments requested behind the last seam are not used
            self.emit("s_nop", None, [I(3)])
            self.phase_a(lastpar ^ 1, mfma=False, softmax=True)
            for i in range(cfg.nks, cfg.nks + 4):
                self.frag_read(i)
            self.emit("s_nop", None, [I(1)], note="freshly packed P -> MFMA operand")
            self.phase_b(lastpar ^ 1, mfma=True, softmax=False, seam=False)
            self.lds_flush()
            self.emit("s_branch", None, [], target=skip)
        # a wave whose rows are done before the workgroup's last step (causal) still owes the others its barriers and its
        # share of the LDS-DMA pieces: steps j = wnt .. nt-1 without arithmetic
        self.label(skip)
        self.emit("s_cmp_ge_i32", None, [SN("j"), SN("nt")])
        self.emit("s_cbranch_scc1", None, [], target=done)
        if "dma" not in cfg.abl:
            for i in range(npc):
                self.dma_piece(i)
            for i in range(npc):
                self.dma_advance(i)
            self.wr_advance()
        self.emit("s_waitcnt", None, [], vmcnt=npc if "dma" not in cfg.abl else 0)
        self.emit("s_barrier")
        self.emit("s_add_u32", SN("j"), [SN("j"), I(1)])
        self.emit("s_branch", None, [], target=skip)
        self.label(done)
        for rb in range(2):
            self.emit("v_add_f32", VN("l%d" % rb), [V(T_LB + rb), VN("l%d" % rb)])
        self.emit("s_waitcnt", None, [], vmcnt=0)
        self.emit("s_branch", None, [], target=fin)
        self.emit_outofline()
        self.label(fin)
        return self.ins


def write_inc(path):
    lines = ["// GENERATED by tools/f256gen.py -- do not edit.  Instruction streams of attn_fwd16_p5 (see the generator's",
             "// header for the register map and the phase tables).", "#pragma once", ""]
    lines.append("#define MFA_P5_OWNED_VGPRS " + ", ".join('"v%d"' % i for i in range(FIRST_OWNED_VGPR, 256)))
    lines.append("")
    lines.append("// X(name, folds Q scale and running maximum into the matrix pipe, stamps the shader clock, head-dimension bucket)")
    lines.append("#define MFA_P5_STREAM_LIST(X) \\")
    for name, cfg in VARIANTS.items():
        lines.append("  X(%s, %d, %d, %d) \\" % (name, cfg.fold, cfg.prof, cfg.D))
    lines.append("")
    lines.append("// transposed K and / or V (developer kernels only): X(name, folds, head-dimension bucket, pattern: bit 0 = K, bit 1 = V)")
    lines.append("#define MFA_P5_TR_STREAM_LIST(X) \\")
    for name, cfg in TR_VARIANTS.items():
        lines.append("  X(%s, %d, %d, %d) \\" % (name, cfg.fold, cfg.D, cfg.tr))
    lines.append("")
    lines.append("")
    for name, cfg in list(VARIANTS.items()) + list(TR_VARIANTS.items()):
        ins = Stream(cfg).build()
        txt = render(ins)
        lines.append("// %s: dtype=%s thr=%g fold=%d prof=%d -- %d instructions" % (name, cfg.dtype, cfg.thr, cfg.fold, cfg.prof, len(txt)))
        lines.append("#define MFA_P5_STREAM_%s \\" % name)
        for t in txt:
            lines.append('  "%s\\n\\t" \\' % t)
        lines.append('  ""')
        lines.append("")
    with open(path, "w") as f:
        f.write("\n".join(lines))


def write_one_operand_inc(path):
    """the K^T-only / V^T-only streams (MODEL_ONLY_VARIANTS) for the DEVELOPER build: generated at build time by `make DEV=1`
    (csrc/attn_fwd16_p5_tr1_stream.inc, git-ignored: ~150 KB of text per stream), included by attn_fwd16_p5_tr.h only"""
    lines = ["// GENERATED by tools/f256gen.py --one-operand at build time (make DEV=1) -- not tracked.  Streams of attn_fwd16_p5_tr for",
             "// ONE transposed operand: X(name, folds, head-dimension bucket, pattern: 1 = K^T, 2 = V^T)", "#pragma once", ""]
    lines.append("#define MFA_P5_TR1_STREAM_LIST(X) \\")
    for name, cfg in MODEL_ONLY_VARIANTS.items():
        lines.append("  X(%s, %d, %d, %d) \\" % (name, cfg.fold, cfg.D, cfg.tr))
    lines.append("")
    lines.append("")
    for name, cfg in MODEL_ONLY_VARIANTS.items():
        txt = render(Stream(cfg).build())
        lines.append("// %s: dtype=%s fold=%d pattern=%d -- %d instructions" % (name, cfg.dtype, cfg.fold, cfg.tr, len(txt)))
        lines.append("#define MFA_P5_STREAM_%s \\" % name)
        for t in txt:
            lines.append('  "%s\\n\\t" \\' % t)
        lines.append('  ""')
        lines.append("")
    with open(path, "w") as f:
        f.write("\n".join(lines))


VARIANTS = {
    "BF16_THR8": Cfg("bf16"),
    "F16_THR8": Cfg("f16"),
    "BF16_FOLD": Cfg("bf16", fold=1),
    "F16_FOLD": Cfg("f16", fold=1),
    "BF16_FOLD_PROF": Cfg("bf16", fold=1, prof=1),
}
VARIANTS["D128_BF16_FOLD"] = Cfg("bf16", fold=1, D=128, xbp=8)     # developer stream: the 32-key-step structure at the headline's D
for _d in (192, 160):       # the head-dimension buckets between 128 and 256
    for _t in ("bf16", "f16"):
        VARIANTS["D%d_%s_THR8" % (_d, _t.upper())] = Cfg(_t, D=_d)
        VARIANTS["D%d_%s_FOLD" % (_d, _t.upper())] = Cfg(_t, fold=1, D=_d)

TR_VARIANTS = {}
for _d in (256, 192, 160):
    for _t in ("bf16", "f16"):
        TR_VARIANTS["D%d_%s_THR8_TR" % (_d, _t.upper())] = Cfg(_t, D=_d, tr=3)
        TR_VARIANTS["D%d_%s_FOLD_TR" % (_d, _t.upper())] = Cfg(_t, fold=1, D=_d, tr=3)
# one operand transposed: verified on the model (tests/test_f256_stream.py); not in the tracked generated file (~150 KB of text per
# stream) -- `make DEV=1` writes them to csrc/attn_fwd16_p5_tr1_stream.inc for the developer kernel (write_one_operand_inc)
MODEL_ONLY_VARIANTS = {}
for _d in (256, 192, 160):
    for _t in ("bf16", "f16"):
        for _sfx, _pat in (("TRK", 1), ("TRV", 2)):
            MODEL_ONLY_VARIANTS["D%d_%s_THR8_%s" % (_d, _t.upper(), _sfx)] = Cfg(_t, D=_d, tr=_pat)
            MODEL_ONLY_VARIANTS["D%d_%s_FOLD_%s" % (_d, _t.upper(), _sfx)] = Cfg(_t, fold=1, D=_d, tr=_pat)

if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    if len(sys.argv) > 2 and sys.argv[1] == "--one-operand":
        write_one_operand_inc(sys.argv[2])
        print("wrote", os.path.normpath(sys.argv[2]), "-", len(MODEL_ONLY_VARIANTS), "streams")
        sys.exit(0)
    out = os.path.join(here, "..", "metal_flash_attention_amd", "csrc", "attn_fwd16_p5_stream.inc")
    write_inc(out)
    ins = Stream(VARIANTS["BF16_THR8"]).build()
    print("wrote", os.path.normpath(out), "-", len(ins), "instructions in the default stream")
