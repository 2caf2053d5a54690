#!/usr/bin/env python3
"""Generator of the hand-placed instruction stream of attn_dq16_p4 (csrc/attn_dq16_p4.h): backwardQuery for D <= 128
with 16-bit Q/K/V/dO.

A workgroup is four waves x 64 query rows (one wave per SIMD, the whole 512-entry register file); it walks the keys in
64-key tiles.  Per tile and wave, 96 matrix instructions (32x32x16), in six groups of 16 -- key block kb = 0, 1 of the
tile, row block rb = 0, 1 of the wave:

    S(kb)  S'^T  = K Q'^T - L     A = K row fragments (LDS), B = Q' = Q * log2(e)/sqrt(D) (registers); the accumulator starts
                                  from a register block holding -L of the wave's rows (they never change): no extra k-step
    P(kb)  dP'^T = V dO^T - D     the same with the D term                          | exp2(S'(kb))
    Q(kb)  dQ^T += K^T dS'^T      A = K^T (ds_read_b64_tr_b16), B = dS' in 16 bits   | dS' = P * dP', pack (in place)

    order: S(0) P(0) S(1) P(1) Q(0) Q(1); VALU work of block kb runs beside the matrix instructions of the next group.

Register map (fixed, see tools/p4gen.py for why):

    a[0:127]    dQ^T accumulators  (rb, db) -> 16 (4 rb + db)         lane = row, registers = head-dimension rows
    a[128:191]  Q' fragments       (rb, ks) -> 128 + 4 (8 rb + ks)
    a[192:255]  dO fragments       (rb, ks) -> 192 + 4 (8 rb + ks)
    v[24:55]    -L blocks, rb -> 24 + 16 rb        v[56:87]  -D blocks, rb -> 56 + 16 rb
    v[88:151]   S'^T / P           (rb, kb) -> 88 + 16 (2 kb + rb)
    v[152:215]  dP'^T / dS'        (rb, kb) -> 152 + 16 (2 kb + rb); the 16-bit dS' fragments (rb, kb, u) are packed in place: + 4 u
    v[216:231]  ring of four A-operand fragments read from LDS, fragment i in slot i % 4
    v[232:239]  temporaries (mask limits, mask constant)
    v[0:23]     left to hipcc (operands of the statement)

LDS: ring of four stages {K tile | V tile}, each tile [D/32][64 keys][32 elements] with the four 16-byte chunks of a 64-byte
row XOR-swizzled by (key >> 2) & 3 (attn_dkv16_rs.h: serves row fragments and transposing reads alike), filled by LDS-DMA
two tiles ahead.  One barrier per tile.

The instruction list is rendered as an asm template and executed by tools/dq4sim.py on the lane-exact model of
tools/p4sim.py: tests/test_dq4_stream.py.

Usage: python tools/dq4gen.py   (rewrites metal_flash_attention_amd/csrc/attn_dq16_p4_stream.inc)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from p4gen import A, F, I, M0, SN, V, VCC, VN, Stream as _P4Stream, render  # noqa: E402

DQ_BASE, Q_BASE, G_BASE = 0, 128, 192
CL, CD, ST, DP, AF = 24, 56, 88, 152, 216
T_TL, T_MASKV = 232, 234
T_TB = 236           # D = 64: transposing-read addresses of the previous tile (v236, v237)
FIRST_OWNED_VGPR = 24

STAGE, VIMG, RING = 32768, 16384, 4
PACKED_MUL = int(os.environ.get('MFA_GEN_PACKED_MUL', '0'))   # v_pk_mul_f32 for dS' = P * dP': no faster on gfx950 (dQ 1.5 % slower), see DESIGN.md
N_MFMA = 96

INOUT_V = ["koff0", "koff1", "koff2", "koff3", "voff0", "voff1", "voff2", "voff3", "ka0", "ka1", "ta0", "ta1"]
TMP_S = ["j", "stg", "delta", "wr", "t0", "t1", "pa", "pb", "pc", "pd", "plast"]
TMP_S64 = ["ptime"]
IN_V = ["negl0", "negl1", "negd0", "negd1", "lim0", "lim1"]
IN_S = ["kres", "vres", "nt", "wnt", "kinc", "vinc", "wr0", "ringend", "maskfrom", "scale2x2"]


class Cfg:
    def __init__(self, dtype="bf16", prof=0, exact=0, abl=(), D=128, defer=None, tr=0):
        """exact: Q stays as stored, -L arrives divided by log2(e)/sqrt(D) and the scale is applied in fp32 before the exp2
        (one packed multiply per two scores more); otherwise Q arrives pre-multiplied, rounded to the 16-bit type"""
        self.dtype, self.prof, self.exact, self.abl, self.D = dtype, prof, exact, frozenset(abl), D
        self.defer = (D == 64) if defer is None else defer     # dQ update of a tile's second key block beside the next tile
        # tr (D = 128; model-verified, NOT yet behind a kernel -- DESIGN.md 10.4): K and V stored TRANSPOSED ([D][keys]).  The
        # images keep the source orientation -- [2 blocks of 32 keys][128 elements][64 bytes], chunks ^ (element >> 2) & 3: the
        # same two read recipes with their roles exchanged.  K and V ROW fragments (A of S' and dP') come from transposing reads
        # (addresses ka0 / ka1: rows + 0 / + 8 of a 16-element step), which return the contraction index in the order of an
        # accumulator block's registers (4 hi + {0..3, 8..11}): the kernel stores the Q' and dO fragments in that order.  K^T
        # fragments (A of dQ^T += K^T dS'^T) are two 8-byte reads of the lane's element row, chunks 2 u and 2 u + 1 at 8 hi
        # (dS' holds its keys in that register order): four addresses ta0..ta3 instead of two.  Whole tiles only (C % 64 == 0):
        # a masked score makes dS' = 0, but 0 x what follows the sequence in a row of V^T / K^T is not.
        self.tr = tr
        assert not (tr and D != 128)


def st_blk(rb, kb):
    return V(ST + 16 * (2 * kb + rb), 16)


def dp_blk(rb, kb):
    return V(DP + 16 * (2 * kb + rb), 16)


def ds16(rb, kb, u):
    return V(DP + 16 * (2 * kb + rb) + 4 * u, 4)


def af(i):
    return V(AF + 4 * (i % 4), 4)


def af_half(i, h):
    return V(AF + 4 * (i % 4) + 2 * h, 2)


def q_frag(rb, ks):
    return A(Q_BASE + 4 * (8 * rb + ks), 4)


def g_frag(rb, ks):
    return A(G_BASE + 4 * (8 * rb + ks), 4)


def dq_acc(rb, db):
    return A(DQ_BASE + 16 * (4 * rb + db), 16)


class Stream(_P4Stream):
    def __init__(self, cfg):
        _P4Stream.__init__(self, cfg)
        self.frag_rid = {}

    # fragment i of a tile: 0..7 K rows kb0 | 8..15 V rows kb0 | 16..23 K rows kb1 | 24..31 V rows kb1 | 32..39 K^T kb0 | 40..47 K^T kb1
    def frag_read(self, i):
        if self.cfg.tr:
            if i < 32:       # rows 16 ks (+ 8) of key block kb's sub-image
                kb, isv, ks = i // 16, (i // 8) & 1, i % 8
                off = (VIMG if isv else 0) + kb * 8192 + ks * 1024
                self.lds_read("ds_read_b64_tr_b16", af_half(i, 0), VN("ka0"), off, note="%s^T rows kb%d ks%d" % ("V" if isv else "K", kb, ks))
                self.frag_rid[i] = self.lds_read("ds_read_b64_tr_b16", af_half(i, 1), VN("ka1"), off)
            else:            # element row 32 db + lane % 32 of sub-image kb: 8 bytes of chunks 2 u, 2 u + 1
                kb, r = divmod(i - 32, 8)
                u, db = divmod(r, 4)
                off = kb * 8192 + db * 2048
                self.lds_read("ds_read_b64", af_half(i, 0), VN("ta%d" % (2 * u)), off, note="K kb%d u%d db%d" % (kb, u, db))
                self.frag_rid[i] = self.lds_read("ds_read_b64", af_half(i, 1), VN("ta%d" % (2 * u + 1)), off)
            return
        if i < 32:
            kb, isv, ks = i // 16, (i // 8) & 1, i % 8
            off = (VIMG if isv else 0) + (ks >> 1) * 4096 + kb * 2048
            self.frag_rid[i] = self.lds_read("ds_read_b128", af(i), VN("ka%d" % (ks & 1)), off,
                                             note="%s rows kb%d ks%d" % ("V" if isv else "K", kb, ks))
        else:
            kb, r = divmod(i - 32, 8)
            u, db = divmod(r, 4)
            off = db * 4096 + kb * 2048 + u * 1024
            self.lds_read("ds_read_b64_tr_b16", af_half(i, 0), VN("ta0"), off, note="K^T kb%d u%d db%d" % (kb, u, db))
            self.frag_rid[i] = self.lds_read("ds_read_b64_tr_b16", af_half(i, 1), VN("ta1"), off)

    # D = 128: pieces 0..3 K, 4..7 V (four 1 KiB pieces per wave and operand tile); D = 64: 0, 1 K, 2, 3 V
    def dma_piece(self, i):
        pw = self.cfg.D // 32
        name, res, base = (("koff%d" % i, "kres", 0) if i < pw else ("voff%d" % (i - pw), "vres", VIMG))
        self.emit("s_add_u32", M0, [SN("wr"), I(base + (i % pw) * 1024)])
        self.emit("buffer_load_dwordx4_lds", None, [VN(name), SN(res, 4)])

    def dma_advance(self, i):
        pw = self.cfg.D // 32
        name, inc = ("koff%d" % i, "kinc") if i < pw else ("voff%d" % (i - pw), "vinc")
        self.emit("v_add_u32_e64", VN(name), [VN(name), SN(inc)], clamp=1)

    def wr_advance(self):
        self.emit("s_add_u32", SN("wr"), [SN("wr"), I(STAGE)])
        self.emit("s_cmp_ge_u32", None, [SN("wr"), SN("ringend")])
        self.emit("s_cselect_b32", SN("t1"), [I(RING * STAGE), I(0)])
        self.emit("s_sub_u32", SN("wr"), [SN("wr"), SN("t1")])

    def stage_delta(self):
        self.emit("s_add_u32", SN("stg"), [SN("stg"), I(1)])
        self.emit("s_and_b32", SN("stg"), [SN("stg"), I(RING - 1)])
        self.emit("s_cmp_eq_u32", None, [SN("stg"), I(0)])
        self.emit("s_cselect_b32", SN("t1"), [I(RING * STAGE), I(0)])
        self.emit("s_sub_u32", SN("delta"), [I(STAGE), SN("t1")])

    def addr_advance(self, names):
        for n in names:
            self.emit("v_add_u32", VN(n), [SN("delta"), VN(n)])

    # ---- arithmetic fillers on block (rb, kb)
    def scale_op(self, rb, kb, r):
        x = V(ST + 16 * (2 * kb + rb) + r, 2)
        self.emit("v_pk_mul_f32", x, [x, SN("scale2x2", 2)])

    def exp_op(self, rb, kb, r):
        x = V(ST + 16 * (2 * kb + rb) + r)
        self.emit("v_exp_f32", x, [x])

    def mul_op(self, rb, kb, r):
        if PACKED_MUL:
            x = V(DP + 16 * (2 * kb + rb) + r, 2)
            self.emit("v_pk_mul_f32", x, [V(ST + 16 * (2 * kb + rb) + r, 2), x])
        else:
            for t in range(2):
                x = V(DP + 16 * (2 * kb + rb) + r + t)
                self.emit("v_mul_f32", x, [V(ST + 16 * (2 * kb + rb) + r + t), x])

    def pack_op(self, rb, kb, u, w):
        r = 8 * u + 2 * w
        b = DP + 16 * (2 * kb + rb)
        self.emit("v_cvt_pk_%s_f32" % self.cfg.dtype, V(b + 4 * u + w), [V(b + r), V(b + r + 1)])

    def mask_section(self, lbl, back, kb):
        """edge / causal mask on the fresh S' blocks of key block kb: key c of row r is visible iff c <= lim[r]"""
        self.label(lbl)
        self.emit("s_lshl_b32", SN("t0"), [SN("j"), I(6)])
        for rb in range(2):
            self.emit("v_subrev_u32", V(T_TL + rb), [SN("t0"), VN("lim%d" % rb)])   # lim - 4 hi - 64 j
        for rb in range(2):
            for r in range(16):
                c = kb * 32 + (r & 3) + 8 * (r >> 2)
                x = V(ST + 16 * (2 * kb + rb) + r)
                self.emit("v_cmp_gt_i32", VCC, [I(c), V(T_TL + rb)])
                self.emit("v_cndmask_b32", x, [x, V(T_MASKV), VCC])
        self.emit("s_branch", None, [], target=back)

    # ---------------------------------------------------------------- one tile
    def tile(self):
        cfg = self.cfg
        fill = [[] for _ in range(N_MFMA)]

        def at(g, fn):
            fill[g].append(fn)

        # fragment i + 4 takes the slot of fragment i once both of its matrix instructions are issued
        for i in range(44):
            if "reads" in cfg.abl and i + 4 >= 4:      # timing-only: fragments 4.. are never refreshed (WRONG RESULTS)
                self.frag_rid[i + 4] = 0
                continue
            at(2 * i + 1, lambda i=i: self.frag_read(i + 4))
        # LDS-DMA of tile j+2: one piece per even gap of S(0); offsets advance in P(0)
        if "dma" not in cfg.abl:
            for i in range(8):
                at(2 * i, lambda i=i: self.dma_piece(i))
                at(16 + 2 * i, lambda i=i: self.dma_advance(i))
            at(64, lambda: self.wr_advance())
        at(48, lambda: self.stage_delta())
        # row-read addresses: the last row fragment (31) is requested in gap 55
        at(56, lambda: self.addr_advance(["ka0", "ka1"]))

        def mask_branch(kb):
            lbl, back = self.newlabel("MASK"), self.newlabel("MASKBACK")
            self.emit("s_cmp_ge_i32", None, [SN("j"), SN("maskfrom")])
            self.emit("s_cbranch_scc1", None, [], target=lbl)
            self.label(back)
            self.outofline.append((lbl, back, kb))
        # ---- key block kb: mask, exp2 beside P(kb) (gaps 16.., 48..), multiply and pack beside the group after it
        for kb in range(2):
            g0 = 16 + 32 * kb
            at(g0 + 2, lambda kb=kb: mask_branch(kb))
            seq = []
            for rb in range(2):
                for u in range(2):
                    if cfg.exact:
                        seq += [lambda rb=rb, kb=kb, r=r: self.scale_op(rb, kb, r) for r in range(8 * u, 8 * u + 8, 2)]
                    seq += [lambda rb=rb, kb=kb, r=r: self.exp_op(rb, kb, r) for r in range(8 * u, 8 * u + 8)]
            per = -(-len(seq) // 13)
            for n, fn in enumerate(seq):
                if "exp" not in cfg.abl:
                    at(g0 + 3 + n // per, fn)
            g1 = g0 + 16           # S(1) after P(0), Q(0) after P(1)
            seq = []
            for rb in range(2):
                seq += [lambda rb=rb, kb=kb, r=r: self.mul_op(rb, kb, r) for r in range(0, 16, 2)]
            for rb in range(2):
                for u in range(2):
                    seq += [lambda rb=rb, kb=kb, u=u, w=w: self.pack_op(rb, kb, u, w) for w in range(4)]
            # even gaps take more of them: the odd ones carry the fragment reads (two per gap beside Q(0)) and their waits
            g, used = g1 + 2, 0
            for fn in seq:
                quota = (5 if len(seq) > 32 else 3) if g % 2 == 0 else 2
                if used == quota:
                    g, used = g + 1, 0
                    quota = (5 if len(seq) > 32 else 3) if g % 2 == 0 else 2
                assert g < g1 + 16
                if "mulpack" not in cfg.abl:
                    at(g, fn)
                used += 1

        # ---- seam (gap 88): own DMA pieces of tile j+1 have landed; barrier; first fragments of tile j+1
        def seam():
            self.emit("s_waitcnt", None, [], vmcnt=8 if "dma" not in cfg.abl else 0)
            self.emit("s_barrier")
            self.addr_advance(["ta0", "ta1", "ta2", "ta3"] if cfg.tr else ["ta0", "ta1"])
        at(88, seam)
        for i in range(4):
            at(89 + 2 * i, lambda i=i: self.frag_read(i))

        # ---- matrix instructions
        mm = []
        for kb in range(2):
            for ks in range(8):
                for rb in range(2):
                    mm.append((st_blk(rb, kb), af(16 * kb + ks), q_frag(rb, ks), V(CL + 16 * rb, 16) if ks == 0 else st_blk(rb, kb), 16 * kb + ks))
            for ks in range(8):
                for rb in range(2):
                    mm.append((dp_blk(rb, kb), af(16 * kb + 8 + ks), g_frag(rb, ks), V(CD + 16 * rb, 16) if ks == 0 else dp_blk(rb, kb), 16 * kb + 8 + ks))
        for kb in range(2):
            for u in range(2):
                for db in range(4):
                    for rb in range(2):
                        fr = 32 + 8 * kb + 4 * u + db
                        mm.append((dq_acc(rb, db), af(fr), ds16(rb, kb, u), dq_acc(rb, db), fr))
        assert len(mm) == N_MFMA
        stamps = {32: "pa", 64: "pb"}
        for g, (d, a_, b_, c_, fr) in enumerate(mm):
            if g in stamps:
                self.stamp(stamps[g])
            self.lds_need(self.frag_rid[fr])
            self.mfma(d, a_, b_, c_)
            for fn in fill[g]:
                fn()
        self.stamp("pc")

    # ---------------------------------------------------------------- one tile, second key block's update deferred
    # Six groups of G = D/8 matrix instructions: S(0) P(0) S(1) Q'(1) P(1) Q(0), where Q'(1) is the dQ update of key block 1
    # of the PREVIOUS tile: the exp2 / multiply / pack work of a tile (1400 VALU cycles beside 1536 of the matrix pipe at
    # D = 64) then has no point where the matrix instructions wait for it -- block 1's runs from the middle of one tile to
    # the middle of the next.  In place as before: S(1) of the next tile overwrites P(1) after the multiplies have read it,
    # P(1) of the next tile overwrites dS'(1) after Q'(1) has read it.
    def frag_read64(self, i):
        nks, ndb = self.cfg.D // 16, self.cfg.D // 32
        grp, r = divmod(i, nks)
        if grp in (0, 1, 2, 4):        # row fragments: K kb0 | V kb0 | K kb1 | V kb1
            kb, isv, ks = (0, 0, r) if grp == 0 else (0, 1, r) if grp == 1 else (1, 0, r) if grp == 2 else (1, 1, r)
            off = (VIMG if isv else 0) + (ks >> 1) * 4096 + kb * 2048
            self.frag_rid[i] = self.lds_read("ds_read_b128", af(i), VN("ka%d" % (ks & 1)), off,
                                             note="%s rows kb%d ks%d" % ("V" if isv else "K", kb, ks))
        else:                          # K^T: kb1 of the previous tile (group 3, addresses v236, v237) | kb0 of this tile (group 5)
            kb = 1 if grp == 3 else 0
            u, db = divmod(r, ndb)
            off = db * 4096 + kb * 2048 + u * 1024
            a0, a1 = (V(T_TB), V(T_TB + 1)) if grp == 3 else (VN("ta0"), VN("ta1"))
            self.lds_read("ds_read_b64_tr_b16", af_half(i, 0), a0, off, note="K^T kb%d u%d db%d%s" % (kb, u, db, " (previous tile)" if grp == 3 else ""))
            self.frag_rid[i] = self.lds_read("ds_read_b64_tr_b16", af_half(i, 1), a1, off)

    def valu_schedule64(self):
        """(gap, fn) of the exp2 / multiply / pack work of ONE tile, gap counted from the tile's first matrix instruction
        (NM.. = beside the next tile's); greedy, earliest gap whose VALU cycles are free"""
        cfg = self.cfg
        G = cfg.D // 8
        NM = 6 * G
        load, out = [0] * NM, []
        total = (64 * 16 + 64 * 4 + 32 * 4 + (32 * 8 if cfg.exact else 0))
        cap = 40 if cfg.exact else -(-total // NM) + 3     # D = 64: two exp2 per gap (+ one packed scale multiply)
        cap = max(cap, 24 if cfg.exact else 20)
        state = {"g": 0}

        def place(release, cost, fn, last):
            g = max(release, state["g"])
            while load[g % NM] + cost > cap:
                g += 1
            assert g <= last, (g, last)
            load[g % NM] += cost
            state["g"] = g
            out.append((g, fn))
        s_end, p_end, q_start, s_next = (G - 1, 3 * G - 1), (2 * G - 1, 5 * G - 1), (5 * G, NM + 3 * G), (NM, NM + 2 * G)
        for kb in range(2):
            state["g"] = 0
            for rb in range(2):
                for u in range(2):
                    if cfg.exact:
                        for r in range(8 * u, 8 * u + 8, 2):
                            place(s_end[kb] + 4, 8, lambda rb=rb, kb=kb, r=r: self.scale_op(rb, kb, r), s_next[kb] - 1)
                    for r in range(8 * u, 8 * u + 8):
                        place(s_end[kb] + 4, 16, lambda rb=rb, kb=kb, r=r: self.exp_op(rb, kb, r), s_next[kb] - 1)
            state["g"] = max(state["g"], p_end[kb] + 3)
            for rb in range(2):
                for r in range(0, 16, 2):
                    place(p_end[kb] + 3, 8, lambda rb=rb, kb=kb, r=r: self.mul_op(rb, kb, r), s_next[kb] - 1)
            for u in range(2):
                for rb in range(2):
                    for w in range(4):
                        place(p_end[kb] + 3, 4, lambda rb=rb, kb=kb, u=u, w=w: self.pack_op(rb, kb, u, w), q_start[kb] - 2)
        return out

    def tile64(self, tail=False):
        """tail: what is left of the last tile after the loop -- its block-1 VALU work and Q'(1)"""
        cfg = self.cfg
        nks, ndb = cfg.D // 16, cfg.D // 32
        G = 2 * nks
        NM = 6 * G
        npieces = 2 * (cfg.D // 32)
        fill = [[] for _ in range(NM)]

        def at(g, fn):
            fill[g].append(fn)
        sched = self.valu_schedule64()
        if tail:
            for g, fn in sched:
                if g >= NM:
                    fn()
            self.emit("s_nop", None, [I(4)])
            for i in range(3 * nks, 3 * nks + 4):
                self.frag_read64(i)
            for i in range(3 * nks, 4 * nks):
                u, db = divmod(i - 3 * nks, ndb)
                for rb in range(2):
                    self.lds_need(self.frag_rid[i])
                    self.mfma(dq_acc(rb, db), af(i), ds16(rb, 1, u), dq_acc(rb, db))
                    if rb == 1 and i + 4 < 4 * nks:
                        self.frag_read64(i + 4)
            return
        # previous tile's work first in a gap: its multiplies must be out of the way of S(1)
        for g, fn in sched:
            if g >= NM:
                at(g - NM, fn)
        nfr = 6 * nks
        for i in range(nfr - 4):
            at(2 * i + 1, lambda i=i: self.frag_read64(i + 4))
        for i in range(npieces):
            at(2 * i, lambda i=i: self.dma_piece(i))
            at(2 * npieces + 2 * i, lambda i=i: self.dma_advance(i))
        at(3 * G + 2, lambda: self.wr_advance())
        at(3 * G, lambda: self.stage_delta())
        at(10 * nks - 8, lambda: self.addr_advance(["ka0", "ka1"]))     # the last row fragment (5 nks - 1) is requested one gap earlier

        def mask_branch(kb):
            lbl, back = self.newlabel("MASK"), self.newlabel("MASKBACK")
            self.emit("s_cmp_ge_i32", None, [SN("j"), SN("maskfrom")])
            self.emit("s_cbranch_scc1", None, [], target=lbl)
            self.label(back)
            self.outofline.append((lbl, back, kb))
        at(G + 2, lambda: mask_branch(0))
        at(3 * G + 2, lambda: mask_branch(1))
        for g, fn in sched:
            if g < NM:
                at(g, fn)

        def seam():
            self.emit("s_waitcnt", None, [], vmcnt=npieces)
            self.emit("s_barrier")
            self.emit("v_mov_b32", V(T_TB), [VN("ta0")])
            self.emit("v_mov_b32", V(T_TB + 1), [VN("ta1")])
            self.addr_advance(["ta0", "ta1"])
        at(NM - 8, seam)
        for i in range(4):
            at(NM - 7 + 2 * i, lambda i=i: self.frag_read64(i))

        mm = []
        for i in range(nfr):
            grp, r = divmod(i, nks)
            for rb in range(2):
                if grp in (0, 2):
                    kb = grp // 2
                    mm.append((st_blk(rb, kb), af(i), q_frag(rb, r), V(CL + 16 * rb, 16) if r == 0 else st_blk(rb, kb), i))
                elif grp in (1, 4):
                    kb = grp // 4
                    mm.append((dp_blk(rb, kb), af(i), g_frag(rb, r), V(CD + 16 * rb, 16) if r == 0 else dp_blk(rb, kb), i))
                else:
                    kb = 1 if grp == 3 else 0
                    u, db = divmod(r, ndb)
                    mm.append((dq_acc(rb, db), af(i), ds16(rb, kb, u), dq_acc(rb, db), i))
        assert len(mm) == NM
        stamps = {2 * G: "pa", 4 * G: "pb"}
        for g, (d, a_, b_, c_, fr) in enumerate(mm):
            if g in stamps:
                self.stamp(stamps[g])
            self.lds_need(self.frag_rid[fr])
            self.mfma(d, a_, b_, c_)
            for fn in fill[g]:
                fn()
        self.stamp("pc")

    # ---------------------------------------------------------------- whole traversal
    def build(self):
        cfg = self.cfg
        d64 = cfg.defer
        npieces = 2 * (cfg.D // 32)
        frag_read, tile = (self.frag_read64, self.tile64) if d64 else (self.frag_read, self.tile)
        self.outofline = []
        for r in range(128):
            if (r // 16) % 4 >= cfg.D // 32:
                continue                                     # (rb, db) -> 16 (4 rb + db) with db < 2
            self.emit("v_accvgpr_write_b32", A(r), [I(0)])
        if d64:
            # block 1 of "the tile before the first": P = exp2(0), dP' = 0 -> dS' = 0; its K^T is read from tile 0
            for rb in range(2):
                for r in range(16):
                    self.emit("v_mov_b32", V(ST + 16 * (2 + rb) + r), [I(0)])
                    self.emit("v_mov_b32", V(DP + 16 * (2 + rb) + r), [I(0)])
            self.emit("v_mov_b32", V(T_TB), [VN("ta0")])
            self.emit("v_mov_b32", V(T_TB + 1), [VN("ta1")])
        for rb in range(2):
            for r in range(16):
                self.emit("v_mov_b32", V(CL + 16 * rb + r), [VN("negl%d" % rb)])
                self.emit("v_mov_b32", V(CD + 16 * rb + r), [VN("negd%d" % rb)])
        self.emit("v_mov_b32", V(T_MASKV), [F(-(0.875 / 1.44269504089) * 3.402823466e+38)])   # +Softmax.swift:242-243
        self.emit("s_mov_b32", SN("wr"), [SN("wr0")])
        for t in range(2):                                   # tiles 0 and 1
            for i in range(npieces):
                self.dma_piece(i)
            for i in range(npieces):
                self.dma_advance(i)
            self.wr_advance()
        self.emit("s_waitcnt", None, [], vmcnt=npieces)
        self.emit("s_barrier")
        self.emit("s_mov_b32", SN("j"), [I(0)])
        self.emit("s_mov_b32", SN("stg"), [I(0)])
        for acc in ("pa", "pb", "pc", "pd"):
            self.emit("s_mov_b32", SN(acc), [I(0)])
        for i in range(4):
            frag_read(i)
        if cfg.prof:
            self.emit("s_memtime", SN("ptime", 2))
            self.emit("s_waitcnt", None, [], lgkmcnt=0)
            self.lds_done = self.lds_issued
            self.emit("s_mov_b64", VCC, [SN("ptime", 2)])
            self.emit("s_mov_b32", SN("plast"), [("vcc_lo",)])
        loop, skip, done, fin = (self.newlabel(x) for x in ("LOOP", "SKIP", "DONE", "FIN"))
        self.label(loop)
        head_out, head_gap = self.lds_issued - self.lds_done, self.lds_issued - self.frag_rid[0]
        tile()
        assert head_gap == (6 if cfg.tr else 3) and self.lds_issued - self.frag_rid[0] == head_gap and \
            self.lds_issued - self.lds_done <= head_out, "loop-carried LDS queue state"
        self.emit("s_add_u32", SN("j"), [SN("j"), I(1)])
        self.emit("s_cmp_lt_i32", None, [SN("j"), SN("wnt")])
        self.emit("s_cbranch_scc1", None, [], target=loop)
        self.emit("s_waitcnt", None, [], lgkmcnt=0)
        self.lds_done = self.lds_issued
        if d64:
            self.tile64(tail=True)
        # a wave whose rows are done before the workgroup's last tile (causal) still owes the others its barriers and its
        # share of the LDS-DMA pieces: tiles j = wnt .. nt-1 without arithmetic (tile j: DMA of tile j+2, barrier of j+1)
        self.label(skip)
        self.emit("s_cmp_ge_i32", None, [SN("j"), SN("nt")])
        self.emit("s_cbranch_scc1", None, [], target=done)
        if "dma" not in cfg.abl:
            for i in range(npieces):
                self.dma_piece(i)
            for i in range(npieces):
                self.dma_advance(i)
            self.wr_advance()
        self.emit("s_waitcnt", None, [], vmcnt=npieces if "dma" not in cfg.abl else 0)
        self.emit("s_barrier")
        self.emit("s_add_u32", SN("j"), [SN("j"), I(1)])
        self.emit("s_branch", None, [], target=skip)
        self.label(done)
        self.emit("s_waitcnt", None, [], vmcnt=0)
        self.emit("s_branch", None, [], target=fin)
        for lbl, back, kb in self.outofline:
            self.mask_section(lbl, back, kb)
        self.label(fin)
        return self.ins


def write_inc(path):
    lines = ["// GENERATED by tools/dq4gen.py -- do not edit.  Instruction streams of attn_dq16_p4 (see the generator's",
             "// header for the register map and the tile table).", "#pragma once", ""]
    lines.append("#define MFA_DQ4_OWNED_VGPRS " + ", ".join('"v%d"' % i for i in range(FIRST_OWNED_VGPR, 256)))
    lines.append("")
    lines.append("// X(name, stamps the shader clock, applies the softmax scale in fp32, head-dimension bucket)")
    lines.append("#define MFA_DQ4_STREAM_LIST(X) \\")
    for name, cfg in VARIANTS.items():
        lines.append("  X(%s, %d, %d, %d) \\" % (name, cfg.prof, cfg.exact, cfg.D))
    lines.append("")
    lines.append("// transposed K / V (model-verified, no kernel yet): X(name, applies the softmax scale in fp32)")
    lines.append("#define MFA_DQ4_TR_STREAM_LIST(X) \\")
    for name, cfg in TR_VARIANTS.items():
        lines.append("  X(%s, %d) \\" % (name, cfg.exact))
    lines.append("")
    lines.append("#define MFA_DQ4_DEV_STREAM_LIST(X) \\")
    for name, cfg in VARIANTS.items():
        if cfg.prof:
            lines.append("  X(%s) \\" % name)
    lines.append("")
    lines.append("")
    for name, cfg in list(VARIANTS.items()) + list(TR_VARIANTS.items()):
        ins = Stream(cfg).build()
        txt = render(ins)
        lines.append("// %s: dtype=%s prof=%d exact=%d D=%d -- %d instructions" % (name, cfg.dtype, cfg.prof, cfg.exact, cfg.D, len(txt)))
        lines.append("#define MFA_DQ4_STREAM_%s \\" % name)
        for t in txt:
            lines.append('  "%s\\n\\t" \\' % t)
        lines.append('  ""')
        lines.append("")
    with open(path, "w") as f:
        f.write("\n".join(lines))


VARIANTS = {
    "BF16_FOLD": Cfg("bf16"),
    "F16_FOLD": Cfg("f16"),
    "BF16_EXACT": Cfg("bf16", exact=1),
    "F16_EXACT": Cfg("f16", exact=1),
    "BF16_FOLD_PROF": Cfg("bf16", prof=1),
    "D64_BF16_FOLD": Cfg("bf16", D=64),
    "D64_F16_FOLD": Cfg("f16", D=64),
    "D64_BF16_EXACT": Cfg("bf16", exact=1, D=64),
    "D64_F16_EXACT": Cfg("f16", exact=1, D=64),
    "D64_BF16_FOLD_PROF": Cfg("bf16", prof=1, D=64),
    "BF16_FOLD_DEFER_PROF": Cfg("bf16", prof=1, defer=True),     # developer: the 64 bucket's group order at D = 128
    # timing-only ablations (WRONG RESULTS; developer build): fillers left out of the tile
    "ABL_DMA": Cfg("bf16", prof=1, abl=("dma",)),
    "ABL_EXP": Cfg("bf16", prof=1, abl=("exp",)),
    "ABL_MULPACK": Cfg("bf16", prof=1, abl=("mulpack",)),
    "ABL_READS": Cfg("bf16", prof=1, abl=("reads",)),
    "ABL_ALL": Cfg("bf16", prof=1, abl=("dma", "exp", "mulpack", "reads")),
}

TR_VARIANTS = {
    "BF16_FOLD_TR": Cfg("bf16", tr=1),
    "F16_FOLD_TR": Cfg("f16", tr=1),
    "BF16_EXACT_TR": Cfg("bf16", exact=1, tr=1),
    "F16_EXACT_TR": Cfg("f16", exact=1, tr=1),
}

if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(here, "..", "metal_flash_attention_amd", "csrc", "attn_dq16_p4_stream.inc")
    write_inc(out)
    ins = Stream(VARIANTS["BF16_FOLD"]).build()
    print("wrote", os.path.normpath(out), "-", len(ins), "instructions in the default stream")
