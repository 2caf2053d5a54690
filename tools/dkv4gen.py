#!/usr/bin/env python3
"""Generator of the hand-placed instruction stream of attn_dkv16_p4 (csrc/attn_dkv16_p4.h): backwardKeyValue for
D <= 128 with 16-bit Q/K/V/dO.

A workgroup is four waves x 64 keys (one wave per SIMD, the whole 512-entry register file); it walks the query rows
in 32-row steps.  Per step and wave, 68 matrix instructions (32x32x16):

    S   (18)  S'  = Q K'^T - L      K' = K * log2(e)/sqrt(D) in registers; L enters through an extra k-step whose A operand
                                    carries (L_hi, L_lo) per row and whose B operand is -1.0 in k-slots 0, 1
    P   (18)  dP' = dO V^T - D/scale   (the same trick with the D term)           | exp2(S') -> P, 16-bit pack of P
    V   (16)  dV^T += dO^T P                                                       | dS' = P * dP', 16-bit pack (in place)
    K   (16)  dK^T += Q^T dS'         (dK = scale * the accumulated sum, applied by the epilogue)

Register map (fixed: hipcc can neither index sub-registers of an asm operand nor place fillers between asm matrix
instructions without a wait state per statement boundary):

    a[0:127]    dV^T accumulators  (db, kb) -> 16 (2 db + kb)        lane = key, registers = head-dimension rows
    a[128:255]  dK^T accumulators  (db, kb) -> 128 + 16 (2 db + kb)
    v[20:23]    L / D values of the next step as loaded, a temporary, the mask constant
    v[24:27]    B operand of the extra k-step  (-1.0 pattern, 0, 0, 0)
    v[28:31]    (L pair, 0, D pair, 0): A operands of the extra k-step are v[28:31] and v[30:33]
    v[32:95]    K' fragments (kb, ks) -> 32 + 4 (8 kb + ks)           B operands of S
    v[96:159]   V  fragments (kb, ks) -> 96 + 4 (8 kb + ks)           B operands of dP
    v[160:191]  S' / P (fp32)      kb -> 160 + 16 kb
    v[192:223]  dP' / dS'          kb -> 192 + 16 kb ; the 16-bit dS' fragments (kb, u) are packed in place: + 4 u
    v[224:239]  16-bit P fragments (kb, u) -> 224 + 4 (2 kb + u)
    v[240:255]  ring of four A-operand fragments read from LDS (Q rows, dO rows, dO^T, Q^T), fragment i in slot i % 4
    v[0:19]     left to hipcc (operands of the statement)

LDS: ring of four stages {Q tile | dO tile}, each tile [D/32][32 rows][32 elements] with the four 16-byte chunks of a
64-byte row XOR-swizzled by (row >> 2) & 3 (the layout of attn_dkv16_rs.h: serves ds_read_b128 row fragments and
ds_read_b64_tr_b16 alike), filled by LDS-DMA two steps ahead.  One barrier per step.

The instruction list is rendered as an asm template and executed by tools/dkv4sim.py (the lane-exact model of
tools/p4sim.py) against a float64 backward pass on the CPU: tests/test_dkv4_stream.py.

Usage: python tools/dkv4gen.py   (rewrites metal_flash_attention_amd/csrc/attn_dkv16_p4_stream.inc)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from p4gen import A, F, I, M0, SN, V, VCC, VN, Stream as _P4Stream, render  # noqa: E402

# ---------------------------------------------------------------- register map
DV_BASE, DK_BASE = 0, 128
T_LRAW, T_DRAW, T_T0, T_MASKV = 20, 21, 22, 23
ONES = 24
LP, DPR = 28, 30
KF, VF = 32, 96
SP, DPB, P16, AF = 160, 192, 224, 240
FIRST_OWNED_VGPR = 20

STAGE, GIMG, RING = 16384, 8192, 4     # bytes per stage, offset of the dO tile in a stage, stages
KVBACK = 32768                         # bytes per wave of the K' / V fragment hand-over area (see build())

# named operands of the asm statement (attn_dkv16_p4.h); order = operand order
INOUT_V = ["qoff0", "qoff1", "goff0", "goff1", "ldoff", "ra0", "ra1", "ta0", "ta1"]
TMP_S = ["j", "stg", "delta", "wr", "t0", "t1", "pa", "pb", "pc", "pd", "plast"]   # "=&s" 32-bit temporaries (p*: PROF streams)
TMP_S64 = ["ptime"]
IN_V = ["onesw", "tk", "kvback"]
IN_S = ["qres", "gres", "lres", "dres", "nsteps", "rscale", "qinc", "ginc", "ldinc", "wr0", "ringend", "maskuntil",
        "rscale2", "scale2x2"]


class Cfg:
    def __init__(self, dtype="bf16", lprec="f32", dprec="f32", prof=0, abl=(), exact=0, gdtype=None, D=128, tr=0):
        """dtype: type of Q, K, V, dO and of the packed P / dS'; lprec / dprec: storage types of L and D.
        exact: K stays as stored and the softmax scale is applied in fp32, P = exp2(scale2 * (Q K^T - L / scale2)) (one packed
        multiply per two scores more); otherwise K arrives pre-multiplied by scale2, rounded to the 16-bit type."""
        self.dtype, self.lprec, self.dprec, self.prof, self.exact = dtype, lprec, dprec, prof, exact
        # gdtype: storage type of dO when it differs from Q / K / V (the reference's own mix: FP16 Q, K, V with BF16 dO,
        # +Precisions.swift:13-17).  The two products that read dO -- dP = dO V^T and dV^T += dO^T P -- then run in dO's type
        # (V is converted once per workgroup, P is packed to that type), S and dK^T += Q^T dS' in the type of Q and K.  The
        # extra k-steps share one B operand: -2.0, the same bit pattern 0xC000 in both types; the pairs carry half the term.
        self.gdtype = gdtype or dtype
        self.mix = self.gdtype != self.dtype
        # D: head-dimension bucket (128, or 64: four k-steps and two head-dimension blocks -- 36 matrix instructions per step,
        # one LDS-DMA piece per wave and operand tile; the register map keeps its D = 128 positions)
        self.D, self.nks, self.ndb = D, D // 16, D // 32
        assert D in (64, 128)
        # tr (D = 128; model-verified, NOT yet behind a kernel -- DESIGN.md 10.4): Q and dO stored TRANSPOSED ([D][rows]).  A step's
        # tile in the source orientation is [128 elements][32 rows x 2 bytes] = the same [4][32][64 bytes] image with the roles
        # of its two read recipes exchanged: Q / dO ROW fragments (A of S' and dP') by transposing reads (addresses ra0 / ra1 =
        # rows + 0 / + 8 of a 16-element step; the contraction index arrives in accumulator-register order, so the kernel parks
        # the K' and V fragments in that order), dO^T / Q^T fragments (A of dV^T and dK^T) as two 8-byte reads of the lane's
        # element row, chunks 2 u and 2 u + 1 at 8 hi (P and dS' hold their rows in that order): four addresses ta0..ta3.
        # Whole steps only (R % 32 == 0): rows beyond R are no longer zeros read past the end of a buffer.
        self.tr = tr
        assert not (tr and D != 128)
        self.abl = frozenset(abl)


def kf(kb, ks):
    return V(KF + 4 * (8 * kb + ks), 4)


def vf(kb, ks):
    return V(VF + 4 * (8 * kb + ks), 4)


def sp_blk(kb):
    return V(SP + 16 * kb, 16)


def dp_blk(kb):
    return V(DPB + 16 * kb, 16)


def p16(kb, u):
    return V(P16 + 4 * (2 * kb + u), 4)


def ds16(kb, u):
    return V(DPB + 16 * kb + 4 * u, 4)


def af(i):
    return V(AF + 4 * (i % 4), 4)


def af_half(i, h):
    return V(AF + 4 * (i % 4) + 2 * h, 2)


def dv_acc(db, kb):
    return A(DV_BASE + 16 * (2 * db + kb), 16)


def dk_acc(db, kb):
    return A(DK_BASE + 16 * (2 * db + kb), 16)


# ---- the 68 matrix instructions of a step.  Fragment i (0..31) of the A-operand ring feeds two consecutive ones.
def frag_first(i):
    return (2 if i < 8 else 4) + 2 * i          # 2 + 2i (S), 20 + 2 (i - 8) (dP), 36 + 2 (i - 16) (dV), 52 + 2 (i - 24) (dK)


PACKED_MUL = int(os.environ.get('MFA_GEN_PACKED_MUL', '0'))   # v_pk_mul_f32 for dS' = P * dP': no faster on gfx950 (dQ 1.5 % slower), see DESIGN.md
N_MFMA = 68


class Stream(_P4Stream):
    def __init__(self, cfg):
        _P4Stream.__init__(self, cfg)
        self.frag_rid = {}

    # ---------------------------------------------------------------- LDS fragment reads
    def frag_read(self, i):
        """issue the LDS read(s) of ring fragment i of the current step (addresses: ra* row reads, ta* transposing reads)"""
        if self.cfg.D == 64:
            return self.frag_read64(i)
        if self.cfg.tr:
            if i < 16:      # rows 16 (ks & 1) (+ 8) of sub-image ks >> 1
                ks = i % 8
                off = (0 if i < 8 else GIMG) + (ks >> 1) * 2048 + (ks & 1) * 1024
                self.lds_read("ds_read_b64_tr_b16", af_half(i, 0), VN("ra0"), off, note="%s^T rows ks%d" % ("Q" if i < 8 else "dO", ks))
                self.frag_rid[i] = self.lds_read("ds_read_b64_tr_b16", af_half(i, 1), VN("ra1"), off)
            else:           # element row 32 db + lane % 32: 8 bytes of chunks 2 u, 2 u + 1
                u, db = divmod(i % 8, 4)
                off = (GIMG if i < 24 else 0) + db * 2048
                self.lds_read("ds_read_b64", af_half(i, 0), VN("ta%d" % (2 * u)), off, note="%s u%d db%d" % ("dO" if i < 24 else "Q", u, db))
                self.frag_rid[i] = self.lds_read("ds_read_b64", af_half(i, 1), VN("ta%d" % (2 * u + 1)), off)
            return
        if i < 16:          # row fragment ks of Q (i < 8) or dO: 16 bytes at chunk (2 ks + hi) ^ swizzle of row (lane & 31)
            ks = i % 8
            img = 0 if i < 8 else GIMG
            self.frag_rid[i] = self.lds_read("ds_read_b128", af(i), VN("ra%d" % (ks & 1)), img + (ks >> 1) * 2048,
                                             note="%s rows ks%d" % ("Q" if i < 8 else "dO", ks))
        else:               # transposed fragment (u, db) of dO (i < 24) or Q: rows 16 u + {0..3 | 8..11} + 4 hi
            u, db = divmod(i % 8, 4)
            img = GIMG if i < 24 else 0
            off = img + db * 2048 + u * 1024
            self.lds_read("ds_read_b64_tr_b16", af_half(i, 0), VN("ta0"), off, note="%s^T u%d db%d" % ("dO" if i < 24 else "Q", u, db))
            self.frag_rid[i] = self.lds_read("ds_read_b64_tr_b16", af_half(i, 1), VN("ta1"), off)

    def frag_read64(self, i):
        """D = 64: fragments 0..3 Q rows (ks), 4..7 dO rows, 8..11 dO^T (u, db), 12..15 Q^T (u, db)"""
        if i < 8:
            ks = i % 4
            img = 0 if i < 4 else GIMG
            self.frag_rid[i] = self.lds_read("ds_read_b128", af(i), VN("ra%d" % (ks & 1)), img + (ks >> 1) * 2048,
                                             note="%s rows ks%d" % ("Q" if i < 4 else "dO", ks))
        else:
            u, db = divmod(i % 4, 2)
            img = GIMG if i < 12 else 0
            off = img + db * 2048 + u * 1024
            self.lds_read("ds_read_b64_tr_b16", af_half(i, 0), VN("ta0"), off, note="%s^T u%d db%d" % ("dO" if i < 12 else "Q", u, db))
            self.frag_rid[i] = self.lds_read("ds_read_b64_tr_b16", af_half(i, 1), VN("ta1"), off)

    # ---------------------------------------------------------------- global -> LDS / registers
    def dma_stage(self, pieces=None):
        if self.cfg.D == 64:      # one 1 KiB piece per wave and operand tile
            for name, res, base in (("qoff0", "qres", 0), ("goff0", "gres", GIMG)):
                if pieces is None or name in pieces:
                    self.emit("s_add_u32", M0, [SN("wr"), I(base)])
                    self.emit("buffer_load_dwordx4_lds", None, [VN(name), SN(res, 4)])
            return
        self._dma_stage128(range(4) if pieces is None else pieces)

    def dma_advance(self, pieces=None):
        if self.cfg.D == 64:
            for name, inc in (("qoff0", "qinc"), ("goff0", "ginc")):
                if pieces is None or name in pieces:
                    self.emit("v_add_u32_e64", VN(name), [VN(name), SN(inc)], clamp=1)
            return
        self._dma_advance128(range(4) if pieces is None else pieces)

    def _dma_stage128(self, pieces=range(4)):
        """this wave's four 1 KiB pieces of the stage `wr` points at: Q pieces 2w, 2w+1, dO pieces 2w, 2w+1"""
        for i in pieces:
            name, res, base = (("qoff%d" % i, "qres", 0) if i < 2 else ("goff%d" % (i - 2), "gres", GIMG))
            self.emit("s_add_u32", M0, [SN("wr"), I(base + (i & 1) * 1024)])
            self.emit("buffer_load_dwordx4_lds", None, [VN(name), SN(res, 4)])

    def _dma_advance128(self, pieces=range(4)):
        for i in pieces:
            name, inc = ("qoff%d" % i, "qinc") if i < 2 else ("goff%d" % (i - 2), "ginc")
            self.emit("v_add_u32_e64", VN(name), [VN(name), SN(inc)], clamp=1)

    def wr_advance(self):
        self.emit("s_add_u32", SN("wr"), [SN("wr"), I(STAGE)])
        self.emit("s_cmp_ge_u32", None, [SN("wr"), SN("ringend")])
        self.emit("s_cselect_b32", SN("t1"), [I(RING * STAGE), I(0)])
        self.emit("s_sub_u32", SN("wr"), [SN("wr"), SN("t1")])

    def ld_load(self):
        """L and D of the rows of the NEXT step, one value per lane (lane & 31 = row); rows past R read as zero"""
        for reg, prec, res in ((T_LRAW, self.cfg.lprec, "lres"), (T_DRAW, self.cfg.dprec, "dres")):
            self.emit("buffer_load_dword" if prec == "f32" else "buffer_load_ushort", V(reg), [VN("ldoff"), SN(res, 4)])
        self.emit("v_add_u32", VN("ldoff"), [SN("ldinc"), VN("ldoff")])

    def ld_convert_ops(self):
        """closures turning the loaded L / D into the 16-bit pairs the extra k-steps consume (exact sums: hi + lo)"""
        cfg = self.cfg
        ops = []
        for raw, prec, dst, isd in ((T_LRAW, cfg.lprec, LP, False), (T_DRAW, cfg.dprec, DPR, True)):
            x, t = V(raw), V(T_T0)
            ptype = cfg.gdtype if isd else cfg.dtype          # the pair travels in the type of the product it joins
            mask = 0xFFFF0000 if ptype == "bf16" else 0xFFFFE000
            if prec == "f16":
                ops.append(lambda x=x: self.emit("v_cvt_f32_f16", x, [x]))
            elif prec == "bf16":
                ops.append(lambda x=x: self.emit("v_lshlrev_b32", x, [I(16), x]))
            if isd:     # the buffer holds D * scale (+Softmax.swift:472-503); dP' needs D itself
                ops.append(lambda x=x: self.emit("v_mul_f32", x, [SN("rscale"), x]))
            elif cfg.exact or cfg.mix:   # S'' = Q K^T - L / scale2 (mix: and / or the half that the -2.0 operand doubles)
                ops.append(lambda x=x: self.emit("v_mul_f32", x, [SN("rscale2"), x]))
            ops.append(lambda x=x, t=t: self.emit("v_and_b32", t, [I(mask), x]))          # hi
            ops.append(lambda x=x, t=t: self.emit("v_sub_f32", x, [x, t]))                # remainder
            ops.append(lambda x=x: self.emit("v_and_b32", x, [I(mask), x]))               # lo
            ops.append(lambda x=x, t=t, dst=dst, ptype=ptype: self.emit("v_cvt_pk_%s_f32" % ptype, V(dst), [t, x]))
        return ops

    def addr_advance(self, names):
        for n in names:
            self.emit("v_add_u32", VN(n), [SN("delta"), VN(n)])

    def stage_delta(self):
        self.emit("s_add_u32", SN("stg"), [SN("stg"), I(1)])
        self.emit("s_and_b32", SN("stg"), [SN("stg"), I(RING - 1)])
        self.emit("s_cmp_eq_u32", None, [SN("stg"), I(0)])
        self.emit("s_cselect_b32", SN("t1"), [I(RING * STAGE), I(0)])
        self.emit("s_sub_u32", SN("delta"), [I(STAGE), SN("t1")])

    # ---------------------------------------------------------------- arithmetic fillers
    def exp_op(self, kb, r):
        x = V(SP + 16 * kb + r)
        self.emit("v_exp_f32", x, [x])

    def packp_op(self, kb, u, w):
        r = 8 * u + 2 * w
        self.emit("v_cvt_pk_%s_f32" % self.cfg.gdtype, V(P16 + 4 * (2 * kb + u) + w), [V(SP + 16 * kb + r), V(SP + 16 * kb + r + 1)])

    def scale_op(self, kb, r):   # exact streams: two scores times scale2
        x = V(SP + 16 * kb + r, 2)
        self.emit("v_pk_mul_f32", x, [x, SN("scale2x2", 2)])

    def mul_op(self, kb, r):     # dS' = P * dP', two at a time
        if PACKED_MUL:
            x = V(DPB + 16 * kb + r, 2)
            self.emit("v_pk_mul_f32", x, [V(SP + 16 * kb + r, 2), x])
        else:
            for t in range(2):
                x = V(DPB + 16 * kb + r + t)
                self.emit("v_mul_f32", x, [V(SP + 16 * kb + r + t), x])

    def packds_op(self, kb, u, w):
        r = 8 * u + 2 * w
        self.emit("v_cvt_pk_%s_f32" % self.cfg.dtype, V(DPB + 16 * kb + 4 * u + w), [V(DPB + 16 * kb + r), V(DPB + 16 * kb + r + 1)])

    def mask_section(self, lbl, back):
        """causal steps: key c of row r contributes iff c <= r + (C - R); tk = key - (C - R) - 4 hi - 32 j' - first row"""
        self.label(lbl)
        self.emit("s_lshl_b32", SN("t0"), [SN("j"), I(5)])
        self.emit("v_subrev_u32", V(T_T0), [SN("t0"), VN("tk")])                     # key - coff - 4 hi - row0(step)
        for kb in range(2):
            if kb:
                self.emit("v_add_u32", V(T_T0), [I(32), V(T_T0)])
            for r in range(16):
                x = V(SP + 16 * kb + r)
                self.emit("v_cmp_lt_i32", VCC, [I((r & 3) + 8 * (r >> 2)), V(T_T0)])   # masked: row < key
                self.emit("v_cndmask_b32", x, [x, V(T_MASKV), VCC])
        self.emit("s_branch", None, [], target=back)

    # ---------------------------------------------------------------- one step
    def step(self, first=False):
        if self.cfg.D == 64:
            return self.step64()
        cfg = self.cfg
        fill = [[] for _ in range(N_MFMA)]

        def at(g, fn):
            fill[g].append(fn)

        # ---- fragment reads: fragment i + 4 takes the slot of fragment i once both of its matrix instructions are issued
        for i in range(28):
            g = frag_first(i) + 1
            if i + 4 >= 16:      # two reads
                at(g, lambda i=i: self.frag_read(i + 4))
            else:
                at(g, lambda i=i: self.frag_read(i + 4))
        # fragments 0..3 of the NEXT step: behind the barrier (gap 60), in the slots of fragments 28..31
        # ---- LDS-DMA of step t+2 (stage wr): one piece per even gap of the S phase (the odd gaps carry the fragment reads);
        # the offsets advance in the following even gaps
        if "dma" not in cfg.abl:
            for i in range(4):
                at(2 + 2 * i, lambda i=i: self._dma_stage128([i]))
                at(10 + 2 * i, lambda i=i: self._dma_advance128([i]))
            at(18, lambda: self.wr_advance())
        # ---- row-read addresses move to the next stage once the last row fragment (15) is requested (gap 27)
        at(0, lambda: self.stage_delta())
        at(28, lambda: self.addr_advance(["ra0", "ra1"]))
        # ---- causal mask on S' before the first exp2
        mask_lbl, mask_back = self.newlabel("MASK"), self.newlabel("MASKBACK")

        def mask_branch():
            self.emit("s_cmp_lt_i32", None, [SN("j"), SN("maskuntil")])
            self.emit("s_cbranch_scc1", None, [], target=mask_lbl)
            self.label(mask_back)
        at(21, mask_branch)
        self.outofline.append((mask_lbl, mask_back))
        # ---- P = exp2(S') from gap 22 on, group (kb, u) by group in the order the dV products consume them; the 16-bit pack
        # of a group trails its exps by a group (a transcendental result is not read back to back)
        order = [(kb, u) for u in range(2) for kb in range(2)]
        seq = []
        for n, (kb, u) in enumerate(order):
            if cfg.exact:
                seq += [lambda kb=kb, r=r: self.scale_op(kb, r) for r in range(8 * u, 8 * u + 8, 2)]
            seq += [lambda kb=kb, r=r: self.exp_op(kb, r) for r in range(8 * u, 8 * u + 8)]
            if n >= 1:
                pkb, pu = order[n - 1]
                seq += [lambda kb=pkb, u=pu, w=w: self.packp_op(kb, u, w) for w in range(4)]
        seq += [lambda kb=order[3][0], u=order[3][1], w=w: self.packp_op(kb, u, w) for w in range(4)]
        # even gaps take more of them: the odd ones carry the transposing reads (two per gap) and their waits
        quota = (5, 3) if cfg.exact else (4, 2)
        g, used = 22, 0
        for fn in seq:
            if used == quota[g & 1]:
                g, used = g + 1, 0
            assert g <= 37
            at(g, fn)
            used += 1
        # ---- dS' = P * dP' (packed multiplies) and its 16-bit packs (in place), four per even gap: u = 0 multiplied in gaps
        # 38, 40 and packed in 44, 46 (the dK products start at 52); u = 1 multiplied in 42, 48 and packed in 52, 54 (needed
        # from 60)
        slots = {("mul", 0): (38, 40), ("mul", 1): (42, 48), ("pack", 0): (44, 46), ("pack", 1): (52, 54)}
        for u in range(2):
            for kb in range(2):
                for n, r in enumerate(range(8 * u, 8 * u + 8, 2)):
                    at(slots[("mul", u)][kb] + (0 if PACKED_MUL else n // 2), lambda kb=kb, r=r: self.mul_op(kb, r))
                for w in range(4):
                    at(slots[("pack", u)][kb], lambda kb=kb, u=u, w=w: self.packds_op(kb, u, w))
        # ---- the seam to the next step (gap 60): own DMA pieces of step t+1 and its L / D have landed; barrier; then the
        # first fragments of step t+1, its L / D pairs, the loads of L / D for t+2, the transposing-read addresses
        def seam():
            self.emit("s_waitcnt", None, [], vmcnt=4 if "dma" not in cfg.abl else 0)
            self.emit("s_barrier")
            self.addr_advance(["ta0", "ta1", "ta2", "ta3"] if cfg.tr else ["ta0", "ta1"])
        at(60, seam)
        conv = self.ld_convert_ops()
        for n, fn in enumerate(conv):
            at(61 + (n * 6) // len(conv), fn)
        at(67, lambda: self.ld_load())
        for i in range(4):
            at(61 + 2 * i, lambda i=i: self.frag_read(i))

        # ---- emit
        mm = []   # (dst, a, b, c, fragment or None)
        for kb in range(2):
            mm.append((sp_blk(kb), V(LP, 4), V(ONES, 4), I(0), None))
        for ks in range(8):
            for kb in range(2):
                mm.append((sp_blk(kb), af(ks), kf(kb, ks), sp_blk(kb), ks))
        for kb in range(2):
            mm.append((dp_blk(kb), V(DPR, 4), V(ONES, 4), I(0), None))
        for ks in range(8):
            for kb in range(2):
                mm.append((dp_blk(kb), af(8 + ks), vf(kb, ks), dp_blk(kb), 8 + ks))
        for u in range(2):
            for db in range(4):
                for kb in range(2):
                    mm.append((dv_acc(db, kb), af(16 + 4 * u + db), p16(kb, u), dv_acc(db, kb), 16 + 4 * u + db))
        for u in range(2):
            for db in range(4):
                for kb in range(2):
                    mm.append((dk_acc(db, kb), af(24 + 4 * u + db), ds16(kb, u), dk_acc(db, kb), 24 + 4 * u + db))
        assert len(mm) == N_MFMA
        stamps = {18: "pa", 36: "pb", 52: "pc"}
        for g, (d, a_, b_, c_, fr) in enumerate(mm):
            if g in stamps:
                self.stamp(stamps[g])
            if fr is not None:
                self.lds_need(self.frag_rid[fr])
            # products 18..51 (dP and dV^T) read dO: they run in its type
            self.emit("v_mfma_f32_32x32x16_" + (cfg.gdtype if 18 <= g < 52 else cfg.dtype), d, [a_, b_, c_])
            for fn in fill[g]:
                fn()
        self.stamp("pd")

    def step64(self):
        """D = 64: 36 matrix instructions -- S 0..9 (two extra k-steps + 4 k-steps x 2 key blocks), dP 10..19, dV^T 20..27
        (u, db, kb), dK^T 28..35.  The same pipeline as D = 128, compressed: the VALU work of a product's result still runs
        beside the next product's matrix instructions."""
        cfg = self.cfg
        NM = 36
        fill = [[] for _ in range(NM)]

        def at(g, fn):
            fill[g].append(fn)

        def first(i):
            return (2 + 2 * i) if i < 4 else (12 + 2 * (i - 4)) if i < 8 else (20 + 2 * (i - 8)) if i < 12 else (28 + 2 * (i - 12))
        for i in range(12):
            at(first(i) + 1, lambda i=i: self.frag_read(i + 4))
        if "dma" not in cfg.abl:
            at(2, lambda: self.dma_stage(["qoff0"]))
            at(4, lambda: self.dma_stage(["goff0"]))
            at(6, lambda: self.dma_advance(["qoff0"]))
            at(8, lambda: self.dma_advance(["goff0"]))
            at(12, lambda: self.wr_advance())
        at(0, lambda: self.stage_delta())
        at(10, lambda: self.addr_advance(["ra0", "ra1"]))          # the last row fragment (7) is requested in gap 9
        mask_lbl, mask_back = self.newlabel("MASK"), self.newlabel("MASKBACK")

        def mask_branch():
            self.emit("s_cmp_lt_i32", None, [SN("j"), SN("maskuntil")])
            self.emit("s_cbranch_scc1", None, [], target=mask_lbl)
            self.label(mask_back)
        at(11, mask_branch)
        self.outofline.append((mask_lbl, mask_back))
        # P = exp2(S') from gap 12, five per gap; the 16-bit packs of u = 0 are done by gap 18 (dV^T starts at 20), of u = 1 by 21
        order = [(kb, u) for u in range(2) for kb in range(2)]
        seq = []
        for n, (kb, u) in enumerate(order):
            if cfg.exact:
                seq += [lambda kb=kb, r=r: self.scale_op(kb, r) for r in range(8 * u, 8 * u + 8, 2)]
            seq += [lambda kb=kb, r=r: self.exp_op(kb, r) for r in range(8 * u, 8 * u + 8)]
            if n >= 1:
                pkb, pu = order[n - 1]
                seq += [lambda kb=pkb, u=pu, w=w: self.packp_op(kb, u, w) for w in range(4)]
        seq += [lambda kb=order[3][0], u=order[3][1], w=w: self.packp_op(kb, u, w) for w in range(4)]
        per = 7 if cfg.exact else 5
        for n, fn in enumerate(seq):
            assert 12 + n // per <= 21
            at(12 + n // per, fn)
        # dS' = P * dP' and its packs: u = 0 in gaps 22..27 (the dK^T products start at 28), u = 1 in 28..31 (needed from 32)
        seq0, seq1 = [], []
        for kb in range(2):
            seq0 += [lambda kb=kb, r=r: self.mul_op(kb, r) for r in range(0, 8, 2)]
        for kb in range(2):
            seq0 += [lambda kb=kb, w=w: self.packds_op(kb, 0, w) for w in range(4)]
        for kb in range(2):
            seq1 += [lambda kb=kb, r=r: self.mul_op(kb, r) for r in range(8, 16, 2)]
        for kb in range(2):
            seq1 += [lambda kb=kb, w=w: self.packds_op(kb, 1, w) for w in range(4)]
        for n, fn in enumerate(seq0):
            at(22 + (n * 6) // len(seq0), fn)
        for n, fn in enumerate(seq1):
            at(28 + (n * 4) // len(seq1), fn)

        def seam():
            self.emit("s_waitcnt", None, [], vmcnt=2 if "dma" not in cfg.abl else 0)
            self.emit("s_barrier")
            self.addr_advance(["ta0", "ta1"])
        at(28, seam)
        conv = self.ld_convert_ops()
        for n, fn in enumerate(conv):
            at(29 + (n * 6) // len(conv), fn)
        at(35, lambda: self.ld_load())
        for i in range(4):
            at(29 + 2 * i, lambda i=i: self.frag_read(i))
        mm = []
        for kb in range(2):
            mm.append((sp_blk(kb), V(LP, 4), V(ONES, 4), I(0), None))
        for ks in range(4):
            for kb in range(2):
                mm.append((sp_blk(kb), af(ks), kf(kb, ks), sp_blk(kb), ks))
        for kb in range(2):
            mm.append((dp_blk(kb), V(DPR, 4), V(ONES, 4), I(0), None))
        for ks in range(4):
            for kb in range(2):
                mm.append((dp_blk(kb), af(4 + ks), vf(kb, ks), dp_blk(kb), 4 + ks))
        for u in range(2):
            for db in range(2):
                for kb in range(2):
                    mm.append((dv_acc(db, kb), af(8 + 2 * u + db), p16(kb, u), dv_acc(db, kb), 8 + 2 * u + db))
        for u in range(2):
            for db in range(2):
                for kb in range(2):
                    mm.append((dk_acc(db, kb), af(12 + 2 * u + db), ds16(kb, u), dk_acc(db, kb), 12 + 2 * u + db))
        assert len(mm) == NM
        stamps = {10: "pa", 20: "pb", 28: "pc"}
        for g, (d, a_, b_, c_, fr) in enumerate(mm):
            if g in stamps:
                self.stamp(stamps[g])
            if fr is not None:
                self.lds_need(self.frag_rid[fr])
            self.emit("v_mfma_f32_32x32x16_" + (cfg.gdtype if 10 <= g < 28 else cfg.dtype), d, [a_, b_, c_])
            for fn in fill[g]:
                fn()
        self.stamp("pd")

    # ---------------------------------------------------------------- whole traversal
    def build(self):
        cfg = self.cfg
        self.outofline = []
        # ---- K' and V fragments: attn_dkv16_p4.h (hipcc: bounds, zero fill, the K prescale) parks them in LDS, lane-linear,
        # 32 x 1 KiB per wave at `kvback`; they move to their fixed registers here
        self.emit("s_waitcnt", None, [], lgkmcnt=0)
        for i in range(32):
            if cfg.D == 64 and (i % 8) >= 4:
                continue           # (kb, ks) with ks >= 4 does not exist; the hand-over slots keep their D = 128 positions
            self.lds_read("ds_read_b128", V((KF if i < 16 else VF) + 4 * (i % 16), 4), VN("kvback"), i * 1024)
        for r in range(256):
            self.emit("v_accvgpr_write_b32", A(r), [I(0)])
        self.emit("v_mov_b32", V(ONES), [VN("onesw")])
        for r in (ONES + 1, ONES + 2, ONES + 3, LP, LP + 1, DPR, DPR + 1):
            self.emit("v_mov_b32", V(r), [I(0)])
        self.emit("v_mov_b32", V(T_MASKV), [F(-(0.875 / 1.44269504089) * 3.402823466e+38)])   # +Softmax.swift:242-243
        self.lds_flush()
        self.emit("s_barrier")                                # every wave has its fragments: the ring may be written
        # ---- stages 0 and 1, L / D of step 0
        self.emit("s_mov_b32", SN("wr"), [SN("wr0")])
        self.dma_stage()
        self.dma_advance()
        self.wr_advance()
        self.ld_load()
        self.dma_stage()
        self.dma_advance()
        self.wr_advance()
        self.emit("s_waitcnt", None, [], vmcnt=4 if cfg.D == 128 else 2)
        for fn in self.ld_convert_ops():
            fn()
        self.emit("s_barrier")
        self.ld_load()
        self.emit("s_mov_b32", SN("j"), [I(0)])
        self.emit("s_mov_b32", SN("stg"), [I(0)])
        for acc in ("pa", "pb", "pc", "pd"):
            self.emit("s_mov_b32", SN(acc), [I(0)])
        for i in range(4):
            self.frag_read(i)
        if cfg.prof:
            self.emit("s_memtime", SN("ptime", 2))
            self.emit("s_waitcnt", None, [], lgkmcnt=0)
            self.lds_done = self.lds_issued
            self.emit("s_mov_b64", VCC, [SN("ptime", 2)])
            self.emit("s_mov_b32", SN("plast"), [("vcc_lo",)])
        loop, fin = self.newlabel("LOOP"), self.newlabel("FIN")
        self.label(loop)
        # bookkeeping at the loop head: exactly the reads of fragments 0..3 may be in flight (same state at the back edge)
        head_issued = self.lds_issued
        head_outstanding = self.lds_issued - self.lds_done
        self.step()
        assert self.lds_issued - self.lds_done <= (8 if cfg.tr else 4) and self.lds_issued - self.frag_rid[0] == (6 if cfg.tr else 3), \
            "loop-carried LDS queue state"
        if head_outstanding < self.lds_issued - self.lds_done:
            raise AssertionError("the loop head assumes fewer reads in flight than the back edge leaves")
        del head_issued
        self.emit("s_add_u32", SN("j"), [SN("j"), I(1)])
        self.emit("s_cmp_lt_i32", None, [SN("j"), SN("nsteps")])
        self.emit("s_cbranch_scc1", None, [], target=loop)
        self.emit("s_waitcnt", None, [], vmcnt=0, lgkmcnt=0)       # run-ahead DMA (zeros past the end), the last L / D loads
        self.emit("s_branch", None, [], target=fin)
        for lbl, back in self.outofline:
            self.mask_section(lbl, back)
        self.label(fin)
        return self.ins


# ---------------------------------------------------------------- rendering
def write_inc(path):
    lines = ["// GENERATED by tools/dkv4gen.py -- do not edit.  Instruction streams of attn_dkv16_p4 (see the generator's",
             "// header for the register map and the step table).", "#pragma once", ""]
    lines.append("#define MFA_DKV4_OWNED_VGPRS " + ", ".join('"v%d"' % i for i in range(FIRST_OWNED_VGPR, 256)))
    lines.append("")
    lines.append("// X(name, stamps the shader clock, applies the softmax scale in fp32, dO is BF16 next to FP16 Q / K / V, head-dimension bucket)")
    lines.append("#define MFA_DKV4_STREAM_LIST(X) \\")
    for name, cfg in VARIANTS.items():
        lines.append("  X(%s, %d, %d, %d, %d) \\" % (name, cfg.prof, cfg.exact, cfg.mix, cfg.D))
    lines.append("")
    lines.append("// transposed Q / dO (developer kernels only): X(name, applies the softmax scale in fp32, dO is BF16 next to FP16 Q / K / V)")
    lines.append("#define MFA_DKV4_TR_STREAM_LIST(X) \\")
    for name, cfg in TR_VARIANTS.items():
        lines.append("  X(%s, %d, %d) \\" % (name, cfg.exact, int(cfg.mix)))
    lines.append("")
    lines.append("")
    for name, cfg in list(VARIANTS.items()) + list(TR_VARIANTS.items()):
        ins = Stream(cfg).build()
        txt = render(ins)
        n_mfma = sum(1 for i in ins if i.op.startswith("v_mfma"))
        lines.append("// %s: dtype=%s L=%s D=%s prof=%d abl=%s -- %d instructions, %d matrix instructions per step"
                     % (name, cfg.dtype, cfg.lprec, cfg.dprec, cfg.prof, ",".join(sorted(cfg.abl)) or "-", len(txt), n_mfma))
        lines.append("#define MFA_DKV4_STREAM_%s \\" % name)
        for t in txt:
            lines.append('  "%s\\n\\t" \\' % t)
        lines.append('  ""')
        lines.append("")
    with open(path, "w") as f:
        f.write("\n".join(lines))


VARIANTS = {
    "BF16_MIXED": Cfg("bf16", "f16", "bf16"),      # the reference's mixed-precision mode: FP16 L, BF16 D (+Precisions.swift)
    "F16_MIXED": Cfg("f16", "f16", "bf16"),
    "BF16_F32": Cfg("bf16", "f32", "f32", exact=1),  # lowPrecisionInputs alone: FP32 L, D and the attention matrix in FP32 registers
    "F16_F32": Cfg("f16", "f32", "f32", exact=1),
    "F16_DOBF16_MIXED": Cfg("f16", "f16", "bf16", gdtype="bf16"),          # the reference's default low-precision mix
    "F16_DOBF16_F32": Cfg("f16", "f32", "f32", exact=1, gdtype="bf16"),
    "BF16_MIXED_PROF": Cfg("bf16", "f16", "bf16", prof=1),
    "D64_BF16_MIXED": Cfg("bf16", "f16", "bf16", D=64),
    "D64_F16_MIXED": Cfg("f16", "f16", "bf16", D=64),
    "D64_BF16_F32": Cfg("bf16", "f32", "f32", exact=1, D=64),
    "D64_F16_F32": Cfg("f16", "f32", "f32", exact=1, D=64),
    "D64_F16_DOBF16_MIXED": Cfg("f16", "f16", "bf16", gdtype="bf16", D=64),
    "D64_F16_DOBF16_F32": Cfg("f16", "f32", "f32", exact=1, gdtype="bf16", D=64),
}
PRODUCT_STREAMS = ("BF16_MIXED", "F16_MIXED", "BF16_F32", "F16_F32", "F16_DOBF16_MIXED", "F16_DOBF16_F32")
TR_VARIANTS = {
    "BF16_MIXED_TR": Cfg("bf16", "f16", "bf16", tr=1),
    "F16_MIXED_TR": Cfg("f16", "f16", "bf16", tr=1),
    "BF16_F32_TR": Cfg("bf16", "f32", "f32", exact=1, tr=1),
    "F16_F32_TR": Cfg("f16", "f32", "f32", exact=1, tr=1),
    "F16_DOBF16_MIXED_TR": Cfg("f16", "f16", "bf16", gdtype="bf16", tr=1),   # (the two products that read dO^T run in BF16)
    "F16_DOBF16_F32_TR": Cfg("f16", "f32", "f32", exact=1, gdtype="bf16", tr=1),
}

if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(here, "..", "metal_flash_attention_amd", "csrc", "attn_dkv16_p4_stream.inc")
    write_inc(out)
    ins = Stream(VARIANTS["BF16_MIXED"]).build()
    print("wrote", os.path.normpath(out), "-", len(ins), "instructions in the default stream")
