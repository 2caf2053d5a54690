#!/usr/bin/env python3
"""Generator of the hand-placed instruction stream of attn_fwd16_p4 (csrc/attn_fwd16_p4.h).

The forward kernel for D <= 128 keeps the whole tile traversal of a 256-row block in ONE inline-asm statement:
four waves x 64 query rows, one wave per SIMD, matrix operands in the accumulator half of the register file.
hipcc cannot place fillers between hand-written matrix instructions without a wait state per asm boundary, and it
cannot index sub-registers of an asm operand, so the stream is emitted here with a fixed register map:

    a[0:127]   O accumulators   (rb, db) -> 16 (4 rb + db)
    a[128:191] Q fragments      (rb, ks) -> 128 + 4 (8 rb + ks)
    a[192:255] K fragments      (kb, ks) -> 192 + 4 (8 kb + ks)          the 64-key tile being multiplied
    v[32:95]   score tile of even tiles,  v[96:159] of odd tiles  (rb, kb) -> 16 (2 rb + kb)
    v[160:191] FOLD streams: -m of the lane's row, sixteen copies per rb -> 160 + 16 rb (the accumulator start of every score
               block; exact-scale streams leave them unused).  P^T fragments (rb, u) are packed in place into their score block
    v[192:223] V^T fragment ring, eight slots: fragment f = 4 u + db lives in slot f % 8
    v[224:255] addresses and softmax temporaries
    v[0:31]    left to hipcc (operands of the statement, its own values)

Per 64-key tile j (parity = j & 1), two phases of 32 matrix instructions, one barrier between them:
    A(j): S(j) = K(j) Q^T        | exp2 / row sum / 16-bit pack of tile j-1, first half of the V^T(j-1) fragment reads
    B(j): O += V^T(j-1) P^T(j-1) | row max of S(j), deferred-rescale decision, s * scale2 - m, K(j+1) fragments,
                                   second half of the V^T(j-1) reads, LDS-DMA of K(j+2) and V(j+1)
Fillers are dealt out per gap between two matrix instructions (tables below).

The same instruction list is rendered as the asm template (render) and executed by tools/p4sim.py, a lane-exact
model of the subset of gfx950 instructions used here, which checks the result against a float64 attention on CPU
(tests/test_p4_stream.py) -- register map, waits and ring discipline are verified without a GPU.

Usage: python tools/p4gen.py   (rewrites metal_flash_attention_amd/csrc/attn_fwd16_p4_stream.inc)
"""
import os
import sys

# ---------------------------------------------------------------- register map
O_BASE, Q_BASE, K_BASE = 0, 128, 192
S_BASE = (32, 96)
CM_BASE, VF_BASE = 160, 192
T_KADDR = 224          # 8: K fragment address per ks
T_VADDR = 232          # V^T read base of the tile being read
T_MX = 233             # 4: running block maxima (rb, kb) -> 2 rb + kb
T_MN = 237             # 2: new row maximum per rb
T_SW = 239             # 2: half-swap temporaries
T_CORR = 241           # 2
T_LB = 243             # 2: second partial row sum per rb
T_MASKV = 245
T_TL = 246             # 2: mask limits relative to the tile
T_THR = 248            # 2: m + THR                      (exact-scale streams)
T_RS = 248             # 8: rescale temporaries v248..v255 (exact-scale streams; FOLD streams borrow V^T ring slots 0, 1)
FIRST_OWNED_VGPR = 28

KSLOT, VBASE, VSLOT, VRING = 16384, 32768, 16384, 3

# named operands of the asm statement (see attn_fwd16_p4.h); order = operand order
OUT_V = ["m0", "m1", "l0", "l1", "koff0", "koff1", "koff2", "koff3", "voff0", "voff1", "voff2", "voff3"]
TMP_S = ["j", "vrd", "vwr", "pend", "t0", "t1", "pa", "pw", "pb", "plast"]   # "=&s" 32-bit temporaries (p*: PROF streams)
TMP_S64 = ["sv", "ptime"]                              # "=&s" 64-bit temporaries
IN_V = ["kbase", "vbase", "lim0", "lim1", "onesw"]
IN_S = ["kres", "vres", "nt", "wnt", "scale2", "kinc", "vinc", "ldsk", "ldsv", "maskfrom"]


class Cfg:
    def __init__(self, dtype="bf16", thr=8.0, xe=0, order_a="kb", pad=0, prof=0, fold=0, xb=40, dma="b", abl=(), xf=64, tr=0, bal=0, cap=7, maxa=1, va0=0, fastdec=0, fdpos=0):
        """tr: bit 0 = K, bit 1 = V stored transposed (attn_fwd16_p4_tr.h)"""
        """fold: Q arrives pre-multiplied by log2(e)/sqrt(D) and the running maximum is subtracted INSIDE the matrix pipe (an
        extra k-step whose A operand is -1.0 and whose B operand carries m as a bf16/f16 pair): no s * scale2 - m per
        score; xb = scores per tile exponentiated in phase B already (FOLD streams only)."""
        self.dtype, self.thr, self.xe, self.order_a, self.pad, self.prof = dtype, float(thr), xe, order_a, pad, prof
        self.fold, self.xb = fold, (xb if fold else 0)
        # exact-scale streams: scores e < xe are exponentiated in phase B (behind their s * scale2 - m), scores e >= xf get
        # that multiply-subtract in phase A(j+1) in front of their exp2 instead of in phase B(j).  xe = xf = 32 keeps the
        # filler COUNT of both phases and moves transcendentals to phase B, where they are cheaper (ablation table)
        self.xf = xf
        # dma = "b": K(j+2), V(j+1) requested in phase B(j) beside the K fragment reads; "a": K(j+1), V(j) requested in the
        # first gaps of phase A(j), where only VALU work is placed (an LDS-DMA issue next to LDS reads costs 2-3x as much)
        self.dma = dma
        # timing-only ablations (WRONG RESULTS; developer builds): fillers left out of the steady-state phases
        self.abl = frozenset(abl)
        # tr = 1: K and V stored TRANSPOSED ([D][keys], transposeState; attn_fwd16_p4_tr.h): the LDS images keep the orientation of
        # the source -- a 16-byte chunk is 8 consecutive KEYS of one head-dimension element -- and the two read recipes change
        # places.  K^T image: [2 blocks of 32 keys][128 elements][64 bytes], a fragment = two ds_read_b64_tr_b16 (rows 16 ks + 8 h
        # of block kb); they return the contraction index in the order of an accumulator block's registers (4 hi + {0..3, 8..11}),
        # so the kernel stores the Q fragments in that order.  V^T image: [128 elements][64 keys], chunks XOR-swizzled by
        # (element >> 1) & 7 (sixteen lanes = sixteen rows then land on 32 distinct banks); P^T holds its keys in that register order anyway, so a fragment's halves are the 8 bytes at 8 hi of
        # chunks 2 u and 2 u + 1: two ds_read_b64 through eight address registers (the K fragment addresses' registers, which the
        # one K^T base does not need), recomputed per tile from the ring position.  The LDS-DMA pieces are the same instructions:
        # where a chunk comes from is the kernel's business (lane offsets, 128 bytes per tile).
        # Either operand alone (tr = 1: K^T, tr = 2: V^T) is the same exchange for that operand; with V^T alone the K fragment
        # addresses keep their eight registers and the V^T chunk addresses of a 16-key step are computed in two scratch registers
        # in front of the step's first read (vta / vtb, operands of the statement).
        self.tr = tr
        self.kt, self.vt = tr & 1, (tr >> 1) & 1
        # bal (round 5, FOLD streams): the fillers of both phases re-dealt by ISSUE SLOTS.  A wave alone on its SIMD issues one
        # instruction per ~4 clocks, a matrix instruction holds the pipe for 32: eight slots per gap, of which the matrix
        # instruction takes one, a transcendental two, anything else one (SQ_ACTIVE_INST_VALU of the round-4 stream: 1429 clocks
        # per tile = 64 x 4 + 163 x 4 + 64 x 8).  The round-2 tables put 1029 clocks of issue into phase B's 1024 and 768 into
        # phase A's, with phase A's transcendentals bunched in its last twelve gaps (up to nine slots).  Here: the row maxima of
        # the first key block move to phase A of their own tile (its score blocks are complete after 16 matrix instructions),
        # the exponentials are dealt out one per gap over the whole of phase A and up to `cap` slots per gap in phase B, the V^T
        # reads of phase A sit in its first sixteen gaps, the K fragment reads of phase B in front of the LDS-DMA pieces (which
        # then sit in gaps without LDS reads), the scalar bookkeeping where slots are free.  No gap above `cap` slots.
        self.bal, self.cap, self.maxa, self.va0 = bal, cap, maxa, va0
        # fastdec (FOLD streams, bal = 2): the rescale decision from ONE half-wave exchange -- the two row blocks' partial maxima
        # swapped against each other give [row maxima of block 0 | of block 1] in the two half-waves, enough for the branch; the
        # per-lane maxima of both blocks are rebuilt in the out-of-line section only (7 issue slots instead of 14 per tile)
        self.fastdec, self.fdpos = fastdec, fdpos   # fdpos: 0 = the gap behind the last row-maximum step, 1 = behind the LDS-DMA gaps, 2 = split over two gaps
        assert not (fastdec and not (fold and bal == 2))
        assert not (bal and (tr or dma != "b")), "bal: row-major K / V"
        assert not (bal == 1 and not fold)
        # number of scores per tile whose exponential phase B takes (the rest: phase A of the next tile)
        self.nexpb = (self.xb if fold else xe) if bal else None
        # ksplit: the second key block's K fragments are requested in the first gaps of phase A (they are first multiplied sixteen
        # matrix instructions later) instead of in phase B of the previous tile, the longer phase.  Always with K^T (32 reads).
        self.ksplit = 1 if self.kt else 0


# ---------------------------------------------------------------- tiny IR
def V(n, cnt=1):
    return ("v", n, cnt)


def A(n, cnt=1):
    return ("a", n, cnt)


def VN(name):
    return ("V", name)


def SN(name, cnt=1):
    return ("S", name, cnt)


def I(v):
    return ("i", int(v))


def F(v):
    return ("f", float(v))


VCC, M0, VCC_LO, VCC_HI = ("vcc",), ("m0",), ("vcc_lo",), ("vcc_hi",)


class Ins:
    __slots__ = ("op", "d", "s", "mod", "note")

    def __init__(self, op, d=None, s=(), mod=None, note=""):
        self.op, self.d, self.s, self.mod, self.note = op, d, tuple(s), dict(mod or {}), note


def s_elem(par, rb, kb, r):
    return V(S_BASE[par] + 16 * (2 * rb + kb) + r)


def s_blk(par, rb, kb):
    return V(S_BASE[par] + 16 * (2 * rb + kb), 16)


# P^T fragment (rb, u) of a tile = 16 keys 16 u .. of the 64: packed IN PLACE into the first eight registers of its own score
# block (kb = u >> 1): the block is not written again before the V^T P^T products that read the fragment have been issued
def p_word(par, rb, u, w):
    return V(S_BASE[par] + 16 * (2 * rb + (u >> 1)) + 4 * (u & 1) + w)


def p_frag(par, rb, u):
    return V(S_BASE[par] + 16 * (2 * rb + (u >> 1)) + 4 * (u & 1), 4)


def cm_blk(rb):
    """FOLD streams: sixteen registers holding -m of the lane's row: the accumulator start of every score block of rb"""
    return V(CM_BASE + 16 * rb, 16)


def vf_frag(f):
    return V(VF_BASE + 4 * (f % 8), 4)


def vf_half(f, h):
    return V(VF_BASE + 4 * (f % 8) + 2 * h, 2)


def o_acc(rb, db):
    return A(O_BASE + 16 * (4 * rb + db), 16)


def q_frag(rb, ks):
    return A(Q_BASE + 4 * (8 * rb + ks), 4)


def k_frag(kb, ks):
    return A(K_BASE + 4 * (8 * kb + ks), 4)


# element e (0..63) of a score tile: blocks in the order (rb0,kb0) (rb1,kb0) (rb0,kb1) (rb1,kb1), register e % 16
def elem(e):
    blk, r = divmod(e, 16)
    return blk & 1, blk >> 1, r   # rb, kb, r


class Stream:
    def __init__(self, cfg):
        self.cfg = cfg
        self.ins = []
        self.uid = 0
        # LDS queue bookkeeping for counted waits: ids of issued LDS reads in order, index of the last one known complete
        self.lds_issued = 0
        self.lds_done = 0
        # round 6 (persistent streams, tools/p4pgen.py): `fast` phases are the copies of the steady-state phases that the branch-free
        # loop runs -- no mask-section test, no block-switch tests, no pending-rescale test; their rare rescale decision continues in
        # the ordinary copy of the same phase (slow_dec_back: parity -> the label behind that copy's decision)
        self.fast = False
        # orow (round 6, persistent streams with fp32 O): the second products run as O = P V (operands exchanged: A = P, B = V^T fragment)
        # -- lane = head-dimension column, register = row -- so that the epilogue stores rows straight from the registers
        self.orow = bool(getattr(cfg, "orow", 0))
        self.slow_dec_back = {}
        # second partial row sum per row block, mask value: module constants unless a stream re-maps them (p4pgen, pksum)
        self.r_lb = [T_LB, T_LB + 1]
        self.r_maskv = T_MASKV

    def emit(self, op, d=None, s=(), note="", **mod):
        self.ins.append(Ins(op, d, s, mod, note))

    def label(self, name):
        self.ins.append(Ins("label", None, (), {"name": name}))

    def newlabel(self, stem):
        self.uid += 1
        return "%s_%d" % (stem, self.uid)

    # ---- LDS reads with exact wait counting (LDS returns in order)
    def lds_read(self, op, d, addr, offset, note="", src=None):
        self.emit(op, d, [addr] + ([src] if src is not None else []), note=note, offset=offset)
        self.lds_issued += 1
        return self.lds_issued      # id = position in issue order (1-based)

    def lds_need(self, rid):
        """make sure LDS read `rid` has returned"""
        if rid <= self.lds_done:
            return
        younger = self.lds_issued - rid
        n = min(younger, 15)
        self.emit("s_waitcnt", None, [], lgkmcnt=n)
        self.lds_done = self.lds_issued - n

    def lds_flush(self):
        if self.lds_done < self.lds_issued:
            self.emit("s_waitcnt", None, [], lgkmcnt=0)
            self.lds_done = self.lds_issued

    def stamp(self, acc):
        """PROF streams: add the shader-clock time since the previous stamp to accumulator `acc` (s_memtime returns through
        lgkmcnt, so a stamp sits only where no LDS read may be in flight afterwards: it waits lgkmcnt(0) itself)"""
        if not self.cfg.prof:
            return
        self.emit("s_memtime", SN("ptime", 2))
        self.emit("s_waitcnt", None, [], lgkmcnt=0)
        self.lds_done = self.lds_issued
        self.emit("s_mov_b64", VCC, [SN("ptime", 2)])       # vcc_lo names the low dword (vcc is dead at a phase seam)
        self.emit("s_sub_u32", SN("t0"), [VCC_LO, SN("plast")])
        self.emit("s_add_u32", SN(acc), [SN(acc), SN("t0")])
        self.emit("s_mov_b32", SN("plast"), [VCC_LO])

    # ---- matrix instructions
    def mfma(self, d, a, b, c):
        self.emit("v_mfma_f32_32x32x16_" + self.cfg.dtype, d, [a, b, c])

    def pv(self, rb, db, f, par, u):
        """one second-product instruction: O^T(db, rb) += V^T P^T, or (orow) O(rb, db) += P V with the same registers"""
        a, b = vf_frag(f), p_frag(par ^ 1, rb, u)
        if self.orow:
            a, b = b, a
        self.mfma(o_acc(rb, db), a, b, o_acc(rb, db))

    def qk_order(self):
        if self.cfg.order_a == "kb":      # kb-major: the kb = 0 blocks complete after 16 instructions
            return [(g // 16, g % 2, (g % 16) // 2) for g in range(32)]          # (kb, rb, ks)
        return [((g % 4) // 2, g % 2, g // 4) for g in range(32)]                # four accumulators in rotation

    # ------------------------------------------------------------ phase A
    def qk_list(self, par):
        """matrix instructions of phase A as (dst, a, b, c) tuples"""
        out = []
        if not self.cfg.fold:
            for kb, rb, ks in self.qk_order():
                c = I(0) if ks == 0 else s_blk(par, rb, kb)
                out.append((s_blk(par, rb, kb), k_frag(kb, ks), q_frag(rb, ks), c))
            return out
        for kb, rb, ks in self.qk_order():   # FOLD: S' = -m + K Q'^T: the running maximum is the accumulator's start value
            c = cm_blk(rb) if ks == 0 else s_blk(par, rb, kb)
            out.append((s_blk(par, rb, kb), k_frag(kb, ks), q_frag(rb, ks), c))
        return out

    def phase_a(self, par, mfma, softmax, zero_o):
        """A(j), par = j & 1: S[par] = K Q^T | finish-softmax of S[par ^ 1] | V^T reads of fragments 0..7"""
        cfg = self.cfg
        if cfg.bal:
            return self.phase_a_bal(par, mfma, softmax, zero_o)
        prev = par ^ 1
        vids = {}
        if softmax and cfg.tr != 3:   # (K^T + V^T: phase B of the previous tile has left the eight V^T addresses)
            self.emit("v_add_u32", V(T_VADDR), [SN("vrd"), VN("vbase")], note="V^T read base of tile j-1")
        mlist = self.qk_list(par)
        ng = len(mlist)
        g0 = (ng - 32) // 2          # element pair i is exponentiated in gap g0 + i and packed one gap later
        v0 = g0 + 12                 # V^T fragments 0..7 (16 reads) in gaps v0 .. v0 + 15: they land before the barrier
        pending_pack = None
        k_late = None
        for g in range(ng):
            if mfma:
                if g == 16 and k_late is not None:
                    self.lds_need(k_late)    # the K^T fragments of key block 1 requested in gaps 0..7
                self.mfma(*mlist[g])
            if cfg.ksplit and mfma and softmax and g < 8:
                # transposed streams: a K^T fragment is two reads -- the second key block's are requested here, where phase B
                # of the previous tile left them out (they are first multiplied sixteen matrix instructions from now)
                k_late = self.k_read(par, 8 + g)
            if zero_o and g < 32:
                for i in range(4):
                    self.emit("v_accvgpr_write_b32", A(O_BASE + 4 * g + i), [I(0)])
            if softmax:
                if pending_pack is not None:
                    self.sum_pack(prev, pending_pack, mfma)
                    pending_pack = None
                i = g - g0
                if not cfg.fold and 0 <= i + 1 < 32:      # s * scale2 - m of the pair exponentiated in the NEXT gap
                    for e in (2 * i + 2, 2 * i + 3):
                        if e >= cfg.xf:
                            self.fma_only(prev, e)
                if 0 <= i < 32:
                    for e in (2 * i, 2 * i + 1):
                        if e >= max(cfg.xe, cfg.xb) and not (mfma and "expa" in cfg.abl):
                            rb_, kb_, r_ = elem(e)
                            x = s_elem(prev, rb_, kb_, r_)
                            self.emit("v_exp_f32", x, [x])
                    pending_pack = 2 * i
                if v0 <= g < v0 + 16 and not (mfma and "vreada" in cfg.abl):
                    vids[g - v0] = self.v_read(g - v0)
                elif v0 <= g < v0 + 16:
                    vids[g - v0] = 0
            if mfma and softmax and cfg.dma == "a":   # steady state only: K(j+1) -> K image (j+1) & 1, V(j) -> V image j % 3
                if g == 0:
                    self.vwr_update()
                if 2 <= g < 6:
                    self.dma_piece("k", par ^ 1, g - 2)
                elif 6 <= g < 10:
                    self.dma_piece("v", par, g - 6)
                if 4 <= g < 8:
                    self.emit("v_add_u32_e64", VN("koff%d" % (g - 4)), [VN("koff%d" % (g - 4)), SN("kinc")], clamp=1)
                elif 8 <= g < 12:
                    self.emit("v_add_u32_e64", VN("voff%d" % (g - 8)), [VN("voff%d" % (g - 8)), SN("vinc")], clamp=1)
        if softmax and pending_pack is not None:
            self.sum_pack(prev, pending_pack, mfma)
        return vids

    def sum_pack(self, prev, e, steady=False):
        rb, kb, r = elem(e)
        x0, x1 = s_elem(prev, rb, kb, r), s_elem(prev, rb, kb, r + 1)
        if not (steady and "sum" in self.cfg.abl):
            self.emit("v_add_f32", VN("l%d" % rb), [x0, VN("l%d" % rb)])
            self.emit("v_add_f32", V(self.r_lb[rb]), [x1, V(self.r_lb[rb])])
        if steady and "pack" in self.cfg.abl:
            return
        # MFMA step u (16 keys) of key block kb uses registers 8 (u & 1) .. + 7
        self.emit("v_cvt_pk_%s_f32" % self.cfg.dtype, p_word(prev, rb, 2 * kb + r // 8, (r % 8) // 2), [x0, x1])

    def v_read(self, i):
        """V^T read i (0..31): fragment f = i // 2 = 4 u + db, half i % 2 (keys +0..3 / +8..11 of the 16-key group)"""
        f, h = divmod(i, 2)
        u, db = divmod(f, 4)
        if self.cfg.vt:   # element 32 db + lane % 32 of the [128][64 keys] image: 8 bytes of chunk 2 u + h
            if self.cfg.kt:
                return self.lds_read("ds_read_b64", vf_half(f, h), V(T_KADDR + 2 * u + h), db * 32 * 128, note="V^T f%d.%d" % (f, h))
            if db == 0 and h == 0:   # the step's two chunk addresses (the XOR only meets the swizzle bits of the read base)
                self.emit("v_xor_b32", VN("vta"), [I((2 * u) << 4), V(T_VADDR)])
                self.emit("v_xor_b32", VN("vtb"), [I((2 * u + 1) << 4), V(T_VADDR)])
            return self.lds_read("ds_read_b64", vf_half(f, h), VN("vtb" if h else "vta"), db * 32 * 128, note="V^T f%d.%d" % (f, h))
        off = (db * 64 + 16 * u) * 64 + h * 8 * 64
        return self.lds_read("ds_read_b64_tr_b16", vf_half(f, h), V(T_VADDR), off, note="V^T f%d.%d" % (f, h))

    # ------------------------------------------------------------ phase B
    def phase_b(self, par, mfma, softmax, vids):
        """B(j): O += V^T(j-1) P^T(j-1) | start-softmax of S[par], K(j+1) fragments, V^T fragments 8..15, DMA"""
        cfg = self.cfg
        if cfg.bal:
            return self.phase_b_bal(par, mfma, softmax, vids)
        if not mfma and softmax:
            self.emit("s_nop", None, [I(15)], note="S(0) is still leaving the matrix pipe")
        if softmax:
            self.mask_section(par, after_mfma=mfma)
        fill = [[] for _ in range(32)]   # closures per gap

        def at(g, fn):
            fill[g].append(fn)

        if mfma:
            # V^T fragments 8..15: fragment f reuses the slot of f-8, free once the two MFMAs of f-8 (gaps 2(f-8), +1) are issued
            for f in range(8, 16):
                g0 = 2 * (f - 8) + 2
                if softmax and "vreadb" in cfg.abl:
                    vids[2 * f] = vids[2 * f + 1] = 0
                    continue
                at(g0, lambda f=f: vids.__setitem__(2 * f, self.v_read(2 * f)))
                at(g0 + 1, lambda f=f: vids.__setitem__(2 * f + 1, self.v_read(2 * f + 1)))
        if softmax:
            for i in range(32):                      # row maxima: gaps 0..7
                if not (mfma and "max" in cfg.abl):
                    at(i // 4, lambda i=i: self.max_op(par, i))
            at(8, lambda: self.decide_1())
            at(9, lambda: self.decide_2())
            dec_lbl = self.newlabel("DEC")
            first = not mfma           # B'(0): the first tile always sets m (the reference starts from the true maximum)
            if cfg.fold:
                at(10, lambda: self.decide_4_fold(dec_lbl, first))
                for e in range(cfg.xb):          # exp2 of the first xb scores right here: S' needs no further arithmetic
                    if not (mfma and "expb" in cfg.abl):
                        at(12 + e // 2 if cfg.xb <= 40 else 11 + (e * 21) // cfg.xb, lambda e=e: self.exp_in_b(par, e))
            else:
                at(10, lambda: self.decide_3())
                at(11, lambda: self.decide_4(dec_lbl))
                # s * scale2 - m : 64 over gaps 12..31 (4 per gap in 12..15, then 3)
                e = 0
                for g in range(12, 32):
                    n = 4 if g < 16 else 3
                    for _ in range(n):
                        if e < 64:
                            at(g, lambda e=e: self.fma_op(par, e))
                            e += 1
                assert e == 64
            # K(j+1) fragments -> a[192:255], one per gap 12..27 (transposed streams: key block 0 here, block 1 in phase A)
            for i in range(8 if cfg.ksplit else 16):
                if not (mfma and "kread" in cfg.abl):
                    at(12 + i, lambda i=i: self.k_read(par ^ 1, i))
            # LDS-DMA: K(j+2) pieces in gaps 20..23, V(j+1) pieces 24..27; their offsets advance in gaps 28..31
            if cfg.vt and cfg.dma == "b":
                at(24, lambda: self.last_v_tile())
            for i in range(4 if cfg.dma == "b" and not (mfma and "dma" in cfg.abl) else 0):
                at(20 + i, lambda i=i: self.dma_piece("k", par, i))
                at(24 + i, lambda i=i: self.dma_piece("v", par, i))
                at(28 + i, lambda i=i: self.emit("v_add_u32_e64", VN("koff%d" % i), [VN("koff%d" % i), SN("kinc")], clamp=1))
                at(28 + i, lambda i=i: self.emit("v_add_u32_e64", VN("voff%d" % i), [VN("voff%d" % i), SN("vinc")], clamp=1))
            if cfg.dma == "b":
                at(0, lambda: self.vwr_update())
            at(1, lambda: self.vrd_advance())
            if cfg.tr == 3:
                # the eight V^T chunk addresses of the tile phase A reads next (vrd has just advanced to it; the last V^T read of
                # THIS phase is in gap 17; the ring position is a multiple of 16 KiB: the XOR only meets the swizzle bits)
                at(18, lambda: self.emit("v_add_u32", V(T_VADDR), [SN("vrd"), VN("vbase")], note="V^T read base of the next phase A"))
                for c in range(8):
                    at(18 + c, lambda c=c: self.emit("v_xor_b32", V(T_KADDR + c), [I(c << 4), V(T_VADDR)]))
            if getattr(self, "persistent", False):   # tools/p4pgen.py: the last tiles' LDS-DMA pieces belong to the next block
                self.b_hook(at, par, mfma)
        for g in range(32):
            if mfma:
                u, db, rb = g // 8, (g % 8) // 2, g % 2
                f = 4 * u + db
                if rb == 0 and f % 2 == 0:
                    self.lds_need(vids[2 * f + 3])   # both halves of fragments f and f + 1 have returned
                self.pv(rb, db, f, par, u)
            for fn in fill[g]:
                fn()
        while self.xe_pending:
            y = self.xe_pending.pop(0)
            self.emit("v_exp_f32", y, [y])
        self.lds_flush()
        if softmax:
            # apply a pending rescale (rare)
            resc, back = self.newlabel("RESC"), self.newlabel("RESCBACK")
            self.emit("s_cmp_eq_u32", None, [SN("pend"), I(0)])
            self.emit("s_cbranch_scc0", None, [], target=resc)
            self.label(back)
            self.outofline.append(("resc", resc, back, par, False))
            self.outofline.append(("dec", dec_lbl, dec_lbl + "_BACK", par, not mfma))

    # ------------------------------------------------------------ slot-balanced phases (cfg.bal)
    def phase_a_bal(self, par, mfma, softmax, zero_o):
        """A(j) dealt by issue slots: per gap one matrix instruction, the sum / sum / pack of one score pair of tile j-1, one
        exponential of tile j-1 (the scores phase B(j-1) left), one V^T read (gaps 0..15), one row-maximum step of the FIRST
        key block of tile j (gaps 17..31: its score blocks are complete behind matrix instruction 15)"""
        cfg = self.cfg
        abl = cfg.abl if (mfma and softmax) else frozenset()
        prev = par ^ 1
        vids = {}
        if softmax:
            self.emit("v_add_u32", V(T_VADDR), [SN("vrd"), VN("vbase")], note="V^T read base of tile j-1")
        mlist = self.qk_list(par)
        assert len(mlist) == 32 and cfg.order_a == "kb"
        ea = list(range(cfg.nexpb, 64))                    # exponentials left to this phase, dealt evenly over the 32 gaps
        exp_gap = {e: (t * 32) // len(ea) for t, e in enumerate(ea)} if ea else {}
        pack_gap, g_prev = {}, 0                            # pair p = elements 2p, 2p + 1: one pair per gap, in order, once ready
        for p in range(32):
            ready = max(exp_gap.get(2 * p, -1), exp_gap.get(2 * p + 1, -1))
            g_prev = min(32, max(g_prev + 1, ready + 1))
            pack_gap[p] = g_prev
        maxa_gap = {k: min(17 + k, 31) for k in range(16 if cfg.maxa else 0)}  # row-maximum steps of score blocks (rb0, kb0), (rb1, kb0) of THIS tile
        for g in range(33):
            if g < 32 and mfma:
                self.mfma(*mlist[g])
            if g < 32 and zero_o:
                for i in range(4):
                    self.emit("v_accvgpr_write_b32", A(O_BASE + 4 * g + i), [I(0)])
            if softmax:
                for p in range(32):
                    if pack_gap[p] == g:
                        self.sum_pack_abl(prev, 2 * p, abl)
                for e in ea:
                    if exp_gap[e] == g and "exp" not in abl:
                        rb_, kb_, r_ = elem(e)
                        x = s_elem(prev, rb_, kb_, r_)
                        self.emit("v_exp_f32", x, [x])
                if cfg.va0 <= g < cfg.va0 + 16:
                    vids[g - cfg.va0] = self.v_read(g - cfg.va0) if "lds" not in abl else 0
            if mfma and g < 32:
                for k in maxa_gap:
                    if maxa_gap[k] == g and "max" not in abl:
                        self.max_op(par, k)
        return vids

    def sum_pack_abl(self, prev, e, abl):
        rb, kb, r = elem(e)
        x0, x1 = s_elem(prev, rb, kb, r), s_elem(prev, rb, kb, r + 1)
        if "sum" not in abl and getattr(self.cfg, "pksum", 0):
            # (l, second partial sum) are an aligned register pair: both additions in one packed instruction
            lp = V(self.r_lb[rb] - 1, 2)
            self.emit("v_pk_add_f32", lp, [V(x0[1], 2), lp])
        elif "sum" not in abl:
            self.emit("v_add_f32", VN("l%d" % rb), [x0, VN("l%d" % rb)])
            self.emit("v_add_f32", V(self.r_lb[rb]), [x1, V(self.r_lb[rb])])
        if "pack" not in abl:
            self.emit("v_cvt_pk_%s_f32" % self.cfg.dtype, p_word(prev, rb, 2 * kb + r // 8, (r % 8) // 2), [x0, x1])

    def phase_b_bal(self, par, mfma, softmax, vids):
        """B(j) dealt by issue slots: row maxima of the second key block (gaps 0..3), decision (4..6), exponentials from gap 7
        wherever a gap has two slots left below `cap`, K(j+1) fragments in gaps 7..22, V^T fragments 8..15 as their ring slots
        fall free, LDS-DMA of K(j+2) in gaps 23..26 and of V(j+1) in 27..30 (gaps without LDS reads)"""
        cfg = self.cfg
        abl = cfg.abl if (mfma and softmax) else frozenset()
        if not mfma and softmax:
            self.emit("s_nop", None, [I(15)], note="S(0) is still leaving the matrix pipe")
        if softmax and "ctl" not in abl and not self.fast:
            self.mask_section(par, after_mfma=mfma)
        fill = [[] for _ in range(32)]
        slots = [1] * 32

        def at(g, fn, cost=1):
            fill[g].append(fn)
            slots[g] += cost

        if mfma:
            for f in range(8, 16):   # fragment f reuses the slot of f - 8, free once the two products of f - 8 (gaps 2 (f - 8), + 1) are issued
                g0 = 2 * (f - 8) + 2
                if "lds" in abl:
                    vids[2 * f] = vids[2 * f + 1] = 0
                    continue
                at(g0, lambda f=f: vids.__setitem__(2 * f, self.v_read(2 * f)))
                at(g0 + 1, lambda f=f: vids.__setitem__(2 * f + 1, self.v_read(2 * f + 1)))
        if softmax and cfg.bal == 1:
            for i in range(16, 32):
                if "max" not in abl:
                    at((i - 16) // 4, lambda i=i: self.max_op(par, i))
            at(4, lambda: self.decide_1(), 4)
            at(5, lambda: self.decide_2(), 5)
            dec_lbl = self.newlabel("DEC")
            first = not mfma
            at(6, lambda: self.decide_4_fold(dec_lbl, first), 5)
            for i in range(16):
                if "lds" not in abl:
                    at(7 + i, lambda i=i: self.k_read(par ^ 1, i))
            at(18, lambda: self.vrd_advance(), 3)
            if getattr(self, "persistent", False):
                self.b_hook(lambda g, fn: at(g, fn, 2), par, mfma)
            if "dma" not in abl:
                at(22, lambda: self.vwr_update(), 3)
                for i in range(4):
                    at(23 + i, lambda i=i: self.dma_piece("k", par, i), 4)
                    at(27 + i, lambda i=i: self.dma_piece("v", par, i), 4)
                    at(27 + i, lambda i=i: self.emit("v_add_u32_e64", VN("koff%d" % i), [VN("koff%d" % i), SN("kinc")], clamp=1))
                    at(31, lambda i=i: self.emit("v_add_u32_e64", VN("voff%d" % i), [VN("voff%d" % i), SN("vinc")], clamp=1))
            exp_from = 7
        elif softmax:
            # bal = 2: the LDS-DMA pieces of K(j+2) and V(j+1) FIRST.  Their images fell free at the barrier in front of this phase
            # (K(j) and V(j-2) were last read in phase B(j-1)) and their deadline is the next barrier: issued here they have two
            # phases of flight instead of one and a quarter.  On all-zero operands (2.4 GHz) the stream without the pieces ran
            # 19 % faster and without the wait + barrier 25 % faster (profiles/r05_p4p_bal_ablations.txt): the wait for pieces
            # issued in the last gaps of this phase was what the loop spent a fifth of its time in
            ksw_g, vsw_g = 0, 4
            if getattr(self, "persistent", False) and "ctl" not in abl and not self.fast:
                self.b_hook(lambda g, fn: at(g, fn, 2), par, mfma, gaps=(ksw_g, vsw_g))
            elif self.fast:   # (the same slot accounting as the ordinary copy: the two copies must agree gap by gap, see slow_dec_back)
                slots[ksw_g] += 2
                slots[vsw_g] += 2
            if "dma" not in abl:
                for n in range(4):
                    at(n, lambda n=n: self.dma_piece("k", par, n), 2)
                at(3, lambda: self.vwr_update(), 3)
                for n in range(4):
                    at(4 + n, lambda n=n: self.dma_piece("v", par, n), 2)
            # row-maximum steps this phase owes (the second key block's, or all four blocks'), greedily under the cap from gap 0
            gm = 0
            for i in range(16 if cfg.maxa else 0, 32):
                while slots[gm] + 1 > cfg.cap:
                    gm += 1
                if "max" not in abl:
                    at(gm, lambda i=i: self.max_op(par, i))
                else:
                    slots[gm] += 1
            dec_lbl = self.newlabel("DEC")
            first = not mfma
            if not cfg.fastdec:
                at(gm + 1, lambda: self.decide_1(), 4)
                at(gm + 2, lambda: self.decide_2(), 5)
            if cfg.fastdec and cfg.fdpos == 2:
                at(gm + 1, lambda: self.decide_fast(dec_lbl, first, part=1), 5)
                at(gm + 2, lambda: self.decide_fast(dec_lbl, first, part=2), 2)
                exp_from = gm + 3
            elif cfg.fastdec:
                gd = max(gm + 1, 8) if cfg.fdpos == 1 else gm + 1
                at(gd, lambda: self.decide_fast(dec_lbl, first), 7)
                exp_from = gd + 1
            elif cfg.fold:
                at(gm + 3, lambda: self.decide_4_fold(dec_lbl, first), 5)
                exp_from = gm + 4
            else:
                at(gm + 3, lambda: self.decide_3(), 4)
                at(gm + 4, lambda: self.decide_4(dec_lbl), 5)
                exp_from = gm + 5
            assert exp_from + 16 <= 32
            for n in range(16):
                if "lds" not in abl:
                    at(exp_from + n, lambda n=n: self.k_read(par ^ 1, n))
            at(26, lambda: self.vrd_advance(), 3)
            if "dma" not in abl and "offs" not in abl and getattr(cfg, "soff", 0):
                # round 6: the tile advance is the SCALAR offset of the loads (part of the range check on gfx950 like the vector offset,
                # tools/probe_soffset.hip): two scalar additions per tile instead of eight vector ones
                at(28, lambda: self.emit("s_add_u32", SN("ksoff"), [SN("ksoff"), SN("kinc")]))
                at(29, lambda: self.emit("s_add_u32", SN("vsoff"), [SN("vsoff"), SN("vinc")]))
            elif "dma" not in abl and "offs" not in abl:
                for n in range(4):
                    at(28 + n, lambda n=n: self.emit("v_add_u32_e64", VN("koff%d" % n), [VN("koff%d" % n), SN("kinc")], clamp=1))
                    at(28 + n, lambda n=n: self.emit("v_add_u32_e64", VN("voff%d" % n), [VN("voff%d" % n), SN("vinc")], clamp=1))
        if softmax:
            # exponentials of the scores e < xb: behind the decision, two slots each, greedily under the cap (then cap + 1, ...)
            if cfg.fold:
                todo = [] if "exp" in abl else [(2, lambda e=e: self.exp_in_b(par, e)) for e in range(cfg.nexpb)]
            else:   # exact-scale streams: s * scale2 - m of every score, the exponential of the first nexpb two scores behind it
                todo = []
                for e in range(64 + 2):
                    if e < 64:
                        todo.append((1, lambda e=e: self.fma_plain(par, e)))
                    if 0 <= e - 2 < cfg.nexpb and "exp" not in abl:
                        todo.append((2, lambda e=e: self.exp_in_b(par, e - 2)))
            # in order (an exponential follows its own multiply-subtract), each gap filled up to the cap; the smallest cap >= cfg.cap
            # under which the whole list fits in front of the phase's end

            def pack(cap):
                g, local, where = exp_from, list(slots), []
                for c, _ in todo:
                    while g <= 31 and local[g] + c > cap:
                        g += 1
                    if g > 31:
                        return None
                    where.append(g)
                    local[g] += c
                return where

            cap = cfg.cap
            where = pack(cap)
            while where is None:
                cap += 1
                where = pack(cap)
            for (c, fn), g in zip(todo, where):
                at(g, fn, c)
        for g in range(32):
            if mfma:
                u, db, rb = g // 8, (g % 8) // 2, g % 2
                f = 4 * u + db
                if rb == 0 and f % 2 == 0:
                    self.lds_need(vids[2 * f + 3])
                self.pv(rb, db, f, par, u)
            for fn in fill[g]:
                fn()
        self.lds_flush()
        if softmax and self.fast:
            # the decision of a fast phase continues behind the decision of the ordinary copy of this phase (same parity, same gap:
            # the two copies issue the same LDS reads and LDS-DMA pieces in the same order, so every counted wait behind that
            # point means the same instructions), which ends with the pending-rescale test
            self.outofline.append(("dec", dec_lbl, None, par, False))
        elif softmax:
            resc, back = self.newlabel("RESC"), self.newlabel("RESCBACK")
            if "ctl" not in abl:
                self.emit("s_cmp_eq_u32", None, [SN("pend"), I(0)])
                self.emit("s_cbranch_scc0", None, [], target=resc)
            self.label(back)
            self.outofline.append(("resc", resc, back, par, False))
            self.outofline.append(("dec", dec_lbl, dec_lbl + "_BACK", par, not mfma))
            if mfma:
                self.slow_dec_back[par] = dec_lbl + "_BACK"

    def max_op(self, par, i):
        # op i: block i // 8 in the order (rb0,kb0) (rb1,kb0) (rb0,kb1) (rb1,kb1); step i % 8 covers 3, 2, ..., 2, 1 values
        blk, st = divmod(i, 8)
        rb, kb = blk & 1, blk >> 1
        mx = V(T_MX + 2 * rb + kb)
        if st == 0:
            self.emit("v_max3_f32", mx, [s_elem(par, rb, kb, 0), s_elem(par, rb, kb, 1), s_elem(par, rb, kb, 2)])
        elif st < 7:
            self.emit("v_max3_f32", mx, [mx, s_elem(par, rb, kb, 2 * st + 1), s_elem(par, rb, kb, 2 * st + 2)])
        else:
            self.emit("v_max_f32", mx, [mx, s_elem(par, rb, kb, 15)])

    def decide_1(self):   # onlineReduceMaximum across the two key blocks, copies for the half swap
        for rb in range(2):
            self.emit("v_max_f32", V(T_MN + rb), [V(T_MX + 2 * rb), V(T_MX + 2 * rb + 1)])
        for rb in range(2):
            self.emit("v_mov_b32", V(T_SW + rb), [V(T_MN + rb)])

    def decide_2(self):   # lanes l and l ^ 32 hold the two halves of a row's keys
        self.emit("s_nop", None, [I(1)], note="VALU write -> permlane read")
        for rb in range(2):
            self.emit("v_permlane32_swap_b32", V(T_SW + rb), [V(T_MN + rb)], swap=1)
        for rb in range(2):
            self.emit("v_max_f32", V(T_MN + rb), [V(T_SW + rb), V(T_MN + rb)])

    def decide_fast(self, lbl, first, part=0):
        if part in (0, 1):
            for rb in range(2):
                self.emit("v_max_f32", V(T_MN + rb), [V(T_MX + 2 * rb), V(T_MX + 2 * rb + 1)])
            self.emit("s_nop", None, [I(1)], note="VALU write -> permlane read")
            self.emit("v_permlane32_swap_b32", V(T_MN), [V(T_MN + 1)], swap=1)
            self.emit("v_max_f32", V(T_SW), [V(T_MN), V(T_MN + 1)])    # [row maxima of row block 0 | of row block 1]
        if part == 1:
            return
        if first:
            self.emit("s_branch", None, [], target=lbl)
        else:
            self.emit("v_cmp_lt_f32", VCC, [F(self.cfg.thr), V(T_SW)])
            self.emit("s_cbranch_vccnz", None, [], target=lbl)
        self.label(lbl + "_BACK")

    def decide_3(self):
        for rb in range(2):
            self.emit("v_mul_f32", V(T_MN + rb), [SN("scale2"), V(T_MN + rb)])
        for rb in range(2):
            self.emit("v_add_f32", V(T_THR + rb), [F(self.cfg.thr), VN("m%d" % rb)])

    def decide_4(self, lbl):   # any row whose block maximum exceeds m + THR -> raise m (out of line)
        self.emit("v_cmp_gt_f32", VCC, [V(T_MN), V(T_THR)])
        self.emit("s_mov_b64", SN("sv", 2), [VCC])
        self.emit("v_cmp_gt_f32", VCC, [V(T_MN + 1), V(T_THR + 1)])
        self.emit("s_or_b64", VCC, [VCC, SN("sv", 2)])
        self.emit("s_cbranch_vccnz", None, [], target=lbl)
        self.label(lbl + "_BACK")

    def decide_4_fold(self, lbl, first):   # the scores are already relative to m: any block maximum above THR raises m
        if first:
            self.emit("s_branch", None, [], target=lbl)
        else:
            self.emit("v_cmp_lt_f32", VCC, [F(self.cfg.thr), V(T_MN)])
            self.emit("s_mov_b64", SN("sv", 2), [VCC])
            self.emit("v_cmp_lt_f32", VCC, [F(self.cfg.thr), V(T_MN + 1)])
            self.emit("s_or_b64", VCC, [VCC, SN("sv", 2)])
            self.emit("s_cbranch_vccnz", None, [], target=lbl)
        self.label(lbl + "_BACK")

    def exp_in_b(self, par, e):
        rb, kb, r = elem(e)
        x = s_elem(par, rb, kb, r)
        self.emit("v_exp_f32", x, [x])

    def fma_plain(self, par, e):
        rb, kb, r = elem(e)
        x = s_elem(par, rb, kb, r)
        self.emit("v_fma_f32", x, [x, SN("scale2"), VN("m%d" % rb)], neg2=1)

    def fma_only(self, par, e):
        rb, kb, r = elem(e)
        x = s_elem(par, rb, kb, r)
        self.emit("v_fma_f32", x, [x, SN("scale2"), VN("m%d" % rb)], neg2=1)

    def fma_op(self, par, e):
        if e >= self.cfg.xf:
            return          # done in phase A of the next tile
        rb, kb, r = elem(e)
        x = s_elem(par, rb, kb, r)
        self.emit("v_fma_f32", x, [x, SN("scale2"), VN("m%d" % rb)], neg2=1)
        if e < self.cfg.xe:   # exp2 already in phase B, two elements behind its own fma
            self.xe_pending.append(x)
        if len(self.xe_pending) > 2:
            y = self.xe_pending.pop(0)
            self.emit("v_exp_f32", y, [y])

    def k_read(self, slot, i):
        kb, ks = divmod(i, 8)
        if self.cfg.kt:   # rows 16 ks (+ 8) of key block kb of the [2][128][64 bytes] image
            for h in range(2):
                rid = self.lds_read("ds_read_b64_tr_b16", A(K_BASE + 4 * (8 * kb + ks) + 2 * h, 2), VN("kbase"),
                                    slot * KSLOT + (kb * 128 + 16 * ks + 8 * h) * 64, note="K^T(%d,%d).%d" % (kb, ks, h))
            return rid
        return self.lds_read("ds_read_b128", k_frag(kb, ks), V(T_KADDR + ks), slot * KSLOT + kb * 8192, note="K(%d,%d)" % (kb, ks))

    def vrd_advance(self):   # the V^T read base of this tile is already in T_VADDR
        self.emit("s_add_u32", SN("vrd"), [SN("vrd"), I(VSLOT)])
        self.emit("s_cmp_ge_u32", None, [SN("vrd"), I(VRING * VSLOT)])
        self.emit("s_cselect_b32", SN("vrd"), [I(0), SN("vrd")])

    def vwr_update(self):   # V(j+1) goes to the image after the one V(j) went to ("vwr" includes this wave's LDS base)
        self.emit("s_add_u32", SN("vwr"), [SN("vwr"), I(VSLOT)])
        self.emit("s_cmp_ge_u32", None, [SN("vwr"), SN("t1")])
        self.emit("s_cselect_b32", SN("vwr"), [SN("ldsv"), SN("vwr")])

    def dma_piece(self, which, par, i):
        so = getattr(self.cfg, "soff", 0)
        if which == "k":   # K(j+2) -> K image j & 1
            self.emit("s_add_u32", M0, [SN("ldsk"), I(par * KSLOT + i * 1024)])
            self.emit("buffer_load_dwordx4_lds", None, [VN("koff%d" % i), SN("kres", 4)] + ([SN("ksoff")] if so else []), pol=getattr(self.cfg, "dmapol", ""))
        else:              # V(j+1) -> V image (j + 1) % 3
            self.emit("s_add_u32", M0, [SN("vwr"), I(i * 1024)])
            self.emit("buffer_load_dwordx4_lds", None, [VN("voff%d" % i), SN("vres", 4)] + ([SN("vsoff")] if so else []), pol=getattr(self.cfg, "dmapol", ""))

    def last_v_tile(self):
        """transposed streams, in front of the pieces of V(j+1): the tile advances ALONG the rows of V^T, so the end of the
        sequence is not the end of the buffer -- the workgroup's LAST tile is fetched through offsets of its own (vlast: out
        of bounds, i.e. zeros, for chunks at or beyond key C; P is 0 there, but 0 x whatever follows the sequence is not).  The
        run-ahead tiles behind it are never read."""
        self.emit("s_cmp_eq_u32", None, [SN("j"), SN("ntm2")])
        self.emit("s_cselect_b64", SN("selv", 2), [I(-1), I(0)])
        for i in range(4):
            self.emit("v_cndmask_b32_e64", VN("voff%d" % i), [VN("voff%d" % i), VN("vlast%d" % i), SN("selv", 2)])

    def mask_section(self, par, after_mfma):
        """edge masks on the fresh score tile: key c of row r is visible iff c <= lim[r] (lim = min(C - 1, causal limit))"""
        skip = self.newlabel("NOMASK")
        self.emit("s_cmp_lt_i32", None, [SN("j"), SN("maskfrom")])
        self.emit("s_cbranch_scc1", None, [], target=skip)
        if after_mfma:
            self.emit("s_nop", None, [I(15)], note="S(j) is still leaving the matrix pipe")
        self.emit("s_lshl_b32", SN("t0"), [SN("j"), I(6)])
        if getattr(self.cfg, "causal", 0) and getattr(self.cfg, "diagmask", 0):
            # round 6, causal streams: the tile ON the wave's diagonal with everything aligned (first row of the wave + C - R = first key
            # of the tile, the tile inside the sequence -- every self-attention launch with N % 64 == 0): which lanes lose which key is
            # a compile-time pattern.  Row block = key block: register r masks rows q < (r & 3) + 8 (r >> 2) (+ 4 in the upper half-wave)
            # -- two scalar moves into vcc and ONE v_cndmask instead of a compare + v_cndmask per score; keys 32..63 against rows 0..31:
            # all masked (a move); keys 0..31 against rows 32..63: all visible (nothing).  48 vector instructions instead of 256, and
            # only one score block's row maxima to take again
            general = self.newlabel("MASKGENERAL")
            self.emit("s_cmp_eq_u32", None, [SN("t0"), SN("wdiag")])
            self.emit("s_cbranch_scc0", None, [], target=general)
            self.emit("s_add_u32", SN("t1x"), [SN("t0"), I(63)])
            self.emit("s_cmp_gt_u32", None, [SN("t1x"), SN("cm1")])
            self.emit("s_cbranch_scc1", None, [], target=general)
            self.emit("v_mov_b32", V(self.r_maskv), [F(-(0.875 / 1.44269504089) * 3.402823466e+38)])
            for rb in range(2):
                for r in range(16):
                    c0 = (r & 3) + 8 * (r >> 2)
                    x = s_elem(par, rb, rb, r)
                    self.emit("s_mov_b32", VCC_LO, [I((1 << c0) - 1)])
                    self.emit("s_mov_b32", VCC_HI, [I((1 << (c0 + 4)) - 1)])
                    self.emit("v_cndmask_b32", x, [x, V(self.r_maskv), VCC])
            for r in range(16):
                self.emit("v_mov_b32", s_elem(par, 0, 1, r), [V(self.r_maskv)])
            if self.cfg.bal and self.cfg.maxa:
                for i in range(8):     # (rb 0, kb 0) is the one first-key-block score block the mask touched
                    self.max_op(par, i)
            self.emit("s_branch", None, [], target=skip)
            self.label(general)
        for rb in range(2):
            self.emit("v_subrev_u32", V(T_TL + rb), [SN("t0"), VN("lim%d" % rb)])   # lim - 4 hi - 64 j
        self.emit("v_mov_b32", V(self.r_maskv), [F(-(0.875 / 1.44269504089) * 3.402823466e+38)])   # +Softmax.swift:242-243
        for rb in range(2):
            for kb in range(2):
                for r in range(16):
                    c = kb * 32 + (r & 3) + 8 * (r >> 2)
                    x = s_elem(par, rb, kb, r)
                    self.emit("v_cmp_gt_i32", VCC, [I(c), V(T_TL + rb)])
                    self.emit("v_cndmask_b32", x, [x, V(self.r_maskv), VCC])
        if self.cfg.bal and self.cfg.maxa:   # the first key block's row maxima were taken in phase A, from the unmasked scores
            for i in range(16):
                self.max_op(par, i)
        self.label(skip)

    # ------------------------------------------------------------ out-of-line sections
    def emit_outofline(self):
        cfg = self.cfg
        rs = VF_BASE if cfg.fold else T_RS      # rescale temporaries (FOLD: V^T ring slots 0, 1 are idle at the end of phase B)
        for kind, lbl, back, par, first in self.outofline:
            if back is None:
                back = self.slow_dec_back[par]
            self.label(lbl)
            if kind == "dec" and cfg.fold:
                if cfg.fastdec:   # v[T_SW] = [maxima of row block 0 | of row block 1] -> both blocks' row maxima in every lane
                    self.emit("v_mov_b32", V(T_MN), [V(T_SW)])
                    self.emit("v_mov_b32", V(T_MN + 1), [V(T_SW)])
                    self.emit("s_nop", None, [I(1)], note="VALU write -> permlane read")
                    self.emit("v_permlane32_swap_b32", V(T_MN), [V(T_MN + 1)], swap=1)
                # m_up = m + max(mx', 0) (first tile: m + mx'); shift = m_up - m re-bases this tile's scores, corr = 2^-shift
                # re-bases O and l at the end of phase B (+Softmax.swift:290-301); the start block of the following tiles = -m_up
                ta, tb = V(T_SW), V(T_SW + 1)
                for rb in range(2):
                    mn, m = V(T_MN + rb), VN("m%d" % rb)
                    if not first:
                        self.emit("v_max_f32", mn, [I(0), mn])
                    self.emit("v_add_f32", ta, [m, mn])
                    self.emit("v_sub_f32", tb, [ta, m])       # shift
                    self.emit("v_mov_b32", m, [ta])
                    self.emit("v_exp_f32", V(T_CORR + rb), [tb], neg0=1)
                    for kb in range(2):
                        for r in range(16):
                            x = s_elem(par, rb, kb, r)
                            self.emit("v_sub_f32", x, [x, tb])
                    for r in range(16):
                        self.emit("v_sub_f32", V(CM_BASE + 16 * rb + r), [I(0), ta])
                self.emit("s_mov_b32", SN("pend"), [I(0 if first else 1)])   # first tile: O and l are still zero, nothing to re-base
                self.emit("s_branch", None, [], target=back)
            elif kind == "dec":   # onlineCorrectO factors (+Softmax.swift:290-301): m_up = max(m, m_new), corr = 2^(m - m_up)
                for rb in range(2):
                    self.emit("v_max_f32", V(T_THR + rb), [VN("m%d" % rb), V(T_MN + rb)])
                for rb in range(2):
                    self.emit("v_sub_f32", V(T_CORR + rb), [VN("m%d" % rb), V(T_THR + rb)])
                for rb in range(2):
                    self.emit("v_mov_b32", VN("m%d" % rb), [V(T_THR + rb)])
                for rb in range(2):
                    self.emit("v_exp_f32", V(T_CORR + rb), [V(T_CORR + rb)])
                self.emit("s_mov_b32", SN("pend"), [I(0 if first else 1)])   # first tile: O and l are still zero
                self.emit("s_branch", None, [], target=back)
            else:               # O, l *= corr once every matrix instruction that accumulates P(j-1) has been issued
                self.emit("s_nop", None, [I(15)])
                self.emit("s_nop", None, [I(7)])
                if self.orow:
                    # register r of a lane belongs to row 8 (r >> 2) + 4 hi + (r & 3): the row's factor comes from the lane that owns the
                    # row (ds_bpermute_b32: lane select = (address + offset) / 4, no LDS memory), four registers of the four blocks at a time
                    self.emit("v_mbcnt_lo_u32_b32", V(rs), [I(-1), I(0)])
                    self.emit("v_mbcnt_hi_u32_b32", V(rs), [I(-1), V(rs)])
                    self.emit("v_lshrrev_b32", V(rs), [I(5), V(rs)])
                    self.emit("v_lshlrev_b32", V(rs), [I(4), V(rs)])          # 16 hi
                    for rb in range(2):
                        for g in range(4):
                            for i in range(3):
                                self.emit("ds_bpermute_b32", V(rs + 1 + i), [V(rs), V(T_CORR + rb)], offset=4 * (8 * g + i))
                            self.emit("ds_bpermute_b32", V(rs + 7), [V(rs), V(T_CORR + rb)], offset=4 * (8 * g + 3))
                            self.emit("s_waitcnt", None, [], lgkmcnt=0)
                            fac = [rs + 1, rs + 2, rs + 3, rs + 7]
                            for db in range(4):
                                for i in range(3):
                                    self.emit("v_accvgpr_read_b32", V(rs + 4 + i), [A(O_BASE + 64 * rb + 16 * db + 4 * g + i)])
                                for i in range(3):
                                    self.emit("v_mul_f32", V(rs + 4 + i), [V(fac[i]), V(rs + 4 + i)])
                                for i in range(3):
                                    self.emit("v_accvgpr_write_b32", A(O_BASE + 64 * rb + 16 * db + 4 * g + i), [V(rs + 4 + i)])
                                # (eight temporaries: the fourth register of the group goes through the fourth factor's register pair)
                                self.emit("v_accvgpr_read_b32", V(rs + 4), [A(O_BASE + 64 * rb + 16 * db + 4 * g + 3)])
                                self.emit("v_mul_f32", V(rs + 4), [V(fac[3]), V(rs + 4)])
                                self.emit("v_accvgpr_write_b32", A(O_BASE + 64 * rb + 16 * db + 4 * g + 3), [V(rs + 4)])
                        self.emit("v_mul_f32", VN("l%d" % rb), [V(T_CORR + rb), VN("l%d" % rb)])
                        self.emit("v_mul_f32", V(self.r_lb[rb]), [V(T_CORR + rb), V(self.r_lb[rb])])
                for rb in range(0 if self.orow else 2):
                    for i0 in range(0, 64, 8):
                        for t in range(8):
                            self.emit("v_accvgpr_read_b32", V(rs + t), [A(O_BASE + 64 * rb + i0 + t)])
                        for t in range(8):
                            self.emit("v_mul_f32", V(rs + t), [V(T_CORR + rb), V(rs + t)])
                        for t in range(8):
                            self.emit("v_accvgpr_write_b32", A(O_BASE + 64 * rb + i0 + t), [V(rs + t)])
                    self.emit("v_mul_f32", VN("l%d" % rb), [V(T_CORR + rb), VN("l%d" % rb)])
                    self.emit("v_mul_f32", V(self.r_lb[rb]), [V(T_CORR + rb), V(self.r_lb[rb])])
                self.emit("s_mov_b32", SN("pend"), [I(0)])
                self.emit("s_nop", None, [I(4)], note="accvgpr write -> MFMA SrcC")
                self.emit("s_branch", None, [], target=back)

    def to16_f32(self, d, x):
        """d = x truncated to the mantissa width of the stream's 16-bit type (still an fp32 value): 8 significant bits for
        bf16, 11 for f16 -- converting d to that type is then exact (x within the type's normal range, which scores in
        log2 units are).  One AND instead of a conversion round trip."""
        self.emit("v_and_b32", d, [I(0xFFFF0000 if self.cfg.dtype == "bf16" else 0xFFFFE000), x])

    # ------------------------------------------------------------ whole traversal
    def build(self):
        self.outofline = []
        self.xe_pending = []
        if self.cfg.pad:
            self.emit("s_nop", None, [I(0)], note="code placement pad")
        # ---- prologue: K(0), V(0) landed (K(1) may be in flight): 12 DMA pieces were issued by the caller
        self.emit("s_waitcnt", None, [], vmcnt=4)
        self.emit("s_barrier")
        for ks in range(8):
            if not self.cfg.kt:
                self.emit("v_xor_b32", V(T_KADDR + ks), [I(ks << 5), VN("kbase")])
        for rb in range(2):
            self.emit("v_mov_b32", V(self.r_lb[rb]), [I(0)])
            self.emit("v_mov_b32", V(T_CORR + rb), [F(1.0)])
        if self.cfg.fold:
            for r in range(32):
                self.emit("v_mov_b32", V(CM_BASE + r), [I(0)])
        self.emit("s_mov_b32", SN("pend"), [I(0)])
        self.emit("s_mov_b32", SN("j"), [I(0)])
        self.emit("s_mov_b32", SN("vrd"), [I(2 * VSLOT)])    # "image of V(-1)"
        self.emit("s_mov_b32", SN("vwr"), [SN("ldsv")])      # V(0) went to image 0: V(1) goes to image 1
        self.emit("s_add_u32", SN("t1"), [SN("ldsv"), I(VRING * VSLOT)])
        for i in range(16):
            self.k_read(0, i)
        self.lds_flush()
        self.phase_a(0, mfma=True, softmax=False, zero_o=True)
        self.emit("s_waitcnt", None, [], vmcnt=0)            # K(1)
        self.emit("s_barrier")
        self.phase_b(0, mfma=False, softmax=True, vids={})
        self.emit("s_mov_b32", SN("j"), [I(1)])
        for acc in ("pa", "pw", "pb"):
            self.emit("s_mov_b32", SN(acc), [I(0)])
        if self.cfg.prof:
            self.emit("s_memtime", SN("ptime", 2))
            self.emit("s_waitcnt", None, [], lgkmcnt=0)
            self.emit("s_mov_b64", VCC, [SN("ptime", 2)])
            self.emit("s_mov_b32", SN("plast"), [VCC_LO])
        loop, end_even, end_odd, done, fin, skip_odd, skip_even = (
            self.newlabel(x) for x in ("LOOP", "ENDEVEN", "ENDODD", "DONE", "FIN", "SKIPODD", "SKIPEVEN"))
        self.label(loop)
        # wnt = key tiles THIS WAVE needs (causal: up to the diagonal of its own last row; otherwise = nt)
        for par, endl in ((1, end_even), (0, end_odd)):
            self.emit("s_cmp_ge_i32", None, [SN("j"), SN("wnt")])
            self.emit("s_cbranch_scc1", None, [], target=endl)
            vids = self.phase_a(par, mfma=True, softmax=True, zero_o=False)
            self.stamp("pa")
            self.lds_flush()
            self.emit("s_waitcnt", None, [], vmcnt=0)        # this wave's pieces of K(j+1) and V(j)
            self.emit("s_barrier")
            self.stamp("pw")
            self.phase_b(par, mfma=True, softmax=True, vids=vids)
            self.stamp("pb")
            self.emit("s_add_u32", SN("j"), [SN("j"), I(1)])
        self.emit("s_branch", None, [], target=loop)
        # tails: finish tile wnt-1 (its scores are in S[last parity]), then keep the other waves company
        for lastpar, lbl, nxt in ((0, end_even, skip_odd), (1, end_odd, skip_even)):
            self.label(lbl)
            vids = self.phase_a(lastpar ^ 1, mfma=False, softmax=True, zero_o=False)
            self.lds_flush()
            self.emit("s_nop", None, [I(1)], note="freshly packed P -> MFMA operand")
            self.phase_b(lastpar ^ 1, mfma=True, softmax=False, vids=vids)
            self.emit("s_branch", None, [], target=nxt)
        # A wave whose rows are done before the workgroup's last tile (causal) still owes the others its barriers and its
        # share of the LDS-DMA pieces: tiles j = wnt .. nt-1 without arithmetic
        for par, lbl in ((1, skip_odd), (0, skip_even)):
            self.label(lbl)
            self.emit("s_cmp_ge_i32", None, [SN("j"), SN("nt")])
            self.emit("s_cbranch_scc1", None, [], target=done)
            self.emit("s_waitcnt", None, [], vmcnt=0)
            self.emit("s_barrier")
            if self.cfg.dma == "b":
                self.vwr_update()
                for i in range(4):
                    self.dma_piece("k", par, i)
                if self.cfg.vt:
                    self.last_v_tile()
                for i in range(4):
                    self.dma_piece("v", par, i)
                for i in range(4):
                    self.emit("v_add_u32_e64", VN("koff%d" % i), [VN("koff%d" % i), SN("kinc")], clamp=1)
                    self.emit("v_add_u32_e64", VN("voff%d" % i), [VN("voff%d" % i), SN("vinc")], clamp=1)
            self.emit("s_add_u32", SN("j"), [SN("j"), I(1)])
            if par == 0:
                self.emit("s_branch", None, [], target=skip_odd)
        self.label(done)
        for rb in range(2):
            self.emit("v_add_f32", VN("l%d" % rb), [V(self.r_lb[rb]), VN("l%d" % rb)])
        self.emit("s_branch", None, [], target=fin)
        self.emit_outofline()
        self.label(fin)
        return self.ins


# ---------------------------------------------------------------- rendering
def fmt(o):
    k = o[0]
    if k in ("v", "a"):
        return "%s%d" % (k, o[1]) if o[2] == 1 else "%s[%d:%d]" % (k, o[1], o[1] + o[2] - 1)
    if k == "V":
        return "%%[%s]" % o[1]
    if k == "S":
        return "%%[%s]" % o[1]
    if k == "sr":       # a fixed scalar register (persistent streams, tools/p4pgen.py)
        return "s%d" % o[1] if o[2] == 1 else "s[%d:%d]" % (o[1], o[1] + o[2] - 1)
    if k == "vcc_lo":
        return "vcc_lo"
    if k == "vcc_hi":
        return "vcc_hi"
    if k == "i":
        return str(o[1]) if -16 <= o[1] <= 64 else hex(o[1] & 0xFFFFFFFF)
    if k == "f":
        import struct
        v = o[1]
        inline = {0.0: "0", 0.5: "0.5", 1.0: "1.0", 2.0: "2.0", 4.0: "4.0", -0.5: "-0.5", -1.0: "-1.0", -2.0: "-2.0", -4.0: "-4.0"}
        if v in inline:
            return inline[v]
        return hex(struct.unpack("<I", struct.pack("<f", v))[0])
    if k == "vcc":
        return "vcc"
    if k == "m0":
        return "m0"
    raise ValueError(o)


def render_one(ins, suffix="%="):
    op, m = ins.op, ins.mod
    if op == "label":
        return "%s_%s:" % (m["name"], suffix)
    if op == "s_waitcnt":
        parts = []
        if "vmcnt" in m:
            parts.append("vmcnt(%d)" % m["vmcnt"])
        if "lgkmcnt" in m:
            parts.append("lgkmcnt(%d)" % m["lgkmcnt"])
        return "s_waitcnt " + " ".join(parts)
    if op == "s_barrier":
        return "s_barrier"
    if op in ("s_cbranch_scc0", "s_cbranch_scc1", "s_cbranch_vccnz", "s_branch"):
        return "%s %s_%s" % (op, m["target"], suffix)
    if op == "buffer_load_dwordx4_lds":
        return "buffer_load_dwordx4 %s, %s, %s offen%s lds" % (fmt(ins.s[0]), fmt(ins.s[1]), fmt(ins.s[2]) if len(ins.s) > 2 else "0", m.get("pol", ""))
    if op in ("buffer_load_dword", "buffer_load_ushort"):
        return "%s %s, %s, %s, 0 offen" % (op, fmt(ins.d), fmt(ins.s[0]), fmt(ins.s[1]))
    if op == "buffer_store_dwordx4":     # s = (four data registers, per-lane byte offset, buffer resource)
        return "buffer_store_dwordx4 %s, %s, %s, 0 offen" % (fmt(ins.s[0]), fmt(ins.s[1]), fmt(ins.s[2]))
    if op == "v_fma_mix_f32":
        return "v_fma_mix_f32 %s, %s, %s, %s op_sel:[%d,%d,%d] op_sel_hi:[%d,%d,%d]" % (
            (fmt(ins.d), fmt(ins.s[0]), fmt(ins.s[1]), fmt(ins.s[2])) + tuple(m["op_sel"]) + tuple(m["op_sel_hi"]))
    if op == "ds_write_b128":           # s = (address register, four data registers)
        return "ds_write_b128 %s, %s offset:%d" % (fmt(ins.s[0]), fmt(ins.s[1]), m["offset"])
    if op in ("ds_read_b128", "ds_read_b64", "ds_read_b64_tr_b16"):
        return "%s %s, %s offset:%d" % (op, fmt(ins.d), fmt(ins.s[0]), m["offset"])
    if op == "ds_bpermute_b32":         # d[lane] = s1[((s0[lane] + offset) >> 2) & 63]
        return "ds_bpermute_b32 %s, %s, %s offset:%d" % (fmt(ins.d), fmt(ins.s[0]), fmt(ins.s[1]), m["offset"])
    if op == "v_fma_f32":
        return "v_fma_f32 %s, %s, %s, -%s" % (fmt(ins.d), fmt(ins.s[0]), fmt(ins.s[1]), fmt(ins.s[2]))
    if op == "v_add_u32_e64":
        return "v_add_u32_e64 %s, %s, %s clamp" % (fmt(ins.d), fmt(ins.s[0]), fmt(ins.s[1]))
    if op == "v_permlane32_swap_b32":
        return "v_permlane32_swap_b32 %s, %s" % (fmt(ins.d), fmt(ins.s[0]))
    if op in ("s_cmp_lt_i32", "s_cmp_ge_i32", "s_cmp_ge_u32", "s_cmp_eq_u32", "s_nop"):
        return "%s %s" % (op, ", ".join(fmt(x) for x in ins.s))
    if op in ("v_cvt_pk_bf16_f32", "v_cvt_pk_f16_f32") and (m.get("neg0") or m.get("neg1")):
        return "%s %s, %s%s, %s%s" % (op, fmt(ins.d), "-" if m.get("neg0") else "", fmt(ins.s[0]), "-" if m.get("neg1") else "", fmt(ins.s[1]))
    if op == "v_exp_f32" and m.get("neg0"):
        return "v_exp_f32_e64 %s, -%s" % (fmt(ins.d), fmt(ins.s[0]))
    if op == "v_cmp_lt_f32":
        return "v_cmp_lt_f32_e32 vcc, %s, %s" % (fmt(ins.s[0]), fmt(ins.s[1]))
    if op in ("v_cmp_gt_f32", "v_cmp_gt_i32"):
        return "%s_e32 vcc, %s, %s" % (op, fmt(ins.s[0]), fmt(ins.s[1]))
    if op == "v_cndmask_b32":
        return "v_cndmask_b32_e32 %s, %s, %s, vcc" % (fmt(ins.d), fmt(ins.s[0]), fmt(ins.s[1]))
    if op == "v_cndmask_b32_e64":   # d = mask ? s1 : s0, the mask a scalar register pair
        return "v_cndmask_b32_e64 %s, %s, %s, %s" % (fmt(ins.d), fmt(ins.s[0]), fmt(ins.s[1]), fmt(ins.s[2]))
    ops = [fmt(ins.d)] if ins.d is not None else []
    ops += [fmt(x) for x in ins.s]
    return "%s %s" % (op, ", ".join(ops))


def render(instrs):
    return [render_one(i) for i in instrs]


def write_inc(path):
    lines = ["// GENERATED by tools/p4gen.py -- do not edit.  Instruction streams of attn_fwd16_p4 (see the generator's",
             "// header for the register map and the phase tables).", "#pragma once", ""]
    vregs = ", ".join('"v%d"' % i for i in range(FIRST_OWNED_VGPR, 256))
    lines.append("#define MFA_P4_OWNED_VGPRS " + vregs)
    lines.append("")
    lines.append("// X(name, folds Q scale and running maximum into the matrix pipe, stamps the shader clock)")
    lines.append("#define MFA_P4_STREAM_LIST(X) \\")
    for name, cfg in VARIANTS.items():
        if not cfg.tr:
            lines.append("  X(%s, %d, %d) \\" % (name, cfg.fold, cfg.prof))
    lines.append("")
    lines.append("// streams of attn_fwd16_p4_tr (K and / or V stored transposed): X(name, folds, pattern: bit 0 = K, bit 1 = V)")
    lines.append("#define MFA_P4_TR_STREAM_LIST(X) \\")
    for name in TR_STREAMS:
        lines.append("  X(%s, %d, %d) \\" % (name, VARIANTS[name].fold, VARIANTS[name].tr))
    lines.append("")
    lines.append("// streams that only the developer build (-DMFA_DEV_VARIANTS) instantiates: MFA_FWD16_IMPL=p4:<1000 + index>")
    lines.append("#define MFA_P4_DEV_STREAM_LIST(X) \\")
    for name, cfg in VARIANTS.items():
        if name not in PRODUCT_STREAMS and cfg.dtype == "bf16" and not cfg.tr:
            lines.append("  X(%s) \\" % name)
    lines.append("")
    lines.append("")
    for name, cfg in VARIANTS.items():
        st = Stream(cfg)
        ins = st.build()
        txt = render(ins)
        n_mfma = sum(1 for i in ins if i.op.startswith("v_mfma"))
        lines.append("// %s: dtype=%s thr=%g xe=%d xf=%d order_a=%s pad=%d prof=%d fold=%d xb=%d dma=%s%s -- %d instructions, %d matrix instructions"
                     % (name, cfg.dtype, cfg.thr, cfg.xe, cfg.xf, cfg.order_a, cfg.pad, cfg.prof, cfg.fold, cfg.xb, cfg.dma,
                        " tr=%d" % cfg.tr if cfg.tr else "", len(txt), n_mfma))
        lines.append("#define MFA_P4_STREAM_%s \\" % name)
        for t in txt:
            lines.append('  "%s\\n\\t" \\' % t)
        lines.append('  ""')
        lines.append("")
    with open(path, "w") as f:
        f.write("\n".join(lines))


VARIANTS = {
    # product streams: the round-5 schedule (bal = 2: LDS-DMA first in phase B, fillers dealt by issue slots)
    "BF16_THR8": Cfg("bf16", 8, xe=32, bal=2, cap=8),
    "BF16_THR0": Cfg("bf16", 0, xe=32, bal=2, cap=8),
    "F16_THR8": Cfg("f16", 8, xe=32, bal=2, cap=8),
    "R4_BF16_THR8": Cfg("bf16", 8, 0),                 # the round-4 schedules (developer library, A/B baselines)
    "R4_BF16_FOLD": Cfg("bf16", 8, fold=1, xb=40),
    "BF16_THR8_XE16": Cfg("bf16", 8, 16),
    "BF16_THR8_ROT": Cfg("bf16", 8, 0, order_a="rot4"),
    "BF16_THR8_PAD": Cfg("bf16", 8, 0, pad=1),
    "BF16_THR8_PROF": Cfg("bf16", 8, 0, prof=1),
    "BF16_FOLD": Cfg("bf16", 8, fold=1, xb=48, bal=2, fastdec=1),
    "F16_FOLD": Cfg("f16", 8, fold=1, xb=48, bal=2, fastdec=1),
    "BF16_FOLD_XB24": Cfg("bf16", 8, fold=1, xb=24),
    "BF16_FOLD_PROF": Cfg("bf16", 8, fold=1, xb=40, prof=1),
    "BF16_THR8_TR": Cfg("bf16", 8, 0, tr=3),
    "F16_THR8_TR": Cfg("f16", 8, 0, tr=3),
    "BF16_FOLD_TR": Cfg("bf16", 8, fold=1, xb=40, tr=3),
    "F16_FOLD_TR": Cfg("f16", 8, fold=1, xb=40, tr=3),
    "BF16_THR8_TRK": Cfg("bf16", 8, 0, tr=1),
    "F16_THR8_TRK": Cfg("f16", 8, 0, tr=1),
    "BF16_FOLD_TRK": Cfg("bf16", 8, fold=1, xb=40, tr=1),
    "F16_FOLD_TRK": Cfg("f16", 8, fold=1, xb=40, tr=1),
    "BF16_THR8_TRV": Cfg("bf16", 8, 0, tr=2),
    "F16_THR8_TRV": Cfg("f16", 8, 0, tr=2),
    "BF16_FOLD_TRV": Cfg("bf16", 8, fold=1, xb=40, tr=2),
    "F16_FOLD_TRV": Cfg("f16", 8, fold=1, xb=40, tr=2),
    "ABL_EXPA": Cfg("bf16", 8, fold=1, prof=1, abl=("expa",)),
    "ABL_SUMPACK": Cfg("bf16", 8, fold=1, prof=1, abl=("sum", "pack")),
    "ABL_VREADA": Cfg("bf16", 8, fold=1, prof=1, abl=("vreada",)),
    "ABL_VREADB": Cfg("bf16", 8, fold=1, prof=1, abl=("vreadb",)),
    "ABL_KREAD": Cfg("bf16", 8, fold=1, prof=1, abl=("kread",)),
    "ABL_MAX": Cfg("bf16", 8, fold=1, prof=1, abl=("max",)),
    "ABL_EXPB": Cfg("bf16", 8, fold=1, prof=1, abl=("expb",)),
    "ABL_DMA": Cfg("bf16", 8, fold=1, prof=1, abl=("dma",)),
    "ABL_ALLB": Cfg("bf16", 8, fold=1, prof=1, abl=("vreadb", "kread", "max", "expb", "dma")),
    "ABL_ALLA": Cfg("bf16", 8, fold=1, prof=1, abl=("expa", "sum", "pack", "vreada")),
}

PRODUCT_STREAMS = ("BF16_THR8", "F16_THR8", "BF16_THR0", "BF16_FOLD", "F16_FOLD")
TR_STREAMS = ("BF16_THR8_TR", "F16_THR8_TR", "BF16_FOLD_TR", "F16_FOLD_TR", "BF16_THR8_TRK", "F16_THR8_TRK", "BF16_FOLD_TRK",
              "F16_FOLD_TRK", "BF16_THR8_TRV", "F16_THR8_TRV", "BF16_FOLD_TRV", "F16_FOLD_TRV")   # product too: their own list (attn_fwd16_p4_tr.h)

if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(here, "..", "metal_flash_attention_amd", "csrc", "attn_fwd16_p4_stream.inc")
    write_inc(out)
    st = Stream(VARIANTS["BF16_THR8"])
    ins = st.build()
    print("wrote", os.path.normpath(out), "-", len(ins), "instructions in the default stream")
