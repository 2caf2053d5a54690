#!/usr/bin/env python3
"""Generator of the PERSISTENT forward stream for head dimensions D <= 64 (csrc/attn_fwd16_p6.h): BASELINE config 2.

Same structure as the D <= 128 forward (tools/p4gen.py, tools/p4pgen.py: four waves x 64 query rows, one wave per SIMD with the
whole register file, 64-key tiles in two phases -- A(j): S(j) = K(j) Q^T beside the softmax finish of tile j - 1, B(j): O^T +=
V^T(j-1) P^T(j-1) beside the softmax start of tile j -- one barrier per tile, the block loop inside the statement), re-derived
for what changes at half the head dimension:

  * A tile is 16 + 16 matrix instructions (1024 clocks of matrix pipe) but the softmax of a 64 x 64 score tile costs what it
    costs at D = 128: 64 exponentials (two issue slots each), 32 packs, 32 row-maximum steps, the decision -- and 64 row-sum
    additions.  A wave alone on its SIMD issues one instruction per ~4 clocks: ~370 slots = 1480 clocks per tile against
    1024 of matrix time, the kernel is ISSUE-bound.  So the row sums move INTO the matrix pipe (`lsum`): a fifth accumulator
    block per row block, L^T += ONES P^T, whose A operand is a constant fragment of 1.0 -- eight more matrix instructions per
    tile (40: 1280 clocks) instead of 64 additions (-56 issue slots), and l sits in the accumulator file beside O^T: a deferred
    rescale multiplies it with O^T, the half-wave exchange of the epilogue disappears (the contraction runs over all 16 keys
    of a step).  The sum is that of the 16-bit P the second product multiplies, which is what the reference's mixed-precision
    mode sums (P lives in 16-bit registers there, +Precisions.swift:149-215); FOLD streams only.
  * Half as long a tile leaves the LDS-DMA half the flight time: the pieces of K(j+3) and V(j+2) lead phase B(j) -- TWO tiles
    ahead (their deadline is the barrier of tile j + 2), rings of four 8 KiB images each.  Image of tile j = j mod 4 for both
    operands, and the loop body is FOUR tiles (a block walks a multiple of four, surplus tiles fully masked): every ring
    position is an immediate -- no ring pointer, no address arithmetic per tile, one loop-exit test and one block-switch test
    of each kind per four tiles.
  * All sixteen V^T fragment reads of a tile sit in phase A (eight fragments = the whole ring of eight slots), none in B.
  * O^T leaves through a staging area of its own (the LDS is far from full): no barrier in front of the epilogue.

Register map (per lane):
    a[0:63]    O^T accumulators      (rb, db) -> 16 (2 rb + db)
    a[64:95]   L^T = ONES P^T        rb -> 64 + 16 rb        (every register holds l of the lane's row)
    a[96:127]  Q' fragments          (rb, ks) -> 96 + 4 (4 rb + ks)
    a[128:159] K fragments           (kb, ks) -> 128 + 4 (4 kb + ks)
    v[32:95] / v[96:159] score tiles of even / odd tiles, v[160:191] -m start blocks (p4gen), v[192:223] V^T fragments (f = 2 u + db)
    v[224:227] K read lane constants per k-step, v[228:231] K read addresses of the tile being fetched, v[232:247] p4gen's
    temporaries, v[248:251] LDS-DMA offsets (K piece 0, 1, V piece 0, 1), v[252:255] the ONES fragment

LDS: K ring 0..32 KiB, V ring 32..64 KiB, the waves' Q images 64..96 KiB (8 KiB each), block table 96..112 KiB, O staging
112..128 KiB (4 KiB per wave).  K-shaped images (K tiles, Q images): rows of 128 bytes, the 16-byte chunk c of row r at
position c ^ ((r >> 1) & 7) (a ds_read_b128 group of sixteen lanes then covers all 64 banks).

The stream runs on the lane-exact model first (tools/p6sim.py, tests/test_p6_stream.py).
Usage: python tools/p6gen.py   (rewrites metal_flash_attention_amd/csrc/attn_fwd16_p6_stream.inc)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import p4gen  # noqa: E402
import p4pgen  # noqa: E402
from p4gen import (A, F, I, M0, SN, V, VCC, VN, Ins, Stream, elem, s_elem, s_blk, p_word, p_frag, cm_blk, CM_BASE, S_BASE,  # noqa: E402
                   T_VADDR, T_MX, T_MN, T_SW, T_CORR, T_LB, T_MASKV, T_TL)
from p4pgen import SR, DESC_FLAGS, OOB  # noqa: E402

O_BASE, LX_BASE, Q_BASE, K_BASE = 0, 64, 96, 128
VF_BASE, T_KC, T_KA, ONES = 192, 224, 228, 252
IMG = 8192                       # one K or V tile image: 64 keys x 64 elements x 2 bytes
RING = 4
KRING, VRING_BASE = 0, RING * IMG
QIMG = 2 * RING * IMG            # 64 KiB: the four waves' Q images
TABLE = QIMG + 4 * IMG           # 96 KiB
TABLE_ENTRIES = 256
STAGE = TABLE + TABLE_ENTRIES * 64   # 112 KiB: O staging, 4 KiB per wave
LDS_BYTES = STAGE + 4 * 4096
NST = 18                         # buffer stores per wave and block: 16 x O, 2 x L
NG_A, NG_B = 16, 24
UNROLL = 4                       # tiles per loop body = ring size: a block walks a multiple of UNROLL tiles

PSGPR = dict(kres=(40, 4), vres=(44, 4), tres=(48, 4), lres=(52, 4),
             qbn=(56, 2), kbn=(58, 2), vbn=(60, 2), obn=(62, 2), lbn=(64, 2), row0n=(66, 1),
             ob=(68, 2), lb=(70, 2), row0=(72, 1), blk=(73, 1), hasnext=(74, 1), ntm2=(75, 1), ntm3=(76, 1),
             j=(77, 1), pend=(80, 1), t0=(81, 1), t1=(82, 1), t2=(83, 1), sv=(84, 2),
             kc0=(86, 1), kstep=(87, 1), vstep=(88, 1), ntb=(89, 1), maskb=(90, 1), q8=(94, 1), t3=(95, 1), t4=(96, 1), qrow=(97, 1),
             # causal ("geometry") streams, round 6: rows / keys of the NEXT table entry, rows / last key / key tiles / diagonal offset of
             # the current block (per-batch lengths travel in the block table, as in tools/p4pgen.py)
             rrn=(91, 1), ccn=(92, 1), rr=(93, 1), cm1=(78, 1), ttot=(79, 1), coff=(67, 1))
FIRST_CLOBBERED_SGPR, LAST_CLOBBERED_SGPR = 40, 97
FIRST_OWNED_VGPR = 28

INOUT_V = ["lim0", "lim1"]
IN_V = ["kbase", "vbase", "kv0", "kv1", "vv", "qv0", "qv1", "ov0", "ov1", "lv", "ewa", "era", "qlane", "hi4"]
IN_S = ["nt", "maskfrom", "scale2", "kinc", "vinc", "ldsk", "ldsv", "ldsq", "qrel", "ldsst", "nblk", "tbl", "wave64", "ldq2", "ldo",
        "nrecq", "nreck", "nrecv", "nreco", "nrecl", "cflag"]   # (nrec*: dense streams; cflag: causal streams -- 1 = causal mask, 0 = lengths only)


class Cfg6(p4gen.Cfg):
    def __init__(self, dtype="bf16", thr=8.0, xb=56, o16=0, l16=1, lsum=1, abl=(), pad=0, vlast=7, kearly=1, fold=1, causal=0, split=0):
        """fold = 0: EXACT-SCALE streams (descriptors that keep the attention matrix in FP32 registers): Q stays as stored, the scale
        is applied in fp32 per score (s * scale2 - m, 64 more vector instructions per tile), the row sums are fp32 additions of
        the unrounded P (no `lsum`), L is stored in FP32.  xb = scores per tile whose exponential phase B takes (both kinds)"""
        p4gen.Cfg.__init__(self, dtype=dtype, thr=thr, fold=fold, xb=xb, xe=0 if fold else xb, abl=abl, pad=pad, bal=2, maxa=1)
        # (bal / maxa: p4gen's mask section then re-takes the first key block's row maxima, which phase A took from the unmasked scores)
        self.xb = xb
        if not fold:
            lsum, l16 = 0, 0
        self.o16, self.l16, self.lsum = o16, l16, lsum
        # causal (extension: row r sees key c iff c <= r + C - R): tile count (a multiple of four), first masked tile and the lanes'
        # mask limits are computed per block inside the stream; EVERY wave walks the block's tiles (the keys beyond a wave's own
        # diagonal are masked like a ragged edge: 1.5 of a block's 4 (rb + 1) tiles on average) -- no per-wave traversal bound, no
        # skip loop.  The block table lists the blocks in pairs (long, short) so that every workgroup walks the same number of tiles
        self.causal = causal
        # split (column-parallel launches of few-workgroup problems: one head, BASELINE config 2 as written): a table entry is one
        # (row block, piece of the key range) -- its K / V bases start at the piece, its O / L bases are the piece's slabs in the
        # caller's workspace -- and the epilogue leaves the UN-normalised O^T (fp32) and (m, l) per row there for attn_fwd_combine
        # (attn_fwd16_v3.h), like the split siblings of the other forward kernels.  Pieces are whole multiples of four tiles
        self.split = split
        if split:
            self.o16, self.l16 = 0, 0
        assert not (split and causal)
        # vlast: last gap of phase A with a V^T read; kearly: the K(j+1) fragment reads right behind the LDS-DMA pieces of phase B
        # (else spread between its exponentials, the last one near the phase's end -- the first GPU runs lost ~250 clocks per tile
        # to the two `lgkmcnt(0)` in front of the phase seams, profiles/r05_p6_ablations_first.txt)
        self.vlast, self.kearly = vlast, kearly


def s(name, cnt=None, off=0):
    base, n = PSGPR[name]
    return SR(base + off, n if cnt is None else cnt)


def o_acc(rb, db):
    return A(O_BASE + 16 * (2 * rb + db), 16)


def lx_acc(rb):
    return A(LX_BASE + 16 * rb, 16)


def q_frag(rb, ks):
    return A(Q_BASE + 4 * (4 * rb + ks), 4)


def k_frag(kb, ks):
    return A(K_BASE + 4 * (4 * kb + ks), 4)


def vf_frag(f):
    return V(VF_BASE + 4 * f, 4)


def vf_half(f, h):
    return V(VF_BASE + 4 * f + 2 * h, 2)


def deal(items, ngaps, first=0):
    """items: ordered [(cost, fn, earliest gap)] -> per-gap lists with the cumulative cost following the straight line from
    gap `first` to the last gap; an item never sits in front of its earliest gap or of an item listed before it"""
    total = sum(c for c, _, _ in items)
    out = [[] for _ in range(ngaps)]
    per = total / float(ngaps - first) if ngaps > first else total
    g, used = first, 0.0
    for c, fn, earliest in items:
        while g < ngaps - 1 and used + c / 2.0 > (g - first + 1) * per:
            g += 1
        if earliest is not None and g < earliest:
            g = min(earliest, ngaps - 1)
            used = max(used, (g - first) * per)
        out[g].append(fn)
        used += c
    return out


class Stream6(Stream):
    persistent = True

    def __init__(self, cfg):
        Stream.__init__(self, cfg)
        d0 = 248 if cfg.fold else CM_BASE
        self.vfixed = {"m0": 28, "m1": 29, "l0": 30, "l1": 31, "koff0": d0, "koff1": d0 + 1, "voff0": d0 + 2, "voff1": d0 + 3}

    def finish(self):
        def m(o):
            if o is None:
                return None
            if o[0] == "V" and o[1] in self.vfixed:
                return V(self.vfixed[o[1]])
            if o[0] == "S" and self.cfg.causal and o[1] in ("nt", "maskfrom"):
                return SR(PSGPR[{"nt": "ntb", "maskfrom": "maskb"}[o[1]]][0], 1)
            if o[0] == "S" and o[1] == "wnt":
                return ("S", "nt", 1)
            if o[0] == "S" and o[1] in PSGPR:
                base, n = PSGPR[o[1]]
                return SR(base, o[2] if len(o) > 2 else n)
            return o
        for ins in self.ins:
            ins.d = m(ins.d)
            ins.s = tuple(m(x) for x in ins.s)
        return self.ins

    # ------------------------------------------------------------ pieces shared with p4pgen (restated for this geometry)
    def desc(self, name, base, nrec):
        self.emit("s_mov_b32", s(name, 1, 0), [s(base, 1, 0)])
        self.emit("s_and_b32", s(name, 1, 1), [s(base, 1, 1), I(0xFFFF)])
        self.nrec(name, nrec)
        self.emit("s_mov_b32", s(name, 1, 3), [I(DESC_FLAGS)])

    def nrec(self, name, which):
        """word 2 of resource `name`: a launch constant in dense streams; rows (or keys) of the block's own batch entry x bytes per row in
        causal ("geometry") streams -- nrecq / nreck / nrecv belong to the NEXT block, nreco / nrecl to the current one"""
        if not self.cfg.causal:
            self.emit("s_mov_b32", s(name, 1, 2), [SN(which)])
            return
        d = s(name, 1, 2)
        if which == "nrecq":
            self.emit("s_mul_i32", d, [s("rrn"), SN("ldq2")])
        elif which in ("nreck", "nrecv"):
            self.emit("s_lshr_b32", d, [SN("kinc" if which == "nreck" else "vinc"), I(6)])    # bytes per row
            self.emit("s_mul_i32", d, [s("ccn"), d])
        elif which == "nreco":
            self.emit("s_mul_i32", d, [s("rr"), SN("ldo")])
        else:
            self.emit("s_lshl_b32", d, [s("rr"), I(1 if self.cfg.l16 else 2)])

    def lds_write(self, op, addr, data, offset):
        self.emit(op, None, [addr, data], offset=offset)
        self.lds_issued += 1

    def load_next(self):
        tv, tb = 96, 100
        self.emit("s_lshl_b32", s("t0"), [s("blk"), I(6)])
        self.emit("s_add_u32", s("t0"), [s("t0"), SN("tbl")])
        self.emit("v_mov_b32", V(tv), [s("t0")])
        ids = [self.lds_read("ds_read_b128", V(tb + 4 * i, 4), V(tv), 16 * i, note="block table") for i in range(4 if self.cfg.causal else 3)]
        self.lds_need(ids[-1])
        self.lds_flush()
        words = [("qbn", 0), ("qbn", 1), ("kbn", 0), ("kbn", 1), ("vbn", 0), ("vbn", 1), ("obn", 0), ("obn", 1), ("lbn", 0), ("lbn", 1), ("row0n", 0)]
        if self.cfg.causal:
            words += [("rrn", 0), ("ccn", 0)]      # rows and keys of the block's batch entry (words 11, 12)
        for i, (name, off) in enumerate(words):
            self.emit("v_readfirstlane_b32", s(name, 1, off), [V(tb + i)])
        self.emit("s_nop", None, [I(4)], note="v_readfirstlane -> SALU / VMEM use of the scalar")

    def switch_k(self):
        """K descriptor and piece offsets of the NEXT block: piece i = rows 16 wave + 8 i .. + 7"""
        self.emit("s_mov_b32", s("kres", 1, 0), [s("kbn", 1, 0)])
        self.emit("s_and_b32", s("kres", 1, 1), [s("kbn", 1, 1), I(0xFFFF)])
        if self.cfg.causal:
            self.nrec("kres", "nreck")
        self.emit("v_add_u32_e64", VN("koff0"), [VN("kv0"), s("kc0")], clamp=1)
        self.emit("s_add_u32", s("t3"), [s("kc0"), s("kstep")])
        self.emit("v_add_u32_e64", VN("koff1"), [VN("kv1"), s("t3")], clamp=1)

    def switch_v(self):
        """V descriptor and piece offsets of the next block: piece i = keys 32 (wave & 1) + 16 i .. + 15 (the lane part holds the 32)"""
        self.emit("s_mov_b32", s("vres", 1, 0), [s("vbn", 1, 0)])
        self.emit("s_and_b32", s("vres", 1, 1), [s("vbn", 1, 1), I(0xFFFF)])
        if self.cfg.causal:
            self.nrec("vres", "nrecv")
        self.emit("v_mov_b32", VN("voff0"), [VN("vv")])
        self.emit("v_add_u32_e64", VN("voff1"), [VN("vv"), s("vstep")], clamp=1)

    def issue_q(self, temps):
        """the wave's 64 rows of the next block's Q' by LDS-DMA into its image: piece i = rows 8 i .. 8 i + 7"""
        self.desc("tres", "qbn", "nrecq")
        self.emit("s_add_u32", s("t0"), [s("row0n"), SN("wave64")])
        self.emit("s_mul_i32", s("qrow"), [s("t0"), SN("ldq2")])
        for i in range(8):
            t = V(temps[i % len(temps)])
            self.emit("v_add_u32_e64", t, [VN("qv%d" % (i & 1)), s("qrow")], clamp=1)
            self.emit("s_add_u32", M0, [SN("ldsq"), I(i * 1024)])
            self.emit("buffer_load_dwordx4_lds", None, [t, s("tres", 4)])
            if i != 7:
                self.emit("s_add_u32", s("qrow"), [s("qrow"), s("q8")])

    def issue_tile(self, which, image):
        """prologue of the first block: both pieces of tile `image` (= its ring image), offsets advanced"""
        for i in range(2):
            if which == "k":
                self.emit("s_add_u32", M0, [SN("ldsk"), I(image * IMG + i * 1024)])
                self.emit("buffer_load_dwordx4_lds", None, [VN("koff%d" % i), s("kres", 4)])
            else:
                self.emit("s_add_u32", M0, [SN("ldsv"), I(image * IMG + i * 1024)])
                self.emit("buffer_load_dwordx4_lds", None, [VN("voff%d" % i), s("vres", 4)])
        for i in range(2):
            if which == "k":
                self.emit("v_add_u32_e64", VN("koff%d" % i), [VN("koff%d" % i), SN("kinc")], clamp=1)
            else:
                self.emit("v_add_u32_e64", VN("voff%d" % i), [VN("voff%d" % i), SN("vinc")], clamp=1)

    def k_read(self, i, img):
        """K fragment i = (kb, ks) of the tile in ring image `img`"""
        kb, ks = divmod(i, 4)
        return self.lds_read("ds_read_b128", k_frag(kb, ks), V(T_KC + ks), img * IMG + kb * 4096, note="K(%d,%d)" % (kb, ks))

    def v_read(self, i, img):
        """V^T read i (0..15) of the tile in ring image `img`: fragment f = i // 2 = 2 u + db, half i % 2 (keys + 0..3 / + 8..11 of
        the 16-key group)"""
        f, h = divmod(i, 2)
        u, db = divmod(f, 2)
        off = img * IMG + (db * 64 + 16 * u) * 64 + h * 8 * 64
        return self.lds_read("ds_read_b64_tr_b16", vf_half(f, h), VN("vbase"), off, note="V^T f%d.%d" % (f, h))

    def q_fragments(self):
        """Q image -> B-operand fragments a[96:127], times log2(e) / sqrt(D) and rounded to the 16-bit type"""
        cfg = self.cfg
        qa, qd, t0, t1 = 96, 32, V(104), V(105)
        for ks in range(4):
            self.emit("v_add_u32", V(qa + ks), [SN("qrel"), V(T_KC + ks)])
        if not cfg.fold:
            for rb in range(2):
                for ks in range(4):
                    self.lds_read("ds_read_b128", q_frag(rb, ks), V(qa + ks), rb * 4096, note="Q(%d,%d)" % (rb, ks))
            self.lds_flush()
            return
        ids = []
        for rb in range(2):
            for ks in range(4):
                ids.append(self.lds_read("ds_read_b128", V(qd + 4 * (4 * rb + ks), 4), V(qa + ks), rb * 4096, note="Q(%d,%d)" % (rb, ks)))
        for n in range(8):
            self.lds_need(ids[n])
            for w in range(4):
                x = V(qd + 4 * n + w)
                if cfg.dtype == "bf16":
                    self.emit("v_lshlrev_b32", t0, [I(16), x])
                    self.emit("v_and_b32", t1, [I(0xFFFF0000), x])
                else:
                    self.emit("v_cvt_f32_f16", t0, [x])
                    self.emit("v_lshrrev_b32", t1, [I(16), x])
                    self.emit("v_cvt_f32_f16", t1, [t1])
                self.emit("v_mul_f32", t0, [SN("scale2"), t0])
                self.emit("v_mul_f32", t1, [SN("scale2"), t1])
                self.emit("v_cvt_pk_%s_f32" % cfg.dtype, x, [t0, t1])
                self.emit("v_accvgpr_write_b32", A(Q_BASE + 4 * n + w), [x])
        self.lds_flush()

    # ------------------------------------------------------------ phase A
    def qk_list(self, par):
        out = []
        for g in range(NG_A):
            kb, rb, ks = g // 8, g % 2, (g % 8) // 2
            c = (cm_blk(rb) if self.cfg.fold else I(0)) if ks == 0 else s_blk(par, rb, kb)
            out.append((s_blk(par, rb, kb), k_frag(kb, ks), q_frag(rb, ks), c))
        return out

    def phase_a(self, par, mfma, softmax, zero_o, t):
        """A(j): S[par] = K(j) Q'^T - m  |  exponentials phase B left, packs (and row sums without `lsum`) of tile j - 1, all 16
        V^T(j-1) fragment reads, the row maxima of the first key block of tile j (complete behind matrix instruction 7); t = j mod 4"""
        cfg = self.cfg
        abl = cfg.abl if (mfma and softmax) else frozenset()
        prev = par ^ 1
        fill = [[] for _ in range(NG_A + 1)]
        vids = {}
        if softmax:
            ea = list(range(cfg.xb, 64))
            exp_gap = {e: (t * NG_A) // len(ea) for t, e in enumerate(ea)} if ea else {}
            for e in ea:
                if "exp" not in abl:
                    fill[exp_gap[e]].append(lambda e=e: self.exp_op(prev, e))
            g_prev = 0
            for p in range(32):
                ready = max(exp_gap.get(2 * p, -1), exp_gap.get(2 * p + 1, -1))
                g_prev = min(NG_A, max(g_prev, p // 2, ready + 1))
                if "pack" not in abl:
                    fill[g_prev].append(lambda p=p: self.sum_pack6(prev, 2 * p))
            for i in range(16):    # done by gap cfg.vlast: the wait in front of the barrier must not expose their latency
                if "lds" not in abl:
                    fill[(i * (cfg.vlast + 1)) // 16].append(lambda i=i: vids.__setitem__(i, self.v_read(i, (t + 3) % 4)))
        if mfma and "max" not in abl:
            for k in range(16):    # (rb0, kb0) is complete behind matrix instruction 6, (rb1, kb0) behind 7: first steps from gap 9 / 12
                fill[9 + (k * 7) // 16].append(lambda k=k: self.max_op(par, k))
        mlist = self.qk_list(par)
        zero = [O_BASE + i for i in range(64)] + ([LX_BASE + i for i in range(32)] if cfg.lsum else [])
        for g in range(NG_A + 1):
            if g < NG_A and mfma:
                self.mfma(*mlist[g])
            if g < NG_A and zero_o:
                for r in zero[g * 6:(g + 1) * 6]:
                    self.emit("v_accvgpr_write_b32", A(r), [I(0)])
            for fn in fill[g]:
                fn()
        return vids

    def exp_op(self, par, e):
        rb, kb, r = elem(e)
        x = s_elem(par, rb, kb, r)
        self.emit("v_exp_f32", x, [x])

    def sum_pack6(self, prev, e):
        rb, kb, r = elem(e)
        x0, x1 = s_elem(prev, rb, kb, r), s_elem(prev, rb, kb, r + 1)
        if not self.cfg.lsum:
            self.emit("v_add_f32", VN("l%d" % rb), [x0, VN("l%d" % rb)])
            self.emit("v_add_f32", V(T_LB + rb), [x1, V(T_LB + rb)])
        self.emit("v_cvt_pk_%s_f32" % self.cfg.dtype, p_word(prev, rb, 2 * kb + r // 8, (r % 8) // 2), [x0, x1])

    # ------------------------------------------------------------ phase B
    def pv_list(self, par):
        """(dst, a, b, c) of phase B: per 16-key step u the four O^T products, then (lsum) the two row-sum products"""
        out = []
        for u in range(4):
            for db in range(2):
                for rb in range(2):
                    out.append((o_acc(rb, db), vf_frag(2 * u + db), p_frag(par ^ 1, rb, u), o_acc(rb, db)))
            if self.cfg.lsum:
                for rb in range(2):
                    out.append((lx_acc(rb), V(ONES, 4), p_frag(par ^ 1, rb, u), lx_acc(rb)))
        return out

    def dma_pieces(self, which, img):
        """this wave's two pieces of K(j+3) or V(j+2) into ring image `img`"""
        for i in range(2):
            if which == "k":
                self.emit("s_add_u32", M0, [SN("ldsk"), I(img * IMG + i * 1024)])
                self.emit("buffer_load_dwordx4_lds", None, [VN("koff%d" % i), s("kres", 4)])
            else:
                self.emit("s_add_u32", M0, [SN("ldsv"), I(img * IMG + i * 1024)])
                self.emit("buffer_load_dwordx4_lds", None, [VN("voff%d" % i), s("vres", 4)])

    def decide_fast(self, lbl, first):
        """any score of the tile above THR (scores are relative to m)?  One half-wave exchange of the two row blocks' partial
        maxima against EACH OTHER gives x = [row maxima of row block 0 | of row block 1] in the two half-waves: enough for the
        branch (one compare); the per-lane row maxima of both blocks are only rebuilt in the out-of-line section (rare)"""
        for rb in range(2):
            self.emit("v_max_f32", V(T_MN + rb), [V(T_MX + 2 * rb), V(T_MX + 2 * rb + 1)])
        self.emit("s_nop", None, [I(1)], note="VALU write -> permlane read")
        self.emit("v_permlane32_swap_b32", V(T_MN), [V(T_MN + 1)], swap=1)
        self.emit("v_max_f32", V(T_SW), [V(T_MN), V(T_MN + 1)])
        if first:
            self.emit("s_branch", None, [], target=lbl)
        else:
            self.emit("v_cmp_lt_f32", VCC, [F(self.cfg.thr), V(T_SW)])
            self.emit("s_cbranch_vccnz", None, [], target=lbl)
        self.label(lbl + "_BACK")

    def check(self, reg, lbl):
        self.emit("s_cmp_eq_u32", None, [SN("j"), s(reg)])
        self.emit("s_cbranch_scc1", None, [], target=lbl)
        self.label(lbl + "_BACK")

    def phase_b(self, par, mfma, softmax, t):
        """B(j): O^T += V^T(j-1) P^T(j-1), L^T += ONES P^T(j-1)  |  LDS-DMA of K(j+3), V(j+2) FIRST, row maxima of the second key block,
        decision, the first xb exponentials, K(j+1) fragments; t = j mod 4: K(j+3) -> image t + 3, V(j+2) -> image t + 2, K(j+1) in
        image t + 1 (mod 4).  The next block's K switch (tile nt - 3) can only fall on t = 1, its V / Q switch (nt - 2) on t = 2"""
        cfg = self.cfg
        abl = cfg.abl if (mfma and softmax) else frozenset()
        if not mfma and softmax:
            self.emit("s_nop", None, [I(15)], note="S(0) is still leaving the matrix pipe")
        if softmax:
            self.mask_section(par, after_mfma=mfma)
        mlist = self.pv_list(par) if mfma else []
        ng = len(mlist) if mfma else NG_B
        items = []
        if softmax:
            ksw, vsw = self.newlabel("KSW"), self.newlabel("VSW")
            if t == 1:
                self.outofline.append(("ksw", ksw, ksw + "_BACK", par, False))
            if t == 2:
                self.outofline.append(("vsw", vsw, vsw + "_BACK", par, False))
            dec_lbl = self.newlabel("DEC")
            first = not mfma
            if t == 1:
                items.append((2, lambda: self.check("ntm3", ksw), None))
            if "dma" not in abl:
                items.append((4, lambda: self.dma_pieces("k", (t + 3) % 4), None))
            if t == 2:
                items.append((2, lambda: self.check("ntm2", vsw), None))
            if "dma" not in abl:
                items.append((4, lambda: self.dma_pieces("v", (t + 2) % 4), None))
            if cfg.kearly:
                for i in range(8):
                    if "lds" not in abl:
                        items.append((1, lambda i=i: self.k_read(i, (t + 1) % 4), None))
            for i in range(16, 32):
                if "max" not in abl:
                    items.append((1, lambda i=i: self.max_op(par, i), None))
            if cfg.fold:
                items.append((7, lambda: self.decide_fast(dec_lbl, first), None))
            else:
                items.append((4, lambda: self.decide_1(), None))
                items.append((5, lambda: self.decide_2(), None))
                items.append((4, lambda: self.decide_3(), None))
                items.append((5, lambda: self.decide_4(dec_lbl), None))
            nexp, nk = cfg.xb, 0
            for e in range(64 if not cfg.fold else nexp):
                if not cfg.fold:      # s * scale2 - m of every score; the exponential of the first xb two scores behind its own
                    items.append((1, lambda e=e: self.fma_plain(par, e), None))
                    e -= 2
                    if e < 0 or e >= nexp:
                        continue
                if "exp" not in abl:
                    items.append((2, lambda e=e: self.exp_in_b(par, e), None))
                if not cfg.kearly and (e + 1) * 8 // nexp > nk:
                    if "lds" not in abl:
                        items.append((1, lambda i=nk: self.k_read(i, (t + 1) % 4), None))
                    nk += 1
            for i in range(nk, 8):
                if not cfg.kearly and "lds" not in abl:
                    items.append((1, lambda i=i: self.k_read(i, (t + 1) % 4), None))
            if not cfg.fold:
                for e in range(62, 64):
                    if e < nexp and "exp" not in abl:
                        items.append((2, lambda e=e: self.exp_in_b(par, e), None))
            for i in range(2):
                items.append((1, lambda i=i: self.emit("v_add_u32_e64", VN("koff%d" % i), [VN("koff%d" % i), SN("kinc")], clamp=1), None))
                items.append((1, lambda i=i: self.emit("v_add_u32_e64", VN("voff%d" % i), [VN("voff%d" % i), SN("vinc")], clamp=1), None))
        fill = deal(items, ng)
        for g in range(ng):
            if mfma:
                self.mfma(*mlist[g])
            for fn in fill[g]:
                fn()
        self.lds_flush()
        if softmax:
            resc, back = self.newlabel("RESC"), self.newlabel("RESCBACK")
            self.emit("s_cmp_eq_u32", None, [SN("pend"), I(0)])
            self.emit("s_cbranch_scc0", None, [], target=resc)
            self.label(back)
            self.outofline.append(("resc", resc, back, par, False))
            self.outofline.append(("dec", dec_lbl, dec_lbl + "_BACK", par, not mfma))

    # ------------------------------------------------------------ out-of-line sections
    def emit_outofline(self):
        cfg = self.cfg
        rs = VF_BASE      # rescale temporaries: the V^T fragment registers are idle at the end of phase B
        for kind, lbl, back, par, first in self.outofline:
            self.label(lbl)
            if kind in ("ksw", "vsw"):
                self.emit("s_cmp_eq_u32", None, [s("hasnext"), I(0)])
                self.emit("s_cbranch_scc1", None, [], target=back)      # last block: the ring runs ahead into zeros (out of range)
                if kind == "ksw":
                    self.switch_k()
                else:
                    self.switch_v()
                    self.issue_q((T_SW, T_SW + 1, T_TL, T_TL + 1, T_MASKV))
                self.emit("s_branch", None, [], target=back)
            elif kind == "dec" and not cfg.fold:   # onlineCorrectO factors (+Softmax.swift:290-301): m_up = max(m, m_new), corr = 2^(m - m_up)
                T_THR = p4gen.T_THR
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
            elif kind == "dec":
                # x = v[T_SW] = [maxima of row block 0 | of row block 1] -> both blocks' row maxima in every lane
                self.emit("v_mov_b32", V(T_MN), [V(T_SW)])
                self.emit("v_mov_b32", V(T_MN + 1), [V(T_SW)])
                self.emit("s_nop", None, [I(1)], note="VALU write -> permlane read")
                self.emit("v_permlane32_swap_b32", V(T_MN), [V(T_MN + 1)], swap=1)
                ta, tb = V(T_SW), V(T_SW + 1)
                for rb in range(2):
                    mn, m = V(T_MN + rb), VN("m%d" % rb)
                    if not first:
                        self.emit("v_max_f32", mn, [I(0), mn])
                    self.emit("v_add_f32", ta, [m, mn])
                    self.emit("v_sub_f32", tb, [ta, m])
                    self.emit("v_mov_b32", m, [ta])
                    self.emit("v_exp_f32", V(T_CORR + rb), [tb], neg0=1)
                    for kb in range(2):
                        for r in range(16):
                            x = s_elem(par, rb, kb, r)
                            self.emit("v_sub_f32", x, [x, tb])
                    for r in range(16):
                        self.emit("v_sub_f32", V(CM_BASE + 16 * rb + r), [I(0), ta])
                self.emit("s_mov_b32", SN("pend"), [I(0 if first else 1)])
                self.emit("s_branch", None, [], target=back)
            else:               # O^T, L^T (or l) *= corr once every matrix instruction that accumulates P(j-1) has been issued
                self.emit("s_nop", None, [I(15)])
                self.emit("s_nop", None, [I(7)])
                for rb in range(2):
                    regs = [O_BASE + 32 * rb + i for i in range(32)] + ([LX_BASE + 16 * rb + i for i in range(16)] if cfg.lsum else [])
                    for i0 in range(0, len(regs), 8):
                        for t in range(8):
                            self.emit("v_accvgpr_read_b32", V(rs + t), [A(regs[i0 + t])])
                        for t in range(8):
                            self.emit("v_mul_f32", V(rs + t), [V(T_CORR + rb), V(rs + t)])
                        for t in range(8):
                            self.emit("v_accvgpr_write_b32", A(regs[i0 + t]), [V(rs + t)])
                    if not cfg.lsum:
                        self.emit("v_mul_f32", VN("l%d" % rb), [V(T_CORR + rb), VN("l%d" % rb)])
                        self.emit("v_mul_f32", V(T_LB + rb), [V(T_CORR + rb), V(T_LB + rb)])
                self.emit("s_mov_b32", SN("pend"), [I(0)])
                self.emit("s_nop", None, [I(4)], note="accvgpr write -> MFMA SrcC")
                self.emit("s_branch", None, [], target=back)

    # ------------------------------------------------------------ epilogue
    EPI_LTOT, EPI_INV = (T_MX, T_MX + 1), (T_MN, T_MN + 1)
    EPI_WA, EPI_RA, EPI_VO = [T_CORR, T_CORR + 1, T_LB, T_LB + 1], T_MASKV, T_TL

    def epilogue(self):
        """O /= l (+Source.swift:165-171), L = m + log2 l (+Caching.swift:373-377); O^T through the wave's 4 KiB staging slice
        (in: lane = row, 16-byte chunks XOR row & 7; out: lane = (row & 7, chunk), eight lanes per 128-byte line)"""
        cfg = self.cfg
        self.emit("s_nop", None, [I(15)], note="the last accumulating MFMAs leave the matrix pipe")
        self.emit("s_nop", None, [I(7)])
        self.desc("tres", "ob", "nreco")
        self.desc("lres", "lb", "nrecl")
        ltot, inv, ta, tb = self.EPI_LTOT, self.EPI_INV, V(T_SW), V(T_SW + 1)
        for rb in range(2):
            lt, iv = V(ltot[rb]), V(inv[rb])
            if cfg.lsum:
                self.emit("v_accvgpr_read_b32", lt, [A(LX_BASE + 16 * rb)])
            else:
                self.emit("v_add_f32", VN("l%d" % rb), [V(T_LB + rb), VN("l%d" % rb)])
                self.emit("v_mov_b32", ta, [VN("l%d" % rb)])
                self.emit("v_mov_b32", tb, [VN("l%d" % rb)])
                self.emit("s_nop", None, [I(1)], note="VALU write -> permlane read")
                self.emit("v_permlane32_swap_b32", ta, [tb], swap=1)
                self.emit("v_add_f32", lt, [ta, tb])
            self.emit("v_add_f32", lt, [I(1), lt], note="+ denorm_min (+Caching.swift:311)")
            if cfg.split:
                continue
            self.emit("v_rcp_f32", iv, [lt])
            self.emit("s_nop", None, [I(0)], note="trans -> VALU")
            self.emit("v_fma_f32", ta, [lt, iv, F(1.0)], neg0=1)
            self.emit("v_fma_f32", iv, [ta, iv, iv])
            self.emit("v_cmp_lt_f32", VCC, [F(1e-30), lt])
            self.emit("v_cndmask_b32", iv, [I(0), iv, VCC])
        wa, ra, vo = self.EPI_WA, self.EPI_RA, V(self.EPI_VO)
        for g in range(4):
            if g:
                self.emit("v_xor_b32", V(wa[g]), [I(g << 5), VN("ewa")])
                self.emit("v_add_u32", V(wa[g]), [SN("ldsst"), V(wa[g])])
            else:
                self.emit("v_add_u32", V(wa[g]), [SN("ldsst"), VN("ewa")])
        self.emit("v_add_u32", V(ra), [SN("ldsst"), VN("era")])
        self.emit("s_lshl_b32", s("t4"), [SN("ldo"), I(3)])          # eight rows
        pending = None

        def stores(rb, db, dst, ids):
            self.lds_need(ids[-1])
            for k in range(4):
                if k == 0:
                    self.emit("s_add_u32", s("t0"), [s("row0"), SN("wave64")])
                    if rb:
                        self.emit("s_add_u32", s("t0"), [s("t0"), I(32)])
                    self.emit("s_mul_i32", s("t0"), [s("t0"), SN("ldo")])
                else:
                    self.emit("s_add_u32", s("t0"), [s("t0"), s("t4")])
                self.emit("v_add_u32_e64", vo, [VN("ov%d" % db), s("t0")], clamp=1)
                if cfg.o16:
                    self.emit("v_cvt_pk_%s_f32" % cfg.dtype, V(dst + 4 * k), [V(dst + 4 * k), V(dst + 4 * k + 1)])
                    self.emit("v_cvt_pk_%s_f32" % cfg.dtype, V(dst + 4 * k + 1), [V(dst + 4 * k + 2), V(dst + 4 * k + 3)])
                    self.emit("buffer_store_dwordx2", None, [V(dst + 4 * k, 2), vo, s("tres", 4)], offset=0)
                else:
                    self.emit("buffer_store_dwordx4", None, [V(dst + 4 * k, 4), vo, s("tres", 4)], offset=0)

        for i, (rb, db) in enumerate([(b // 2, b % 2) for b in range(4)]):
            src, dst = S_BASE[0] + 16 * i, S_BASE[1] + 16 * i
            for r in range(16):
                self.emit("v_accvgpr_read_b32", V(src + r), [A(O_BASE + 16 * (2 * rb + db) + r)])
            if not cfg.split:
                for r in range(16):
                    self.emit("v_mul_f32", V(src + r), [V(inv[rb]), V(src + r)])
            for g in range(4):
                self.lds_write("ds_write_b128", V(wa[g]), V(src + 4 * g, 4), 0)
            ids = [self.lds_read("ds_read_b128", V(dst + 4 * k, 4), V(ra), 1024 * k, note="O(%d,%d) rows %d.." % (rb, db, 8 * k)) for k in range(4)]
            if pending is not None:
                stores(*pending)
            pending = (rb, db, dst, ids)
        stores(*pending)
        self.lds_flush()
        for rb in range(2):
            if cfg.split:     # (m, l) of the lane's row, two floats at 8 bytes per row (attn_fwd_combine merges the pieces)
                self.emit("s_add_u32", s("t0"), [s("row0"), SN("wave64")])
                if rb:
                    self.emit("s_add_u32", s("t0"), [s("t0"), I(32)])
                self.emit("s_lshl_b32", s("t0"), [s("t0"), I(3)])
                self.emit("v_add_u32_e64", vo, [VN("lv"), s("t0")], clamp=1)
                # (one 8-byte store: an instruction offset on top of an out-of-range lane offset would wrap past 2^32 into the buffer)
                pair = T_CORR + 1      # v242:243 -- the staging addresses are dead here; an EVEN-aligned pair (gfx950: 64-bit VGPR tuples)
                assert pair % 2 == 0
                self.emit("v_mov_b32", V(pair), [VN("m%d" % rb)])
                self.emit("v_mov_b32", V(pair + 1), [V(ltot[rb])])
                self.emit("buffer_store_dwordx2", None, [V(pair, 2), vo, s("lres", 4)], offset=0)
                continue
            x = V(T_SW + rb)
            self.emit("v_log_f32", x, [V(ltot[rb])])
            self.emit("s_nop", None, [I(0)], note="trans -> VALU")
            self.emit("v_add_f32", x, [VN("m%d" % rb), x])
            self.emit("s_add_u32", s("t0"), [s("row0"), SN("wave64")])
            if rb:
                self.emit("s_add_u32", s("t0"), [s("t0"), I(32)])
            self.emit("s_lshl_b32", s("t0"), [s("t0"), I(1 if cfg.l16 else 2)])
            self.emit("v_add_u32_e64", vo, [VN("lv"), s("t0")], clamp=1)
            if cfg.l16:
                self.emit("v_cvt_f16_f32", x, [x])
                self.emit("buffer_store_short", None, [x, vo, s("lres", 4)], offset=0)
            else:
                self.emit("buffer_store_dword", None, [x, vo, s("lres", 4)], offset=0)

    # ------------------------------------------------------------ block loop
    def block_head(self, nonext):
        for name in ("ob", "lb"):
            self.emit("s_mov_b32", s(name, 1, 0), [s(name + "n", 1, 0)])
            self.emit("s_mov_b32", s(name, 1, 1), [s(name + "n", 1, 1)])
        self.emit("s_mov_b32", s("row0"), [s("row0n")])
        if self.cfg.causal:
            # rows / last key / key tiles / diagonal offset of THIS block's batch entry (the *n registers still hold its table entry); without
            # the causal mask the offset is out of reach of any row and the same arithmetic yields the dense geometry (tools/p4pgen.py)
            self.emit("s_mov_b32", s("rr"), [s("rrn")])
            self.emit("s_sub_u32", s("cm1"), [s("ccn"), I(1)])
            self.emit("s_add_u32", s("ttot"), [s("ccn"), I(63)])
            self.emit("s_lshr_b32", s("ttot"), [s("ttot"), I(6)])
            self.emit("s_max_u32", s("ttot"), [s("ttot"), I(1)])
            self.emit("s_sub_u32", s("coff"), [s("ccn"), s("rrn")])
            self.emit("s_max_i32", s("coff"), [s("coff"), I(0)])
            self.emit("s_cmp_eq_u32", None, [SN("cflag"), I(0)])
            self.emit("s_cselect_b32", s("coff"), [I(0x40000000), s("coff")])
            self.block_geometry()
        self.emit("s_add_u32", s("blk"), [s("blk"), I(1)])
        self.emit("s_mov_b32", s("hasnext"), [I(0)])
        self.emit("s_cmp_ge_u32", None, [s("blk"), SN("nblk")])
        self.emit("s_cbranch_scc1", None, [], target=nonext)
        self.emit("s_mov_b32", s("hasnext"), [I(1)])
        self.load_next()
        self.label(nonext)

    def block_geometry(self):
        """causal streams, per block: tile count (made a multiple of four), first tile that needs masking for this wave, the lanes'
        mask limits min(C - 1, row + C - R) - 4 hi"""
        t0, t2, t3 = s("t0"), s("t2"), s("t3")
        self.emit("s_add_u32", t0, [s("row0"), I(256)])
        self.emit("s_min_u32", t0, [t0, SN("rr")])
        self.emit("s_sub_u32", t0, [t0, I(1)])                       # last row of the block
        self.emit("s_add_u32", t0, [t0, SN("coff")])
        self.emit("s_lshr_b32", t0, [t0, I(6)])
        self.emit("s_add_u32", t0, [t0, I(1)])
        self.emit("s_min_u32", t0, [t0, SN("ttot")])                 # tiles the block's last row can see
        self.emit("s_add_u32", t0, [t0, I(3)])
        self.emit("s_and_b32", s("ntb"), [t0, I(0xFFFFFFFC)])
        self.emit("s_sub_u32", s("ntm2"), [s("ntb"), I(2)])
        self.emit("s_sub_u32", s("ntm3"), [s("ntb"), I(3)])
        self.emit("s_add_u32", t2, [s("row0"), SN("wave64")])        # first row of the wave
        self.emit("s_add_u32", t3, [t2, SN("coff")])
        self.emit("s_min_u32", t0, [t3, SN("cm1")])
        self.emit("s_add_u32", t0, [t0, I(1)])
        self.emit("s_lshr_b32", s("maskb"), [t0, I(6)])              # first tile with a key the wave's first row may not see (or >= C)
        for rb in range(2):
            if rb:
                self.emit("s_add_u32", t3, [t3, I(32)])
            lim = VN("lim%d" % rb)
            self.emit("v_add_u32", lim, [t3, VN("qlane")])
            self.emit("v_min_u32", lim, [SN("cm1"), lim])
            self.emit("v_sub_u32", lim, [lim, VN("hi4")])

    def block_init(self):
        cfg = self.cfg
        for rb in range(2):
            self.emit("v_mov_b32", V(T_LB + rb), [I(0)])
            self.emit("v_mov_b32", V(T_CORR + rb), [F(1.0)])
            self.emit("v_mov_b32", VN("l%d" % rb), [I(0)])
            self.emit("v_mov_b32", VN("m%d" % rb), [F(0.0) if cfg.fold else F(-3.402823466e+38)])
        if cfg.fold:
            for r in range(32):
                self.emit("v_mov_b32", V(CM_BASE + r), [I(0)])
        self.emit("s_mov_b32", SN("pend"), [I(0)])
        self.emit("s_mov_b32", SN("j"), [I(0)])

    def build(self):
        cfg = self.cfg
        self.outofline = []
        blk_lbl, loop, end_lbl, fin, nonext = (self.newlabel(x) for x in ("BLOCK", "LOOP", "END", "FIN", "NONEXT"))
        # ---- once per workgroup
        if cfg.pad:
            self.emit("s_nop", None, [I(0)], note="code placement pad")
        for ks in range(4):
            self.emit("v_xor_b32", V(T_KC + ks), [I(ks << 5), VN("kbase")])
        one = 0x3F803F80 if cfg.dtype == "bf16" else 0x3C003C00
        for r in range(4):
            self.emit("v_mov_b32", V(ONES + r), [I(one)])
        if not cfg.causal:
            self.emit("s_sub_u32", s("ntm2"), [SN("nt"), I(2)])
            self.emit("s_sub_u32", s("ntm3"), [SN("nt"), I(3)])
        # scalar parts of the LDS-DMA start offsets: K piece i begins at row 16 wave + 8 i, V piece i at key 16 i (+ the lane part)
        self.emit("s_lshr_b32", s("t0"), [SN("kinc"), I(6)])                 # 2 ld(K)
        self.emit("s_lshr_b32", s("t2"), [SN("wave64"), I(2)])               # 16 wave
        self.emit("s_mul_i32", s("kc0"), [s("t0"), s("t2")])
        self.emit("s_lshl_b32", s("kstep"), [s("t0"), I(3)])                 # eight rows of K
        self.emit("s_lshr_b32", s("vstep"), [SN("vinc"), I(2)])              # sixteen keys of V
        self.emit("s_lshl_b32", s("q8"), [SN("ldq2"), I(3)])
        for name, nrec in (("kres", "nreck"), ("vres", "nrecv")):
            if not cfg.causal:     # (causal streams: per block, switch_k / switch_v)
                self.emit("s_mov_b32", s(name, 1, 2), [SN(nrec)])
            self.emit("s_mov_b32", s(name, 1, 3), [I(DESC_FLAGS)])
        # ---- first block: Q', K(0), V(0), K(1), V(1), K(2) are requested here; later blocks find theirs requested by their predecessor
        self.emit("s_mov_b32", s("blk"), [I(0)])
        self.load_next()
        self.issue_q((S_BASE[0] + 0, S_BASE[0] + 1, S_BASE[0] + 2, S_BASE[0] + 3))
        self.switch_k()
        self.switch_v()
        self.issue_tile("k", 0)
        self.issue_tile("v", 0)
        self.issue_tile("k", 1)
        self.issue_tile("v", 1)
        self.issue_tile("k", 2)
        self.emit("s_waitcnt", None, [], vmcnt=0)
        # ================= block loop =================
        self.label(blk_lbl)
        self.block_head(nonext)
        # this wave's Q image and its pieces of K(0), V(0), K(1), V(1), K(2): everything older than the previous block's stores
        self.emit("s_waitcnt", None, [], vmcnt=NST)
        self.q_fragments()
        self.emit("s_barrier")
        self.block_init()
        for i in range(8):     # K(0): a block starts in ring image 0 (it walks a multiple of four tiles)
            self.k_read(i, 0)
        self.lds_flush()
        self.phase_a(0, mfma=True, softmax=False, zero_o=True, t=0)
        self.emit("s_waitcnt", None, [], vmcnt=NST)          # K(1) (older than the stores)
        self.emit("s_barrier")
        self.phase_b(0, mfma=False, softmax=True, t=0)
        self.emit("s_mov_b32", SN("j"), [I(1)])
        self.label(loop)
        for t in (1, 2, 3, 0):
            par = t & 1
            if t == 0:           # the only exit: behind tile nt - 1 = 3 (mod 4)
                self.emit("s_cmp_ge_i32", None, [SN("j"), SN("nt")])
                self.emit("s_cbranch_scc1", None, [], target=end_lbl)
            self.phase_a(par, mfma=True, softmax=True, zero_o=False, t=t)
            self.lds_flush()
            if "bar" not in cfg.abl:
                # K(j+1), V(j) were requested in phase B(j-2); only the four pieces of B(j-1) (K(j+2), V(j+1)) may still fly
                self.emit("s_waitcnt", None, [], vmcnt=4)
                self.emit("s_barrier")
            self.phase_b(par, mfma=True, softmax=True, t=t)
            self.emit("s_add_u32", SN("j"), [SN("j"), I(1)])
        self.emit("s_branch", None, [], target=loop)
        # tail: finish tile nt - 1 (its scores are in the odd score tile, its V in image 3)
        self.label(end_lbl)
        self.phase_a(0, mfma=False, softmax=True, zero_o=False, t=0)
        self.lds_flush()
        self.emit("s_nop", None, [I(1)], note="freshly packed P -> MFMA operand")
        self.phase_b(0, mfma=True, softmax=False, t=0)
        self.epilogue()
        self.emit("s_cmp_eq_u32", None, [s("hasnext"), I(1)])
        self.emit("s_cbranch_scc1", None, [], target=blk_lbl)
        self.emit("s_waitcnt", None, [], vmcnt=0)
        self.emit("s_branch", None, [], target=fin)
        self.emit_outofline()
        self.label(fin)
        return self.finish()


# ---------------------------------------------------------------- rendering
render_one = p4pgen.render_one


def render(instrs):
    return [render_one(i) for i in instrs]


VARIANTS = {
    # name: cfg            (X-macro columns: 16-bit type is f16, folds the softmax scale into Q, O in the 16-bit type, L in FP16)
    "BF16_FOLD_L16": Cfg6("bf16", 8, l16=1),                  # config 2: mixed-precision mode, fp32 O, FP16 L
    "BF16_FOLD_O16_L16": Cfg6("bf16", 8, o16=1, l16=1),
    "F16_FOLD_L16": Cfg6("f16", 8, l16=1),
    "F16_FOLD_O16_L16": Cfg6("f16", 8, o16=1, l16=1),
    "BF16_FOLD_L16_CAUSAL": Cfg6("bf16", 8, l16=1, causal=1),
    "BF16_FOLD_O16_L16_CAUSAL": Cfg6("bf16", 8, o16=1, l16=1, causal=1),
    "F16_FOLD_L16_CAUSAL": Cfg6("f16", 8, l16=1, causal=1),
    "F16_FOLD_O16_L16_CAUSAL": Cfg6("f16", 8, o16=1, l16=1, causal=1),
    "BF16_EXACT_CAUSAL": Cfg6("bf16", 8, fold=0, xb=8, causal=1),
    "BF16_EXACT_O16_CAUSAL": Cfg6("bf16", 8, fold=0, xb=8, o16=1, causal=1),
    "F16_EXACT_CAUSAL": Cfg6("f16", 8, fold=0, xb=8, causal=1),
    "F16_EXACT_O16_CAUSAL": Cfg6("f16", 8, fold=0, xb=8, o16=1, causal=1),
    "BF16_FOLD_SPLIT": Cfg6("bf16", 8, split=1),               # column-parallel pieces: un-normalised O and (m, l) into the workspace
    "F16_FOLD_SPLIT": Cfg6("f16", 8, split=1),
    "BF16_EXACT_SPLIT": Cfg6("bf16", 8, fold=0, xb=8, split=1),
    "F16_EXACT_SPLIT": Cfg6("f16", 8, fold=0, xb=8, split=1),
    "BF16_EXACT": Cfg6("bf16", 8, fold=0, xb=8),               # lowPrecisionInputs only: scale in fp32, fp32 row sums, FP32 L
    "BF16_EXACT_O16": Cfg6("bf16", 8, fold=0, xb=8, o16=1),
    "F16_EXACT": Cfg6("f16", 8, fold=0, xb=8),
    "F16_EXACT_O16": Cfg6("f16", 8, fold=0, xb=8, o16=1),
    # developer streams (libmfa_hip_dev.so, tools/p6_ab.py): placements, and timing-only ablations (ABL_*, NOBAR: WRONG RESULTS)
    "BF16_EXACT_XB24": Cfg6("bf16", 8, fold=0, xb=24),
    "BF16_EXACT_XB40": Cfg6("bf16", 8, fold=0, xb=40),
    "BF16_FOLD_L16_VSUM": Cfg6("bf16", 8, l16=1, lsum=0),        # row sums as 64 VALU additions per tile (A/B of `lsum`)
    "BF16_FOLD_L16_XB32": Cfg6("bf16", 8, xb=32, l16=1),
    "BF16_FOLD_L16_XB44": Cfg6("bf16", 8, xb=44, l16=1),
    "BF16_FOLD_L16_XB60": Cfg6("bf16", 8, xb=60, l16=1),
    "BF16_FOLD_L16_V5": Cfg6("bf16", 8, l16=1, vlast=5),
    "BF16_FOLD_L16_V10": Cfg6("bf16", 8, l16=1, vlast=10),
    "BF16_FOLD_L16_R5": Cfg6("bf16", 8, xb=44, l16=1, vlast=15, kearly=0),   # the first GPU version's read placement
    "BF16_FOLD_L16_NOBAR": Cfg6("bf16", 8, l16=1, abl=("bar",)),
    "ABL_DMA": Cfg6("bf16", 8, l16=1, abl=("dma",)),
    "ABL_EXP": Cfg6("bf16", 8, l16=1, abl=("exp",)),
    "ABL_MAX": Cfg6("bf16", 8, l16=1, abl=("max",)),
    "ABL_PACK": Cfg6("bf16", 8, l16=1, abl=("pack",)),
    "ABL_LDS": Cfg6("bf16", 8, l16=1, abl=("lds",)),
    "ABL_ALL": Cfg6("bf16", 8, l16=1, abl=("dma", "exp", "max", "pack", "lds")),
    "ABL_ALL_NOBAR": Cfg6("bf16", 8, l16=1, abl=("dma", "exp", "max", "pack", "lds", "bar")),
}
PRODUCT_STREAMS = tuple(n for n in VARIANTS if n.split("_")[0] in ("BF16", "F16") and n.split("_")[1] in ("FOLD", "EXACT") and
                        all(t in ("BF16", "F16", "FOLD", "EXACT", "O16", "L16", "CAUSAL", "SPLIT") for t in n.split("_")))


def write_inc(path):
    lines = ["// GENERATED by tools/p6gen.py -- do not edit.  Persistent instruction streams of attn_fwd16_p6 (D <= 64; see the generator's",
             "// header for the phases, the register map and the LDS map).", "#pragma once", ""]
    lines.append("#define MFA_P6_OWNED_VGPRS " + ", ".join('"v%d"' % i for i in range(FIRST_OWNED_VGPR, 256)))
    lines.append("#define MFA_P6_OWNED_SGPRS " + ", ".join('"s%d"' % i for i in range(FIRST_CLOBBERED_SGPR, LAST_CLOBBERED_SGPR + 1)))
    for name, val in (("VRING", VRING_BASE), ("QIMG", QIMG), ("TABLE", TABLE), ("TABLE_ENTRIES", TABLE_ENTRIES), ("STAGE", STAGE),
                      ("LDS_BYTES", LDS_BYTES)):
        lines.append("#define MFA_P6_%s %d" % (name, val))
    lines.append("")
    lines.append("// X(name, 16-bit type is f16, folds the softmax scale into Q, O in the 16-bit type, L in FP16, causal, split)")
    lines.append("#define MFA_P6_PRODUCT_STREAM_LIST(X) \\")
    for name in PRODUCT_STREAMS:
        cfg = VARIANTS[name]
        lines.append("  X(%s, %d, %d, %d, %d, %d, %d) \\" % (name, cfg.dtype == "f16", cfg.fold, cfg.o16, cfg.l16, cfg.causal, cfg.split))
    lines.append("")
    lines.append("// streams that only the developer build (-DMFA_DEV_VARIANTS) instantiates")
    lines.append("#define MFA_P6_DEV_STREAM_LIST(X) \\")
    for name, cfg in VARIANTS.items():
        if name not in PRODUCT_STREAMS:
            lines.append("  X(%s, %d, %d, %d, %d, %d, %d) \\" % (name, cfg.dtype == "f16", cfg.fold, cfg.o16, cfg.l16, cfg.causal, cfg.split))
    lines.append("")
    lines.append("#ifdef MFA_DEV_VARIANTS")
    lines.append("#define MFA_P6_STREAM_LIST(X) MFA_P6_PRODUCT_STREAM_LIST(X) MFA_P6_DEV_STREAM_LIST(X)")
    lines.append("#else")
    lines.append("#define MFA_P6_STREAM_LIST(X) MFA_P6_PRODUCT_STREAM_LIST(X)")
    lines.append("#endif")
    lines.append("")
    for name, cfg in VARIANTS.items():
        ins = Stream6(cfg).build()
        txt = render(ins)
        if name not in PRODUCT_STREAMS:
            lines.append("#ifdef MFA_DEV_VARIANTS")
        n_mfma = sum(1 for i in ins if i.op.startswith("v_mfma"))
        lines.append("// %s: dtype=%s thr=%g fold=%d xb=%d o16=%d l16=%d lsum=%d causal=%d -- %d instructions, %d matrix instructions"
                     % (name, cfg.dtype, cfg.thr, cfg.fold, cfg.xb, cfg.o16, cfg.l16, cfg.lsum, cfg.causal, len(txt), n_mfma))
        lines.append("#define MFA_P6_STREAM_%s \\" % name)
        for t in txt:
            lines.append('  "%s\\n\\t" \\' % t)
        lines.append('  ""')
        if name not in PRODUCT_STREAMS:
            lines.append("#endif")
        lines.append("")
    with open(path, "w") as f:
        f.write("\n".join(lines))


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(here, "..", "metal_flash_attention_amd", "csrc", "attn_fwd16_p6_stream.inc")
    write_inc(out)
    print("wrote", os.path.normpath(out), "-", len(Stream6(VARIANTS["BF16_FOLD_L16"]).build()), "instructions in the default stream")
