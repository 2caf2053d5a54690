#!/usr/bin/env python3
"""Generator of the hand-placed instruction stream of attn_dkv16_p5 (csrc/attn_dkv16_p5.h): backwardKeyValue for the
head-dimension buckets 160, 192 and 256 with 16-bit Q/K/V/dO -- ROLE-SPLIT wave pairs, 64 keys per wave.

At D > 128 one wave cannot hold dV^T, dK^T and the K', V fragments of 64 keys (3 D registers), and with 32 keys per wave every
Q / dO fragment read from LDS feeds ONE matrix instruction (attn_dkv16_rs.h: ~2 LDS instructions per matrix instruction, 0.26-0.34
of the roof).  Here a workgroup is four waves = two pairs x 64 keys, one wave per SIMD with the whole register file:

    V-role wave:  S' = Q K'^T - L  ->  P = exp2(S')  ->  dV^T += dO^T P            holds K' fragments (D/2 registers) + dV^T (D)
    K-role wave:  dP' = dO V^T - D ->  dS' = P dP'   ->  dK^T += Q^T dS'           holds V  fragments              + dK^T

so every fragment read from LDS feeds the two key blocks of its wave (the ratio of attn_dkv16_p4) and nothing is recomputed:
4 products per (row block, key block) as the reference (+Source.swift:244-293).  The only coupling is P: the V-role wave leaves
its packed 16-bit P fragments (what its own dV product consumes; the reference holds P in 16-bit registers with low-precision
inputs, +Precisions.swift:149-215) in an LDS exchange buffer, the K-role wave of the pair picks them up one barrier later.

Iteration i of BOTH roles (one barrier per iteration, n + 1 iterations for n row blocks of 32):

    V-role:  phase A  S'(i)  = Q(i) K'^T - L(i)          | -
             phase B  dV^T  += dO^T(i-1) P(i-1)           | P(i) = exp2(S'(i)), packed into the other P buffer, written to the exchange
    K-role:  phase A  dP'(i) = dO(i) V^T - D(i)           | dS'(i-1) = P(i-1) dP'(i-1), packed in place (the other score set)
             phase B  dK^T  += Q^T(i-1) dS'(i-1)          | -

i.e. the VALU work of a block always runs beside matrix instructions that do not depend on it, a ring stage (block i) is read in
iterations i and i + 1 only (four stages with the LDS-DMA two blocks ahead), and the two register sets alternate by the parity of
i (the loop body is two iterations).  The first iteration has no phase B, the last no phase A.

Register map (fixed; v[0:27] left to hipcc):
    a[0:32 ndb)   the wave's accumulator (dV^T or dK^T)  (db, kb) -> 16 (2 db + kb)      lane = key, registers = head-dimension rows
    v[28:31]      L / D value as loaded, two temporaries, the mask constant
    v[32:35]      B operand of the extra k-step (-1.0 pattern, 0, 0, 0)        v[36:39]  its A operand (L or D pair, 0, 0, 0)
    v[40:55]      ring of four A-operand fragments read from LDS, fragment k in slot k % 4
    v[56:63]      K-role: eight received P words
    v[64:95]      V-role: two buffers of packed P fragments (parity, kb, u) -> 64 + 16 parity + 4 (2 kb + u); K-role: score set 1
    v[96:127]     score set 0 (V-role: S' / P of the block; K-role: dP' / dS' by parity), kb -> + 16 kb
    v[128:255]    cached B-operand fragments (K' or V)  (kb, ks) -> 128 + 4 (16 kb + ks)

LDS: ring of four stages {Q tile | dO tile}, each tile [DI/32][32 rows][32 elements] (DI = 192 for the buckets 160 and 192, 256
for 256) with the 16-byte chunks of a 64-byte row XOR-swizzled by (row >> 2) & 3 (attn_dkv16_rs.h), filled by LDS-DMA two blocks
ahead; behind it the exchange buffer [pair][parity][4 fragments][64 lanes x 16 bytes].

The instruction list is rendered as an asm template and executed by tools/dkv5sim.py (lane-exact model of tools/p4sim.py):
tests/test_dkv5_stream.py.

Usage: python tools/dkv5gen.py   (rewrites metal_flash_attention_amd/csrc/attn_dkv16_p5_stream.inc)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from p4gen import A, F, I, M0, SN, V, VCC, VN, Stream as _P4Stream, render  # noqa: E402

T_RAW, T_T0, T_T1, T_MASKV = 28, 29, 30, 31
ONES, PAIR, AF, PR, X1, SB, CF = 32, 36, 40, 56, 64, 96, 128
FIRST_OWNED_VGPR = 28
RING = 4
XPAR = 4096                 # bytes of one parity of a pair's exchange buffer: 4 fragments x 64 lanes x 16 bytes
WAIT_AHEAD = int(os.environ.get("MFA_GEN_WAIT_AHEAD", "3"))      # matrix instructions whose fragments one s_waitcnt may cover (0: exact waits)

# named operands of the asm statement (attn_dkv16_p5.h); order = operand order
INOUT_V = ["qoff0", "qoff1", "qoff2", "qoff3", "goff0", "goff1", "goff2", "goff3", "ldoff", "ra0", "ra1", "ta0", "ta1"]
TMP_S = ["j", "stg", "delta", "deltat", "wr", "t0", "t1", "pa", "pb", "pc", "pd", "plast"]   # p*: PROF streams
TMP_S64 = ["ptime"]
IN_V = ["onesw", "tk", "kvback", "xaddr"]
IN_S = ["qres", "gres", "ldres", "nsteps", "rscale", "rscale2", "qinc", "ginc", "ldinc", "wr0", "ringend", "maskuntil", "scale2x2", "role"]


class Cfg:
    def __init__(self, dtype="bf16", lprec="f32", dprec="f32", exact=0, gdtype=None, D=256, abl=(), prof=0):
        """dtype: type of Q, K, V (and of the packed dS'); gdtype: storage type of dO (the reference's own mix: FP16 Q, K, V with
        BF16 dO, +Precisions.swift:13-17): the two products that read dO -- dP' and dV^T -- run in it (V converted once by the
        kernel, P packed to it).  lprec / dprec: storage types of L and D.  exact: K stays as stored, the softmax scale is applied
        in fp32 (one packed multiply per two scores); otherwise K arrives pre-multiplied by log2(e)/sqrt(D), rounded to dtype."""
        assert D in (160, 192, 256)
        self.dtype, self.lprec, self.dprec, self.exact, self.D = dtype, lprec, dprec, exact, D
        self.gdtype = gdtype or dtype
        self.mix = self.gdtype != self.dtype
        self.nks, self.ndb = D // 16, D // 32
        self.DI = 192 if D <= 192 else 256          # geometry of the LDS images
        self.TI = 64 * self.DI                      # bytes of one operand tile (32 rows x DI x 2)
        self.STAGE, self.GIMG = 2 * self.TI, self.TI
        self.NPW = self.DI // 64                    # 1 KiB LDS-DMA pieces per wave and operand tile
        self.F = 2 * self.nks                       # fragments of a full iteration: nks row fragments, 2 ndb transposed ones
        self.NM = 2 + 2 * self.F                    # matrix instructions of a full iteration
        self.prof = prof      # developer streams: shader-clock sums per wave -- pa: behind the barrier .. end of phase A, pb: phase B up to
        self.abl = frozenset(abl)   # the seam, pc: the seam's waits + barrier (full iterations only; every stamp costs an lgkmcnt(0))


def af(k):
    return V(AF + 4 * (k % 4), 4)


def af_half(k, h):
    return V(AF + 4 * (k % 4) + 2 * h, 2)


def cf(kb, ks):
    return V(CF + 4 * (16 * kb + ks), 4)


def acc(db, kb):
    return A(16 * (2 * db + kb), 16)


def sset(q):
    return SB if q == 0 else X1


def p16(par, kb, u):          # V-role
    return V(X1 + 16 * par + 4 * (2 * kb + u), 4)


def ds16(q, kb, u):           # K-role: packed in place in score set q
    return V(sset(q) + 16 * kb + 4 * u, 4)


class Stream(_P4Stream):
    def __init__(self, cfg):
        _P4Stream.__init__(self, cfg)
        self.frag_rid = {}

    def lds_write(self, addr, data, offset):
        self.emit("ds_write_b128", None, [addr, data], offset=offset)
        self.lds_issued += 1

    def stamp0(self):
        """PROF streams: start the clock (no accumulation)"""
        if not self.cfg.prof:
            return
        self.emit("s_memtime", SN("ptime", 2))
        self.emit("s_waitcnt", None, [], lgkmcnt=0)
        self.lds_done = self.lds_issued
        self.emit("s_mov_b64", VCC, [SN("ptime", 2)])
        self.emit("s_mov_b32", SN("plast"), [("vcc_lo",)])

    def need(self, frags, ahead=()):
        """the reads of fragments `frags` have returned; when that takes a wait, the same wait also covers the fragments of the
        next matrix instructions that are already in flight: an s_waitcnt costs an issue slot of the only wave of the SIMD"""
        now = max(self.frag_rid[k] for k in frags)
        if now <= self.lds_done:
            return
        more = [self.frag_rid[k] for k in ahead if k in self.frag_rid and self.lds_done < self.frag_rid[k] <= self.lds_issued]
        self.lds_need(max([now] + more))

    # ---------------------------------------------------------------- LDS fragment reads
    def frag_read(self, role, k):
        """fragment k of an iteration: k < nks = row fragment ks of block i (V-role: Q, K-role: dO; addresses ra*), else the
        transposed fragment (u, db) of block i - 1 (V-role: dO^T, K-role: Q^T; addresses ta*)"""
        cfg = self.cfg
        if k < cfg.nks:
            img = 0 if role == 0 else cfg.GIMG
            self.frag_rid[k] = self.lds_read("ds_read_b128", af(k), VN("ra%d" % (k & 1)), img + (k >> 1) * 2048,
                                             note="%s rows ks%d" % ("Q" if role == 0 else "dO", k))
        else:
            u, db = divmod(k - cfg.nks, cfg.ndb)
            img = cfg.GIMG if role == 0 else 0
            off = img + db * 2048 + u * 1024
            self.lds_read("ds_read_b64_tr_b16", af_half(k, 0), VN("ta0"), off, note="%s^T u%d db%d" % ("dO" if role == 0 else "Q", u, db))
            self.frag_rid[k] = self.lds_read("ds_read_b64_tr_b16", af_half(k, 1), VN("ta1"), off)

    # ---------------------------------------------------------------- global -> LDS / registers
    def dma_piece(self, n):
        """piece n of this wave's share of the stage `wr` points at: n < NPW Q pieces, then the dO pieces"""
        npw = self.cfg.NPW
        name, res, base = (("qoff%d" % n, "qres", 0) if n < npw else ("goff%d" % (n - npw), "gres", self.cfg.GIMG))
        self.emit("s_add_u32", M0, [SN("wr"), I(base + (n % npw) * 1024)])
        self.emit("buffer_load_dwordx4_lds", None, [VN(name), SN(res, 4)])

    def dma_advance(self, n):
        npw = self.cfg.NPW
        name, inc = ("qoff%d" % n, "qinc") if n < npw else ("goff%d" % (n - npw), "ginc")
        self.emit("v_add_u32_e64", VN(name), [VN(name), SN(inc)], clamp=1)

    def wr_advance(self):
        self.emit("s_add_u32", SN("wr"), [SN("wr"), I(self.cfg.STAGE)])
        self.emit("s_cmp_ge_u32", None, [SN("wr"), SN("ringend")])
        self.emit("s_cselect_b32", SN("t1"), [I(RING * self.cfg.STAGE), I(0)])
        self.emit("s_sub_u32", SN("wr"), [SN("wr"), SN("t1")])

    def ld_load(self, role):
        """L (V-role) or D (K-role) of the rows of the block after the next, one value per lane (lane & 31 = row); rows past R read as zero"""
        prec = self.cfg.lprec if role == 0 else self.cfg.dprec
        self.emit("buffer_load_dword" if prec == "f32" else "buffer_load_ushort", V(T_RAW), [VN("ldoff"), SN("ldres", 4)])
        self.emit("v_add_u32", VN("ldoff"), [SN("ldinc"), VN("ldoff")])

    def ld_convert_ops(self, role):
        """closures turning the loaded value into the 16-bit pair the extra k-step consumes (hi + lo: exact to 16 bits of mantissa)"""
        cfg = self.cfg
        ops = []
        x, t = V(T_RAW), V(T_T0)
        prec = cfg.lprec if role == 0 else cfg.dprec
        ptype = cfg.dtype if role == 0 else cfg.gdtype           # the pair travels in the type of the product it joins
        mask = 0xFFFF0000 if ptype == "bf16" else 0xFFFFE000
        if prec == "f16":
            ops.append(lambda: self.emit("v_cvt_f32_f16", x, [x]))
        elif prec == "bf16":
            ops.append(lambda: self.emit("v_lshlrev_b32", x, [I(16), x]))
        if role == 1:       # the buffer holds D * scale (+Softmax.swift:472-503); dP' needs D itself
            ops.append(lambda: self.emit("v_mul_f32", x, [SN("rscale"), x]))
        elif cfg.exact:     # S'' = Q K^T - L / scale2
            ops.append(lambda: self.emit("v_mul_f32", x, [SN("rscale2"), x]))
        ops.append(lambda: self.emit("v_and_b32", t, [I(mask), x]))          # hi
        ops.append(lambda: self.emit("v_sub_f32", x, [x, t]))                # remainder
        ops.append(lambda: self.emit("v_and_b32", x, [I(mask), x]))          # lo
        ops.append(lambda: self.emit("v_cvt_pk_%s_f32" % ptype, V(PAIR), [t, x]))
        return ops

    def stage_delta(self):
        """deltat: what the transposing-read addresses (block i - 1) advance by = what the row-read addresses advanced by an
        iteration ago; delta: this iteration's advance of the row-read addresses (block i -> i + 1)"""
        self.emit("s_mov_b32", SN("deltat"), [SN("delta")])
        self.emit("s_add_u32", SN("stg"), [SN("stg"), I(1)])
        self.emit("s_and_b32", SN("stg"), [SN("stg"), I(RING - 1)])
        self.emit("s_cmp_eq_u32", None, [SN("stg"), I(0)])
        self.emit("s_cselect_b32", SN("t1"), [I(RING * self.cfg.STAGE), I(0)])
        self.emit("s_sub_u32", SN("delta"), [I(self.cfg.STAGE), SN("t1")])

    def addr_advance(self, names, by):
        for n in names:
            self.emit("v_add_u32", VN(n), [SN(by), VN(n)])

    # ---------------------------------------------------------------- V-role arithmetic on the fresh S' (score set 0)
    def v_ops(self, par):
        """scale (exact streams), exp2 and the 16-bit packs of block i into P buffer `par`, group (kb, u) by group in the order
        the dV products consume them; a group's packs trail its exps by a group (a transcendental result is not read back to back)"""
        cfg = self.cfg
        order = [(kb, u) for u in range(2) for kb in range(2)]
        seq = []

        def scale(kb, r):
            x = V(SB + 16 * kb + r, 2)
            return lambda: self.emit("v_pk_mul_f32", x, [x, SN("scale2x2", 2)])

        def exp(kb, r):
            x = V(SB + 16 * kb + r)
            return lambda: self.emit("v_exp_f32", x, [x])

        def pack(kb, u, w):
            r = 8 * u + 2 * w
            return lambda: self.emit("v_cvt_pk_%s_f32" % cfg.gdtype, V(X1 + 16 * par + 4 * (2 * kb + u) + w), [V(SB + 16 * kb + r), V(SB + 16 * kb + r + 1)])
        for n, (kb, u) in enumerate(order):
            if cfg.exact:
                seq += [scale(kb, r) for r in range(8 * u, 8 * u + 8, 2)]
            seq += [exp(kb, r) for r in range(8 * u, 8 * u + 8)]
            if n >= 1:
                pkb, pu = order[n - 1]
                seq += [pack(pkb, pu, w) for w in range(4)]
        seq += [pack(order[3][0], order[3][1], w) for w in range(4)]
        return seq

    def v_writes(self, par):
        """the packed P fragments of block i -> the pair's exchange buffer (parity `par`), lane-linear"""
        return [lambda kb=kb, u=u: self.lds_write(VN("xaddr"), p16(par, kb, u), par * XPAR + (2 * kb + u) * 1024)
                for kb in range(2) for u in range(2)]

    def mask_section(self, lbl, back):
        """causal blocks: key c of row r contributes iff c <= r + (C - R); tk = key - (C - R) - 4 hi - first row of the traversal"""
        self.label(lbl)
        self.emit("s_lshl_b32", SN("t0"), [SN("j"), I(5)])
        self.emit("v_subrev_u32", V(T_T0), [SN("t0"), VN("tk")])                     # key - coff - 4 hi - row0(block)
        for kb in range(2):
            if kb:
                self.emit("v_add_u32", V(T_T0), [I(32), V(T_T0)])
            for r in range(16):
                x = V(SB + 16 * kb + r)
                self.emit("v_cmp_lt_i32", VCC, [I((r & 3) + 8 * (r >> 2)), V(T_T0)])   # masked: row < key
                self.emit("v_cndmask_b32", x, [x, V(T_MASKV), VCC])
        self.emit("s_branch", None, [], target=back)

    # ---------------------------------------------------------------- K-role arithmetic on the previous block's dP' (score set q)
    def k_ops(self, q, par_prev):
        """dS' = P dP' on score set q with the partner's packed P of the previous block (exchange parity par_prev), packed in place.
        Returns a list of (kind, fn): kind "read" = LDS read (issue), "valu" = arithmetic"""
        cfg = self.cfg
        g16 = cfg.gdtype == "f16"
        out = []
        base = sset(q)
        rid = {}
        packs = {0: [], 1: []}
        for u in range(2):
            for kb in range(2):
                def rd(kb=kb, u=u):
                    rid[(kb, u)] = self.lds_read("ds_read_b128", V(PR + 4 * kb, 4), VN("xaddr"), par_prev * XPAR + (2 * kb + u) * 1024,
                                                 note="P kb%d u%d" % (kb, u))
                out.append(("read", rd))
            if u == 1:          # the packs of the first half cover the flight of the second half's P words
                out += packs[0]
            for kb in range(2):
                for w in range(4):
                    word = V(PR + 4 * kb + w)
                    r = 8 * u + 2 * w
                    lo, hi = V(base + 16 * kb + r), V(base + 16 * kb + r + 1)

                    def first(kb=kb, u=u, word=word, lo=lo):
                        self.lds_need(rid[(kb, u)])
                        if g16:       # FP16 half x FP32 in one instruction (v_cvt + v_mul: two)
                            self.emit("v_fma_mix_f32", lo, [word, lo, I(0)], op_sel=(0, 0, 0), op_sel_hi=(1, 0, 0))
                            return
                        self.emit("v_lshlrev_b32", V(T_T0), [I(16), word])
                        self.emit("v_mul_f32", lo, [V(T_T0), lo])

                    def second(word=word, hi=hi):
                        if g16:
                            self.emit("v_fma_mix_f32", hi, [word, hi, I(0)], op_sel=(1, 0, 0), op_sel_hi=(1, 0, 0))
                            return
                        self.emit("v_and_b32", V(T_T1), [I(0xFFFF0000), word])
                        self.emit("v_mul_f32", hi, [V(T_T1), hi])
                    out.append(("valu", first))
                    out.append(("valu", second))
            for kb in range(2):
                for w in range(4):
                    r = 8 * u + 2 * w
                    packs[u].append(("valu", lambda kb=kb, u=u, w=w, r=r: self.emit(
                        "v_cvt_pk_%s_f32" % cfg.dtype, V(base + 16 * kb + 4 * u + w), [V(base + 16 * kb + r), V(base + 16 * kb + r + 1)])))
        return out + packs[1]

    # ---------------------------------------------------------------- one iteration
    def iteration(self, role, par, phase_a, phase_b, alt_label):
        """role 0 / 1; par = parity of the iteration index; alt_label: where the seam branches when the next iteration is the last
        one (None: this IS the last one -- no seam)"""
        cfg = self.cfg
        nks, ndb, NPW = cfg.nks, cfg.ndb, cfg.NPW
        q_cur, q_prev = par, 1 - par
        # ---- the matrix instructions: (dst, a, b, c, fragment, type)
        mm = []
        ta, tb = (cfg.dtype, cfg.gdtype) if role == 0 else (cfg.gdtype, cfg.dtype)
        sa = SB if role == 0 else sset(q_cur)                  # where phase A accumulates
        if phase_a:
            for kb in range(2):
                mm.append((V(sa + 16 * kb, 16), V(PAIR, 4), V(ONES, 4), I(0), None, ta))
            for ks in range(nks):
                for kb in range(2):
                    mm.append((V(sa + 16 * kb, 16), af(ks), cf(kb, ks), V(sa + 16 * kb, 16), ks, ta))
        nA = len(mm)
        if phase_b:
            for u in range(2):
                for db in range(ndb):
                    for kb in range(2):
                        k = nks + u * ndb + db
                        b = p16(q_prev, kb, u) if role == 0 else ds16(q_prev, kb, u)
                        mm.append((acc(db, kb), af(k), b, acc(db, kb), k, tb))
        NM = len(mm)
        fill = [[] for _ in range(NM + 1)]          # fill[g]: after matrix instruction g; fill[NM]: not used

        def at(g, fn):
            fill[min(max(g, 0), NM - 1)].append(fn)

        frags = [m[4] for m in mm if m[4] is not None][::2]          # fragments in order of use
        first_g = {}
        for g, m in enumerate(mm):
            if m[4] is not None and m[4] not in first_g:
                first_g[m[4]] = g
        full = phase_a and phase_b
        # ---- fragment reads: four in the ring; fragment n + 4 (in order of use) takes the slot of fragment n once both of its
        # matrix instructions are issued.  A full iteration starts with fragments 0, 1 in flight (issued behind the previous seam)
        # and issues 2, 3 in its first two gaps; the others start from an empty queue and issue their first four up front.
        pre = []
        if full:
            at(0, lambda: self.frag_read(role, frags[2]))
            at(1, lambda: self.frag_read(role, frags[3]))
        else:
            pre = frags[:4]
        for n in range(len(frags) - 4):
            at(first_g[frags[n]] + 1, lambda n=n: self.frag_read(role, frags[n + 4]))
        # the seam sits four matrix instructions before the end of a full iteration (the first two fragments of the next one are
        # requested behind its barrier); iteration 0 has nothing to run behind it: everything first, then the seam
        seam_g = None if alt_label is None else (NM - 4 if full else NM - 1)
        if phase_a:
            # ---- LDS-DMA of block i + 2 (stage wr): one piece per even gap; the offsets advance in the following even gaps
            # (the K-role wave, whose phase A carries the dS' arithmetic -- 4.5 instructions per matrix-instruction gap with the
            # LDS-DMA work --, issues its pieces in phase B instead, which has nothing but fragment reads)
            if "dma" not in cfg.abl and not (role == 1 and phase_b):
                for n in range(2 * NPW):
                    at(2 + 2 * n, lambda n=n: self.dma_piece(n))
                    at(2 + 4 * NPW + 2 * n, lambda n=n: self.dma_advance(n))
                at(2 + 8 * NPW, lambda: self.wr_advance())
            elif "dma" not in cfg.abl:
                ops = [lambda n=n: self.dma_piece(n) for n in range(2 * NPW)] + [lambda n=n: self.dma_advance(n) for n in range(2 * NPW)] + \
                      [lambda: self.wr_advance()]
                lo, hi = nA + 1, NM - 5
                for n, fn in enumerate(ops):
                    at(lo + (n * (hi - lo + 1)) // len(ops), fn)
            at(0, lambda: self.stage_delta())
            # row-read addresses move to the next stage once the last row fragment is requested
            at(first_g[frags[nks - 5]] + 2 if nks >= 5 else 2, lambda: self.addr_advance(["ra0", "ra1"], "delta"))
        # ---- arithmetic
        if role == 0 and phase_a:
            ops = self.v_ops(q_cur)
            mask_lbl, mask_back = self.newlabel("MASK"), self.newlabel("MASKBACK")

            def mask_branch():
                self.emit("s_cmp_lt_i32", None, [SN("j"), SN("maskuntil")])
                self.emit("s_cbranch_scc1", None, [], target=mask_lbl)
                self.label(mask_back)
            self.outofline.append((mask_lbl, mask_back))
            writes = self.v_writes(q_cur)
            if phase_b:
                g0, g1 = nA + 1, seam_g - 2          # exps / packs in [g0 + 1, g1], the exchange writes in g1 + 1
                at(g0, mask_branch)
                span = g1 - g0
                for n, fn in enumerate(ops):
                    at(g0 + 1 + (n * span) // len(ops), fn)
                for fn in writes:
                    at(g1 + 1, fn)
            else:                                   # iteration 0: nothing to hide behind; before the seam
                for fn in [mask_branch] + ops + writes:
                    at(NM - 1, fn)
        if role == 1 and phase_b:
            ops = self.k_ops(q_prev, q_prev)
            if phase_a:      # the first two P fragments are requested at once, the arithmetic starts three gaps later
                for n, (kind, fn) in enumerate(ops[:2]):
                    at(0, fn)
                g0, g1 = 3, nA - 2
                span = g1 - g0 + 1
                for n, (kind, fn) in enumerate(ops[2:]):
                    at(g0 + (n * span) // len(ops[2:]), fn)
            else:
                pre_ops = ops
        # ---- the seam to the next iteration
        if alt_label is not None:
            def seam():
                # (address arithmetic in front of the barrier: the last transposing read of the iteration is issued)
                self.addr_advance(["ta0", "ta1"], "deltat")
                self.emit("s_add_u32", SN("j"), [SN("j"), I(1)])
                if full:
                    self.stamp("pb")
                self.emit("s_waitcnt", None, [], vmcnt=2 * NPW if "dma" not in cfg.abl else 0, lgkmcnt=0)
                self.lds_done = self.lds_issued
                self.emit("s_barrier")
                if full:
                    self.stamp("pc")
                self.emit("s_cmp_ge_i32", None, [SN("j"), SN("nsteps")])
                self.emit("s_cbranch_scc1", None, [], target=alt_label)
                self.alt_capture = (self.lds_issued, self.lds_done, dict(self.frag_rid))
            at(seam_g, seam)
            if full:
                at(nA - 1, lambda: self.stamp("pa"))
            conv = self.ld_convert_ops(role)
            for n, fn in enumerate(conv):
                at(seam_g + 1 + (n * 2) // len(conv), fn)
            at(NM - 1, lambda: self.ld_load(role))
            at(seam_g + 1, lambda: self.frag_read(role, 0))
            at(NM - 1, lambda: self.frag_read(role, 1))
        # ---- emit
        for k in pre:
            self.frag_read(role, k)
        if role == 1 and phase_b and not phase_a:       # last iteration of a K-role wave: its arithmetic has nothing to hide behind
            for kind, fn in pre_ops:
                fn()
        alt_state = None
        for g, (d, a_, b_, c_, fr, typ) in enumerate(mm):
            if fr is not None:
                self.need([fr], [m[4] for m in mm[g + 1:g + 1 + WAIT_AHEAD] if m[4] is not None])
            self.emit("v_mfma_f32_32x32x16_" + typ, d, [a_, b_, c_])
            for fn in fill[g]:
                fn()
            if g == seam_g:
                alt_state = self.alt_capture + (mm[g + 1:],)
        return alt_state

    def alt_tail(self, label, state, target):
        """the matrix instructions behind the seam when the next iteration is the last one: no prefetch for it"""
        issued, done, rid, rest = state
        save = (self.lds_issued, self.lds_done, self.frag_rid)
        self.lds_issued, self.lds_done, self.frag_rid = issued, done, rid
        self.label(label)
        for d, a_, b_, c_, fr, typ in rest:
            if fr is not None:
                self.lds_need(self.frag_rid[fr])
            self.emit("v_mfma_f32_32x32x16_" + typ, d, [a_, b_, c_])
        self.lds_flush()
        self.emit("s_branch", None, [], target=target)
        self.lds_issued, self.lds_done, self.frag_rid = save

    def enter_full(self):
        """bookkeeping at the head of a full iteration: exactly the reads of fragments 0 and 1 are in flight"""
        self.lds_done = self.lds_issued - 2
        self.frag_rid = {0: self.lds_issued - 1, 1: self.lds_issued}

    # ---------------------------------------------------------------- whole traversal of one role
    def role_stream(self, role):
        cfg = self.cfg
        nks, NPW = cfg.nks, cfg.NPW
        tag = "V" if role == 0 else "K"
        # ---- cached fragments: attn_dkv16_p5.h (hipcc: bounds, zero fill, the K prescale, the V conversion of mix streams) parks
        # them in LDS, lane-linear, (nks kb + ks) x 1 KiB per wave at `kvback`; they move to their fixed registers here
        self.emit("s_waitcnt", None, [], lgkmcnt=0)
        for kb in range(2):
            for ks in range(nks):
                self.lds_read("ds_read_b128", cf(kb, ks), VN("kvback"), (nks * kb + ks) * 1024)
        for r in range(32 * cfg.ndb):
            self.emit("v_accvgpr_write_b32", A(r), [I(0)])
        self.emit("v_mov_b32", V(ONES), [VN("onesw")])
        for r in (ONES + 1, ONES + 2, ONES + 3, PAIR, PAIR + 1, PAIR + 2, PAIR + 3):
            self.emit("v_mov_b32", V(r), [I(0)])
        self.emit("v_mov_b32", V(T_MASKV), [F(-(0.875 / 1.44269504089) * 3.402823466e+38)])   # +Softmax.swift:242-243
        self.lds_flush()
        self.emit("s_barrier")                                # every wave has its fragments: the ring may be written
        # ---- stages 0 and 1, the per-row term of block 0
        self.emit("s_mov_b32", SN("wr"), [SN("wr0")])
        for t in range(2):
            for n in range(2 * NPW):
                self.dma_piece(n)
            for n in range(2 * NPW):
                self.dma_advance(n)
            self.wr_advance()
            if t == 0:
                self.ld_load(role)
        self.emit("s_waitcnt", None, [], vmcnt=2 * NPW)       # stage 0 and the term of block 0 have landed
        for fn in self.ld_convert_ops(role):
            fn()
        self.emit("s_barrier")
        self.ld_load(role)
        self.emit("s_mov_b32", SN("j"), [I(0)])
        self.emit("s_mov_b32", SN("stg"), [I(0)])
        self.emit("s_mov_b32", SN("delta"), [I(cfg.STAGE)])
        for d in ("pa", "pb", "pc", "pd"):
            self.emit("s_mov_b32", SN(d), [I(0)])
        lbl = {n: self.newlabel(tag + n) for n in ("LOOP", "ALT0", "ALT1", "ALTP", "EPI0", "EPI1", "END")}
        # ---- iteration 0 (parity 0): phase A only
        st = self.iteration(role, 0, True, False, lbl["ALTP"])
        self.stamp0()
        self.enter_full()
        # ---- the loop: iterations 1, 3, ... (parity 1) and 2, 4, ... (parity 0)
        self.label(lbl["LOOP"])
        st1 = self.iteration(role, 1, True, True, lbl["ALT1"])
        self.enter_full()
        st0 = self.iteration(role, 0, True, True, lbl["ALT0"])
        self.enter_full()
        self.emit("s_branch", None, [], target=lbl["LOOP"])
        # ---- the last iteration (phase B only), by parity; entered from the seam of the iteration before it
        self.alt_tail(lbl["ALTP"], st, lbl["EPI1"])
        self.alt_tail(lbl["ALT1"], st1, lbl["EPI0"])
        self.alt_tail(lbl["ALT0"], st0, lbl["EPI1"])
        for par in (0, 1):
            self.label(lbl["EPI%d" % par])
            self.lds_done = self.lds_issued
            self.iteration(role, par, False, True, None)
            self.lds_flush()
            self.emit("s_branch", None, [], target=lbl["END"])
        for mask_lbl, mask_back in self.outofline:
            self.mask_section(mask_lbl, mask_back)
        self.outofline = []
        self.label(lbl["END"])
        self.emit("s_waitcnt", None, [], vmcnt=0, lgkmcnt=0)       # run-ahead DMA (zeros past the end), the last term loads

    def build(self):
        self.outofline = []
        krole, fin = self.newlabel("KROLE"), self.newlabel("FIN")
        self.emit("s_cmp_eq_u32", None, [SN("role"), I(1)])
        self.emit("s_cbranch_scc1", None, [], target=krole)
        self.role_stream(0)
        self.emit("s_branch", None, [], target=fin)
        self.label(krole)
        self.lds_issued = self.lds_done = 0
        self.role_stream(1)
        self.label(fin)
        return self.ins


# ---------------------------------------------------------------- rendering
def write_inc(path):
    lines = ["// GENERATED by tools/dkv5gen.py -- do not edit.  Instruction streams of attn_dkv16_p5 (see the generator's",
             "// header for the register map and the iteration table).", "#pragma once", ""]
    lines.append("#define MFA_DKV5_OWNED_VGPRS " + ", ".join('"v%d"' % i for i in range(FIRST_OWNED_VGPR, 256)))
    lines.append("")
    lines.append("// X(name, applies the softmax scale in fp32, dO is BF16 next to FP16 Q / K / V, head-dimension bucket, stamps the shader clock)")
    lines.append("#define MFA_DKV5_STREAM_LIST(X) \\")
    for name, cfg in VARIANTS.items():
        lines.append("  X(%s, %d, %d, %d, %d) \\" % (name, cfg.exact, int(cfg.mix), cfg.D, cfg.prof))
    lines.append("")
    lines.append("")
    for name, cfg in VARIANTS.items():
        ins = Stream(cfg).build()
        txt = render(ins)
        lines.append("// %s: dtype=%s dO=%s L=%s D=%s bucket %d -- %d instructions, %d matrix instructions per full iteration"
                     % (name, cfg.dtype, cfg.gdtype, cfg.lprec, cfg.dprec, cfg.D, len(txt), cfg.NM))
        lines.append("#define MFA_DKV5_STREAM_%s \\" % name)
        for t in txt:
            lines.append('  "%s\\n\\t" \\' % t)
        lines.append('  ""')
        lines.append("")
    with open(path, "w") as f:
        f.write("\n".join(lines))


def _variants():
    out = {}
    for D in (160, 192, 256):
        for dt in ("bf16", "f16"):
            T = dt.upper()
            out["D%d_%s_MIXED" % (D, T)] = Cfg(dt, "f16", "bf16", D=D)             # the reference's mixed-precision mode: FP16 L, BF16 D
            out["D%d_%s_F32" % (D, T)] = Cfg(dt, "f32", "f32", exact=1, D=D)       # lowPrecisionInputs alone: FP32 L, D, scale in fp32
        out["D%d_F16_DOBF16_MIXED" % D] = Cfg("f16", "f16", "bf16", gdtype="bf16", D=D)   # the reference's default low-precision mix
        out["D%d_F16_DOBF16_F32" % D] = Cfg("f16", "f32", "f32", exact=1, gdtype="bf16", D=D)
    out["D256_BF16_MIXED_PROF"] = Cfg("bf16", "f16", "bf16", D=256, prof=1)       # developer library only (tools/bwd5_prof.py)
    out["D160_BF16_MIXED_PROF"] = Cfg("bf16", "f16", "bf16", D=160, prof=1)
    return out


VARIANTS = _variants()

if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(here, "..", "metal_flash_attention_amd", "csrc", "attn_dkv16_p5_stream.inc")
    write_inc(out)
    ins = Stream(VARIANTS["D256_BF16_MIXED"]).build()
    print("wrote", os.path.normpath(out), "-", len(ins), "instructions in the D = 256 BF16 stream")
