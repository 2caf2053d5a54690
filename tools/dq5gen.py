#!/usr/bin/env python3
"""Generator of the hand-placed instruction stream of attn_dq16_p5 (csrc/attn_dq16_p5.h): backwardQuery for the head-dimension
buckets 160, 192 and 256 with 16-bit Q/K/V/dO -- ROLE-SPLIT wave pairs, 64 rows per pair.

At D > 128 one wave cannot hold dQ^T and the Q', dO fragments of 64 rows (2 D + D registers ... 512 at D = 256 before any score
tile), and with 32 rows per wave every K / V fragment read from LDS feeds ONE matrix instruction (attn_bwd16.h attn_dq16: 0.35-0.37
of the roof, LDS-bound).  Here a workgroup is four waves = two pairs x 64 rows, one wave per SIMD with the whole register file:

    S-role wave:  S'^T = K Q'^T - L  ->  P = exp2(S'^T)                       holds Q' fragments (D/2 registers) + dQ^T[first half of D]
    P-role wave:  dP'^T = V dO^T - D ->  dS' = P dP'                          holds dO fragments                 + dQ^T[second half]
    both:         dQ^T[own half of the head dimension] += K^T dS'^T

Every K / V fragment feeds the two row blocks of the wave, nothing is recomputed: 3 products per (row block, key block) as the
reference (+Source.swift:202-242).  Two exchanges through LDS, each picked up one barrier later: the S-role wave's packed 16-bit P
fragments -> its partner; the partner's packed dS' fragments -> back (the B operand of both waves' dQ update).

Keys advance in blocks of 32.  Iteration i (one barrier per iteration, n + 2 iterations for n key blocks):

    S-role:  phase A  S'^T(i)   row block 0, then row block 1 (the K row fragments wait in a buffer of their own), the exp2 / pack
                                work of row block 0 beside the matrix instructions of row block 1, of row block 1 beside phase B
             phase B  dQ^T_s += K^T(i-2) dS'^T(i-2)
    P-role:  phase A  dP'^T(i)                                          | dS'(i-1) = P(i-1) dP'(i-1), packed in place (other score set)
             phase B  dQ^T_p += K^T(i-1) dS'^T(i-1)

K block i is read in iterations i .. i + 2 (ring of five stages with the LDS-DMA two blocks ahead), V block i in iteration i only
(ring of three).  The rows of a wave never change, so -L and -D are register blocks that START the accumulation
(attn_dq16_p4.h): exactly the 6 N^2 D flops of the algorithm.

Register map (fixed; v[0:63] left to hipcc):
    a[0:127]      the wave's half of dQ^T  (rb, local d block) -> 16 (4 rb + dbl)         lane = row, registers = head-dimension rows
    a[128:255]    cached B-operand fragments (Q' or dO)  (rb, ks) -> 128 + 4 (16 rb + ks)
    v[64:95]      start blocks (-L or -D of the lane's row), rb -> + 16 rb
    v[96:127]     score set 0 (S-role: S'^T / P; P-role: dP' / dS' by parity), rb -> + 16 rb
    v[128:159]    P-role: score set 1.   S-role: packed P fragments (rb, u) -> 128 + 4 (2 rb + u); received dS' fragments -> 144 + ...
    v[160:223]    S-role: K row fragments of the block, ks -> 160 + 4 ks.   P-role: v[160:175] received P fragments
    v[224:239]    ring of four A-operand fragments read from LDS
    v[240:244]    mask limits (2), mask constant, two temporaries

LDS: K ring (five tiles) | V ring (three tiles) | P exchange [pair][parity][4 x 1 KiB] | dS' exchange [pair][parity][4 x 1 KiB];
a tile is [DI/32][32 keys][32 elements] (DI = 192 for the buckets 160 and 192, 256 for 256), 16-byte chunks XOR-swizzled by
(key >> 2) & 3 (attn_dkv16_rs.h): 160 KiB at DI = 256, exactly the LDS of a compute unit.

The instruction list is rendered as an asm template and executed by tools/dq5sim.py (lane-exact model of tools/p4sim.py):
tests/test_dq5_stream.py.

Usage: python tools/dq5gen.py   (csrc/attn_dq16_p5_stream.inc is written by tools/gen_streams.py)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from p4gen import A, F, I, M0, SN, V, VCC, VN, Stream as _P4Stream, render  # noqa: E402

CT, S0, S1, P16, DSR, PRB, KF, AF = 64, 96, 128, 128, 144, 160, 160, 224
T_TL, T_MASKV, T_T0, T_T1 = 240, 242, 243, 244
FIRST_OWNED_VGPR = 64
KRING, VRING = 5, 3
XPAR = 4096
WAIT_AHEAD = int(os.environ.get("MFA_GEN_WAIT_AHEAD", "2"))      # matrix instructions whose fragments one s_waitcnt may cover (0: exact waits)

INOUT_V = ["koff0", "koff1", "koff2", "koff3", "voff0", "voff1", "voff2", "voff3", "ra0", "ra1", "ta0", "ta1"]
TMP_S = ["j", "sk", "sv", "kd0", "kd1", "kd2", "vd0", "wrk", "wrv", "t0", "t1", "pa", "pb", "pc", "pd", "plast"]   # p*: PROF streams
TMP_S64 = ["ptime"]
IN_V = ["negt0", "negt1", "lim0", "lim1", "xp", "xs"]
IN_S = ["kres", "vres", "nt", "kinc", "vinc", "wrk0", "wrv0", "kend", "vend", "maskfrom", "scale2x2", "role"]


class Cfg:
    def __init__(self, dtype="bf16", exact=0, D=256, abl=(), prof=0, xwe=0, pv0=4, wa=None):
        """exact: Q stays as stored, -L arrives divided by log2(e)/sqrt(D) and the scale is applied in fp32 before the exp2 (one
        packed multiply per two scores); otherwise Q arrives pre-multiplied, rounded to the 16-bit type.  dO arrives in dtype (the
        kernel converts BF16 gradients next to FP16 operands while it loads the fragments, as attn_dq16_p4.h)."""
        assert D in (160, 192, 256)
        self.dtype, self.exact, self.D = dtype, exact, D
        self.nks, self.ndb = D // 16, D // 32
        self.ndbs = (self.ndb + 1) // 2               # head-dimension blocks of dQ^T the S-role wave owns; the P-role wave the rest
        self.DI = 192 if D <= 192 else 256
        self.TI = 64 * self.DI                        # bytes of one tile (32 keys x DI x 2)
        self.NPW = self.DI // 64                      # 1 KiB LDS-DMA pieces per wave and operand tile
        self.VR0 = KRING * self.TI                    # offsets of the V ring and of the exchanges from the K ring
        self.XP0 = (KRING + VRING) * self.TI
        self.XS0 = self.XP0 + 4 * XPAR
        self.LDS = self.XS0 + 4 * XPAR
        self.wa = WAIT_AHEAD if wa is None else wa   # matrix instructions whose fragments one s_waitcnt may cover
        self.pv0 = pv0        # P-role: first gap of the dS' arithmetic (the partner's P fragments are requested in gaps 0, 1)
        self.xwe = xwe        # S-role: the second row block's exp2 / pack work ends, and its two exchange writes go out, xwe gaps earlier
        self.prof = prof      # developer streams: shader-clock sums per wave -- pa: behind the barrier .. end of phase A, pb: phase B up to
        self.abl = frozenset(abl)   # the seam, pc: the seam's waits + barrier (full iterations only; every stamp costs an lgkmcnt(0))

    def share(self, role):
        return self.ndbs if role == 0 else self.ndb - self.ndbs


def acc(rb, dbl):
    return A(16 * (4 * rb + dbl), 16)


def cfrag(rb, ks):
    return A(128 + 4 * (16 * rb + ks), 4)


def blk(base, rb):
    return V(base + 16 * rb, 16)


def af(k):
    return V(AF + 4 * (k % 4), 4)


def af_half(k, h):
    return V(AF + 4 * (k % 4) + 2 * h, 2)


def kf(ks):
    return V(KF + 4 * ks, 4)


def sset(q):
    return S0 if q == 0 else S1


class Stream(_P4Stream):
    def __init__(self, cfg):
        _P4Stream.__init__(self, cfg)
        self.rid = {}
        self.in_loop = False      # (ablations only touch the full iterations)

    # timing-only ablations (developer streams; results are garbage): classes of instructions that are simply not emitted.  The
    # wait bookkeeping stays as generated -- an s_waitcnt for a read that was never issued returns at once.
    ABL = {"rowrd": lambda op, note: op == "ds_read_b128" and note.startswith("rows"),
           "trrd": lambda op, note: op == "ds_read_b64_tr_b16",
           "xr": lambda op, note: op == "ds_read_b128" and not note.startswith("rows"),
           "xw": lambda op, note: op == "ds_write_b128",
           "exp": lambda op, note: op == "v_exp_f32",
           "valu": lambda op, note: op in ("v_exp_f32", "v_pk_mul_f32", "v_mul_f32", "v_lshlrev_b32", "v_and_b32", "v_fma_mix_f32") or op.startswith("v_cvt_pk"),
           "bar": lambda op, note: op == "s_barrier",
           "wait": lambda op, note: op == "s_waitcnt"}

    def emit(self, op, d=None, s=(), note="", **mod):
        for a in self.cfg.abl:
            if a in self.ABL and self.in_loop and self.ABL[a](op, note):
                return
        _P4Stream.emit(self, op, d, s, note=note, **mod)

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

    def need(self, keys, ahead=()):
        """the reads behind `keys` have returned; when that takes a wait, one wait also covers the reads of the next matrix
        instructions that are already in flight (`ahead`): an s_waitcnt costs an issue slot of the only wave of the SIMD -- the
        PROF streams, whose stamps flush the queue twice per iteration, ran 3.5 % FASTER than one exact wait per fragment)"""
        now = max(self.rid[k] for k in keys)
        if now <= self.lds_done:
            return
        more = [self.rid[k] for k in ahead if k in self.rid and self.lds_done < self.rid[k] <= self.lds_issued]
        self.lds_need(max([now] + more))

    # ---------------------------------------------------------------- LDS reads
    def row_read(self, dst, ks, key):
        """row fragment ks of the tile the row-read addresses point at (S-role: K block i; P-role: V block i)"""
        self.rid[key] = self.lds_read("ds_read_b128", dst, VN("ra%d" % (ks & 1)), (ks >> 1) * 2048, note="rows ks%d" % ks)

    def tr_read(self, k, t, role):
        """transposed fragment t = (u, local d block) of the K tile the transposing-read addresses point at, into ring slot k % 4"""
        u, dbl = divmod(t, self.cfg.share(role))
        db = dbl if role == 0 else self.cfg.ndbs + dbl
        off = db * 2048 + u * 1024
        self.lds_read("ds_read_b64_tr_b16", af_half(k, 0), VN("ta0"), off, note="K^T u%d db%d" % (u, db))
        self.rid[("t", t)] = self.lds_read("ds_read_b64_tr_b16", af_half(k, 1), VN("ta1"), off)

    # ---------------------------------------------------------------- global -> LDS
    def dma_piece(self, n):
        npw = self.cfg.NPW
        name, res, wr = (("koff%d" % n, "kres", "wrk") if n < npw else ("voff%d" % (n - npw), "vres", "wrv"))
        self.emit("s_add_u32", M0, [SN(wr), I((n % npw) * 1024)])
        self.emit("buffer_load_dwordx4_lds", None, [VN(name), SN(res, 4)])

    def dma_advance(self, n):
        npw = self.cfg.NPW
        name, inc = ("koff%d" % n, "kinc") if n < npw else ("voff%d" % (n - npw), "vinc")
        self.emit("v_add_u32_e64", VN(name), [VN(name), SN(inc)], clamp=1)

    def wr_advance(self):
        for wr, end, stages in (("wrk", "kend", KRING), ("wrv", "vend", VRING)):
            self.emit("s_add_u32", SN(wr), [SN(wr), I(self.cfg.TI)])
            self.emit("s_cmp_ge_u32", None, [SN(wr), SN(end)])
            self.emit("s_cselect_b32", SN("t1"), [I(stages * self.cfg.TI), I(0)])
            self.emit("s_sub_u32", SN(wr), [SN(wr), SN("t1")])

    def deltas(self):
        """kd0 / kd1 / kd2: what a K-ring address moves by from block i / i - 1 / i - 2 to the next; vd0: the V ring's"""
        ti = self.cfg.TI
        self.emit("s_mov_b32", SN("kd2"), [SN("kd1")])
        self.emit("s_mov_b32", SN("kd1"), [SN("kd0")])
        for cnt, dst, stages in (("sk", "kd0", KRING), ("sv", "vd0", VRING)):
            self.emit("s_add_u32", SN(cnt), [SN(cnt), I(1)])
            self.emit("s_cmp_eq_u32", None, [SN(cnt), I(stages)])
            self.emit("s_cselect_b32", SN("t1"), [I(stages * ti), I(0)])
            self.emit("s_cselect_b32", SN(cnt), [I(0), SN(cnt)])
            self.emit("s_sub_u32", SN(dst), [I(ti), SN("t1")])

    def seam(self, role, pieces, target=None, prof=False):
        """end of an iteration: own LDS-DMA pieces of the next block have landed, the exchange writes are out; barrier; the read
        addresses move on; then the exit test (cond = 'ge': branch to `target` when the next iteration index >= n)"""
        # (the address arithmetic in front of the barrier: every read of this iteration is issued, and a wave that arrives early
        # does it while it would wait)
        self.deltas()
        for n in ("ra0", "ra1"):
            self.emit("v_add_u32", VN(n), [SN("kd0" if role == 0 else "vd0"), VN(n)])
        for n in ("ta0", "ta1"):
            self.emit("v_add_u32", VN(n), [SN("kd2" if role == 0 else "kd1"), VN(n)])
        self.emit("s_add_u32", SN("j"), [SN("j"), I(1)])
        if prof:
            self.stamp("pb")
        self.emit("s_waitcnt", None, [], vmcnt=pieces, lgkmcnt=0)
        self.lds_done = self.lds_issued
        self.emit("s_barrier")
        if prof:
            self.stamp("pc")
        if target is not None:
            self.emit("s_cmp_ge_i32", None, [SN("j"), SN("nt")])
            self.emit("s_cbranch_scc1", None, [], target=target)

    def dma_fill(self, at, NM, window=None):
        """the wave's LDS-DMA pieces of block i + 2, their offset advances and the write pointers' advance.  window = None: one
        piece per odd gap from the start of the iteration (the role whose phase A is light); (lo, hi): dealt out over those gaps
        (the P-role wave: behind its phase-A arithmetic -- phase A carried 4.9 instructions per matrix-instruction gap with them,
        profiles/r04_call6)"""
        npw = self.cfg.NPW
        if "dma" in self.cfg.abl:
            return
        if window is None:
            for n in range(2 * npw):
                at(1 + 2 * n, lambda n=n: self.dma_piece(n))
                at(1 + 4 * npw + 2 * n, lambda n=n: self.dma_advance(n))
            at(2 + 8 * npw, lambda: self.wr_advance())
            return
        lo, hi = window
        ops = [lambda n=n: self.dma_piece(n) for n in range(2 * npw)] + [lambda n=n: self.dma_advance(n) for n in range(2 * npw)] + \
              [lambda: self.wr_advance()]
        for n, fn in enumerate(ops):
            at(lo + (n * (hi - lo + 1)) // len(ops), fn)

    # ---------------------------------------------------------------- S-role
    def s_valu(self, rb):
        """scale (exact streams), exp2 and 16-bit packs of row block rb of the fresh S'^T; the packs of half u trail its exps"""
        cfg = self.cfg
        seq = []
        for u in range(2):
            if cfg.exact:
                seq += [lambda r=r: self.emit("v_pk_mul_f32", V(S0 + 16 * rb + r, 2), [V(S0 + 16 * rb + r, 2), SN("scale2x2", 2)])
                        for r in range(8 * u, 8 * u + 8, 2)]
            seq += [lambda r=r: self.emit("v_exp_f32", V(S0 + 16 * rb + r), [V(S0 + 16 * rb + r)]) for r in range(8 * u, 8 * u + 8)]
            if u == 1:
                seq += [lambda w=w: self.emit("v_cvt_pk_%s_f32" % cfg.dtype, V(P16 + 4 * (2 * rb) + w),
                                              [V(S0 + 16 * rb + 2 * w), V(S0 + 16 * rb + 2 * w + 1)]) for w in range(4)]
        seq += [lambda w=w: self.emit("v_cvt_pk_%s_f32" % cfg.dtype, V(P16 + 4 * (2 * rb + 1) + w),
                                      [V(S0 + 16 * rb + 8 + 2 * w), V(S0 + 16 * rb + 8 + 2 * w + 1)]) for w in range(4)]
        return seq

    def s_writes(self, rb, par):
        return [lambda u=u: self.lds_write(VN("xp"), V(P16 + 4 * (2 * rb + u), 4), par * XPAR + (2 * rb + u) * 1024) for u in range(2)]

    def mask_section(self, lbl, back, rb):
        """edge / causal mask on the fresh S'^T of row block rb: key c of the lane's row is visible iff c <= lim (lim - 4 hi arrives)"""
        self.label(lbl)
        self.emit("s_lshl_b32", SN("t0"), [SN("j"), I(5)])
        self.emit("v_subrev_u32", V(T_TL + rb), [SN("t0"), VN("lim%d" % rb)])   # lim - 4 hi - 32 j
        for r in range(16):
            x = V(S0 + 16 * rb + r)
            self.emit("v_cmp_gt_i32", VCC, [I((r & 3) + 8 * (r >> 2)), V(T_TL + rb)])
            self.emit("v_cndmask_b32", x, [x, V(T_MASKV), VCC])
        self.emit("s_branch", None, [], target=back)

    def mask_branch(self, rb):
        lbl, back = self.newlabel("MASK"), self.newlabel("MASKBACK")
        self.emit("s_cmp_ge_i32", None, [SN("j"), SN("maskfrom")])
        self.emit("s_cbranch_scc1", None, [], target=lbl)
        self.label(back)
        self.outofline.append((lbl, back, rb))

    def s_iteration(self, par, phase_a, phase_b, seam_target, last=False, prefetch=True):
        """S-role iteration of parity par.  Enters with K row fragments 0..5 in flight when phase_a (self.rid), else with nothing."""
        cfg = self.cfg
        nks, nd = cfg.nks, cfg.share(0)
        mm = []     # (dst, a, b, c, read key)
        if phase_a:
            for rb in range(2):
                for ks in range(nks):
                    mm.append((blk(S0, rb), kf(ks), cfrag(rb, ks), blk(CT, rb) if ks == 0 else blk(S0, rb), [("k", ks)]))
        nA = len(mm)
        if phase_b:
            for t in range(2 * nd):
                u, dbl = divmod(t, nd)
                for rb in range(2):
                    mm.append((acc(rb, dbl), af(t), V(DSR + 4 * (2 * rb + u), 4), acc(rb, dbl), [("t", t), ("d", 2 * rb + u)]))
        NM = len(mm)
        fill = [[] for _ in range(NM)]

        def at(g, fn):
            fill[min(max(g, 0), NM - 1)].append(fn)
        pre = []
        if phase_a:
            for k in range(6, nks):
                at(k - 6, lambda k=k: self.row_read(kf(k), k, ("k", k)))
            self.dma_fill(at, NM)
            # row block 0: mask, scale, exp2, packs beside the matrix instructions of row block 1; its two fragments go out at once
            at(nks, lambda: self.mask_branch(0))
            ops = self.s_valu(0)
            lo, hi = nks + 1, 2 * nks - 2
            for n, fn in enumerate(ops):
                at(lo + (n * (hi - lo + 1)) // len(ops), fn)
            for fn in self.s_writes(0, par):
                at(2 * nks - 1, fn)
        if phase_b:
            # the partner's dS' fragments of block i - 2 (exchange parity par) and the first four K^T fragments
            def dsr_read(n):
                self.rid[("d", n)] = self.lds_read("ds_read_b128", V(DSR + 4 * n, 4), VN("xs"), par * XPAR + n * 1024, note="dS' %d" % n)
            if phase_a:
                for n in range(4):
                    at(nks - 5 + n, lambda n=n: dsr_read(n))
                for t in range(min(4, 2 * nd)):
                    at(nks + 1 + 2 * t, lambda t=t: self.tr_read(t, t, 0))
            else:
                pre = [lambda n=n: dsr_read(n) for n in range(4)] + [lambda t=t: self.tr_read(t, t, 0) for t in range(min(4, 2 * nd))]
            for t in range(2 * nd - 4):
                at(nA + 2 * t + 1, lambda t=t: self.tr_read(t + 4, t + 4, 0))
        if phase_a:
            ops = [lambda: self.mask_branch(1)] + self.s_valu(1)
            writes = self.s_writes(1, par)
            if phase_b:
                # (the writes must be PERFORMED before the seam's lgkmcnt(0) + barrier at NM - 4: issued one gap ahead, their
                # latency is the wave's -- and so the workgroup's -- wait; cfg.xwe moves them and the arithmetic up)
                lo, hi = 2 * nks + 1, max(2 * nks + 4, NM - 6 - cfg.xwe)
                for n, fn in enumerate(ops):
                    at(lo + (n * (hi - lo + 1)) // len(ops), fn)
                for fn in writes:
                    at(hi + 1, fn)
            else:                       # iterations 0 and 1: nothing to hide row block 1's work behind
                for fn in ops + writes:
                    at(NM - 1, fn)
        seam_g = None
        if not last:
            seam_g = NM - 4 if (phase_a and phase_b) else NM - 1
            at(seam_g, lambda: self.seam(0, 2 * cfg.NPW if (phase_a and "dma" not in cfg.abl) else 0, seam_target, prof=phase_a and phase_b))
            if phase_a and phase_b:
                at(nA - 1, lambda: self.stamp("pa"))

            def capture():
                self.alt_capture = (self.lds_issued, self.lds_done, dict(self.rid))
            at(seam_g, capture)
            if prefetch:                # K row fragments 0..5 of the next block, two per gap behind the barrier
                for k in range(6):
                    at(seam_g + 1 + k // 2 if (phase_a and phase_b) else NM - 1, lambda k=k: self.row_read(kf(k), k, ("k", k)))
        for fn in pre:
            fn()
        state = None
        for g, (d, a_, b_, c_, keys) in enumerate(mm):
            self.need(keys, [k for m in mm[g + 1:g + 1 + cfg.wa] for k in m[4]])
            self.emit("v_mfma_f32_32x32x16_" + cfg.dtype, d, [a_, b_, c_])
            for fn in fill[g]:
                fn()
            if g == seam_g:
                state = self.alt_capture + (mm[g + 1:],)
        return state

    # ---------------------------------------------------------------- P-role
    def p_valu(self, q, par_prev):
        """dS' = P dP' on score set q with the partner's packed P of the previous block (exchange parity par_prev), packed in place,
        and written to the dS' exchange (same parity).  List of closures in program order."""
        cfg = self.cfg
        f16 = cfg.dtype == "f16"
        base = sset(q)
        out = []
        rid = {}
        for n in range(4):
            def rd(n=n):
                rid[n] = self.lds_read("ds_read_b128", V(PRB + 4 * n, 4), VN("xp"), par_prev * XPAR + n * 1024, note="P %d" % n)
            out.append(rd)
        for rb in range(2):
            for u in range(2):
                n = 2 * rb + u
                for w in range(4):
                    word = V(PRB + 4 * n + w)
                    r = 8 * u + 2 * w
                    lo, hi = V(base + 16 * rb + r), V(base + 16 * rb + r + 1)

                    def first(n=n, word=word, lo=lo):
                        self.lds_need(rid[n])
                        if f16:       # FP16 half x FP32 in one instruction (v_cvt + v_mul: two)
                            self.emit("v_fma_mix_f32", lo, [word, lo, I(0)], op_sel=(0, 0, 0), op_sel_hi=(1, 0, 0))
                            return
                        self.emit("v_lshlrev_b32", V(T_T0), [I(16), word])
                        self.emit("v_mul_f32", lo, [V(T_T0), lo])

                    def second(word=word, hi=hi):
                        if f16:
                            self.emit("v_fma_mix_f32", hi, [word, hi, I(0)], op_sel=(1, 0, 0), op_sel_hi=(1, 0, 0))
                            return
                        self.emit("v_and_b32", V(T_T1), [I(0xFFFF0000), word])
                        self.emit("v_mul_f32", hi, [V(T_T1), hi])
                    out += [first, second]
                for w in range(4):
                    r = 8 * u + 2 * w
                    out.append(lambda rb=rb, u=u, w=w, r=r: self.emit(
                        "v_cvt_pk_%s_f32" % cfg.dtype, V(base + 16 * rb + 4 * u + w), [V(base + 16 * rb + r), V(base + 16 * rb + r + 1)]))
            for u in range(2):
                out.append(lambda rb=rb, u=u: self.lds_write(VN("xs"), V(base + 16 * rb + 4 * u, 4), par_prev * XPAR + (2 * rb + u) * 1024))
        return out

    def p_iteration(self, par, phase_a, phase_b, seam_target, prefetch=True):
        """P-role iteration of parity par.  Enters with V row fragments 0, 1 in flight when phase_a, else with nothing."""
        cfg = self.cfg
        nks, nd = cfg.nks, cfg.share(1)
        q_cur, q_prev = par, 1 - par
        mm = []
        if phase_a:
            for ks in range(nks):
                for rb in range(2):
                    mm.append((blk(sset(q_cur), rb), af(ks), cfrag(rb, ks), blk(CT, rb) if ks == 0 else blk(sset(q_cur), rb), [("f", ks)]))
        nA = len(mm)
        if phase_b:
            for t in range(2 * nd):
                u, dbl = divmod(t, nd)
                for rb in range(2):
                    mm.append((acc(rb, dbl), af(nks + t), V(sset(q_prev) + 16 * rb + 4 * u, 4), acc(rb, dbl), [("f", nks + t)]))
        NM = len(mm)
        fill = [[] for _ in range(NM)]

        def at(g, fn):
            fill[min(max(g, 0), NM - 1)].append(fn)

        def fread(k):
            if k < nks:
                self.row_read(af(k), k, ("f", k))
            else:
                self.tr_read(k, k - nks, 1)
                self.rid[("f", k)] = self.rid[("t", k - nks)]
        frags = [m[4][0][1] for m in mm][::2]
        first_g = {}
        for g, m in enumerate(mm):
            first_g.setdefault(m[4][0][1], g)
        pre = []
        if phase_a:
            at(0, lambda: fread(frags[2]))
            at(1, lambda: fread(frags[3]))
            self.dma_fill(at, NM, (nA - 2, NM - 5) if phase_b else None)
        else:
            pre = [lambda k=k: fread(k) for k in frags[:4]]
        for n in range(len(frags) - 4):
            at(first_g[frags[n]] + 1, lambda n=n: fread(frags[n + 4]))
        valu = self.p_valu(q_prev, q_prev) if phase_b else []
        if phase_b and phase_a:      # the partner's four P fragments are requested at once, the arithmetic starts four gaps later
            for n, fn in enumerate(valu[:4]):
                at(n // 2, fn)
            lo, hi = cfg.pv0, nA - 3
            for n, fn in enumerate(valu[4:]):
                at(lo + (n * (hi - lo + 1)) // len(valu[4:]), fn)
        seam_g = NM - 4 if (phase_a and phase_b) else NM - 1
        at(seam_g, lambda: self.seam(1, 2 * cfg.NPW if (phase_a and "dma" not in cfg.abl) else 0, seam_target, prof=phase_a and phase_b))
        if phase_a and phase_b:
            at(nA - 1, lambda: self.stamp("pa"))

        def capture():
            self.alt_capture = (self.lds_issued, self.lds_done, dict(self.rid))
        at(seam_g, capture)
        if prefetch and phase_a:
            at(seam_g + 1 if phase_b else NM - 1, lambda: self.row_read(af(0), 0, ("f", 0)))
            at(NM - 1, lambda: self.row_read(af(1), 1, ("f", 1)))
        if phase_b and not phase_a:         # the last block: nothing to hide the arithmetic behind
            for fn in valu:
                fn()
        for fn in pre:
            fn()
        state = None
        for g, (d, a_, b_, c_, keys) in enumerate(mm):
            self.need(keys, [k for m in mm[g + 1:g + 1 + cfg.wa] for k in m[4]])
            self.emit("v_mfma_f32_32x32x16_" + cfg.dtype, d, [a_, b_, c_])
            for fn in fill[g]:
                fn()
            if g == seam_g:
                state = self.alt_capture + (mm[g + 1:],)
        return state

    # ---------------------------------------------------------------- control flow
    def alt_tail(self, label, state, target):
        """the matrix instructions behind a seam whose exit test fired: the next iteration has no first product to prefetch for"""
        issued, done, rid, rest = state
        save = (self.lds_issued, self.lds_done, self.rid)
        self.lds_issued, self.lds_done, self.rid = issued, done, rid
        self.label(label)
        for d, a_, b_, c_, keys in rest:
            for key in keys:
                self.lds_need(self.rid[key])
            self.emit("v_mfma_f32_32x32x16_" + self.cfg.dtype, d, [a_, b_, c_])
        self.lds_flush()
        self.emit("s_branch", None, [], target=target)
        self.lds_issued, self.lds_done, self.rid = save

    def enter(self, n_reads, keys):
        self.lds_done = self.lds_issued - n_reads
        self.rid = {k: self.lds_issued - n_reads + 1 + i for i, k in enumerate(keys)}

    def prologue(self, role):
        cfg = self.cfg
        for r in range(16 * 4 * 2):
            if (r // 16) % 4 >= cfg.share(role):
                continue
            self.emit("v_accvgpr_write_b32", A(r), [I(0)])
        for rb in range(2):
            for r in range(16):
                self.emit("v_mov_b32", V(CT + 16 * rb + r), [VN("negt%d" % rb)])
        self.emit("v_mov_b32", V(T_MASKV), [F(-(0.875 / 1.44269504089) * 3.402823466e+38)])   # +Softmax.swift:242-243
        self.emit("s_mov_b32", SN("wrk"), [SN("wrk0")])
        self.emit("s_mov_b32", SN("wrv"), [SN("wrv0")])
        for t in range(2):                                   # blocks 0 and 1
            for n in range(2 * cfg.NPW):
                self.dma_piece(n)
            for n in range(2 * cfg.NPW):
                self.dma_advance(n)
            self.wr_advance()
        self.emit("s_waitcnt", None, [], vmcnt=2 * cfg.NPW)
        self.emit("s_barrier")
        self.emit("s_mov_b32", SN("j"), [I(0)])
        self.emit("s_mov_b32", SN("sk"), [I(0)])
        self.emit("s_mov_b32", SN("sv"), [I(0)])
        for d in ("kd0", "kd1", "kd2", "vd0"):
            self.emit("s_mov_b32", SN(d), [I(cfg.TI)])
        for d in ("pa", "pb", "pc", "pd"):
            self.emit("s_mov_b32", SN(d), [I(0)])

    def role_stream(self, role):
        cfg = self.cfg
        tag = "S" if role == 0 else "P"
        L = {n: self.newlabel(tag + n) for n in ("LOOP", "ALTA", "ALTB", "ALT0", "ALT1", "T0", "T1", "IDLE", "END")}
        self.prologue(role)
        if role == 0:
            keys = [("k", k) for k in range(6)]
            for k in range(6):
                self.row_read(kf(k), k, ("k", k))
            st_a = self.s_iteration(0, True, False, L["ALTA"])                 # i = 0
            self.enter(6, keys)
            st_b = self.s_iteration(1, True, False, L["ALTB"])                 # i = 1 (n >= 2)
            self.stamp0()
            self.enter(6, keys)
            self.label(L["LOOP"])
            self.in_loop = True
            st0 = self.s_iteration(0, True, True, L["ALT0"])                   # i even
            self.enter(6, keys)
            st1 = self.s_iteration(1, True, True, L["ALT1"])                   # i odd
            self.in_loop = False
            self.enter(6, keys)
            self.emit("s_branch", None, [], target=L["LOOP"])
            # exits.  i = 1 >= n (n = 1): an idle iteration, then block 0's update (iteration 2)
            self.alt_tail(L["ALTA"], st_a, L["IDLE"])
            self.alt_tail(L["ALTB"], st_b, L["T0"])                            # i = 2 = n
            self.alt_tail(L["ALT0"], st0, L["T1"])                             # next i odd = n
            self.alt_tail(L["ALT1"], st1, L["T0"])                             # next i even = n
            self.label(L["IDLE"])
            self.lds_done = self.lds_issued
            self.seam(0, 0)
            self.rid = {}
            self.s_iteration(0, False, True, None, last=True)
            self.lds_flush()
            self.emit("s_branch", None, [], target=L["END"])
            for par in (0, 1):                                                 # i = n (block n - 2), then i = n + 1 (block n - 1)
                self.label(L["T%d" % par])
                self.lds_done = self.lds_issued
                self.rid = {}
                self.s_iteration(par, False, True, None, prefetch=False)
                self.rid = {}
                self.s_iteration(1 - par, False, True, None, last=True)
                self.lds_flush()
                self.emit("s_branch", None, [], target=L["END"])
        else:
            keys = [("f", 0), ("f", 1)]
            for k in range(2):
                self.row_read(af(k), k, ("f", k))
            st_a = self.p_iteration(0, True, False, L["ALTA"])                 # i = 0
            self.stamp0()
            self.enter(2, keys)
            self.label(L["LOOP"])
            self.in_loop = True
            st1 = self.p_iteration(1, True, True, L["ALT1"])                   # i odd
            self.enter(2, keys)
            st0 = self.p_iteration(0, True, True, L["ALT0"])                   # i even
            self.in_loop = False
            self.enter(2, keys)
            self.emit("s_branch", None, [], target=L["LOOP"])
            self.alt_tail(L["ALTA"], st_a, L["T1"])                            # i = 1 = n
            self.alt_tail(L["ALT1"], st1, L["T0"])
            self.alt_tail(L["ALT0"], st0, L["T1"])
            for par in (0, 1):                                                 # i = n: the last block's dS' and update
                self.label(L["T%d" % par])
                self.lds_done = self.lds_issued
                self.rid = {}
                self.p_iteration(par, False, True, None, prefetch=False)
                self.lds_flush()
                self.emit("s_branch", None, [], target=L["END"])
        for lbl, back, rb in self.outofline:
            self.mask_section(lbl, back, rb)
        self.outofline = []
        self.label(L["END"])
        self.emit("s_waitcnt", None, [], vmcnt=0, lgkmcnt=0)

    def build(self):
        self.outofline = []
        prole, fin = self.newlabel("PROLE"), self.newlabel("FIN")
        self.emit("s_cmp_eq_u32", None, [SN("role"), I(1)])
        self.emit("s_cbranch_scc1", None, [], target=prole)
        self.role_stream(0)
        self.emit("s_branch", None, [], target=fin)
        self.label(prole)
        self.lds_issued = self.lds_done = 0
        self.role_stream(1)
        self.label(fin)
        return self.ins


# ---------------------------------------------------------------- rendering
def write_inc(path):
    lines = ["// GENERATED by tools/dq5gen.py -- do not edit.  Instruction streams of attn_dq16_p5 (see the generator's",
             "// header for the register map and the iteration table).", "#pragma once", ""]
    lines.append("#define MFA_DQ5_OWNED_VGPRS " + ", ".join('"v%d"' % i for i in range(FIRST_OWNED_VGPR, 256)))
    lines.append("")
    lines.append("// X(name, applies the softmax scale in fp32, head-dimension bucket, stamps the shader clock)")
    lines.append("#define MFA_DQ5_STREAM_LIST(X) \\")
    for name, cfg in VARIANTS.items():
        lines.append("  X(%s, %d, %d, %d) \\" % (name, cfg.exact, cfg.D, cfg.prof))
    lines.append("")
    lines.append("// timing-only ablations of the D = 256 BF16 stream (developer library: MFA_DQ5_DEV_STREAM=<name>, tools/dq5_ab.py)")
    lines.append("#define MFA_DQ5_DEV_ABL_LIST(X) " + " ".join("X(D256_BF16_FOLD_%s)" % n for n in list(DEV_ABLATIONS) + ["XWE2", "XWE4", "PV8", "XWE4_PV8", "PV12", "PV16", "WA4", "PV12_WA4"]))
    lines.append("")
    for name, cfg in VARIANTS.items():
        ins = Stream(cfg).build()
        txt = render(ins)
        lines.append("// %s: dtype=%s exact=%d bucket %d -- %d instructions" % (name, cfg.dtype, cfg.exact, cfg.D, len(txt)))
        lines.append("#define MFA_DQ5_STREAM_%s \\" % name)
        for t in txt:
            lines.append('  "%s\\n\\t" \\' % t)
        lines.append('  ""')
        lines.append("")
    with open(path, "w") as f:
        f.write("\n".join(lines))


DEV_ABLATIONS = {"ABL_ROWRD": ("rowrd",), "ABL_TRRD": ("trrd",), "ABL_XCH": ("xr", "xw"), "ABL_VALU": ("valu",), "ABL_BAR": ("bar",),
                 "ABL_DMA": ("dma",), "ABL_LDS": ("rowrd", "trrd", "xr", "xw"), "ABL_LDS_VALU": ("rowrd", "trrd", "xr", "xw", "valu"),
                 "ABL_ALL": ("rowrd", "trrd", "xr", "xw", "valu", "dma", "bar", "wait")}


def _variants():
    out = {}
    for D in (160, 192, 256):
        for dt in ("bf16", "f16"):
            out["D%d_%s_FOLD" % (D, dt.upper())] = Cfg(dt, D=D)
            out["D%d_%s_EXACT" % (D, dt.upper())] = Cfg(dt, exact=1, D=D)
    out["D256_BF16_FOLD_PROF"] = Cfg("bf16", D=256, prof=1)       # developer library only (tools/bwd5_prof.py)
    out["D160_BF16_FOLD_PROF"] = Cfg("bf16", D=160, prof=1)
    for e in (2, 4):                                            # developer schedules: the S-role's second pair of exchange writes earlier
        out["D256_BF16_FOLD_XWE%d" % e] = Cfg("bf16", D=256, xwe=e)
    out["D256_BF16_FOLD_PV8"] = Cfg("bf16", D=256, pv0=8)
    out["D256_BF16_FOLD_PV12"] = Cfg("bf16", D=256, pv0=12)
    out["D256_BF16_FOLD_PV16"] = Cfg("bf16", D=256, pv0=16)
    out["D256_BF16_FOLD_WA4"] = Cfg("bf16", D=256, wa=4)
    out["D256_BF16_FOLD_PV12_WA4"] = Cfg("bf16", D=256, pv0=12, wa=4)
    out["D256_BF16_FOLD_XWE4_PV8"] = Cfg("bf16", D=256, xwe=4, pv0=8)
    for name, abl in DEV_ABLATIONS.items():                     # developer library only: timing-only ablations (tools/dq5_ab.py)
        out["D256_BF16_FOLD_" + name] = Cfg("bf16", D=256, abl=abl)
    return out


VARIANTS = _variants()

if __name__ == "__main__":
    ins = Stream(VARIANTS["D256_BF16_FOLD"]).build()
    print(len(ins), "instructions in the D = 256 BF16 stream")
