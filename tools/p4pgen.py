#!/usr/bin/env python3
"""Generator of the PERSISTENT instruction stream of the D <= 128 forward kernel (csrc/attn_fwd16_p4p.h).

attn_fwd16_p4 (tools/p4gen.py) keeps the traversal of ONE 256-row block in an asm statement and pays the block's fixed
cost in the open: argument decoding, the Q loads, the first three LDS-DMA tiles, the epilogue through LDS and the drain of
its stores -- 28.5 k of a 64-tile block's 211 k shader clocks, because a kernel that owns all 512 registers
runs one workgroup per compute unit.  Here the BLOCK LOOP is inside the statement: one workgroup per compute unit walks the
blocks `blockIdx, blockIdx + gridDim, ...` and

  * the first three K / V tiles of block n + 1 are simply the ring's next tiles: the LDS-DMA fillers of block n's last two
    tiles switch to the next block's descriptors (K in phase B(nt - 2), V in B(nt - 1));
  * Q of block n + 1 arrives by LDS-DMA in images of its own (4 x 16 KiB behind the ring), requested in B(nt - 1);
  * O of block n leaves the accumulator registers directly: 1 / l, 32 buffer stores per wave (lane = row, four consecutive
    d per register group), L = m + log2 l; nothing passes through LDS and nothing waits for the stores -- gfx950 counts
    loads and stores in one in-order vmcnt, so the first waits of block n + 1 are `vmcnt(34)`: every LDS-DMA piece older
    than the 34 stores has landed;
  * what hipcc did per workgroup (block decode, descriptors) is a table of 64-byte entries in LDS that the C++ prologue
    fills once per workgroup; the stream reads entry n + 1 while block n starts.

The tile traversal itself is p4gen's (phase tables, register map, ring discipline, deferred rescale); a block always walks
an EVEN number of tiles (an odd count gets one fully masked tile), so the two K images and the score-tile parity line up
from block to block.  Dense launches only (no causal mask, no per-batch lengths, no block mask): those keep attn_fwd16_p4.

Registers the statement owns beyond p4gen's map:
    v28..v31    m0, m1, l0, l1
    v248..v255  LDS-DMA offsets koff0..3, voff0..3 (exact-scale streams: v160..v167, their -m blocks are unused)
    s40..s99    descriptors, block state, loop counters (PSGPR below); hipcc passes inputs only, nothing is live after it

The same instruction list runs on the lane-exact model (tools/p4psim.py, tests/test_p4p_stream.py): several blocks per
workgroup, LDS-DMA landing early / late, stores retiring late.

Usage: python tools/p4pgen.py   (rewrites metal_flash_attention_amd/csrc/attn_fwd16_p4p_stream.inc)
"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import p4gen  # noqa: E402
from p4gen import (A, F, I, M0, SN, V, VCC, VN, Cfg, Ins, Stream, KSLOT, VBASE, VSLOT, VRING, O_BASE, Q_BASE, CM_BASE,  # noqa: E402
                   T_KADDR, T_LB, T_CORR, T_MX, T_MN, T_SW, T_TL, T_MASKV, S_BASE)

QIMG = VBASE + VRING * VSLOT          # 80 KiB: the four waves' Q images (16 KiB each)
TABLE = QIMG + 4 * 16384              # 144 KiB: block table, 64 bytes per entry
TABLE_ENTRIES = 256
LDS_BYTES = TABLE + TABLE_ENTRIES * 64
assert LDS_BYTES == 160 * 1024
NST = 34                               # buffer stores per wave and block: 32 x O, 2 x L
DESC_FLAGS = 0x00020000                # raw buffer, 32-bit data format (the words hipcc's make_buffer_rsrc uses)
OOB = 0xFFFFFF00


def SR(n, cnt=1):
    return ("sr", n, cnt)


# ---------------------------------------------------------------- fixed scalar registers (clobbered by the statement)
PSGPR = dict(kres=(40, 4), vres=(44, 4), tres=(48, 4), lres=(52, 4),
             qbn=(56, 2), kbn=(58, 2), vbn=(60, 2), obn=(62, 2), lbn=(64, 2), row0n=(66, 1),
             ob=(68, 2), lb=(70, 2), row0=(72, 1), blk=(73, 1), hasnext=(74, 1), ntm1=(75, 1), ntm2=(76, 1),
             j=(77, 1), vrd=(78, 1), vwr=(79, 1), pend=(80, 1), t0=(81, 1), t1=(82, 1), t2=(83, 1), sv=(84, 2),
             kc0=(86, 1), kstep=(87, 1), vstep=(88, 1), ntb=(89, 1), wntb=(90, 1), maskb=(91, 1), ntu=(92, 1), t5=(93, 1),
             q4=(94, 1), t3=(95, 1), t4=(96, 1), qrow=(97, 1), plast=(100, 1), qswj=(101, 1),
             fend=(101, 1), fcnt=(93, 1), ksoff=(67, 1), vsoff=(98, 1),
             wdiag=(99, 1), t1x=(85, 1),
             # causal ("geometry") streams, round 6: the block's own sequence lengths travel in its table entry (per-batch lengths) -- rows
             # and keys of the NEXT entry, rows / last key / tile count / diagonal offset of the current block
             rrn=(36, 1), ccn=(37, 1), rr=(38, 1), cm1=(39, 1), ttot=(67, 1), coff=(98, 1))   # causal streams: first row of the wave + C - R (the first key of its diagonal tile when aligned); a temporary   # branch-free loop (dense streams without the merged block switch): first tile it does not take, iterations left
FIRST_CLOBBERED_SGPR, LAST_CLOBBERED_SGPR = 36, 101

# inputs of the statement (hipcc allocates them below s40 / v28)
INOUT_V = ["lim0", "lim1"]          # mask limits per row block: constant in dense streams, rewritten per block by causal ones
IN_V = ["kbase", "vbase", "kv0", "kv1", "kv2", "kv3", "vv", "qv0", "qv1", "qv2", "qv3", "ov0", "ov1", "ov2", "ov3",
        "lv", "ewa", "era", "qlane", "hi4"]
IN_S = ["nt", "maskfrom", "scale2", "kinc", "vinc", "ldsk", "ldsv", "ldsq", "qrel", "nblk", "tbl", "wave64", "ldq2", "ldo",
        "nrecq", "nreck", "nrecv", "nreco", "nrecl", "dr", "cflag"]   # (nrec*: dense streams; cflag: causal streams -- 1 = causal mask, 0 = lengths only)


class PCfg(Cfg):
    """o16: O leaves in the stream's 16-bit type (lowPrecisionOutputs); l16: L is stored in FP16 (mixed-precision mode)"""

    def __init__(self, dtype="bf16", thr=8.0, fold=0, xb=40, o16=0, l16=0, pprof=0, causal=0, merge=0, fuse=None, bal=0, cap=7, abl=(), pad=0, maxa=1, va0=0, xe=0, fastdec=0, fdpos=0,
                 fastloop=0, align=0, soff=0, pksum=0, dmapol="", diagmask=0, split=0, orow=0, stpol="", qearly=0):
        Cfg.__init__(self, dtype=dtype, thr=thr, fold=fold, xb=xb, xe=xe, bal=bal, cap=cap, abl=abl, pad=pad, maxa=maxa, va0=va0, fastdec=fastdec, fdpos=fdpos)
        self.o16, self.l16 = o16, l16
        # fastloop (round 6, dense streams): the timing-only ablation that dropped the per-tile tests of the loop -- block switch
        # (two not-taken branches), mask section (one TAKEN branch), pending rescale (one not-taken branch) -- ran 9.6 % faster on
        # all-zero operands (profiles/r06_p4p_ablations.txt, ABL6_CTL): a wave alone on its SIMD has nobody to hide an instruction
        # fetch redirect or a branch's issue bubble behind.  The tiles that cannot need any of those tests -- j < maskfrom and
        # j <= nt - 3 -- run in a BRANCH-FREE copy of the loop: two tiles per iteration, one back-edge, and the rescale decision
        # (the one test online softmax cannot lose), whose rare taken path continues in the ordinary copy of the same phase.
        # fastloop = N: N tile pairs per iteration (1 or 2); align: the loop head on a 64-byte line
        self.fastloop, self.align = fastloop, align
        # soff (round 6): the LDS-DMA loads take the tile advance as their SCALAR offset; the lanes' offsets are then constants of the
        # workgroup (set once), a block switch is a new resource base and a scalar offset of zero
        self.soff = soff
        assert not (soff and pprof == 1)
        # pksum (round 6 experiment): the two partial row sums of a row block live in an aligned register pair (l0 | v31, v244 | v245)
        # and a score pair is added with ONE v_pk_add_f32: 32 vector instructions per tile less
        self.pksum = pksum
        # dmapol (round 6 experiment): cache-policy bits on the steady-state LDS-DMA loads of K / V (" nt", " sc1", ...): every line of a
        # tile is read once per compute unit, the vector L1 never hits
        self.dmapol = dmapol
        # diagmask (round 6, causal streams): compile-time lane masks for the aligned diagonal tile (p4gen.mask_section)
        self.diagmask = diagmask
        # split (round 6; column-parallel launches of few-workgroup problems -- one head, the reference's own benchmark shape): a table
        # entry is (row block, piece of the key range), its K / V bases start at the piece; the epilogue leaves the UN-NORMALISED O^T
        # (fp32, leading dimension D) and (m, l) in the piece's slabs of the caller's workspace, which attn_fwd_combine merges
        # (attn_fwd16_v3.h), like the split streams of tools/p6gen.py.  Pieces are whole multiples of two tiles
        self.split = split
        assert not (split and (causal or o16 or merge or fuse))
        # orow (round 6, fp32 O): the second products accumulate O = P V with lane = column, register = row (p4gen.Stream.pv): the
        # epilogue then stores rows straight from the registers -- 128 buffer_store_dword per wave and block, each two full 128-byte
        # lines -- instead of turning O^T through LDS (32 ds_write_b128 + 32 ds_read_b128 + waits per wave: 5.7 k clocks per block,
        # profiles/r06_p4p_epilogue.txt).  The price: a row's factor (deferred rescale, 1 / l) is needed per REGISTER, fetched from the
        # lane that owns the row by ds_bpermute_b32.  No LDS staging, so no ordering against the V ring either
        self.orow = orow
        self.stpol = stpol     # (round 6 experiment) cache-policy bits on the stores of O (" nt", " sc1", " sc0 sc1")
        # qearly (round 6 experiment): the next block's Q image (64 KiB per workgroup, the largest of its first requests) is asked for one
        # tile earlier -- with K'(0) under tile nt - 2 instead of with V'(0) / K'(1) under the last tile -- so that it has landed before the
        # epilogue's 128 KiB of stores want the same path (its LDS image is free since the block's own Q fragments were read)
        self.qearly = qearly
        assert not (qearly and merge)
        assert not (orow and (o16 or merge or fuse or pprof == 1))
        if split:
            self.l16 = 0
        assert not (diagmask and (not causal or pprof))
        assert not (pksum and bal != 2)
        assert not (fastloop and (merge or bal != 2))
        # causal (extension, row r sees key c iff c <= r + C - R): tile counts, mask limits and the per-wave traversal bound are
        # computed per block inside the stream; the block table lists the blocks in pairs (long, short) like attn_fwd16_p4's
        # causal launch, so that every workgroup walks the same number of tiles
        self.causal = causal
        # merge (dense streams; developer experiment that LOST, profiles/r03_p4p_merged_block_switch.txt): the softmax finish of a
        # block's last tile runs beside the next block's first K Q^T products instead of beside nothing -- the pipeline's drain and
        # fill share one phase at every block switch (-1.4 k clocks), but O must then be zeroed in the epilogue and its staging
        # registers halve (+2.6 k)
        self.merge = merge
        assert not (merge and causal)
        # fuse (dense streams; developer experiment that did NOT pay, profiles/r03_p4p_fused_tail.txt: the LDS round trips and
        # store issues stall the products they sit between as long as they take on their own): the last tile's P V products run
        # in (head-dimension block, key step, row block) order, so that a
        # 32 x 32 block of O^T is complete after every fourth product; its share of the epilogue (accumulator reads, 1 / l, the
        # trip through LDS, four stores) is dealt out as fillers of the products that follow instead of running behind them.
        # Causal streams keep the separate epilogue: their waves reach the last tile at different times, and the barrier in
        # front of the LDS staging must be the same barrier for every wave
        self.fuse = 0 if fuse is None else fuse
        assert not (self.fuse and (causal or merge))
        # pprof (developer builds, exact-scale streams only: their -m blocks v168.. are free): shader-clock sums per segment of
        # the block loop in v168..v183, written to O[first row of the wave's last block][0:16] when the workgroup ends
        # pprof = 2 (round 6; any dense stream): the sums live in SCALAR registers the dense streams leave unused (SPROF_ACC) -- the
        # product schedule itself can be stamped, FOLD streams included: phase A | wait for this wave's LDS-DMA pieces | barrier | phase B
        self.pprof = pprof
        assert not (pprof == 1 and fold)
        assert not (pprof == 2 and (causal or merge))


PROF_ACC = 168
PROF_MAGIC = 0x50524F46
# pprof = 2: name -> scalar accumulator (s67 is free; s89..s93 are the causal streams' block geometry, unused by dense streams)
SPROF_NAMES = ["loop_a", "loop_vm", "loop_bar", "loop_b", "rest", "blocks"]
SPROF_ACC = {"loop_a": 99, "loop_vm": 89, "loop_bar": 90, "loop_b": 91, "rest": 92, "blocks": 73}   # (blocks: the table index `blk` itself)
PROF_NAMES = ["table", "wait_q", "qfrag", "tile0_a", "tile0_b", "loop_a", "loop_wait", "loop_b", "tile1_wait", "tail", "epilogue", "blocks"]


def s(name, cnt=None, off=0):
    base, n = PSGPR[name]
    return SR(base + off, n if cnt is None else cnt)


class PStream(Stream):
    persistent = True

    def __init__(self, cfg):
        Stream.__init__(self, cfg)
        dma_base = 248 if cfg.fold else 160
        self.vfixed = {"m0": 28, "m1": 29, "l0": 30, "l1": 31}
        if cfg.pksum:
            self.vfixed["l1"] = 244
            self.r_lb = [31, 245]
            self.r_maskv = 243
        for i in range(4):
            self.vfixed["koff%d" % i] = dma_base + i
            self.vfixed["voff%d" % i] = dma_base + 4 + i

    # ---- operand mapping: p4gen's named temporaries become fixed registers
    def finish(self):
        def m(o):
            if o is None:
                return None
            if o[0] == "V" and o[1] in self.vfixed:
                return V(self.vfixed[o[1]])
            if o[0] == "S" and self.cfg.causal and o[1] in ("nt", "wnt", "maskfrom"):
                return SR(PSGPR[{"nt": "ntb", "wnt": "wntb", "maskfrom": "maskb"}[o[1]]][0], 1)
            if o[0] == "S" and o[1] == "wnt":
                return ("S", "nt", 1)            # dense: every wave walks the whole block
            if o[0] == "S" and o[1] in PSGPR:
                base, n = PSGPR[o[1]]
                return SR(base, o[2] if len(o) > 2 else n)
            return o
        for ins in self.ins:
            ins.d = m(ins.d)
            ins.s = tuple(m(x) for x in ins.s)
        return self.ins

    # ---- hooks called by p4gen.Stream.phase_b
    def b_hook(self, at, par, mfma, gaps=(16, 17)):
        """phase B(j) of the persistent stream: the LDS-DMA pieces of the block's last two tiles belong to the NEXT block"""
        ksw, vsw = self.newlabel("KSW"), self.newlabel("VSW")

        def check(reg, lbl):
            self.emit("s_cmp_eq_u32", None, [SN("j"), s(reg)])
            self.emit("s_cbranch_scc1", None, [], target=lbl)
            self.label(lbl + "_BACK")
        at(gaps[0], lambda: check("ntm2", ksw))
        at(gaps[1], lambda: check("ntm1", vsw))
        self.outofline.append(("ksw", ksw, ksw + "_BACK", par, False))
        self.outofline.append(("vsw", vsw, vsw + "_BACK", par, False))
        if self.cfg.merge:      # the next block's Q is needed right behind this block's last tile: requested three tiles earlier
            qsw = self.newlabel("QSW")
            at(15, lambda: check("qswj", qsw))
            self.outofline.append(("qsw", qsw, qsw + "_BACK", par, False))

    def pstamp(self, name, first=False):
        """developer streams: add the shader-clock time since the previous stamp to accumulator `name` (s_memtime returns
        through lgkmcnt: a stamp sits only where no LDS read is in flight)"""
        if not self.cfg.pprof:
            return
        if self.cfg.pprof == 2:   # (s[84:85] = `sv`: the fast-decision FOLD streams do not use it; s98 is the V loads' scalar offset)
            assert self.cfg.fastdec
            self.emit("s_memtime", SR(84, 2))
            self.emit("s_waitcnt", None, [], lgkmcnt=0)
            self.lds_done = self.lds_issued
            if not first:
                acc = SR(SPROF_ACC.get(name, SPROF_ACC["rest"]))
                self.emit("s_sub_u32", s("t3"), [SR(84), s("plast")])
                self.emit("s_add_u32", acc, [acc, s("t3")])
            self.emit("s_mov_b32", s("plast"), [SR(84)])
            return
        self.emit("s_memtime", SR(98, 2))
        self.emit("s_waitcnt", None, [], lgkmcnt=0)
        self.lds_done = self.lds_issued
        if not first:
            self.emit("s_sub_u32", s("t3"), [SR(98), s("plast")])
            acc = V(PROF_ACC + PROF_NAMES.index(name))
            self.emit("v_add_u32", acc, [s("t3"), acc])
        self.emit("s_mov_b32", s("plast"), [SR(98)])

    def lds_write(self, op, addr, data, offset):
        """LDS writes share the in-order LDS queue (and lgkmcnt) with the reads"""
        self.emit(op, None, [addr, data], offset=offset)
        self.lds_issued += 1

    # ---- pieces
    def desc(self, name, base, nrec):
        """128-bit buffer resource `name` = (64-bit base in SGPR pair `base`, byte count `nrec`)"""
        self.emit("s_mov_b32", s(name, 1, 0), [s(base, 1, 0)])
        self.emit("s_and_b32", s(name, 1, 1), [s(base, 1, 1), I(0xFFFF)])
        self.nrec(name, nrec)
        self.emit("s_mov_b32", s(name, 1, 3), [I(DESC_FLAGS)])

    def nrec(self, name, which):
        """word 2 of resource `name`: the byte count of the operand -- a launch constant in dense streams, rows (or keys) of the block's
        own batch entry x bytes per row in causal ("geometry") streams.  which: nrecq / nreck / nrecv (NEXT block), nreco / nrecl (current)"""
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

    def load_next(self):
        """entry `blk` of the block table -> the *n registers (Q, K, V, O, L bases and the first row of the block)"""
        tv, tb = 96, 100                       # address register, twelve data registers v100..v111 (score tile, free here)
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

    def switch_k(self, init=False):
        if not init:
            self.emit("s_mov_b32", s("kres", 1, 0), [s("kbn", 1, 0)])
            self.emit("s_and_b32", s("kres", 1, 1), [s("kbn", 1, 1), I(0xFFFF)])
            if self.cfg.causal:
                self.nrec("kres", "nreck")
            if self.cfg.soff:
                self.emit("s_mov_b32", s("ksoff"), [I(0)])
                return
        self.emit("s_mov_b32", s("t3"), [s("kc0")])
        for i in range(4):
            self.emit("v_add_u32_e64", VN("koff%d" % i), [VN("kv%d" % i), s("t3")], clamp=1)
            if i != 3:
                self.emit("s_add_u32", s("t3"), [s("t3"), s("kstep")])

    def switch_v(self, init=False):
        if not init:
            self.emit("s_mov_b32", s("vres", 1, 0), [s("vbn", 1, 0)])
            self.emit("s_and_b32", s("vres", 1, 1), [s("vbn", 1, 1), I(0xFFFF)])
            if self.cfg.causal:
                self.nrec("vres", "nrecv")
            if self.cfg.soff:
                self.emit("s_mov_b32", s("vsoff"), [I(0)])
                return
        self.emit("s_mov_b32", s("t3"), [I(0)])
        for i in range(4):
            self.emit("v_add_u32_e64", VN("voff%d" % i), [VN("vv"), s("t3")], clamp=1)
            if i != 3:
                self.emit("s_add_u32", s("t3"), [s("t3"), s("vstep")])

    def block_geometry(self):
        """causal streams, per block: tile count (made even), this wave's traversal bound, first tile that needs masking, the
        lanes' mask limits -- what attn_fwd16_p4's C++ prologue computes per workgroup (attn_fwd16_p4.h, 'traversal range')"""
        t0, t2, t3 = s("t0"), s("t2"), s("t3")
        self.emit("s_add_u32", t0, [s("row0"), I(256)])
        self.emit("s_min_u32", t0, [t0, SN("rr")])
        self.emit("s_sub_u32", t0, [t0, I(1)])                       # last row of the block
        self.emit("s_add_u32", t0, [t0, SN("coff")])
        self.emit("s_lshr_b32", t0, [t0, I(6)])
        self.emit("s_add_u32", t0, [t0, I(1)])
        self.emit("s_min_u32", s("ntu"), [t0, SN("ttot")])           # tiles the block's last row can see
        self.emit("s_add_u32", t2, [s("row0"), SN("wave64")])        # first row of the wave
        self.emit("s_add_u32", t3, [t2, I(64)])
        self.emit("s_min_u32", t3, [t3, SN("rr")])
        self.emit("s_sub_u32", t3, [t3, I(1)])                       # its last row
        self.emit("s_add_u32", t3, [t3, SN("coff")])
        self.emit("s_lshr_b32", t3, [t3, I(6)])
        self.emit("s_add_u32", t3, [t3, I(1)])
        self.emit("s_min_u32", t3, [t3, s("ntu")])
        self.emit("s_max_u32", t3, [t3, I(1)])
        self.emit("s_cmp_ge_u32", None, [t2, SN("rr")])              # a wave beyond the last row: one tile, nothing stored
        self.emit("s_cselect_b32", s("wntb"), [I(1), t3])
        self.emit("s_add_u32", t3, [t2, SN("coff")])
        if self.cfg.diagmask:
            self.emit("s_mov_b32", s("wdiag"), [t3])
        self.emit("s_min_u32", t0, [t3, SN("cm1")])
        self.emit("s_add_u32", t0, [t0, I(1)])
        self.emit("s_lshr_b32", s("maskb"), [t0, I(6)])
        self.emit("s_and_b32", t0, [s("ntu"), I(1)])
        self.emit("s_add_u32", s("ntb"), [s("ntu"), t0])             # an odd count walks one fully masked tile more
        self.emit("s_sub_u32", s("ntm1"), [s("ntb"), I(1)])
        self.emit("s_sub_u32", s("ntm2"), [s("ntb"), I(2)])
        for rb in range(2):
            if rb:
                self.emit("s_add_u32", t3, [t3, I(32)])
            lim = VN("lim%d" % rb)
            self.emit("v_add_u32", lim, [t3, VN("qlane")])
            self.emit("v_min_u32", lim, [SN("cm1"), lim])
            self.emit("v_sub_u32", lim, [lim, VN("hi4")])

    def skip_tile(self, par):
        """one tile of a wave whose own rows are done (causal): the workgroup's barrier and this wave's share of the LDS-DMA"""
        self.emit("s_waitcnt", None, [], vmcnt=0)
        self.emit("s_barrier")
        self.vwr_update()
        for kind, reg in (("k", "ntm2"), ("v", "ntm1")):
            over = self.newlabel("NOSW")
            self.emit("s_cmp_eq_u32", None, [SN("j"), s(reg)])
            self.emit("s_cbranch_scc0", None, [], target=over)
            self.emit("s_cmp_eq_u32", None, [s("hasnext"), I(0)])
            self.emit("s_cbranch_scc1", None, [], target=over)
            if kind == "k":
                self.switch_k()
                if self.cfg.qearly:
                    self.issue_q((T_SW, T_SW + 1, T_TL, T_TL + 1, self.r_maskv))
            else:
                self.switch_v()
                if not self.cfg.qearly:
                    self.issue_q((T_SW, T_SW + 1, T_TL, T_TL + 1, self.r_maskv))
            self.label(over)
        for i in range(4):
            self.dma_piece("k", par, i)
        for i in range(4):
            self.dma_piece("v", par, i)
        if self.cfg.soff:
            self.emit("s_add_u32", s("ksoff"), [s("ksoff"), SN("kinc")])
            self.emit("s_add_u32", s("vsoff"), [s("vsoff"), SN("vinc")])
        for i in range(0 if self.cfg.soff else 4):
            self.emit("v_add_u32_e64", VN("koff%d" % i), [VN("koff%d" % i), SN("kinc")], clamp=1)
            self.emit("v_add_u32_e64", VN("voff%d" % i), [VN("voff%d" % i), SN("vinc")], clamp=1)
        self.emit("s_add_u32", SN("j"), [SN("j"), I(1)])

    def issue_q(self, temps):
        """the wave's 64 rows of the next block's Q by LDS-DMA into its own image: piece i = rows 4 i .. 4 i + 3"""
        self.desc("tres", "qbn", "nrecq")
        self.emit("s_add_u32", s("t0"), [s("row0n"), SN("wave64")])
        self.emit("s_mul_i32", s("qrow"), [s("t0"), SN("ldq2")])
        for i in range(16):
            t = V(temps[i % len(temps)])
            self.emit("v_add_u32_e64", t, [VN("qv%d" % (i & 3)), s("qrow")], clamp=1)
            self.emit("s_add_u32", M0, [SN("ldsq"), I(i * 1024)])
            self.emit("buffer_load_dwordx4_lds", None, [t, s("tres", 4)])
            if i != 15:
                self.emit("s_add_u32", s("qrow"), [s("qrow"), s("q4")])

    def issue_tile(self, which, image, advance=True):
        so = self.cfg.soff
        for i in range(4):
            if which == "k":
                self.emit("s_add_u32", M0, [SN("ldsk"), I(image * KSLOT + i * 1024)])
                self.emit("buffer_load_dwordx4_lds", None, [VN("koff%d" % i), s("kres", 4)] + ([s("ksoff")] if so else []))
            else:
                self.emit("s_add_u32", M0, [SN("vwr"), I(i * 1024)])
                self.emit("buffer_load_dwordx4_lds", None, [VN("voff%d" % i), s("vres", 4)] + ([s("vsoff")] if so else []))
        if so:
            name = "k" if which == "k" else "v"
            self.emit("s_add_u32", s(name + "soff"), [s(name + "soff"), SN(name + "inc")])
            return
        for i in range(4):
            if which == "k":
                self.emit("v_add_u32_e64", VN("koff%d" % i), [VN("koff%d" % i), SN("kinc")], clamp=1)
            else:
                self.emit("v_add_u32_e64", VN("voff%d" % i), [VN("voff%d" % i), SN("vinc")], clamp=1)

    def q_fragments(self, qa=96, tmp=(104, 105)):
        """Q image -> B-operand fragments a[128:191] (FOLD: times log2(e)/sqrt(D), rounded to the 16-bit type); eight address
        registers from `qa`, two temporaries, 64 data registers v32..v95 (the even tiles' score registers)"""
        cfg = self.cfg
        qd = 32
        for ks in range(8):
            self.emit("v_add_u32", V(qa + ks), [SN("qrel"), V(T_KADDR + ks)])
        ids = []
        for b in range(2):
            for ks in range(8):
                ids.append(self.lds_read("ds_read_b128", V(qd + 4 * (8 * b + ks), 4), V(qa + ks), b * 8192, note="Q(%d,%d)" % (b, ks)))
        t0, t1 = V(tmp[0]), V(tmp[1])
        for n in range(16):
            self.lds_need(ids[n])
            for w in range(4):
                x = V(qd + 4 * n + w)
                if cfg.fold:
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

    EPI_LTOT, EPI_INV = (T_MX, T_MX + 1), (T_MN, T_MN + 1)
    EPI_VO = T_TL   # (re-initialised by the next block / dead after the loop)
    EPI_WA = property(lambda self: [T_CORR, T_CORR + 1, self.r_lb[0], self.r_lb[1]])
    EPI_RA = property(lambda self: self.r_maskv)

    def epi_prepare(self):
        """what the per-block work of the epilogue needs: resources of O and L, l of the row (half swap), 1 / l, the LDS staging
        addresses.  O leaves through a 4 KiB slice of LDS per wave (its share of the V image the ring is not using: the one after
        V'(0)'s), one 32 x 32 block at a time: lane = row with four consecutive columns per register group goes in (16-byte
        chunks, chunk index XOR row & 7), lane = (row & 7, chunk) comes out -- eight lanes then cover one 128-byte line of a row
        and a store instruction touches 8 lines instead of 32 (measured: 3.4 k instead of 9.0 k clocks per block for the 32 stores)"""
        self.desc("tres", "ob", "nreco")
        self.desc("lres", "lb", "nrecl")
        ltot, inv, ta, tb = self.EPI_LTOT, self.EPI_INV, V(T_SW), V(T_SW + 1)
        for rb in range(2):
            lt, iv = V(ltot[rb]), V(inv[rb])
            self.emit("v_mov_b32", ta, [VN("l%d" % rb)])
            self.emit("v_mov_b32", tb, [VN("l%d" % rb)])
            self.emit("s_nop", None, [I(1)], note="VALU write -> permlane read")
            self.emit("v_permlane32_swap_b32", ta, [tb], swap=1)
            self.emit("v_add_f32", lt, [ta, tb])
            self.emit("v_add_f32", lt, [I(1), lt], note="+ denorm_min (+Caching.swift:311)")
            if self.cfg.split:     # (the pieces stay un-normalised: attn_fwd_combine divides by the merged l)
                continue
            self.emit("v_rcp_f32", iv, [lt])
            self.emit("s_nop", None, [I(0)], note="trans -> VALU")
            self.emit("v_fma_f32", ta, [lt, iv, F(1.0)], neg0=1)      # e = 1 - l r
            self.emit("v_fma_f32", iv, [ta, iv, iv])                   # r += e r
            self.emit("v_cmp_lt_f32", VCC, [F(1e-30), lt])
            self.emit("v_cndmask_b32", iv, [I(0), iv, VCC])            # a row without keys: O = 0
        if self.cfg.orow:
            return
        wa, ra = self.EPI_WA, self.EPI_RA
        self.emit("s_add_u32", s("t2"), [SN("vwr"), I(VSLOT)])
        self.emit("s_cmp_ge_u32", None, [s("t2"), SN("t1")])
        self.emit("s_cselect_b32", s("t2"), [SN("ldsv"), s("t2")])
        for g in range(4):
            if g:
                self.emit("v_xor_b32", V(wa[g]), [I(g << 5), VN("ewa")])
                self.emit("v_add_u32", V(wa[g]), [s("t2"), V(wa[g])])
            else:
                self.emit("v_add_u32", V(wa[g]), [s("t2"), VN("ewa")])
        self.emit("v_add_u32", V(ra), [s("t2"), VN("era")])
        self.emit("s_lshl_b32", s("t4"), [SN("ldo"), I(3)])          # eight rows

    def epi_block_work(self, blocks, regs):
        """the per-block work as a list of (position in `blocks`, closure) in execution order: accumulator reads, 1 / l, four
        ds_write_b128, four ds_read_b128, and -- one block later, behind the counted wait for those reads -- four stores.
        blocks: (rb, db) in processing order; regs(i) -> (source registers, destination registers) of the i-th block."""
        cfg = self.cfg
        wa, ra, vo, inv = self.EPI_WA, self.EPI_RA, V(self.EPI_VO), self.EPI_INV
        work = []

        def stores(i, rb, db, dst, ids):
            out = [(i, lambda: self.lds_need(ids[-1]))]
            for k in range(4):
                if k == 0:
                    out.append((i, lambda: self.emit("s_add_u32", s("t0"), [s("row0"), SN("wave64")])))
                    if rb:
                        out.append((i, lambda: self.emit("s_add_u32", s("t0"), [s("t0"), I(32)])))
                    out.append((i, lambda: self.emit("s_mul_i32", s("t0"), [s("t0"), SN("ldo")])))
                else:
                    out.append((i, lambda: self.emit("s_add_u32", s("t0"), [s("t0"), s("t4")])))
                out.append((i, lambda: self.emit("v_add_u32_e64", vo, [VN("ov%d" % db), s("t0")], clamp=1)))
                if "epi_st" in cfg.abl:
                    continue
                if cfg.o16:
                    out.append((i, lambda k=k: self.emit("v_cvt_pk_%s_f32" % cfg.dtype, V(dst + 4 * k), [V(dst + 4 * k), V(dst + 4 * k + 1)])))
                    out.append((i, lambda k=k: self.emit("v_cvt_pk_%s_f32" % cfg.dtype, V(dst + 4 * k + 1), [V(dst + 4 * k + 2), V(dst + 4 * k + 3)])))
                    out.append((i, lambda k=k: self.emit("buffer_store_dwordx2", None, [V(dst + 4 * k, 2), vo, s("tres", 4)], offset=0)))
                else:
                    out.append((i, lambda k=k: self.emit("buffer_store_dwordx4", None, [V(dst + 4 * k, 4), vo, s("tres", 4)], offset=0, pol=cfg.stpol)))
            return out

        pending = None
        for i, (rb, db) in enumerate(blocks):
            src, dst = regs(i)
            b = 4 * rb + db
            for r in range(0 if "epi_valu" in cfg.abl else 16):
                work.append((i, lambda r=r, src=src, b=b: self.emit("v_accvgpr_read_b32", V(src + r), [A(O_BASE + 16 * b + r)])))
            if cfg.merge:       # the next block's first phase multiplied K Q^T beside this block's softmax: O is zeroed here
                for r in range(16):
                    work.append((i, lambda r=r, b=b: self.emit("v_accvgpr_write_b32", A(O_BASE + 16 * b + r), [I(0)])))
            for r in range(0 if (cfg.split or "epi_valu" in cfg.abl) else 16):
                work.append((i, lambda r=r, src=src, rb=rb: self.emit("v_mul_f32", V(src + r), [V(inv[rb]), V(src + r)])))
            for g in range(4):
                work.append((i, lambda g=g, src=src: self.lds_write("ds_write_b128", V(wa[g]), V(src + 4 * g, 4), 0)))
            ids = []
            if pending is not None and regs(i)[1] == pending[3]:     # one destination set: the previous block's stores go first
                work += stores(*pending)
                pending = None
            for k in range(4):
                work.append((i, lambda k=k, dst=dst, ids=ids, rb=rb, db=db: ids.append(
                    self.lds_read("ds_read_b128", V(dst + 4 * k, 4), V(ra), 1024 * k, note="O(%d,%d) rows %d.." % (rb, db, 8 * k)))))
            if pending is not None:
                work += stores(*pending)
            pending = (i, rb, db, dst, ids)
        work += stores(*pending)
        return work

    def epi_finish(self):
        cfg = self.cfg
        vo = V(self.EPI_VO)
        for rb in range(2):
            if cfg.split:     # (m, l) of the lane's row, two floats at 8 bytes per row (one 8-byte store: an instruction offset on top of an
                self.emit("s_add_u32", s("t0"), [s("row0"), SN("wave64")])   # out-of-range lane offset would wrap past 2^32 into the buffer)
                if rb:
                    self.emit("s_add_u32", s("t0"), [s("t0"), I(32)])
                self.emit("s_lshl_b32", s("t0"), [s("t0"), I(3)])
                self.emit("v_add_u32_e64", vo, [VN("lv"), s("t0")], clamp=1)
                pair = T_SW - 1      # v238:239 (T_MN + 1, T_SW: dead behind the loop); an EVEN-aligned pair (gfx950: 64-bit VGPR tuples)
                assert pair % 2 == 0
                self.emit("v_mov_b32", V(pair), [VN("m%d" % rb)])
                self.emit("v_mov_b32", V(pair + 1), [V(self.EPI_LTOT[rb])])
                self.emit("buffer_store_dwordx2", None, [V(pair, 2), vo, s("lres", 4)], offset=0)
                continue
            x = V(T_SW + rb)
            self.emit("v_log_f32", x, [V(self.EPI_LTOT[rb])])
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

    def epilogue_rows(self):
        """orow streams: O(rb, db) holds column 32 db + n of rows 8 (r >> 2) + 4 hi + (r & 3) in register r.  1 / l of the row per
        register (ds_bpermute_b32 from the lane that owns the row), then sixteen row stores per block: lane offset = 4 (32 db + n)
        + 4 hi ld(O) (out of range beyond D: `ov<db>`), scalar offset = the row's (part of the range check like the lane's)."""
        cfg = self.cfg
        inv, bpa = self.EPI_INV, V(T_CORR)
        fac = S_BASE[1] + 32                      # v128..v159: 1 / l per (rb, r)
        sets = [S_BASE[0] + 16 * i for i in range(4)] + [S_BASE[1], S_BASE[1] + 16]
        if not cfg.split:
            self.emit("v_mbcnt_lo_u32_b32", bpa, [I(-1), I(0)])
            self.emit("v_mbcnt_hi_u32_b32", bpa, [I(-1), bpa])
            self.emit("v_lshrrev_b32", bpa, [I(5), bpa])
            self.emit("v_lshlrev_b32", bpa, [I(4), bpa])               # 16 hi
            ids = []
            for rb in range(2):
                for r in range(16):
                    ids.append(self.lds_read("ds_bpermute_b32", V(fac + 16 * rb + r), bpa, 4 * (8 * (r >> 2) + (r & 3)), src=V(inv[rb])))
        self.emit("s_add_u32", s("t0"), [s("row0"), SN("wave64")])
        self.emit("s_mul_i32", s("t0"), [s("t0"), SN("ldo")])           # byte offset of the wave's first row
        self.emit("s_mul_i32", s("t4"), [SN("ldo"), I(5)])              # rows 3 -> 8 of a register group
        for i, (rb, db) in enumerate((b // 4, b % 4) for b in range(8)):
            t = sets[i % len(sets)]
            b = 4 * rb + db
            if rb and not db:
                self.emit("s_lshl_b32", s("t2"), [SN("ldo"), I(5)])
                self.emit("s_add_u32", s("t0"), [s("t0"), s("t2")])
            for r in range(16):
                self.emit("v_accvgpr_read_b32", V(t + r), [A(O_BASE + 16 * b + r)])
            if not cfg.split:
                if i == 0 or (rb and not db):
                    self.lds_need(ids[16 * rb + 15])
                for r in range(16):
                    self.emit("v_mul_f32", V(t + r), [V(fac + 16 * rb + r), V(t + r)])
            self.emit("s_mov_b32", s("t2"), [s("t0")])
            for r in range(16):
                self.emit("buffer_store_dword", None, [V(t + r), VN("ov%d" % db), s("tres", 4), s("t2")], offset=0, pol=cfg.stpol)
                if r != 15:
                    self.emit("s_add_u32", s("t2"), [s("t2"), s("t4") if (r & 3) == 3 else SN("ldo")])
        self.lds_flush()

    def epilogue(self):
        """O /= l (+Source.swift:165-171) and L = m + log2 l (+Caching.swift:373-377), straight from the registers"""
        cfg = self.cfg
        self.emit("s_nop", None, [I(15)], note="the last accumulating MFMAs leave the matrix pipe")
        self.emit("s_nop", None, [I(7)])
        self.epi_prepare()
        if cfg.orow:
            if "epi" not in cfg.abl:
                self.epilogue_rows()
            self.epi_finish()
            return
        # staging registers: four 16-register sets each way (an instruction reads its registers when it issues); the merged
        # block switch has the odd tiles' score registers only (the even ones hold the next block's tile 0): two sets
        if cfg.merge:
            regs = lambda i: (S_BASE[1] + 16 * (i & 1), S_BASE[1] + 32 + 16 * (i & 1))
        else:
            regs = lambda i: (S_BASE[0] + 16 * (i & 3), S_BASE[1] + 16 * (i & 3))
        if "epi" not in cfg.abl:     # (timing-only ablation: O stays in the registers -- what the staging and the stores of a block cost)
            for _, fn in self.epi_block_work([(b // 4, b % 4) for b in range(8)], regs):
                fn()
        self.lds_flush()
        self.epi_finish()

    def fused_tail(self):
        """dense streams, behind the last (odd) tile: softmax finish, then P V with the epilogue dealt out between the products"""
        cfg = self.cfg
        lastpar, par = 1, 0
        vids = self.phase_a(par, mfma=False, softmax=True, zero_o=False)       # softmax finish + V^T fragments 0..7 (ring)
        # V^T fragments 8..15 have no ring slot to wait for: they go to registers that are dead behind the last tile (FOLD: the -m
        # start blocks; exact-scale streams: their free registers and the rescale temporaries)
        spare = [CM_BASE + 4 * k for k in range(8)] if cfg.fold else [CM_BASE + 8 + 4 * k for k in range(6)] + [p4gen.T_RS, p4gen.T_RS + 4]
        if cfg.pprof:     # (the clock sums live in the exact-scale streams' free registers: the fragments take half of the staging area)
            spare = [S_BASE[0] + 32 + 4 * k for k in range(8)]
        for f in range(8, 16):
            u, db = divmod(f, 4)
            for h in range(2):
                self.lds_read("ds_read_b64_tr_b16", V(spare[f - 8] + 2 * h, 2), V(p4gen.T_VADDR), (db * 64 + 16 * u) * 64 + h * 8 * 64,
                              note="V^T f%d.%d" % (f, h))
        self.lds_flush()
        for rb in range(2):
            self.emit("v_add_f32", VN("l%d" % rb), [V(self.r_lb[rb]), VN("l%d" % rb)])
        self.emit("s_barrier")       # every wave is done with the V image O is staged in (last read in phase B(nt-1))
        self.epi_prepare()
        regs = lambda i: (S_BASE[0] + 16 * (i & 1), S_BASE[0] + 32 + 16 * (i & 1))   # the even tiles' score registers are free
        if cfg.pprof:
            regs = lambda i: (S_BASE[0], S_BASE[0] + 16)
        blocks = [(rb, db) for db in range(4) for rb in range(2)]                     # completion order
        work = self.epi_block_work(blocks, regs)
        # products in (db, u, rb) order: block (rb, db) is complete behind product 8 db + 6 + rb; its work may start two products later
        fill = [[] for _ in range(32)]
        for db in range(3):
            mine = [fn for i, fn in work if i // 2 == db]
            gaps = list(range(8 * (db + 1) + 1, 8 * (db + 2)))
            for n, fn in enumerate(mine):
                fill[gaps[n * len(gaps) // len(mine)]].append(fn)
        rest = [fn for i, fn in work if i // 2 == 3]
        self.emit("s_nop", None, [I(1)], note="freshly packed P -> MFMA operand")
        for g in range(32):
            db, u, rb = g // 8, (g % 8) // 2, g % 2
            f = 4 * u + db
            afrag = p4gen.vf_frag(f) if f < 8 else V(spare[f - 8], 4)
            self.mfma(p4gen.o_acc(rb, db), afrag, p4gen.p_frag(lastpar, rb, u), p4gen.o_acc(rb, db))
            for fn in fill[g]:
                fn()
        self.pstamp("tail")
        self.emit("s_nop", None, [I(15)], note="the last accumulating MFMAs leave the matrix pipe")
        self.emit("s_nop", None, [I(7)])
        for fn in rest:
            fn()
        self.lds_flush()
        self.epi_finish()

    def emit_outofline(self):
        mine = [x for x in self.outofline if x[0] in ("ksw", "vsw", "qsw")]
        self.outofline = [x for x in self.outofline if x[0] not in ("ksw", "vsw", "qsw")]
        Stream.emit_outofline(self)
        for kind, lbl, back, par, _ in mine:
            self.label(lbl)
            self.emit("s_cmp_eq_u32", None, [s("hasnext"), I(0)])
            self.emit("s_cbranch_scc1", None, [], target=back)      # last block: the ring runs ahead into zeros (out of range)
            if kind == "ksw":
                self.switch_k()
                if self.cfg.qearly:
                    self.issue_q((T_SW, T_SW + 1, T_TL, T_TL + 1, self.r_maskv))
            elif kind == "vsw":
                self.switch_v()
                if not self.cfg.merge and not self.cfg.qearly:
                    self.issue_q((T_SW, T_SW + 1, T_TL, T_TL + 1, self.r_maskv))
            else:
                self.issue_q((T_SW, T_SW + 1, T_TL, T_TL + 1, self.r_maskv))
            self.emit("s_branch", None, [], target=back)

    def block_head(self, nonext):
        """the block whose operands were requested under the previous one becomes the current one; read the table entry after it"""
        cfg = self.cfg
        for name in ("ob", "lb"):
            self.emit("s_mov_b32", s(name, 1, 0), [s(name + "n", 1, 0)])
            self.emit("s_mov_b32", s(name, 1, 1), [s(name + "n", 1, 1)])
        self.emit("s_mov_b32", s("row0"), [s("row0n")])
        if cfg.causal:
            # rows / last key / key tiles / diagonal offset of THIS block's batch entry (the *n registers still hold its table entry):
            # row r sees key c iff c <= r + max(C - R, 0) (include/mfa.h: with per-batch lengths the offset is each entry's own, clamped
            # at 0); without the causal mask the offset is out of reach of any row and the same arithmetic yields the dense geometry
            self.emit("s_mov_b32", s("rr"), [s("rrn")])
            self.emit("s_sub_u32", s("cm1"), [s("ccn"), I(1)])
            self.emit("s_add_u32", s("ttot"), [s("ccn"), I(63)])
            self.emit("s_lshr_b32", s("ttot"), [s("ttot"), I(6)])
            self.emit("s_max_u32", s("ttot"), [s("ttot"), I(1)])       # (an entry without keys: one fully masked tile pair, O = 0)
            self.emit("s_sub_u32", s("coff"), [s("ccn"), s("rrn")])
            self.emit("s_max_i32", s("coff"), [s("coff"), I(0)])
            self.emit("s_cmp_eq_u32", None, [SN("cflag"), I(0)])
            self.emit("s_cselect_b32", s("coff"), [I(0x40000000), s("coff")])
            self.block_geometry()
            if cfg.fastloop:
                self.fast_end()
            # waves that skipped tiles did not walk the V read pointer: V(-1)'s image is the one before V(0)'s
            self.emit("s_sub_u32", s("t0"), [s("vwr"), SN("ldsv")])
            self.emit("s_sub_u32", s("t0"), [s("t0"), I(VSLOT)])
            self.emit("s_add_u32", s("t2"), [s("t0"), I(VRING * VSLOT)])
            self.emit("s_cmp_lt_i32", None, [s("t0"), I(0)])
            self.emit("s_cselect_b32", s("vrd"), [s("t2"), s("t0")])
        self.emit("s_add_u32", s("blk"), [s("blk"), I(1)])
        self.emit("s_mov_b32", s("hasnext"), [I(0)])
        self.emit("s_cmp_ge_u32", None, [s("blk"), SN("nblk")])
        self.emit("s_cbranch_scc1", None, [], target=nonext)
        self.emit("s_mov_b32", s("hasnext"), [I(1)])
        self.load_next()
        self.label(nonext)

    def block_init(self):
        cfg = self.cfg
        for rb in range(2):
            self.emit("v_mov_b32", V(self.r_lb[rb]), [I(0)])
            self.emit("v_mov_b32", V(T_CORR + rb), [F(1.0)])
            self.emit("v_mov_b32", VN("l%d" % rb), [I(0)])
            self.emit("v_mov_b32", VN("m%d" % rb), [F(0.0) if cfg.fold else F(-3.402823466e+38)])
        if cfg.fold:
            for r in range(32):
                self.emit("v_mov_b32", V(CM_BASE + r), [I(0)])
        self.emit("s_mov_b32", SN("pend"), [I(0)])
        self.emit("s_mov_b32", SN("j"), [I(0)])

    def fast_end(self):
        """fend = the largest odd number <= min(maskfrom, nt - 2): tiles 1 .. fend - 1 run in the branch-free loop (causal streams:
        per block and per wave -- its first masked tile, and its own traversal bound for a wave beyond the last row)"""
        self.emit("s_sub_u32", s("fend"), [SN("nt"), I(2)])
        self.emit("s_min_i32", s("fend"), [s("fend"), SN("maskfrom")])
        if self.cfg.causal:
            self.emit("s_min_i32", s("fend"), [s("fend"), SN("wnt")])
        self.emit("s_sub_u32", s("fend"), [s("fend"), I(1)])
        self.emit("s_or_b32", s("fend"), [s("fend"), I(1)])

    def fast_pair(self, cfg):
        """two tiles (parities 1, 0) of the branch-free loop"""
        self.fast = True
        for par in (1, 0):
            vids = self.phase_a(par, mfma=True, softmax=True, zero_o=False)
            self.lds_flush()
            self.pstamp("loop_a")
            if "bar" not in cfg.abl:
                if "vm" not in cfg.abl:
                    self.emit("s_waitcnt", None, [], vmcnt=0)
                if cfg.pprof == 2:
                    self.pstamp("loop_vm")
                self.emit("s_barrier")
            self.pstamp("loop_bar" if cfg.pprof == 2 else "loop_wait")
            self.phase_b(par, mfma=True, softmax=True, vids=vids)
            self.pstamp("loop_b")
            self.emit("s_add_u32", SN("j"), [SN("j"), I(1)])
        self.fast = False

    # ------------------------------------------------------------ whole stream
    def build(self):
        cfg = self.cfg
        self.outofline = []
        self.xe_pending = []
        self.first_tiles = False
        blk_lbl, loop, end_even, end_odd, done, fin, nonext, after_epi = (
            self.newlabel(x) for x in ("BLOCK", "LOOP", "ENDEVEN", "ENDODD", "DONE", "FIN", "NONEXT", "AFTEREPI"))
        # ---- once per workgroup
        if cfg.pad:     # developer streams: every instruction behind it moves by four bytes (MI355X_MICROARCH: code placement)
            self.emit("s_nop", None, [I(0)], note="code placement pad")
        for ks in range(8):
            self.emit("v_xor_b32", V(T_KADDR + ks), [I(ks << 5), VN("kbase")])
        if not cfg.causal:
            self.emit("s_sub_u32", s("ntm1"), [SN("nt"), I(1)])
            self.emit("s_sub_u32", s("ntm2"), [SN("nt"), I(2)])
        if cfg.merge:
            self.emit("s_sub_u32", s("t0"), [SN("nt"), I(3)])
            self.emit("s_max_i32", s("qswj"), [s("t0"), I(0)])   # tile whose phase B requests the next block's Q
        if cfg.fastloop and not cfg.causal:
            self.fast_end()
        self.emit("s_mov_b32", s("vrd"), [I(2 * VSLOT)])     # "image of V(-1)"
        self.emit("s_mov_b32", s("vwr"), [SN("ldsv")])       # V(0) goes to image 0
        self.emit("s_add_u32", s("t1"), [SN("ldsv"), I(VRING * VSLOT)])
        # scalar parts of the LDS-DMA start offsets: K piece i begins at row 16 wave + 4 i, V piece i at key 16 i
        self.emit("s_lshr_b32", s("t0"), [SN("kinc"), I(6)])                 # 2 ld(K)
        self.emit("s_lshr_b32", s("t2"), [SN("wave64"), I(2)])
        self.emit("s_mul_i32", s("kc0"), [s("t0"), s("t2")])
        self.emit("s_lshl_b32", s("kstep"), [s("t0"), I(2)])                 # four rows of K
        self.emit("s_lshr_b32", s("vstep"), [SN("vinc"), I(2)])              # sixteen keys of V
        self.emit("s_lshl_b32", s("q4"), [SN("ldq2"), I(2)])
        for name, nrec in (("kres", "nreck"), ("vres", "nrecv")):
            if not cfg.causal:     # (causal streams: per block, switch_k / switch_v)
                self.emit("s_mov_b32", s(name, 1, 2), [SN(nrec)])
            self.emit("s_mov_b32", s(name, 1, 3), [I(DESC_FLAGS)])
        # ---- first block: its Q, K(0), V(0), K(1) are requested here; later blocks find theirs requested by their predecessor
        self.emit("s_mov_b32", s("blk"), [I(0)])
        self.load_next()
        self.issue_q((S_BASE[0] + 0, S_BASE[0] + 1, S_BASE[0] + 2, S_BASE[0] + 3))
        if cfg.soff:
            self.switch_k(init=True)
            self.switch_v(init=True)
        self.switch_k()
        self.issue_tile("k", 0)
        self.switch_v()
        self.issue_tile("v", 0)
        self.issue_tile("k", 1)
        self.emit("s_waitcnt", None, [], vmcnt=0)
        # ================= block loop =================
        if cfg.pprof == 1:
            for i in range(16):
                self.emit("v_mov_b32", V(PROF_ACC + i), [I(0)])
        elif cfg.pprof == 2:
            for name, r in SPROF_ACC.items():
                if name != "blocks":
                    self.emit("s_mov_b32", SR(r), [I(0)])
        self.label(blk_lbl)
        self.pstamp("table", first=True)
        self.block_head(nonext)
        self.pstamp("table")
        # this wave's Q image, and its pieces of K(0), V(0), K(1): everything older than the previous block's stores
        if not cfg.orow:
            self.emit("s_waitcnt", None, [], vmcnt=NST)
        self.pstamp("wait_q")
        if "qfrag" not in cfg.abl:
            self.q_fragments()
        self.emit("s_barrier")
        self.pstamp("qfrag")
        self.block_init()
        for i in range(16):
            self.k_read(0, i)
        self.lds_flush()
        self.first_tiles = True
        self.phase_a(0, mfma=True, softmax=False, zero_o=True)
        if not cfg.orow:
            self.emit("s_waitcnt", None, [], vmcnt=NST)          # K(1) (older than the stores)
        self.emit("s_barrier")
        self.pstamp("tile0_a")
        self.phase_b(0, mfma=False, softmax=True, vids={})
        self.first_tiles = False
        self.emit("s_mov_b32", SN("j"), [I(1)])
        self.pstamp("tile0_b")
        self.label(loop)
        if cfg.fastloop:
            # (j is odd here: the tile about to run has parity 1)
            slow, fast = self.newlabel("SLOW"), self.newlabel("FAST")
            self.emit("s_sub_u32", s("fcnt"), [s("fend"), SN("j")])
            self.emit("s_cmp_lt_i32", None, [s("fcnt"), I(2)])
            self.emit("s_cbranch_scc1", None, [], target=slow)
            self.emit("s_lshr_b32", s("fcnt"), [s("fcnt"), I(cfg.fastloop)])     # iterations of fastloop tile pairs ...
            if cfg.fastloop == 2:   # ... (an odd number of pairs: the first pair on its own)
                odd = self.newlabel("FASTODD")
                self.emit("s_sub_u32", s("t3"), [s("fend"), SN("j")])
                self.emit("s_bitcmp1_b32", None, [s("t3"), I(1)])
                self.emit("s_cbranch_scc0", None, [], target=fast)
                self.fast_pair(cfg)
                self.emit("s_cmp_eq_u32", None, [s("fcnt"), I(0)])
                self.emit("s_cbranch_scc1", None, [], target=slow)
            if cfg.align:
                self.emit("align", None, [I(6)])
            self.label(fast)
            for _ in range(cfg.fastloop):
                self.fast_pair(cfg)
            self.emit("s_sub_u32", s("fcnt"), [s("fcnt"), I(1)])
            self.emit("s_cmp_lg_u32", None, [s("fcnt"), I(0)])
            self.emit("s_cbranch_scc1", None, [], target=fast)
            self.label(slow)
        for par, endl in ((1, end_even), (0, end_odd)):
            self.emit("s_cmp_ge_i32", None, [SN("j"), SN("wnt")])     # (dense streams: wnt = nt)
            self.emit("s_cbranch_scc1", None, [], target=endl)
            vids = self.phase_a(par, mfma=True, softmax=True, zero_o=False)
            self.lds_flush()
            self.pstamp("loop_a")
            if "bar" not in cfg.abl:
                if "vm" not in cfg.abl:
                    self.emit("s_waitcnt", None, [], vmcnt=0)        # this wave's pieces of K(j+1) and V(j) (and, in tile 1, the stores)
                if cfg.pprof == 2:
                    self.pstamp("loop_vm")
                self.emit("s_barrier")
            self.pstamp("loop_bar" if cfg.pprof == 2 else "loop_wait")
            if cfg.pprof == 1 and par == 1:                      # the wait of tile 1 separately (it includes the previous block's stores)
                self.emit("s_cmp_eq_u32", None, [SN("j"), I(1)])
                self.emit("s_cselect_b32", s("t3"), [s("t3"), I(0)])
                acc = V(PROF_ACC + PROF_NAMES.index("tile1_wait"))
                self.emit("v_add_u32", acc, [s("t3"), acc])
            self.phase_b(par, mfma=True, softmax=True, vids=vids)
            self.pstamp("loop_b")
            self.emit("s_add_u32", SN("j"), [SN("j"), I(1)])
        self.emit("s_branch", None, [], target=loop)
        skip_odd, skip_even = self.newlabel("SKIPODD"), self.newlabel("SKIPEVEN")
        for lastpar, lbl, nxt in ((0, end_even, skip_odd), (1, end_odd, skip_even)):
            self.label(lbl)
            if cfg.merge and lastpar == 1:
                # A block ends behind an odd tile (even count).  With a block to follow, the softmax finish of that tile runs
                # beside the NEXT block's first K Q^T products (its Q arrived three tiles ago, K'(0)'s fragments were fetched by
                # phase B(nt-1) like any K(j+1)'s): the drain of this block's pipeline and the fill of the next share a phase
                plain = self.newlabel("PLAINTAIL")
                self.emit("s_cmp_eq_u32", None, [s("hasnext"), I(0)])
                self.emit("s_cbranch_scc1", None, [], target=plain)
                if cfg.fold:
                    for r in range(32):
                        self.emit("v_mov_b32", V(CM_BASE + r), [I(0)])
                self.q_fragments(qa=T_MX, tmp=(T_TL, T_TL + 1))
                vids = self.phase_a(0, mfma=True, softmax=True, zero_o=False)
                self.lds_flush()
                self.emit("s_nop", None, [I(1)], note="freshly packed P -> MFMA operand")
                self.phase_b(0, mfma=True, softmax=False, vids=vids)
                self.emit("s_branch", None, [], target=done)
                self.label(plain)
            if cfg.fuse and lastpar == 1:       # (a dense block ends behind an odd tile: its count is even)
                self.fused_tail()
                self.emit("s_branch", None, [], target=after_epi)
                continue
            vids = self.phase_a(lastpar ^ 1, mfma=False, softmax=True, zero_o=False)
            self.lds_flush()
            self.emit("s_nop", None, [I(1)], note="freshly packed P -> MFMA operand")
            self.phase_b(lastpar ^ 1, mfma=True, softmax=False, vids=vids)
            self.emit("s_branch", None, [], target=nxt if cfg.causal else done)
        if cfg.causal:
            # a wave whose rows end before the block's last tile still owes the others its barriers and its LDS-DMA pieces
            for par, lbl in ((1, skip_odd), (0, skip_even)):
                self.label(lbl)
                self.emit("s_cmp_ge_i32", None, [SN("j"), SN("nt")])
                self.emit("s_cbranch_scc1", None, [], target=done)
                self.skip_tile(par)
                if par == 0:
                    self.emit("s_branch", None, [], target=skip_odd)
        self.label(done)
        if cfg.merge or cfg.orow:    # (orow: 130 stores per block are beyond a counted wait: the next block's operands land HERE)
            # a merged switch has no barrier between this block's end and the next block's phase B(0), which reads K'(1): every
            # wave's pieces of the next block's first tiles (requested two phases ago) land before the barrier below
            self.emit("s_waitcnt", None, [], vmcnt=0)
        self.emit("s_barrier")       # every wave is done with the V image the epilogue stages O in (last read in phase B(nt-1))
        self.pstamp("tail")
        for rb in range(2):
            self.emit("v_add_f32", VN("l%d" % rb), [V(self.r_lb[rb]), VN("l%d" % rb)])
        self.epilogue()
        if cfg.fuse:
            self.label(after_epi)
        self.pstamp("epilogue")
        if cfg.pprof == 1:
            acc = V(PROF_ACC + PROF_NAMES.index("blocks"))
            self.emit("v_add_u32", acc, [I(1), acc])

        post = self.newlabel("POST")
        self.emit("s_cmp_eq_u32", None, [s("hasnext"), I(1)])
        self.emit("s_cbranch_scc1", None, [], target=post if cfg.merge else blk_lbl)
        self.emit("s_waitcnt", None, [], vmcnt=0)
        if cfg.pprof:   # lane 0 leaves the sums in O[first row of the wave's last block][0:16] (fp32 O)
            self.emit("s_add_u32", s("t0"), [s("row0"), SN("wave64")])
            self.emit("s_mul_i32", s("t0"), [s("t0"), SN("ldo")])
            self.emit("v_mov_b32", V(T_TL), [s("t0")])
            pacc = PROF_ACC
            if cfg.pprof == 2:   # (the score registers are dead behind the last block)
                pacc = S_BASE[0]
                for i in range(15):
                    self.emit("v_mov_b32", V(pacc + i), [SR(SPROF_ACC[SPROF_NAMES[i]]) if i < len(SPROF_NAMES) else I(0)])
            self.emit("v_mov_b32", V(pacc + 15), [I(PROF_MAGIC)])
            self.emit("s_mov_b64", ("exec",), [I(1)])
            for i in range(4):
                self.emit("buffer_store_dwordx4", None, [V(pacc + 4 * i, 4), V(T_TL), s("tres", 4)], offset=16 * i)
            self.emit("s_mov_b64", ("exec",), [I(-1)])
            self.emit("s_waitcnt", None, [], vmcnt=0)
        self.emit("s_branch", None, [], target=fin)
        if cfg.merge:
            # behind a merged block switch: tile 0 of the new block has its scores; its softmax start follows the old block's epilogue
            self.label(post)
            nonext2 = self.newlabel("NONEXT")
            self.pstamp("table", first=True)
            self.block_head(nonext2)
            self.pstamp("table")
            self.block_init()
            self.phase_b(0, mfma=False, softmax=True, vids={})   # (K(1) landed before the barrier in front of the epilogue)
            self.emit("s_mov_b32", SN("j"), [I(1)])
            self.pstamp("tile0_b")
            self.emit("s_branch", None, [], target=loop)
        self.emit_outofline()
        self.label(fin)
        return self.finish()


# ---------------------------------------------------------------- rendering
def render_one(ins):
    op, m = ins.op, ins.mod
    f = p4gen.fmt
    if op in ("buffer_store_dwordx4", "buffer_store_dwordx2", "buffer_store_dword", "buffer_store_short"):
        off = " offset:%d" % m["offset"] if m.get("offset") else ""
        return "%s %s, %s, %s, %s offen%s%s" % (op, f(ins.s[0]), f(ins.s[1]), f(ins.s[2]), f(ins.s[3]) if len(ins.s) > 3 else "0", off, m.get("pol", ""))
    if op == "v_fma_f32" and not m.get("neg2"):
        return "v_fma_f32 %s, %s%s, %s, %s" % (f(ins.d), "-" if m.get("neg0") else "", f(ins.s[0]), f(ins.s[1]), f(ins.s[2]))
    if op == "s_cmp_lt_u32":
        return "s_cmp_lt_u32 %s, %s" % (f(ins.s[0]), f(ins.s[1]))
    if op == "s_memtime":
        return "s_memtime %s" % f(ins.d)
    if op == "align":
        return ".p2align %d" % ins.s[0][1]
    if op in ("s_cmp_lg_u32", "s_bitcmp1_b32", "s_cmp_gt_u32"):
        return "%s %s, %s" % (op, f(ins.s[0]), f(ins.s[1]))
    if op == "s_mov_b64" and ins.d == ("exec",):
        return "s_mov_b64 exec, %s" % f(ins.s[0])
    if op == "ds_write_b128":
        return "ds_write_b128 %s, %s offset:%d" % (f(ins.s[0]), f(ins.s[1]), m["offset"])
    if op in ("v_cvt_f32_f16", "v_cvt_f16_f32", "v_rcp_f32", "v_log_f32"):
        return "%s_e32 %s, %s" % (op, f(ins.d), f(ins.s[0]))
    return p4gen.render_one(ins)


def render(instrs):
    return [render_one(i) for i in instrs]


VARIANTS = {
    # name: cfg            (X-macro columns: folds the scale, 16-bit O, FP16 L)
    "BF16_FOLD_L16": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, fastloop=1, align=1),          # headline: mixed-precision mode, fp32 O, FP16 L
    "BF16_FOLD_O16_L16": PCfg("bf16", 8, fold=1, o16=1, l16=1, bal=2, xb=48, fastdec=1, fastloop=1, align=1),
    "BF16_EXACT": PCfg("bf16", 8, fold=0, bal=2, xe=32, cap=8, fastloop=1, align=1),                    # lowPrecisionInputs only: scale in fp32, fp32 O and L
    "BF16_EXACT_O16": PCfg("bf16", 8, fold=0, o16=1, bal=2, xe=32, cap=8, fastloop=1, align=1),
    "F16_FOLD_L16": PCfg("f16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, fastloop=1, align=1),
    "F16_FOLD_O16_L16": PCfg("f16", 8, fold=1, o16=1, l16=1, bal=2, xb=48, fastdec=1, fastloop=1, align=1),
    "F16_EXACT": PCfg("f16", 8, fold=0, bal=2, xe=32, cap=8, fastloop=1, align=1),
    "F16_EXACT_O16": PCfg("f16", 8, fold=0, o16=1, bal=2, xe=32, cap=8, fastloop=1, align=1),
    "BF16_FOLD_SPLIT": PCfg("bf16", 8, fold=1, bal=2, xb=48, fastdec=1, fastloop=1, align=1, split=1),   # column-parallel pieces: un-normalised O and (m, l) into the workspace
    "BF16_EXACT_SPLIT": PCfg("bf16", 8, fold=0, bal=2, xe=32, cap=8, fastloop=1, align=1, split=1),
    "F16_FOLD_SPLIT": PCfg("f16", 8, fold=1, bal=2, xb=48, fastdec=1, fastloop=1, align=1, split=1),
    "F16_EXACT_SPLIT": PCfg("f16", 8, fold=0, bal=2, xe=32, cap=8, fastloop=1, align=1, split=1),
    "BF16_FOLD_L16_CAUSAL": PCfg("bf16", 8, fold=1, l16=1, causal=1, bal=2, xb=48, fastdec=1, fastloop=1, align=1),
    "BF16_FOLD_O16_L16_CAUSAL": PCfg("bf16", 8, fold=1, o16=1, l16=1, causal=1, bal=2, xb=48, fastdec=1, fastloop=1, align=1),
    "BF16_EXACT_CAUSAL": PCfg("bf16", 8, fold=0, causal=1, bal=2, xe=32, cap=8, fastloop=1, align=1),
    "BF16_EXACT_O16_CAUSAL": PCfg("bf16", 8, fold=0, o16=1, causal=1, bal=2, xe=32, cap=8, fastloop=1, align=1),
    "F16_FOLD_L16_CAUSAL": PCfg("f16", 8, fold=1, l16=1, causal=1, bal=2, xb=48, fastdec=1, fastloop=1, align=1),
    "F16_FOLD_O16_L16_CAUSAL": PCfg("f16", 8, fold=1, o16=1, l16=1, causal=1, bal=2, xb=48, fastdec=1, fastloop=1, align=1),
    "F16_EXACT_CAUSAL": PCfg("f16", 8, fold=0, causal=1, bal=2, xe=32, cap=8, fastloop=1, align=1),
    "F16_EXACT_O16_CAUSAL": PCfg("f16", 8, fold=0, o16=1, causal=1, bal=2, xe=32, cap=8, fastloop=1, align=1),
    "BF16_EXACT_PROF": PCfg("bf16", 8, fold=0, pprof=1),      # developer builds only (tools/p4p_prof.py)
    "BF16_FOLD_L16_MERGE": PCfg("bf16", 8, fold=1, l16=1, merge=1),   # developer builds only: merged block switch (lost)
    "BF16_FOLD_L16_FUSE": PCfg("bf16", 8, fold=1, l16=1, fuse=1),     # developer builds only: epilogue dealt out under the last P V (no gain)
    "R4_BF16_FOLD_L16": PCfg("bf16", 8, fold=1, l16=1),        # the round-4 schedule of the headline stream (A/B baseline)
    "R5_BF16_FOLD_L16": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1),   # the round-5 product stream (per-tile tests in the loop)
    "R5_BF16_EXACT": PCfg("bf16", 8, fold=0, bal=2, xe=32, cap=8),
    "R4_BF16_EXACT": PCfg("bf16", 8, fold=0),
    # round 5, developer builds: slot-balanced phases (p4gen Cfg.bal) and their timing-only ablations (WRONG RESULTS)
    "BF16_FOLD_L16_BAL32": PCfg("bf16", 8, fold=1, l16=1, bal=1, xb=32),
    "BF16_FOLD_L16_BAL40": PCfg("bf16", 8, fold=1, l16=1, bal=1, xb=40),
    "BF16_FOLD_L16_BAL24": PCfg("bf16", 8, fold=1, l16=1, bal=1, xb=24),
    "BF16_FOLD_L16_BAL32C6": PCfg("bf16", 8, fold=1, l16=1, bal=1, xb=32, cap=6),
    "BF16_FOLD_L16_BAL32C8": PCfg("bf16", 8, fold=1, l16=1, bal=1, xb=32, cap=8),
    "BF16_FOLD_L16_BAL2_32": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=32),
    "BF16_FOLD_L16_BAL2_40": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=40),
    "BF16_FOLD_L16_BAL2_24": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=24),
    "BF16_FOLD_L16_BAL2_32_PAD": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=32, pad=1),
    "ABL_BAL2_EXP": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=32, abl=("exp",)),
    "ABL_BAL2_MAX": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=32, abl=("max",)),
    "ABL_BAL2_SUMPACK": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=32, abl=("sum", "pack")),
    "ABL_BAL2_LDS": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=32, abl=("lds",)),
    "ABL_BAL2_DMA": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=32, abl=("dma",)),
    "ABL_BAL2_BAR": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=32, abl=("bar",)),
    "ABL_BAL2_ALL": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=32, abl=("exp", "max", "sum", "pack", "lds", "dma")),
    "BAL2_40_FD": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=40, fastdec=1),
    "BAL2_48_FD": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1),
    "BAL2_48_FD1": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, fdpos=1),
    "BAL2_48_FD2": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, fdpos=2),
    "BAL2_56_FD": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=56, fastdec=1),
    "BAL2_48": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48),
    "BAL2_56": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=56),
    "BAL2_40_C8": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=40, cap=8),
    "BAL2_48_C8": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, cap=8),
    "BAL2_40_C6": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=40, cap=6),
    "BAL2_40_MB": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=40, maxa=0),
    "BAL2_48_MB": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, maxa=0),
    "BAL2_32_MB": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=32, maxa=0),
    "BAL2_40_MB_V16": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=40, maxa=0, va0=16),
    "BAL2_40_MB_V8": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=40, maxa=0, va0=8),
    "BAL2_40_PAD": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=40, pad=1),
    "EXACT_BAL2_XE16": PCfg("bf16", 8, fold=0, bal=2, xe=16),
    "EXACT_BAL2_XE24": PCfg("bf16", 8, fold=0, bal=2, xe=24),
    "EXACT_BAL2_XE32": PCfg("bf16", 8, fold=0, bal=2, xe=32),
    "EXACT_BAL2_XE24_C8": PCfg("bf16", 8, fold=0, bal=2, xe=24, cap=8),
    "EXACT_BAL2_XE32_C8": PCfg("bf16", 8, fold=0, bal=2, xe=32, cap=8),
    "EXACT_BAL2_XE40_C8": PCfg("bf16", 8, fold=0, bal=2, xe=40, cap=8),
    # round 6, developer builds: the product schedule stamped per phase (scalar accumulators) and its timing-only ablations (WRONG RESULTS)
    "BF16_FOLD_L16_SPROF": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, pprof=2),
    "ABL6_EXP": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, abl=("exp",)),
    "ABL6_MAX": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, abl=("max",)),
    "ABL6_SUMPACK": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, abl=("sum", "pack")),
    "ABL6_SUM": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, abl=("sum",)),
    "ABL6_LDS": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, abl=("lds",)),
    "ABL6_DMA": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, abl=("dma",)),
    "ABL6_BAR": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, abl=("bar",)),
    "ABL6_VM": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, abl=("vm",)),
    "ABL6_CTL": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, abl=("ctl",)),
    "ABL6_OFFS": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, abl=("offs",)),
    "ABL6_VALU": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, abl=("exp", "max", "sum", "pack")),
    "ABL6_ALL": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, abl=("exp", "max", "sum", "pack", "lds", "dma")),
    "ABL6_ALL_BAR": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, abl=("exp", "max", "sum", "pack", "lds", "dma", "bar", "ctl")),
    "BF16_FOLD_L16_OROW": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, fastloop=1, align=1, orow=1),
    "BF16_EXACT_OROW": PCfg("bf16", 8, fold=0, bal=2, xe=32, cap=8, fastloop=1, align=1, orow=1),
    "BF16_FOLD_L16_CAUSAL_OROW": PCfg("bf16", 8, fold=1, l16=1, causal=1, bal=2, xb=48, fastdec=1, fastloop=1, align=1, orow=1),
    "BF16_FOLD_L16_ST_NT": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, fastloop=1, align=1, stpol=" nt"),
    "BF16_FOLD_L16_ST_SC1": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, fastloop=1, align=1, stpol=" sc1"),
    "BF16_FOLD_L16_ST_SC01": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, fastloop=1, align=1, stpol=" sc0 sc1"),
    "BF16_FOLD_L16_ST_SC1NT": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, fastloop=1, align=1, stpol=" sc1 nt"),
    "BF16_FOLD_L16_OROW_ST_NT": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, fastloop=1, align=1, orow=1, stpol=" nt"),
    "BF16_FOLD_L16_CAUSAL_ST_NT": PCfg("bf16", 8, fold=1, l16=1, causal=1, bal=2, xb=48, fastdec=1, fastloop=1, align=1, stpol=" nt"),
    "BF16_FOLD_L16_QE": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, fastloop=1, align=1, qearly=1),
    "BF16_EXACT_QE": PCfg("bf16", 8, fold=0, bal=2, xe=32, cap=8, fastloop=1, align=1, qearly=1),
    "BF16_FOLD_L16_CAUSAL_QE": PCfg("bf16", 8, fold=1, l16=1, causal=1, bal=2, xb=48, fastdec=1, fastloop=1, align=1, qearly=1),
    "ABL7_EPI_ST": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, fastloop=1, align=1, abl=("epi_st",)),
    "ABL7_EPI_VALU": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, fastloop=1, align=1, abl=("epi_valu",)),
    "ABL7_EPI_ST_VALU": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, fastloop=1, align=1, abl=("epi_st", "epi_valu")),
    "ABL7_EPI": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, fastloop=1, align=1, abl=("epi",)),
    "ABL7_EPI_QFRAG": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, fastloop=1, align=1, abl=("epi", "qfrag")),
    "ABL7_EPI_CAUSAL": PCfg("bf16", 8, fold=1, l16=1, causal=1, bal=2, xb=48, fastdec=1, fastloop=1, align=1, abl=("epi",)),
    "BF16_FOLD_L16_FL1": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, fastloop=1),
    "BF16_FOLD_L16_FL2": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, fastloop=2),
    "BF16_FOLD_L16_FL1_AL": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, fastloop=1, align=1),
    "BF16_FOLD_L16_FL1_SPROF": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, fastloop=1, pprof=2),
    "BF16_EXACT_FL1": PCfg("bf16", 8, fold=0, bal=2, xe=32, cap=8, fastloop=1),
    "BF16_FOLD_L16_NT": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, fastloop=1, align=1, dmapol=" nt"),
    "BF16_FOLD_L16_SC1": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, fastloop=1, align=1, dmapol=" sc1"),
    "BF16_FOLD_L16_SC0": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, fastloop=1, align=1, dmapol=" sc0"),
    "BF16_FOLD_L16_SC1NT": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, fastloop=1, align=1, dmapol=" sc1 nt"),
    "BF16_FOLD_L16_SOFF": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, soff=1),
    "BF16_FOLD_L16_PKS": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, pksum=1),
    "BF16_FOLD_L16_FL1_SOFF": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, fastloop=1, soff=1),
    "BF16_FOLD_L16_FL1_SOFF_PKS": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, fastloop=1, soff=1, pksum=1),
    "BF16_FOLD_L16_FL2_SOFF": PCfg("bf16", 8, fold=1, l16=1, bal=2, xb=48, fastdec=1, fastloop=2, soff=1),
    "BF16_EXACT_FL1_SOFF": PCfg("bf16", 8, fold=0, bal=2, xe=32, cap=8, fastloop=1, soff=1),
    "BF16_FOLD_L16_CAUSAL_DM": PCfg("bf16", 8, fold=1, l16=1, causal=1, bal=2, xb=48, fastdec=1, fastloop=1, align=1, diagmask=1),
    "BF16_EXACT_CAUSAL_DM": PCfg("bf16", 8, fold=0, causal=1, bal=2, xe=32, cap=8, fastloop=1, align=1, diagmask=1),
    "R5_BF16_FOLD_L16_CAUSAL": PCfg("bf16", 8, fold=1, l16=1, causal=1, bal=2, xb=48, fastdec=1),   # the round-5 causal stream (A/B baseline)
    "BF16_FOLD_L16_PAD": PCfg("bf16", 8, fold=1, l16=1, pad=1),
    "BF16_FOLD_L16_BAL32_PAD": PCfg("bf16", 8, fold=1, l16=1, bal=1, xb=32, pad=1),
    "BF16_FOLD_L16_BAL40_PAD": PCfg("bf16", 8, fold=1, l16=1, bal=1, xb=40, pad=1),
    "ABL_BAL32_EXP": PCfg("bf16", 8, fold=1, l16=1, bal=1, xb=32, abl=("exp",)),
    "ABL_BAL32_MAX": PCfg("bf16", 8, fold=1, l16=1, bal=1, xb=32, abl=("max",)),
    "ABL_BAL32_SUMPACK": PCfg("bf16", 8, fold=1, l16=1, bal=1, xb=32, abl=("sum", "pack")),
    "ABL_BAL32_LDS": PCfg("bf16", 8, fold=1, l16=1, bal=1, xb=32, abl=("lds",)),
    "ABL_BAL32_DMA": PCfg("bf16", 8, fold=1, l16=1, bal=1, xb=32, abl=("dma",)),
    "ABL_BAL32_BAR": PCfg("bf16", 8, fold=1, l16=1, bal=1, xb=32, abl=("bar",)),
    "ABL_BAL32_ALL": PCfg("bf16", 8, fold=1, l16=1, bal=1, xb=32, abl=("exp", "max", "sum", "pack", "lds", "dma")),
}
PRODUCT_STREAMS = tuple(n for n, c in VARIANTS.items() if re.match(r'^(BF16|F16)_(FOLD|EXACT)', n) and not c.pprof and not c.merge and not c.fuse and not c.abl and not c.pad and not re.search(r'BAL|_FL\d|_SOFF|_PKS|_NT$|_SC\d|_DM$|_OROW$|_ST_|_QE$', n))
assert "BF16_FOLD_L16_SPROF" not in PRODUCT_STREAMS


def write_inc(path):
    lines = ["// GENERATED by tools/p4pgen.py -- do not edit.  Persistent instruction streams of attn_fwd16_p4p (see the generator's",
             "// header for the block loop, the register map and the block table).", "#pragma once", ""]
    lines.append("#define MFA_P4P_OWNED_VGPRS " + ", ".join('"v%d"' % i for i in range(p4gen.FIRST_OWNED_VGPR, 256)))
    lines.append("#define MFA_P4P_OWNED_SGPRS " + ", ".join('"s%d"' % i for i in range(FIRST_CLOBBERED_SGPR, LAST_CLOBBERED_SGPR + 1)))
    lines.append("#define MFA_P4P_QIMG %d" % QIMG)
    lines.append("#define MFA_P4P_TABLE %d" % TABLE)
    lines.append("#define MFA_P4P_TABLE_ENTRIES %d" % TABLE_ENTRIES)
    lines.append("#define MFA_P4P_LDS_BYTES %d" % LDS_BYTES)
    lines.append("")
    lines.append("// X(name, 16-bit type is f16, folds the softmax scale into Q, O in the 16-bit type, L in FP16, causal [2 = column-parallel pieces])")
    lines.append("#define MFA_P4P_PRODUCT_STREAM_LIST(X) \\")
    for name, cfg in VARIANTS.items():
        if name in PRODUCT_STREAMS:
            lines.append("  X(%s, %d, %d, %d, %d, %d) \\" % (name, cfg.dtype == "f16", cfg.fold, cfg.o16, cfg.l16, (2 if cfg.split else cfg.causal) | (4 if cfg.orow else 0)))
    lines.append("")
    lines.append("// streams that only the developer build (-DMFA_DEV_VARIANTS) instantiates")
    lines.append("#define MFA_P4P_DEV_STREAM_LIST(X) \\")
    for name, cfg in VARIANTS.items():
        if name not in PRODUCT_STREAMS:
            lines.append("  X(%s, %d, %d, %d, %d, %d) \\" % (name, cfg.dtype == "f16", cfg.fold, cfg.o16, cfg.l16, (2 if cfg.split else cfg.causal) | (4 if cfg.orow else 0)))
    lines.append("")
    lines.append("#ifdef MFA_DEV_VARIANTS")
    lines.append("#define MFA_P4P_STREAM_LIST(X) MFA_P4P_PRODUCT_STREAM_LIST(X) MFA_P4P_DEV_STREAM_LIST(X)")
    lines.append("#else")
    lines.append("#define MFA_P4P_STREAM_LIST(X) MFA_P4P_PRODUCT_STREAM_LIST(X)")
    lines.append("#endif")
    lines.append("")
    lines.append("")
    for name, cfg in VARIANTS.items():
        ins = PStream(cfg).build()
        txt = render(ins)
        if name not in PRODUCT_STREAMS:
            lines.append("#ifdef MFA_DEV_VARIANTS")
        n_mfma = sum(1 for i in ins if i.op.startswith("v_mfma"))
        lines.append("// %s: dtype=%s thr=%g fold=%d xb=%d o16=%d l16=%d causal=%d -- %d instructions, %d matrix instructions"
                     % (name, cfg.dtype, cfg.thr, cfg.fold, cfg.xb, cfg.o16, cfg.l16, cfg.causal, len(txt), n_mfma))
        lines.append("#define MFA_P4P_STREAM_%s \\" % name)
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
    out = os.path.join(here, "..", "metal_flash_attention_amd", "csrc", "attn_fwd16_p4p_stream.inc")
    write_inc(out)
    print("wrote", os.path.normpath(out), "-", len(PStream(VARIANTS["BF16_FOLD_L16"]).build()), "instructions in the headline stream")
