#!/usr/bin/env python3
"""Lane-exact model of the instruction subset used by tools/p4gen.py, for checking the hand-placed stream of
attn_fwd16_p4 on the CPU: register map, fragment layouts, counted waits, LDS ring discipline, the deferred rescale.

Not a timing model.  What it is strict about:
  * a register that is the destination of an LDS read still in flight is poisoned until an s_waitcnt retires the read
    (reading it raises);
  * LDS-DMA data land either at issue ("early") or only when the issuing wave's s_waitcnt vmcnt retires them ("late");
    a correct stream gives the right result under both, with the four waves run in either order between barriers;
  * LDS returns in order, vmcnt retires in order.
The C++ part of the kernel (Q fragments, the first three DMA tiles, the epilogue) is restated in `run_block`.
"""
import struct

import numpy as np

from p4gen import KSLOT, VBASE, VSLOT, Cfg, Stream

LDS_BYTES = 163840
FLT_MAX = np.float32(3.402823466e+38)


def f32_to_bf16_rne(x):
    b = x.view(np.uint32).astype(np.uint64)
    r = (b + 0x7FFF + ((b >> 16) & 1)) >> 16
    return (r & 0xFFFF).astype(np.uint32)


def bf16_to_f32(h):
    return (h.astype(np.uint32) << 16).view(np.float32)


class Poison(Exception):
    pass


class Wave:
    def __init__(self, wid, wg):
        self.id, self.wg = wid, wg
        self.v = np.zeros((256, 64), np.uint32)
        self.a = np.zeros((256, 64), np.uint32)
        self.vn, self.sn = {}, {}
        self.vcc = np.zeros(64, bool)
        self.scc = 0
        self.m0 = 0
        self.lds_q = []    # (dest regs [(kind, idx)], data uint32 [n][64])
        self.vm_q = []     # (lds addresses [64], data uint8 [64][16])
        self.poison = set()
        self.count = {}

    # ---- operand access
    def regs(self, o):
        return [(o[0], o[1] + i) for i in range(o[2])]

    def rd(self, o):
        k = o[0]
        if k in ("v", "a"):
            assert o[2] == 1
            if (k, o[1]) in self.poison:
                raise Poison("wave %d reads %s%d while its LDS read is in flight" % (self.id, k, o[1]))
            return (self.v if k == "v" else self.a)[o[1]]
        if k == "V":
            return self.vn[o[1]]
        if k == "S":
            return np.full(64, self.sn[o[1]], np.uint32) if not isinstance(self.sn[o[1]], float) else \
                np.full(64, np.float32(self.sn[o[1]])).view(np.uint32)
        if k == "i":
            return np.full(64, o[1] & 0xFFFFFFFF, np.uint32)
        if k == "f":
            return np.full(64, np.float32(o[1])).view(np.uint32)
        raise ValueError(o)

    def rdf(self, o):
        return self.rd(o).view(np.float32)

    def rd_multi(self, o):
        out = []
        for kind, idx in self.regs(o):
            if (kind, idx) in self.poison:
                raise Poison("wave %d reads %s%d while its LDS read is in flight" % (self.id, kind, idx))
            out.append((self.v if kind == "v" else self.a)[idx])
        return np.stack(out)

    def wr(self, o, val):
        val = np.asarray(val)
        if val.dtype == np.float32:
            val = val.view(np.uint32)
        val = val.astype(np.uint32)
        k = o[0]
        if k == "v":
            self.v[o[1]] = val
        elif k == "a":
            self.a[o[1]] = val
        elif k == "V":
            self.vn[o[1]] = val.copy()
        else:
            raise ValueError(o)

    def srd(self, o):
        if o[0] == "S":
            return self.sn[o[1]]
        if o[0] == "i":
            return o[1]
        if o[0] == "m0":
            return self.m0
        raise ValueError(o)

    def swr(self, o, val):
        if o[0] == "S":
            self.sn[o[1]] = int(val) & 0xFFFFFFFF
        elif o[0] == "m0":
            self.m0 = int(val) & 0xFFFFFFFF
        else:
            raise ValueError(o)

    # ---- queues
    def retire_lds(self, keep):
        while len(self.lds_q) > keep:
            dests, data = self.lds_q.pop(0)
            if isinstance(dests, tuple) and dests and dests[0] == "write":     # ds_write: the data reach LDS as late as the waits allow
                self.wg.lds_write16(dests[1], data)
                continue
            for (kind, idx), row in zip(dests, data):
                (self.v if kind == "v" else self.a)[idx] = row
                self.poison.discard((kind, idx))

    def retire_vm(self, keep):
        while len(self.vm_q) > keep:
            addrs, data = self.vm_q.pop(0)
            if isinstance(addrs, tuple):                 # plain buffer load: (kind, idx) <- data
                (self.v if addrs[0] == "v" else self.a)[addrs[1]] = data
                self.poison.discard(addrs)
            elif addrs is not None:
                self.wg.lds_write16(addrs, data)


def frag16(regs4):
    """4 registers x 64 lanes of packed bf16 -> float32 [32 (l & 31)][16 (k = 8 (l >> 5) + t)]"""
    out = np.zeros((32, 16), np.float32)
    for l in range(64):
        i, hi = l & 31, l >> 5
        for w in range(4):
            word = int(regs4[w][l])
            out[i, 8 * hi + 2 * w] = bf16_to_f32(np.array([word & 0xFFFF], np.uint32))[0]
            out[i, 8 * hi + 2 * w + 1] = bf16_to_f32(np.array([word >> 16], np.uint32))[0]
    return out


def h16_to_f32(h, f16):
    return h.astype(np.uint16).view(np.float16).astype(np.float32) if f16 else bf16_to_f32(h)


def f32_to_h16(x, f16):
    return x.astype(np.float16).view(np.uint16).astype(np.uint32) if f16 else f32_to_bf16_rne(x)


def frag16_fast(regs4, f16=False):
    r = np.asarray(regs4)                       # [4][64]
    lo = h16_to_f32(r & 0xFFFF, f16)            # element 2 w
    hi = h16_to_f32(r >> 16, f16)               # element 2 w + 1
    out = np.zeros((32, 16), np.float32)
    for h in range(2):
        for w in range(4):
            out[:, 8 * h + 2 * w] = lo[w, 32 * h:32 * h + 32]
            out[:, 8 * h + 2 * w + 1] = hi[w, 32 * h:32 * h + 32]
    return out


ROWMAP = np.array([[(r & 3) + 8 * (r >> 2) + 4 * h for h in range(2)] for r in range(16)])   # [r][hi] -> row


class Workgroup:
    def __init__(self, instrs, dma_mode="late"):
        self.instrs = instrs
        self.labels = {ins.mod["name"]: i for i, ins in enumerate(instrs) if ins.op == "label"}
        self.lds = np.zeros(LDS_BYTES, np.uint8)
        self.mem = {}
        self.dma_mode = dma_mode
        self.waves = [Wave(w, self) for w in range(4)]

    def lds_write16(self, addrs, data):
        for l in range(64):
            a = int(addrs[l])
            self.lds[a:a + 16] = data[l]

    def lds_read(self, addrs, n):
        idx = addrs.astype(np.int64)[:, None] + np.arange(n)[None, :]
        return self.lds[idx]

    # ---- one wave until its next barrier (or the end); returns False when the stream has ended
    def run_wave(self, w, limit=10_000_000):
        ins_list = self.instrs
        pc = w.pc
        while pc < len(ins_list):
            ins = ins_list[pc]
            pc += 1
            op = ins.op
            if op == "label":
                continue
            w.count[op] = w.count.get(op, 0) + 1
            if op == "s_barrier":
                w.pc = pc
                return True
            npc = self.execute(w, ins)
            if npc is not None:
                pc = npc
        w.pc = pc
        return False

    def execute(self, w, ins):
        op, d, s, m = ins.op, ins.d, ins.s, ins.mod
        if op.startswith("v_mfma"):
            f16 = op.endswith("_f16")
            A = frag16_fast(w.rd_multi(s[0]), f16)     # [i][k]
            B = frag16_fast(w.rd_multi(s[1]), f16)     # [j][k]
            acc = np.zeros((32, 32), np.float32)
            if s[2][0] != "i":
                c = w.rd_multi(s[2]).view(np.float32)     # [16][64]
                for r in range(16):
                    for h in range(2):
                        acc[ROWMAP[r][h], :] = c[r, 32 * h:32 * h + 32]
            res = (A.astype(np.float64) @ B.astype(np.float64).T + acc).astype(np.float32)
            out = np.zeros((16, 64), np.float32)
            for r in range(16):
                for h in range(2):
                    out[r, 32 * h:32 * h + 32] = res[ROWMAP[r][h], :]
            for (kind, idx), row in zip(w.regs(d), out.view(np.uint32)):
                if (kind, idx) in w.poison:
                    raise Poison("MFMA writes poisoned register")
                (w.v if kind == "v" else w.a)[idx] = row
        elif op == "s_waitcnt":
            if "lgkmcnt" in m:
                w.retire_lds(m["lgkmcnt"])
            if "vmcnt" in m:
                w.retire_vm(m["vmcnt"])
        elif op == "s_nop":
            pass
        elif op == "ds_read_b128":
            addr = w.rd(s[0]) + m["offset"]
            data = self.lds_read(addr, 16).copy().view(np.uint32)     # [64][4]
            dests = w.regs(d)
            for t in dests:
                w.poison.add(t)
            w.lds_q.append((dests, data.T.copy()))
        elif op == "ds_write_b128":
            # s = (address register, four data registers); in the wave's in-order LDS queue (lgkmcnt) like a read.  The data are read
            # at issue and land when a wait retires the entry ("late": a barrier without the wait leaves other waves the old bytes)
            addr = (w.rd(s[0]).astype(np.int64) + int(m.get("offset", 0)))
            words = w.rd_multi(s[1]).astype(np.uint32)                 # [4][64]
            data = np.ascontiguousarray(words.T).view(np.uint8).reshape(64, 16).copy()
            if getattr(self, "lds_write_mode", "late") == "early":
                self.lds_write16(addr, data)
                w.lds_q.append(([], np.zeros((0, 64), np.uint32)))
            else:
                w.lds_q.append((("write", addr, None), data))
        elif op == "v_fma_mix_f32":
            # d = s0 * s1 + s2 in fp32; source i is an FP16 half of its register when op_sel_hi[i] (the high half when op_sel[i]), else FP32
            vals = []
            for i in range(3):
                raw = w.rd(s[i])
                if m["op_sel_hi"][i]:
                    half = (raw >> 16) if m["op_sel"][i] else (raw & 0xFFFF)
                    vals.append(h16_to_f32(half & 0xFFFF, True).astype(np.float64))
                else:
                    vals.append(raw.view(np.float32).astype(np.float64))
            with np.errstate(over="ignore", invalid="ignore"):
                w.wr(d, (vals[0] * vals[1] + vals[2]).astype(np.float32))
        elif op == "v_lshrrev_b32":
            w.wr(d, w.rd(s[1]) >> np.uint32(int(s[0][1])))
        elif op == "ds_read_b64":
            addr = w.rd(s[0]) + m["offset"]
            data = self.lds_read(addr, 8).copy().view(np.uint32)      # [64][2]
            dests = w.regs(d)
            for t in dests:
                w.poison.add(t)
            w.lds_q.append((dests, data.T.copy()))
        elif op == "ds_read_b64_tr_b16":
            addr = (w.rd(s[0]) + m["offset"]).astype(np.int64)
            raw = self.lds_read(addr, 8).copy().view(np.uint16)       # [64 lanes][4 elements] at each lane's address
            res = np.zeros((64, 4), np.uint16)
            for l in range(64):
                g, i = l & ~15, l & 15
                for k in range(4):
                    res[l, k] = raw[g + 4 * k + (i >> 2), i & 3]
            data = res.view(np.uint32)                                # [64][2]
            dests = w.regs(d)
            for t in dests:
                w.poison.add(t)
            w.lds_q.append((dests, data.T.copy()))
        elif op == "buffer_load_dwordx4_lds":
            off = w.rd(s[0]).astype(np.int64)
            buf, nrec = w.sn[s[1][1]]
            data = np.zeros((64, 16), np.uint8)
            for l in range(64):
                o = int(off[l])
                if o + 16 <= nrec:
                    data[l] = buf[o:o + 16]
            addrs = (w.m0 + 16 * np.arange(64)).astype(np.int64)
            if self.dma_mode == "early":
                self.lds_write16(addrs, data)
                w.vm_q.append((None, None))
            else:
                w.vm_q.append((addrs, data))
        elif op in ("buffer_load_dword", "buffer_load_ushort"):
            off = w.rd(s[0]).astype(np.int64)
            buf, nrec = w.sn[s[1][1]]
            n = 4 if op == "buffer_load_dword" else 2
            data = np.zeros(64, np.uint32)
            for l in range(64):
                o = int(off[l])
                if o + n <= nrec:
                    data[l] = int.from_bytes(bytes(buf[o:o + n]), "little")
            dest = (d[0], d[1])
            w.poison.add(dest)
            w.vm_q.append((dest, data))
        elif op == "buffer_store_dwordx4":
            # data registers are read at issue; the write counts in vmcnt like a load (gfx950 has no separate store counter)
            regs = w.regs(s[0])
            off = w.rd(s[1]).astype(np.int64)
            buf, nrec = w.sn[s[2][1]]
            words = np.stack([(w.v if kind == "v" else w.a)[idx] for kind, idx in regs], axis=1).astype(np.uint32)   # [64][4]
            for l in range(64):
                o = int(off[l])
                if 0 <= o and o + 16 <= nrec:
                    buf[o:o + 16] = words[l].view(np.uint8)
            w.vm_q.append((None, None))
        elif op == "v_exp_f32":
            x = w.rdf(s[0]).astype(np.float64)
            if m.get("neg0"):
                x = -x
            with np.errstate(over="ignore", under="ignore"):
                w.wr(d, np.exp2(x).astype(np.float32))
        elif op == "v_fma_f32":
            a, b, c = w.rdf(s[0]).astype(np.float64), w.rdf(s[1]).astype(np.float64), w.rdf(s[2]).astype(np.float64)
            if m.get("neg2"):
                c = -c
            with np.errstate(over="ignore"):
                w.wr(d, (a * b + c).astype(np.float32))
        elif op in ("v_add_f32", "v_sub_f32", "v_mul_f32", "v_max_f32"):
            a, b = w.rdf(s[0]), w.rdf(s[1])
            with np.errstate(over="ignore", invalid="ignore"):
                r = {"v_add_f32": a + b, "v_sub_f32": a - b, "v_mul_f32": a * b, "v_max_f32": np.maximum(a, b)}[op]
            w.wr(d, r.astype(np.float32))
        elif op in ("v_pk_mul_f32", "v_pk_add_f32"):
            x, y = w.rd_multi(s[0]).view(np.float32), (w.rd_multi(s[1]) if s[1][0] != "S" else np.stack([w.rd(("S", s[1][1]))] * 2)).view(np.float32)
            with np.errstate(over="ignore", invalid="ignore"):
                res = ((x * y) if op == "v_pk_mul_f32" else (x + y)).astype(np.float32).view(np.uint32)
            for (kind, idx), row in zip(w.regs(d), res):
                (w.v if kind == "v" else w.a)[idx] = row
        elif op == "v_max3_f32":
            w.wr(d, np.maximum(np.maximum(w.rdf(s[0]), w.rdf(s[1])), w.rdf(s[2])))
        elif op in ("v_cvt_pk_bf16_f32", "v_cvt_pk_f16_f32"):
            f16 = op == "v_cvt_pk_f16_f32"
            x0, x1 = w.rdf(s[0]).copy(), w.rdf(s[1]).copy()
            if m.get("neg0"):
                x0 = -x0
            if m.get("neg1"):
                x1 = -x1
            with np.errstate(over="ignore"):
                lo, hi = f32_to_h16(x0, f16), f32_to_h16(x1, f16)
            w.wr(d, lo | (hi << 16))
        elif op == "v_cvt_f16_f32":
            with np.errstate(over="ignore"):
                w.wr(d, f32_to_h16(w.rdf(s[0]).copy(), True))
        elif op == "v_cvt_f32_f16":
            w.wr(d, h16_to_f32(w.rd(s[0]) & 0xFFFF, True))
        elif op == "v_lshlrev_b32":
            w.wr(d, (w.rd(s[1]).astype(np.uint64) << int(s[0][1])) & 0xFFFFFFFF)
        elif op == "v_cmp_lt_f32":
            w.vcc = w.rdf(s[0]) < w.rdf(s[1])
        elif op == "v_mov_b32":
            w.wr(d, w.rd(s[0]).copy())
        elif op == "v_and_b32":
            w.wr(d, w.rd(s[0]) & w.rd(s[1]))
        elif op == "v_xor_b32":
            w.wr(d, w.rd(s[0]) ^ w.rd(s[1]))
        elif op == "v_add_u32":
            w.wr(d, (w.rd(s[0]).astype(np.uint64) + w.rd(s[1]).astype(np.uint64)) & 0xFFFFFFFF)
        elif op == "v_add_u32_e64":
            w.wr(d, np.minimum(w.rd(s[0]).astype(np.uint64) + w.rd(s[1]).astype(np.uint64), 0xFFFFFFFF))
        elif op == "v_subrev_u32":
            w.wr(d, (w.rd(s[1]).astype(np.int64) - w.rd(s[0]).astype(np.int64)) & 0xFFFFFFFF)
        elif op == "v_cmp_gt_f32":
            w.vcc = w.rdf(s[0]) > w.rdf(s[1])
        elif op == "v_cmp_le_i32":
            w.vcc = w.rd(s[0]).view(np.int32) <= w.rd(s[1]).view(np.int32)
        elif op == "v_cmp_lt_i32":
            w.vcc = w.rd(s[0]).view(np.int32) < w.rd(s[1]).view(np.int32)
        elif op == "v_cmp_gt_i32":
            w.vcc = w.rd(s[0]).view(np.int32) > w.rd(s[1]).view(np.int32)
        elif op == "v_cndmask_b32":
            w.wr(d, np.where(w.vcc, w.rd(s[1]), w.rd(s[0])))
        elif op == "v_permlane32_swap_b32":
            x, y = w.rd(d).copy(), w.rd(s[0]).copy()
            x2, y2 = x.copy(), y.copy()
            x2[32:] = y[:32]
            y2[:32] = x[32:]
            w.wr(d, x2)
            w.wr(s[0], y2)
        elif op == "v_accvgpr_write_b32":
            w.wr(d, w.rd(s[0]).copy())
        elif op == "v_accvgpr_read_b32":
            w.wr(d, w.rd(s[0]).copy())
        elif op == "s_mov_b32":
            w.swr(d, w.srd(s[0]))
        elif op in ("s_add_u32", "s_sub_u32"):
            a, b = w.srd(s[0]), w.srd(s[1])
            w.swr(d, a + b if op == "s_add_u32" else a - b)
        elif op == "s_and_b32":
            w.swr(d, w.srd(s[0]) & w.srd(s[1]))
        elif op == "s_lshl_b32":
            w.swr(d, w.srd(s[0]) << w.srd(s[1]))
        elif op in ("s_cmp_lt_i32", "s_cmp_ge_i32"):
            a, b = np.int32(np.uint32(w.srd(s[0]))), np.int32(np.uint32(w.srd(s[1])))
            w.scc = int(a < b) if op == "s_cmp_lt_i32" else int(a >= b)
        elif op == "s_cmp_ge_u32":
            w.scc = int((w.srd(s[0]) & 0xFFFFFFFF) >= (w.srd(s[1]) & 0xFFFFFFFF))
        elif op == "s_cmp_eq_u32":
            w.scc = int((w.srd(s[0]) & 0xFFFFFFFF) == (w.srd(s[1]) & 0xFFFFFFFF))
        elif op == "s_cselect_b64":
            w.sn[d[1]] = np.full(64, bool(w.srd(s[0]) if w.scc else w.srd(s[1])))
        elif op == "v_cndmask_b32_e64":
            w.wr(d, np.where(w.sn[s[2][1]], w.rd(s[1]), w.rd(s[0])))
        elif op == "s_cselect_b32":
            w.swr(d, w.srd(s[0]) if w.scc else w.srd(s[1]))
        elif op == "s_mov_b64":
            if d[0] == "S":
                w.sn[d[1]] = w.vcc.copy()
            else:
                raise ValueError(ins)
        elif op == "s_or_b64":
            w.vcc = w.vcc | w.sn[s[1][1]]
        elif op == "s_cbranch_scc0":
            if not w.scc:
                return self.labels[m["target"]]
        elif op == "s_cbranch_scc1":
            if w.scc:
                return self.labels[m["target"]]
        elif op == "s_cbranch_vccnz":
            if w.vcc.any():
                return self.labels[m["target"]]
        elif op == "s_branch":
            return self.labels[m["target"]]
        else:
            raise NotImplementedError(op)
        return None

    def run(self, order=(0, 1, 2, 3)):
        for w in self.waves:
            w.pc = 0
        alive = True
        while alive:
            states = [self.run_wave(self.waves[i]) for i in order]
            assert all(states) or not any(states), "waves disagree on the number of barriers"
            alive = states[0]


def run_block(q, k, v, rblk, cfg=None, causal=False, dma_mode="late", order=(0, 1, 2, 3), scale=None, stream=None, tr_pad=0):
    """One 256-row block: q [R][128], k / v [C][128] as uint16 bf16 bit patterns.  Returns O [256][128] f32, L [256].
    cfg.tr: K and V are handed to the stream TRANSPOSED ([128][C + tr_pad] in memory, C % 8 == 0; the padding columns hold NaN:
    what follows the sequence in a row must neither be multiplied nor summed), as attn_fwd16_p4_tr.h does."""
    cfg = cfg or Cfg()
    tr = bool(getattr(cfg, "tr", 0))
    ktr, vtr = bool(getattr(cfg, "kt", 0)), bool(getattr(cfg, "vt", 0))
    f16 = cfg.dtype == "f16"
    R, C, D = q.shape[0], k.shape[0], 128
    assert q.shape[1] == D
    instrs = stream if stream is not None else Stream(cfg).build()
    wg = Workgroup(instrs, dma_mode)
    kb, vb, qb = k.reshape(-1).view(np.uint8), v.reshape(-1).view(np.uint8), q.reshape(-1).view(np.uint8)
    if tr:
        assert C % 8 == 0, "the transposed streams take whole 16-byte chunks"
        nan16 = 0x7E00 if f16 else 0x7FC0
        kt, vt = (np.concatenate([x.T, np.full((D, tr_pad), nan16, np.uint16)], axis=1) for x in (k, v))
        if ktr:
            kb = np.ascontiguousarray(kt).reshape(-1).view(np.uint8)
        if vtr:
            vb = np.ascontiguousarray(vt).reshape(-1).view(np.uint8)
    ld2 = D * 2
    ldt2 = (C + tr_pad) * 2          # leading dimension of K^T / V^T, bytes
    knrec = D * ldt2 if ktr else C * ld2
    vnrec = D * ldt2 if vtr else C * ld2
    nt_total = (C + 63) // 64
    coff = C - R
    nt = nt_total
    if causal:
        last_row = min(R, (rblk + 1) * 256) - 1
        nt = min(nt_total, (last_row + coff) // 64 + 1)
    ragged = (C % 64 != 0) and nt == nt_total
    scale = scale if scale is not None else 1.0 / np.sqrt(np.float32(D))
    scale2 = float(np.float32(1.44269504089) * np.float32(scale))
    OOB = 0xFFFFFF00
    if cfg.fold:   # the C++ prologue of FOLD streams stores Q * scale2, rounded to the 16-bit type, in a[128:191]
        qf = h16_to_f32(q.astype(np.uint32).reshape(-1), f16).reshape(q.shape)
        q = f32_to_h16((qf * np.float32(scale2)).astype(np.float32).reshape(-1), f16).astype(np.uint16).reshape(q.shape)
        qb = q.reshape(-1).view(np.uint8)
    for w in wg.waves:
        wave = w.id
        r0 = rblk * 256 + wave * 64
        lane = np.arange(64)
        qq, hi = lane & 31, lane >> 5
        # Q fragments -> a[128:191]
        for b in range(2):
            for s in range(8):
                for l in range(64):
                    row = r0 + b * 32 + int(qq[l])
                    d0 = 16 * s + 8 * int(hi[l])
                    if row < R and ktr:   # elements 4 hi + {0..3, 8..11} of the step: the order the K^T fragments arrive in
                        d0 = 16 * s + 4 * int(hi[l])
                        chunk = np.concatenate([qb[row * ld2 + d0 * 2: row * ld2 + d0 * 2 + 8],
                                                qb[row * ld2 + (d0 + 8) * 2: row * ld2 + (d0 + 8) * 2 + 8]]).view(np.uint32)
                    elif row < R:
                        chunk = qb[row * ld2 + d0 * 2: row * ld2 + d0 * 2 + 16].view(np.uint32)
                    else:
                        chunk = np.zeros(4, np.uint32)
                    for t in range(4):
                        w.a[128 + 4 * (b * 8 + s) + t][l] = chunk[t]
        # DMA offsets and the three tiles issued by the C++ prologue (K(0), V(0), K(1))
        koff, voff = [], []
        for i in range(4):
            p = (wave * 4 + i) * 64 + lane
            krow, kc = p >> 4, (p & 15) ^ ((p >> 4) & 15)
            vkey, vc = (p >> 2) & 63, (p >> 8) * 4 + (p & 3)
            # K^T image [2 blocks of 32 keys][128 elements][4 chunks]; V^T image [128 elements][8 chunks ^ (element >> 1 & 7)]
            if ktr:
                koff.append((((p >> 2) & 127) * ldt2 + ((p >> 9) * 32 + (p & 3) * 8) * 2).astype(np.uint32))
            else:
                koff.append((krow * ld2 + kc * 16).astype(np.uint32))
            if vtr:
                voff.append(((p >> 3) * ldt2 + ((p & 7) ^ ((p >> 4) & 7)) * 16).astype(np.uint32))
            else:
                voff.append((vkey * ld2 + vc * 16).astype(np.uint32))

        def dma(buf, nrec, off, ldsbase):
            data = np.zeros((64, 16), np.uint8)
            for l in range(64):
                o = int(off[l])
                if o + 16 <= nrec:
                    data[l] = buf[o:o + 16]
            wg.lds_write16(ldsbase + 16 * np.arange(64), data)

        kinc, vinc = 128 if ktr else 64 * ld2, 128 if vtr else 64 * ld2
        vlast = []
        if vtr:   # offsets of the workgroup's LAST tile: chunks at or beyond key C are not fetched (zeros)
            for i in range(4):
                p = (wave * 4 + i) * 64 + lane
                key0 = ((p & 7) ^ ((p >> 4) & 7)) * 8 + 64 * (nt - 1)
                vlast.append(np.where(key0 < C, voff[i].astype(np.uint64) + (nt - 1) * vinc, OOB).astype(np.uint32))
                w.vn["vlast%d" % i] = vlast[i]
        for i in range(4):
            dma(kb, knrec, koff[i], 0 * KSLOT + (wave * 4 + i) * 1024)
            koff[i] = np.minimum(koff[i].astype(np.uint64) + kinc, 0xFFFFFFFF).astype(np.uint32)
        for i in range(4):
            dma(vb, vnrec, vlast[i] if vtr and nt == 1 else voff[i], VBASE + (wave * 4 + i) * 1024)
            voff[i] = np.minimum(voff[i].astype(np.uint64) + vinc, 0xFFFFFFFF).astype(np.uint32)
        for i in range(4):
            dma(kb, knrec, koff[i], 1 * KSLOT + (wave * 4 + i) * 1024)
            koff[i] = np.minimum(koff[i].astype(np.uint64) + kinc, 0xFFFFFFFF).astype(np.uint32)
        w.vm_q = [(None, None)] * 12
        n16 = lane & 15
        w.vn.update({
            "m0": np.full(64, 0.0 if cfg.fold else -FLT_MAX, np.float32).view(np.uint32),
            "m1": np.full(64, 0.0 if cfg.fold else -FLT_MAX, np.float32).view(np.uint32),
            "onesw": np.where(lane < 32, 0xBC00BC00 if f16 else 0xBF80BF80, 0).astype(np.uint32),
            "l0": np.zeros(64, np.uint32), "l1": np.zeros(64, np.uint32),
            "kbase": (qq * 256 + ((hi ^ (qq & 15)) << 4)).astype(np.uint32),
            "vbase": (VBASE + ((n16 >> 2) + 4 * hi) * 64 + (((lane >> 4) & 1) * 16 + 4 * (n16 & 3)) * 2).astype(np.uint32),
        })
        if ktr:   # K^T: the transposing-read lane term
            w.vn["kbase"] = (((n16 >> 2) + 4 * hi) * 64 + (((lane >> 4) & 1) * 16 + 4 * (n16 & 3)) * 2).astype(np.uint32)
        if vtr:   # V^T: row lane % 32, the swizzle's XOR mask, + 8 hi
            w.vn["vbase"] = (VBASE + qq * 128 + (((qq >> 1) & 7) << 4) + 8 * hi).astype(np.uint32)
            w.vn["vta"] = w.vn["vtb"] = np.zeros(64, np.uint32)
        for i in range(4):
            w.vn["koff%d" % i], w.vn["voff%d" % i] = koff[i], voff[i]
        for b in range(2):
            row = r0 + b * 32 + qq
            lim = np.minimum(C - 1, row + coff) if causal else np.full(64, C - 1)
            w.vn["lim%d" % b] = (lim - 4 * hi).astype(np.int64).astype(np.uint32)
        minlim = min(C - 1, r0 + coff) if causal else C - 1
        maskfrom = (minlim + 1) // 64 if (causal or ragged) else nt
        wnt = nt
        if causal:   # tiles this wave's own rows can see
            wlast = min(R, r0 + 64) - 1
            wnt = max(1, min(nt, (wlast + coff) // 64 + 1)) if wlast >= r0 else 1
        w.sn.update({"kres": (kb, knrec), "vres": (vb, vnrec), "nt": nt, "wnt": wnt, "scale2": scale2, "kinc": kinc, "vinc": vinc,
                     "ldsk": wave * 4096, "ldsv": VBASE + wave * 4096, "maskfrom": maskfrom, "ntm2": nt - 2})
    wg.run(order)
    O = np.zeros((256, D), np.float32)
    L = np.zeros(256, np.float32)
    for w in wg.waves:
        assert not w.lds_q, "LDS reads left in flight"
        for b in range(2):
            l = w.vn["l%d" % b].view(np.float32)
            mm = w.vn["m%d" % b].view(np.float32)
            ltot = l[:32] + l[32:] + np.float32(1.401298464e-45)
            for db in range(4):
                for r in range(16):
                    reg = w.a[16 * (4 * b + db) + r].view(np.float32)
                    for h in range(2):
                        dcol = 32 * db + ROWMAP[r][h]
                        O[w.id * 64 + b * 32 + np.arange(32), dcol] = reg[32 * h:32 * h + 32] / ltot
            L[w.id * 64 + b * 32 + np.arange(32)] = mm[:32] + np.log2(ltot)
    return O, L, wg


def reference(q, k, v, causal=False, f16=False):
    """float64 attention on the 16-bit inputs; returns O, L (base-2 log-sum-exp of the scaled scores)"""
    qf, kf, vf = (h16_to_f32(x.astype(np.uint32).reshape(-1), f16).reshape(x.shape).astype(np.float64) for x in (q, k, v))
    R, C = qf.shape[0], kf.shape[0]
    s = qf @ kf.T / np.sqrt(qf.shape[1])
    if causal:
        mask = np.arange(C)[None, :] > (np.arange(R)[:, None] + max(C - R, 0))   # (the offset is clamped at 0: include/mfa.h, rowLengths)
        s = np.where(mask, -np.inf, s)
    mx = s.max(axis=1, keepdims=True)
    p = np.exp(s - mx)
    lsum = p.sum(axis=1, keepdims=True)
    return (p @ vf) / lsum, (mx[:, 0] + np.log(lsum[:, 0])) * 1.44269504089


def rand_bf16(shape, rng, scale=1.0, f16=False):
    x = (rng.standard_normal(shape) * scale).astype(np.float32)
    return f32_to_h16(x.reshape(-1), f16).astype(np.uint16).reshape(shape)


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    R, C = 256, 192
    q, k, v = rand_bf16((R, 128), rng), rand_bf16((C, 128), rng), rand_bf16((C, 128), rng)
    O, L, wg = run_block(q, k, v, 0)
    Oref, Lref = reference(q, k, v)
    print("max |dO|", np.abs(O - Oref).max(), "max |dL|", np.abs(L - Lref).max())
    print({k_: v_ for k_, v_ in sorted(wg.waves[0].count.items())})
