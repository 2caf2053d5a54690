#!/usr/bin/env python3
"""Lane-exact model of the persistent forward stream (tools/p4pgen.py): p4sim's workgroup plus what the block loop adds --
fixed scalar registers, buffer resources assembled in SGPRs and resolved against a flat global memory, 16 / 32 / 64-bit buffer
stores with instruction offsets, v_readfirstlane, v_rcp / v_log, and the block table in LDS that the C++ prologue of
attn_fwd16_p4p fills (restated in `run_workgroup`).

Stores are queued like loads (one in-order vmcnt): with stores="late" their data reach memory only when the issuing wave's
s_waitcnt retires them (or at the end of the stream), which is how a vmcnt(34) that lets an LDS-DMA piece slip is caught.
"""
import numpy as np

import p4pgen
from p4gen import KSLOT, VBASE
from p4sim import (FLT_MAX, Poison, Workgroup, bf16_to_f32, f32_to_bf16_rne, f32_to_h16, h16_to_f32, rand_bf16,  # noqa: F401
                   reference)


class GlobalMem:
    """flat byte-addressed memory made of named allocations (base address -> uint8 array)"""

    def __init__(self):
        self.allocs = []
        self.next = 0x7F0000001000

    def alloc(self, array_u8, pad=4096):
        base = self.next
        self.allocs.append((base, array_u8))
        self.next = (base + array_u8.size + pad + 255) & ~255
        return base

    def find(self, addr):
        for base, arr in self.allocs:
            if base <= addr < base + arr.size:
                return arr, addr - base
        raise Poison("address 0x%x is in no allocation" % addr)

    def window(self, base, nrec):
        """bytes [base, base + nrec) as (array, offset); the window may end beyond the allocation only if nrec says so"""
        if nrec == 0:
            return None, 0
        arr, off = self.find(base)
        return arr, off


class PWorkgroup(Workgroup):
    def __init__(self, instrs, mem, dma_mode="late", stores="late"):
        Workgroup.__init__(self, instrs, dma_mode)
        self.gmem = mem
        self.stores = stores
        for w in self.waves:
            w.sr = np.zeros(128, np.uint64)

    # ---- scalar operands
    def sval(self, w, o):
        k = o[0]
        if k == "sr":
            return int(w.sr[o[1]])
        if k == "S":
            return w.sn[o[1]]
        if k == "i":
            return o[1]
        if k == "m0":
            return w.m0
        raise ValueError(o)

    def sset(self, w, o, val):
        if o[0] == "sr":
            w.sr[o[1]] = int(val) & 0xFFFFFFFF
        elif o[0] == "m0":
            w.m0 = int(val) & 0xFFFFFFFF
        else:
            raise ValueError(o)

    def resource(self, w, o):
        """(array, offset of the base inside it, nrec) of the 128-bit resource in four SGPRs"""
        assert o[0] == "sr" and o[2] == 4 and o[1] % 4 == 0, o
        w0, w1, nrec, flags = (int(w.sr[o[1] + i]) for i in range(4))
        assert flags == p4pgen.DESC_FLAGS, "resource word 3 = 0x%x" % flags
        assert w1 >> 16 == 0, "stride / swizzle bits set in resource word 1"
        arr, off = self.gmem.window(w0 | (w1 << 32), nrec)
        return arr, off, nrec

    def vsrc(self, w, o):
        if o[0] == "sr":
            return np.full(64, int(w.sr[o[1]]) & 0xFFFFFFFF, np.uint32)
        return w.rd(o)

    def execute(self, w, ins):
        op, d, s, m = ins.op, ins.d, ins.s, ins.mod
        scalar_kinds = ("sr",)
        if op == "align":
            return None
        if op == "s_mov_b32" and d[0] in ("vcc_lo", "vcc_hi"):   # a scalar constant into one half of the lane mask
            val = int(self.sval(w, s[0])) & 0xFFFFFFFF
            bits = np.array([(val >> i) & 1 for i in range(32)], bool)
            w.vcc = w.vcc.copy()
            if d[0] == "vcc_lo":
                w.vcc[:32] = bits
            else:
                w.vcc[32:] = bits
            return None
        if op in ("s_mov_b32", "s_add_u32", "s_sub_u32", "s_and_b32", "s_or_b32", "s_lshl_b32", "s_lshr_b32", "s_mul_i32", "s_min_u32", "s_max_u32", "s_max_i32", "s_min_i32"):
            vals = [self.sval(w, x) for x in s]
            if op == "s_mov_b32":
                r = vals[0]
                if isinstance(r, float):
                    r = int(np.float32(r).view(np.uint32))
            else:
                a, b = int(vals[0]) & 0xFFFFFFFF, int(vals[1]) & 0xFFFFFFFF
                sgn = lambda x: int(np.int32(np.uint32(x)))
                r = {"s_add_u32": a + b, "s_sub_u32": a - b, "s_and_b32": a & b, "s_or_b32": a | b, "s_lshl_b32": a << (b & 31),
                     "s_min_i32": min(sgn(a), sgn(b)),
                     "s_lshr_b32": a >> (b & 31), "s_mul_i32": a * b, "s_min_u32": min(a, b), "s_max_u32": max(a, b),
                     "s_max_i32": max(int(np.int32(np.uint32(a))), int(np.int32(np.uint32(b))))}[op]
                if op == "s_add_u32":
                    w.scc = int(a + b > 0xFFFFFFFF)
            if d[0] in ("sr", "m0"):
                self.sset(w, d, r)
            else:
                w.swr(d, r)
            return None
        if op == "s_bitcmp1_b32":
            w.scc = (int(self.sval(w, s[0])) >> (int(self.sval(w, s[1])) & 31)) & 1
            return None
        if op in ("s_cmp_lt_i32", "s_cmp_ge_i32", "s_cmp_ge_u32", "s_cmp_eq_u32", "s_cmp_lt_u32", "s_cmp_lg_u32", "s_cmp_gt_u32"):
            a, b = int(self.sval(w, s[0])) & 0xFFFFFFFF, int(self.sval(w, s[1])) & 0xFFFFFFFF
            if op.endswith("i32"):
                a, b = int(np.int32(np.uint32(a))), int(np.int32(np.uint32(b)))
            w.scc = int({"s_cmp_lt_i32": a < b, "s_cmp_ge_i32": a >= b, "s_cmp_ge_u32": a >= b, "s_cmp_eq_u32": a == b,
                         "s_cmp_lt_u32": a < b, "s_cmp_lg_u32": a != b, "s_cmp_gt_u32": a > b}[op])
            return None
        if op == "s_cselect_b32":
            self.sset(w, d, self.sval(w, s[0]) if w.scc else self.sval(w, s[1]))
            return None
        if op == "s_mov_b64" and d[0] == "sr":
            w.sr_mask = getattr(w, "sr_mask", {})
            w.sr_mask[d[1]] = w.vcc.copy()
            return None
        if op == "s_or_b64" and s[1][0] == "sr":
            w.vcc = w.vcc | w.sr_mask[s[1][1]]
            return None
        if op == "v_readfirstlane_b32":
            self.sset(w, d, int(w.rd(s[0])[0]))
            return None
        if op in ("v_mov_b32", "v_add_u32", "v_add_u32_e64", "v_mul_f32", "v_subrev_u32") and any(x[0] in scalar_kinds for x in s):
            vals = [self.vsrc(w, x) for x in s]
            if op == "v_mov_b32":
                w.wr(d, vals[0].copy())
            elif op == "v_add_u32":
                w.wr(d, (vals[0].astype(np.uint64) + vals[1].astype(np.uint64)) & 0xFFFFFFFF)
            elif op == "v_add_u32_e64":
                w.wr(d, np.minimum(vals[0].astype(np.uint64) + vals[1].astype(np.uint64), 0xFFFFFFFF))
            elif op == "v_subrev_u32":
                w.wr(d, (vals[1].astype(np.int64) - vals[0].astype(np.int64)) & 0xFFFFFFFF)
            else:
                w.wr(d, (vals[0].view(np.float32) * vals[1].view(np.float32)).astype(np.float32))
            return None
        if op == "buffer_load_dwordx4_lds" and s[1][0] == "sr":
            off = w.rd(s[0]).astype(np.int64)
            arr, base, nrec = self.resource(w, s[1])
            soff = (int(self.sval(w, s[2])) & 0xFFFFFFFF) if len(s) > 2 else 0   # scalar offset: part of the range check, no 32-bit wrap (tools/probe_soffset.hip)
            data = np.zeros((64, 16), np.uint8)
            for l in range(64):
                o = int(off[l]) + soff
                if o + 16 <= nrec:
                    data[l] = arr[base + o:base + o + 16]
            addrs = (w.m0 + 16 * np.arange(64)).astype(np.int64)
            if self.dma_mode == "early":
                self.lds_write16(addrs, data)
                w.vm_q.append((None, None))
            else:
                w.vm_q.append((addrs, data))
            return None
        if op in ("buffer_store_dwordx4", "buffer_store_dwordx2", "buffer_store_dword", "buffer_store_short") and s[2][0] == "sr":
            nbytes = {"buffer_store_dwordx4": 16, "buffer_store_dwordx2": 8, "buffer_store_dword": 4, "buffer_store_short": 2}[op]
            regs = w.regs(s[0])
            for t in regs:
                if t in w.poison:
                    raise Poison("store of a register whose load is in flight")
            words = np.stack([(w.v if kind == "v" else w.a)[idx] for kind, idx in regs], axis=1).astype(np.uint32)   # [64][n]
            off = (w.rd(s[1]).astype(np.int64) + int(m.get("offset", 0))) & 0xFFFFFFFF   # the address adder wraps at 32 bits
            if len(s) > 3:      # scalar offset: part of the range check, no 32-bit wrap (as for the loads)
                off = off + (int(self.sval(w, s[3])) & 0xFFFFFFFF)
            arr, base, nrec = self.resource(w, s[2])
            writes = []
            for l in range(64):
                o = int(off[l])
                if 0 <= o and o + nbytes <= nrec:
                    writes.append((base + o, words[l].view(np.uint8)[:nbytes].copy()))
            if self.stores == "early":
                for a, b in writes:
                    arr[a:a + len(b)] = b
                w.vm_q.append((None, None))
            else:
                w.vm_q.append((("store", arr, writes), None))
            return None
        if op == "ds_bpermute_b32":     # d[lane] = s1[((s0[lane] + offset) >> 2) & 63]; returns through the LDS queue like a read
            sel = ((w.rd(s[0]).astype(np.int64) + int(m.get("offset", 0))) >> 2) & 63
            data = w.rd(s[1])[sel].astype(np.uint32)
            dests = w.regs(d)
            for t in dests:
                w.poison.add(t)
            w.lds_q.append((dests, data.reshape(1, 64).copy()))
            return None
        if op in ("v_mbcnt_lo_u32_b32", "v_mbcnt_hi_u32_b32"):      # (mask -1: the lane's index, low 32 lanes / the rest)
            assert s[0] == ("i", -1)
            lanes = np.arange(64)
            cnt = np.minimum(lanes, 32) if op.startswith("v_mbcnt_lo") else np.maximum(lanes - 32, 0)
            w.wr(d, (cnt + self.vsrc(w, s[1]).astype(np.int64)).astype(np.uint32))
            return None
        if op == "ds_write_b128":
            addr = w.rd(s[0]).astype(np.int64) + int(m.get("offset", 0))
            words = w.rd_multi(s[1]).astype(np.uint32)          # [4][64]
            self.lds_write16(addr, np.ascontiguousarray(words.T).view(np.uint8).reshape(64, 16))
            w.lds_q.append(([], np.zeros((0, 64), np.uint32)))   # counts in lgkmcnt like a read
            return None
        if op == "v_sub_u32":
            w.wr(d, (self.vsrc(w, s[0]).astype(np.int64) - self.vsrc(w, s[1]).astype(np.int64)) & 0xFFFFFFFF)
            return None
        if op == "v_min_u32":
            w.wr(d, np.minimum(self.vsrc(w, s[0]), self.vsrc(w, s[1])))
            return None
        if op == "v_rcp_f32":
            with np.errstate(divide="ignore", over="ignore"):
                w.wr(d, (np.float32(1.0) / w.rdf(s[0])).astype(np.float32))
            return None
        if op == "v_log_f32":
            with np.errstate(divide="ignore", invalid="ignore"):
                w.wr(d, np.log2(w.rdf(s[0]).astype(np.float64)).astype(np.float32))
            return None
        if op == "v_lshrrev_b32":
            w.wr(d, w.rd(s[1]) >> np.uint32(int(s[0][1])))
            return None
        if op == "v_fma_f32" and (m.get("neg0") or s[2][0] == "f"):
            a, b, c = (w.rdf(x).astype(np.float64) for x in s)
            if m.get("neg0"):
                a = -a
            if m.get("neg2"):
                c = -c
            w.wr(d, (a * b + c).astype(np.float32))
            return None
        if op == "s_waitcnt" and "vmcnt" in m:
            assert 0 <= m["vmcnt"] <= 63, "vmcnt is a six-bit counter"
            self.retire_vm(w, m["vmcnt"])
            if "lgkmcnt" in m:
                w.retire_lds(m["lgkmcnt"])
            return None
        return Workgroup.execute(self, w, ins)

    def retire_vm(self, w, keep):
        while len(w.vm_q) > keep:
            addrs, data = w.vm_q.pop(0)
            if isinstance(addrs, tuple) and addrs and addrs[0] == "store":
                _, arr, writes = addrs
                for a, b in writes:
                    arr[a:a + len(b)] = b
            elif isinstance(addrs, tuple):
                (w.v if addrs[0] == "v" else w.a)[addrs[1]] = data
                w.poison.discard(addrs)
            elif addrs is not None:
                self.lds_write16(addrs, data)


def run_workgroup(q, k, v, blocks, cfg, D=128, dma_mode="late", stores="late", order=(0, 1, 2, 3), stream=None, ld=None, causal=None,
                  lengths=None, cflag=None, splits=1):
    # lengths (causal / "geometry" streams): {head: (rows, keys)} of the head's batch entry, at most the array shapes (per-batch lengths);
    # cflag: 1 = causal mask (default for causal streams), 0 = lengths only
    """One persistent workgroup over `blocks` = [(head, row block), ...].  q [H][R][D], k / v [H][C][D] uint16 bit patterns.
    Restates the C++ prologue of attn_fwd16_p4p (block table, lane constants, scalar inputs).  Returns O [H][R][D] float32
    (or the 16-bit patterns as float32 when cfg.o16), L [H][R] float32 (log2 units), the workgroup."""
    f16 = cfg.dtype == "f16"
    H, R, _ = q.shape
    C = k.shape[1]
    ldq = ldk = ldv = ldo = ld or D
    instrs = stream if stream is not None else p4pgen.PStream(cfg).build()
    mem = GlobalMem()

    def padded(x, width):     # rows of `width` elements (leading dimension >= D)
        out = np.zeros(x.shape[:-1] + (width,), x.dtype)
        out[..., :x.shape[-1]] = x
        return out
    qm, km, vm = (np.ascontiguousarray(padded(x, ldq)).reshape(-1).view(np.uint8).copy() for x in (q, k, v))
    osz = 2 if cfg.o16 else 4
    lsz = 2 if cfg.l16 else 4
    split = bool(getattr(cfg, "split", 0))
    if split:      # blocks = [(head, row block, piece)]: wsO [splits][H][R][D] fp32, wsML [splits][H][R][2]
        assert C % (128 * splits) == 0 and ldo == D
        lsz, piece = 8, C // splits
        om = np.full(splits * H * R * D * 4, 0xCD, np.uint8)
        lm = np.full(splits * H * R * 8, 0xCD, np.uint8)
    else:
        om = np.full(H * R * ldo * osz, 0xCD, np.uint8)
        lm = np.full(H * R * lsz, 0xCD, np.uint8)
    qb, kb, vb, ob, lb = (mem.alloc(x) for x in (qm, km, vm, om, lm))
    wg = PWorkgroup(instrs, mem, dma_mode, stores)
    # block table (64 bytes per entry): Q, K, V, O, L base of the head, first row of the block
    table = np.zeros((len(blocks), 16), np.uint32)
    for n, blk in enumerate(blocks):
        h, rblk = blk[0], blk[1]
        if split:
            sp = blk[2]
            bases = (qb + h * R * ldq * 2, kb + (h * C + sp * piece) * ldk * 2, vb + (h * C + sp * piece) * ldv * 2,
                     ob + (sp * H + h) * R * D * 4, lb + (sp * H + h) * R * 8)
        else:
            bases = (qb + h * R * ldq * 2, kb + h * C * ldk * 2, vb + h * C * ldv * 2, ob + h * R * ldo * osz, lb + h * R * lsz)
        for i, a in enumerate(bases):
            table[n, 2 * i], table[n, 2 * i + 1] = a & 0xFFFFFFFF, a >> 32
        table[n, 10] = rblk * 256
        table[n, 11], table[n, 12] = (lengths or {}).get(h, (R, C))
    tb = table.reshape(-1).view(np.uint8)
    wg.lds[p4pgen.TABLE:p4pgen.TABLE + tb.size] = tb
    nt = (C + 63) // 64
    nt += nt & 1
    Ck = C
    if split:      # every workgroup sees its piece as the key range
        Ck, nt = piece, piece // 64
    scale2 = float(np.float32(1.44269504089) * np.float32(1.0 / np.sqrt(np.float32(D))))
    lane = np.arange(64)
    qq, hi = lane & 31, lane >> 5
    n16 = lane & 15
    ldq2, ldk2, ldv2 = ldq * 2, ldk * 2, ldv * 2
    for w in wg.waves:
        wave = w.id
        kv, qv = [], []
        for i in range(4):
            kc = (lane & 15) ^ ((4 * i + (lane >> 4)) & 15)
            kv.append(np.where(kc * 8 < D, (lane >> 4) * ldk2 + kc * 16, p4pgen.OOB).astype(np.uint32))
            qv.append(np.where(kc * 8 < D, (lane >> 4) * ldq2 + kc * 16, p4pgen.OOB).astype(np.uint32))
        vc = wave * 4 + (lane & 3)
        vv = np.where(vc * 8 < D, (lane >> 2) * ldv2 + vc * 16, p4pgen.OOB).astype(np.uint32)
        w.vn.update({
            "kbase": (qq * 256 + ((hi ^ (qq & 15)) << 4)).astype(np.uint32),
            "vbase": (VBASE + ((n16 >> 2) + 4 * hi) * 64 + (((lane >> 4) & 1) * 16 + 4 * (n16 & 3)) * 2).astype(np.uint32),
            "lim0": (Ck - 1 - 4 * hi).astype(np.int64).astype(np.uint32), "lim1": (Ck - 1 - 4 * hi).astype(np.int64).astype(np.uint32),
            "vv": vv, "lv": np.where(hi == 0, qq * lsz, p4pgen.OOB).astype(np.uint32),   # (split: lsz = 8, the (m, l) pair of the row)
            # epilogue: in as lane = row (16-byte chunks, chunk index XOR row & 7), out as lane = (row & 7, chunk)
            "qlane": qq.astype(np.uint32), "hi4": (4 * hi).astype(np.uint32),
            "ewa": (qq * 128 + ((hi ^ (qq & 7)) << 4)).astype(np.uint32),
            "era": ((lane >> 3) * 128 + (((lane & 7) ^ (lane >> 3)) << 4)).astype(np.uint32),
        })
        for db in range(4):
            col = 32 * db + 4 * (lane & 7)
            w.vn["ov%d" % db] = np.where(col < D, (lane >> 3) * ldo * osz + col * osz, p4pgen.OOB).astype(np.uint32)
            if getattr(cfg, "orow", 0):     # lane = column 32 db + n of rows 4 hi + ...: one fp32 element per lane and store
                ocol = 32 * db + qq
                w.vn["ov%d" % db] = np.where(ocol < D, 4 * ocol + hi * 4 * ldo * 4, p4pgen.OOB).astype(np.uint32)
        for i in range(4):
            w.vn["kv%d" % i], w.vn["qv%d" % i] = kv[i], qv[i]
        w.sn.update({"nt": nt, "maskfrom": Ck // 64, "scale2": scale2, "kinc": 64 * ldk2, "vinc": 64 * ldv2,
                     "ldsk": wave * 4096, "ldsv": VBASE + wave * 4096, "ldsq": p4pgen.QIMG + wave * 16384,
                     "qrel": p4pgen.QIMG + wave * 16384, "nblk": len(blocks), "tbl": p4pgen.TABLE, "wave64": wave * 64,
                     "ldq2": ldq2, "ldo": ldo * osz, "nrecq": R * ldq2, "nreck": Ck * ldk2, "nrecv": Ck * ldv2,
                     "nreco": R * ldo * osz, "nrecl": R * lsz, "dr": D, "cflag": int(bool(cfg.causal)) if cflag is None else int(cflag)})
    wg.run(order)
    for w in wg.waves:
        assert not w.lds_q, "LDS reads left in flight"
        wg.retire_vm(w, 0)
    if split:      # the merge of attn_fwd_combine, restated: m* = max m_s, w_s = 2^(m_s - m*), O = sum w_s O_s / sum w_s l_s, L = m* + log2 l*
        Os = om.view(np.float32).reshape(splits, H, R, D).astype(np.float64)
        ml = lm.view(np.float32).reshape(splits, H, R, 2).astype(np.float64)
        mstar = ml[..., 0].max(axis=0)
        wgt = np.exp2(ml[..., 0] - mstar[None])
        lstar = (wgt * ml[..., 1]).sum(axis=0)
        O = ((wgt[..., None] * Os).sum(axis=0) / lstar[..., None]).astype(np.float32)
        L = (mstar + np.log2(lstar)).astype(np.float32)
        return O, L, wg, (om, lm)
    if cfg.o16:
        O = h16_to_f32(om.view(np.uint16).astype(np.uint32), f16).reshape(H, R, ldo)[..., :D]
    else:
        O = om.view(np.float32).reshape(H, R, ldo)[..., :D]
    L = lm.view(np.float16).astype(np.float32).reshape(H, R) if cfg.l16 else lm.view(np.float32).reshape(H, R)
    return O, L, wg, (om, lm)
