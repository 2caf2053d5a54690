#!/usr/bin/env python3
"""CPU harness for the instruction stream of attn_dkv16_p4 (tools/dkv4gen.py) on the lane-exact model of
tools/p4sim.py: one workgroup = 256 keys, all row steps.  The C++ part of the kernel (attn_dkv16_p4.h: the K' / V
fragments parked in LDS, the operands of the asm statement, the epilogue's dK scale) is restated in `run_block`."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dkv4gen import Cfg, RING, STAGE, Stream  # noqa: E402
from p4sim import ROWMAP, Workgroup, f32_to_h16, h16_to_f32  # noqa: E402

LOG2E = 1.44269504089


def to_f32(x, f16):
    return h16_to_f32(x.astype(np.uint32).reshape(-1), f16).reshape(x.shape)


def store_prec(x, prec):
    """values as the forward / dQ kernels leave them in memory: bytes + element size"""
    x = np.asarray(x, np.float32)
    if prec == "f32":
        return x.copy().view(np.uint8), 4
    return f32_to_h16(x.copy(), prec == "f16").astype(np.uint16).view(np.uint8), 2


def load_prec(buf, prec):
    if prec == "f32":
        return buf.view(np.float32).astype(np.float64)
    return h16_to_f32(buf.view(np.uint16).astype(np.uint32), prec == "f16").astype(np.float64)


def reference(q, k, v, do, f16=False, causal=False, scale=None, g16=None):
    """float64 backward pass on the 16-bit inputs: L (base-2 log-sum-exp of the scaled scores), D * scale, dV, dK"""
    qf, kf, vf = (to_f32(x, f16).astype(np.float64) for x in (q, k, v))
    gf = to_f32(do, f16 if g16 is None else g16).astype(np.float64)
    R, C = qf.shape[0], kf.shape[0]
    scale = scale if scale is not None else 1.0 / np.sqrt(qf.shape[1])
    s = qf @ kf.T * scale
    if causal:
        s = np.where(np.arange(C)[None, :] > (np.arange(R)[:, None] + (C - R)), -np.inf, s)
    mx = s.max(axis=1, keepdims=True)
    p = np.exp(s - mx)
    lsum = p.sum(axis=1, keepdims=True)
    p /= lsum
    L = (mx[:, 0] + np.log(lsum[:, 0])) * LOG2E
    o = p @ vf
    dterm = (gf * o).sum(axis=1)
    dv = p.T @ gf
    dp = gf @ vf.T
    ds = p * (dp - dterm[:, None]) * scale
    dk = ds.T @ qf
    return L, dterm * scale, dv, dk


def run_block(q, k, v, do, L, Dt, cblk=0, cfg=None, causal=False, dma_mode="late", order=(0, 1, 2, 3), scale=None, stream=None):
    """q, do [R][128], k, v [C][128] as uint16 bit patterns; L, Dt float arrays [R] (stored in cfg.lprec / cfg.dprec).
    Returns dV, dK [256][128] float32 of key block `cblk`."""
    cfg = cfg or Cfg()
    f16 = cfg.dtype == "f16"
    R, C, D = q.shape[0], k.shape[0], cfg.D
    nks, ndb, pw = cfg.nks, cfg.ndb, (2 if cfg.D == 128 else 1)
    instrs = stream if stream is not None else Stream(cfg).build()
    wg = Workgroup(instrs, dma_mode)
    ld2 = D * 2
    qb, gb = q.reshape(-1).view(np.uint8), do.reshape(-1).view(np.uint8)
    tr = bool(getattr(cfg, "tr", 0))
    if tr:   # Q^T, dO^T: [D][R] in memory
        assert R % 32 == 0 and D == 128
        qb, gb = np.ascontiguousarray(q.T).reshape(-1).view(np.uint8), np.ascontiguousarray(do.T).reshape(-1).view(np.uint8)
    ldt2 = R * 2
    lbuf, lesz = store_prec(L, cfg.lprec)
    dbuf, desz = store_prec(Dt, cfg.dprec)
    assert lesz == desz
    scale = np.float32(scale if scale is not None else 1.0 / np.sqrt(np.float32(D)))
    scale2 = np.float32(LOG2E) * scale
    coff = C - R
    c0 = cblk * 256
    row_first = 0
    if causal:      # the first 32-row block that sees the workgroup's first key
        row_first = max(0, c0 - coff) // 32 * 32
    nsteps = max(1, (R - row_first + 31) // 32)
    kfl, vfl = to_f32(k, f16), v
    kp = f32_to_h16((kfl * scale2).astype(np.float32).reshape(-1), f16).astype(np.uint16).reshape(k.shape)   # K' = K * scale2
    if cfg.exact:
        kp = k
    if cfg.mix:     # V in dO's type (attn_dkv16_p4.h converts the fragments once)
        vfl = f32_to_h16(to_f32(v, f16).reshape(-1), cfg.gdtype == "f16").astype(np.uint16).reshape(v.shape)
    lane = np.arange(64)
    kc, hi, n16 = lane & 31, lane >> 5, lane & 15
    for w in wg.waves:
        wave = w.id
        back = wave * 32768   # overlaps the ring: the stream reads the fragments back, then a barrier, before the first DMA
        for i in range(32):
            src = kp if i < 16 else vfl
            kb_, ks = divmod(i % 16, 8)
            if ks >= nks:
                continue
            data = np.zeros((64, 16), np.uint8)
            for l in range(64):
                col = c0 + 64 * wave + 32 * kb_ + int(kc[l])
                d0 = 16 * ks + 8 * int(hi[l])
                if col < C and tr:   # elements 4 hi + {0..3, 8..11}: the order the transposing reads of Q^T / dO^T return
                    d0 = 16 * ks + 4 * int(hi[l])
                    data[l] = np.concatenate([src[col, d0:d0 + 4], src[col, d0 + 8:d0 + 12]]).view(np.uint8)
                elif col < C:
                    data[l] = src[col, d0:d0 + 8].view(np.uint8)
            wg.lds_write16(back + i * 1024 + 16 * lane, data)
        # DMA source offsets: piece i of wave w fills 16-byte positions (2 w + i) * 64 + lane of a tile
        offs = []
        for i in range(2):
            p = (pw * wave + i) * 64 + lane
            db, row, slot = p >> 7, (p >> 2) & 31, p & 3
            chunk = db * 4 + (slot ^ ((row >> 2) & 3))
            if tr:   # image [128 elements][4 chunks of 8 rows ^ (element >> 2) & 3]
                d_ = p >> 2
                offs.append((d_ * ldt2 + (row_first + ((p & 3) ^ ((d_ >> 2) & 3)) * 8) * 2).astype(np.uint32))
                continue
            offs.append(np.where((i < pw) & (chunk * 8 < D), (row_first + row) * ld2 + chunk * 16, 0xFFFFFF00).astype(np.uint32))
        trow = (n16 >> 2) + 4 * hi
        tchunk = 2 * ((lane >> 4) & 1) + ((n16 & 3) >> 1)
        thalf = (n16 & 3) & 1
        ra0 = kc * 64 + ((hi ^ ((kc >> 2) & 3)) * 16)
        w.vn.update({
            "qoff0": offs[0].copy(), "qoff1": offs[1].copy(), "goff0": offs[0].copy(), "goff1": offs[1].copy(),
            "ldoff": ((row_first + kc) * lesz).astype(np.uint32),
            "ra0": ra0.astype(np.uint32), "ra1": (ra0 ^ 32).astype(np.uint32),
            "ta0": (trow * 64 + ((tchunk ^ (hi & 3)) * 16) + thalf * 8).astype(np.uint32),
            "ta1": ((trow + 8) * 64 + ((tchunk ^ ((hi + 2) & 3)) * 16) + thalf * 8).astype(np.uint32),
            "onesw": np.where(lane < 32, 0xC000C000 if cfg.mix else (0xBC00BC00 if f16 else 0xBF80BF80), 0).astype(np.uint32),
            "tk": ((c0 + 64 * wave + kc) - coff - 4 * hi - row_first).astype(np.int64).astype(np.uint32),
            "kvback": (back + 16 * lane).astype(np.uint32),
        })
        if tr:
            ta = [(kc * 64 + ((c ^ ((kc >> 2) & 3)) * 16) + 8 * hi).astype(np.uint32) for c in range(4)]
            w.vn.update({"ra0": w.vn["ta0"], "ra1": w.vn["ta1"]})          # rows + 0 / + 8 of a 16-element step (transposing reads)
            w.vn.update({"ta%d" % c: ta[c] for c in range(4)})
        maskuntil = 0
        if causal:   # steps whose rows do not all see this wave's last key
            maskuntil = max(0, -(-(c0 + 64 * wave + 63 - coff - row_first) // 32))
        w.sn.update({"qres": (qb, R * ld2), "gres": (gb, R * ld2), "lres": (lbuf, R * lesz), "dres": (dbuf, R * desz),
                     "nsteps": nsteps, "rscale": float(np.float32(0.5 if cfg.mix else 1.0) / scale),
                     "qinc": 64 if tr else 32 * ld2, "ginc": 64 if tr else 32 * ld2,
                     "ldinc": 32 * lesz, "wr0": wave * pw * 1024, "ringend": RING * STAGE, "maskuntil": maskuntil,
                     "rscale2": float(np.float32(0.5 if cfg.mix else 1.0) / (scale2 if cfg.exact else np.float32(1.0))), "scale2x2": float(scale2)})
    wg.run(order)
    dV = np.zeros((256, D), np.float32)
    dK = np.zeros((256, D), np.float32)
    for w in wg.waves:
        assert not w.lds_q and not w.vm_q, "memory operations left in flight"
        for kb_ in range(2):
            keys = 64 * w.id + 32 * kb_ + np.arange(32)
            for db in range(ndb):
                for r in range(16):
                    for h in range(2):
                        dcol = 32 * db + ROWMAP[r][h]
                        dV[keys, dcol] = w.a[16 * (2 * db + kb_) + r].view(np.float32)[32 * h:32 * h + 32]
                        dK[keys, dcol] = w.a[128 + 16 * (2 * db + kb_) + r].view(np.float32)[32 * h:32 * h + 32] * scale
    return dV, dK, wg


def rand16(shape, rng, scale=1.0, f16=False):
    x = (rng.standard_normal(shape) * scale).astype(np.float32)
    return f32_to_h16(x.reshape(-1), f16).astype(np.uint16).reshape(shape)


def check(R=96, C=256, cfg=None, causal=False, seed=0, cblk=0, **kw):
    cfg = cfg or Cfg()
    f16 = cfg.dtype == "f16"
    rng = np.random.default_rng(seed)
    q, k, v = (rand16((n, cfg.D), rng, f16=f16) for n in (R, C, C))
    do = rand16((R, cfg.D), rng, f16=cfg.gdtype == "f16")
    L, Dt, dv, dk = reference(q, k, v, do, f16, causal, g16=cfg.gdtype == "f16")
    # the kernel sees L and D as stored
    Ls = load_prec(store_prec(L, cfg.lprec)[0], cfg.lprec)
    Ds = load_prec(store_prec(Dt, cfg.dprec)[0], cfg.dprec)
    dV, dK, wg = run_block(q, k, v, do, Ls, Ds, cblk, cfg, causal, **kw)
    n = min(256, C - cblk * 256)
    sl = slice(cblk * 256, cblk * 256 + n)
    return (np.abs(dV[:n] - dv[sl]).max(), np.abs(dK[:n] - dk[sl]).max(), np.abs(dv[sl]).max(), np.abs(dk[sl]).max(), wg)


if __name__ == "__main__":
    ev, ek, mv, mk, wg = check()
    print("max |ddV| %.3e (|dV| max %.2f)   max |ddK| %.3e (|dK| max %.2f)" % (ev, mv, ek, mk))
    print({k_: v_ for k_, v_ in sorted(wg.waves[0].count.items())})
