#!/usr/bin/env python3
"""CPU harness for the instruction stream of attn_dkv16_p5 (tools/dkv5gen.py) on the lane-exact model of tools/p4sim.py: one
workgroup = two wave pairs x 64 keys (waves 0, 1: V-role, dV; waves 2, 3: K-role, dK), all row blocks.  The C++ part of the kernel
(attn_dkv16_p5.h: the cached fragments parked in LDS, the operands of the asm statement, the epilogue's dK scale) is restated
in `run_block`."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dkv4sim import load_prec, rand16, reference, store_prec, to_f32  # noqa: E402
from dkv5gen import RING, XPAR, Cfg, Stream  # noqa: E402
from p4sim import ROWMAP, Workgroup, f32_to_h16  # noqa: E402

LOG2E = 1.44269504089
WGKEYS = 128


def run_block(q, k, v, do, L, Dt, cblk=0, cfg=None, causal=False, dma_mode="late", order=(0, 1, 2, 3), scale=None, stream=None, Dr=None):
    """q, do [R][Dr], k, v [C][Dr] as uint16 bit patterns (Dr <= cfg.D: the head dimension inside the bucket); L, Dt float arrays
    [R] (stored in cfg.lprec / cfg.dprec).  Returns dV, dK [128][cfg.D] float32 of key block `cblk`."""
    cfg = cfg or Cfg()
    f16, g16 = cfg.dtype == "f16", cfg.gdtype == "f16"
    R, C, D = q.shape[0], k.shape[0], cfg.D
    Dr = Dr or q.shape[1]
    assert q.shape[1] == Dr and Dr <= D and Dr % 8 == 0
    nks, ndb, npw = cfg.nks, cfg.ndb, cfg.NPW
    instrs = stream if stream is not None else Stream(cfg).build()
    wg = Workgroup(instrs, dma_mode)
    ld2 = Dr * 2
    qb, gb = q.reshape(-1).view(np.uint8), do.reshape(-1).view(np.uint8)
    lbuf, lesz = store_prec(L, cfg.lprec)
    dbuf, desz = store_prec(Dt, cfg.dprec)
    assert lesz == desz
    scale = np.float32(scale if scale is not None else 1.0 / np.sqrt(np.float32(Dr)))
    scale2 = np.float32(LOG2E) * scale
    coff = C - R
    row_first = 0
    if causal:      # the first 32-row block that sees the workgroup's first key
        row_first = max(0, cblk * WGKEYS - coff) // 32 * 32
    nsteps = max(1, (R - row_first + 31) // 32)
    kp = f32_to_h16((to_f32(k, f16) * scale2).astype(np.float32).reshape(-1), f16).astype(np.uint16).reshape(k.shape)   # K' = K * scale2
    if cfg.exact:
        kp = k
    vg = v
    if cfg.mix:     # V in dO's type (attn_dkv16_p5.h converts the fragments once)
        vg = f32_to_h16(to_f32(v, f16).reshape(-1), g16).astype(np.uint16).reshape(v.shape)
    lane = np.arange(64)
    kc, hi, n16 = lane & 31, lane >> 5, lane & 15
    xb = RING * cfg.STAGE
    for w in wg.waves:
        wave = w.id
        pair, role = wave & 1, wave >> 1
        c0 = cblk * WGKEYS + 64 * pair
        back = wave * (2 * nks * 1024)   # overlaps the ring: the stream reads the fragments back, then a barrier, before the first DMA
        src = kp if role == 0 else vg
        for kb_ in range(2):
            for ks in range(nks):
                data = np.zeros((64, 16), np.uint8)
                for l in range(64):
                    col = c0 + 32 * kb_ + int(kc[l])
                    d0 = 16 * ks + 8 * int(hi[l])
                    if col < C and d0 < Dr:
                        data[l] = src[col, d0:d0 + 8].view(np.uint8)
                wg.lds_write16(back + (nks * kb_ + ks) * 1024 + 16 * lane, data)
        offs = []
        for i in range(4):
            p = (npw * wave + i) * 64 + lane
            db, row, slot = p >> 7, (p >> 2) & 31, p & 3
            chunk = db * 4 + (slot ^ ((row >> 2) & 3))
            offs.append(np.where((i < npw) & (chunk * 8 < Dr), (row_first + row) * ld2 + chunk * 16, 0xFFFFFF00).astype(np.uint32))
        trow = (n16 >> 2) + 4 * hi
        tchunk = 2 * ((lane >> 4) & 1) + ((n16 & 3) >> 1)
        thalf = (n16 & 3) & 1
        ra0 = kc * 64 + ((hi ^ ((kc >> 2) & 3)) * 16)
        ones = (0xBC00BC00 if (f16 if role == 0 else g16) else 0xBF80BF80)
        back_stage = (1 << 32) - cfg.STAGE          # the transposing reads start one stage behind the row reads (block -1)
        w.vn.update({"qoff%d" % i: offs[i].copy() for i in range(4)})
        w.vn.update({"goff%d" % i: offs[i].copy() for i in range(4)})
        w.vn.update({
            "ldoff": ((row_first + kc) * lesz).astype(np.uint32),
            "ra0": ra0.astype(np.uint32), "ra1": (ra0 ^ 32).astype(np.uint32),
            "ta0": ((trow * 64 + ((tchunk ^ (hi & 3)) * 16) + thalf * 8 + back_stage) & 0xFFFFFFFF).astype(np.uint32),
            "ta1": (((trow + 8) * 64 + ((tchunk ^ ((hi + 2) & 3)) * 16) + thalf * 8 + back_stage) & 0xFFFFFFFF).astype(np.uint32),
            "onesw": np.where(lane < 32, ones, 0).astype(np.uint32),
            "tk": ((c0 + kc) - coff - 4 * hi - row_first).astype(np.int64).astype(np.uint32),
            "kvback": (back + 16 * lane).astype(np.uint32),
            "xaddr": (xb + pair * 2 * XPAR + 16 * lane).astype(np.uint32),
        })
        maskuntil = 0
        if causal:   # blocks whose rows do not all see this wave's last key
            maskuntil = max(0, -(-(c0 + 63 - coff - row_first) // 32))
        w.sn.update({"qres": (qb, R * ld2), "gres": (gb, R * ld2), "ldres": ((lbuf, R * lesz) if role == 0 else (dbuf, R * desz)),
                     "nsteps": nsteps, "rscale": float(np.float32(1.0) / scale),
                     "rscale2": float(np.float32(1.0) / (scale2 if cfg.exact else np.float32(1.0))),
                     "qinc": 32 * ld2, "ginc": 32 * ld2, "ldinc": 32 * lesz, "wr0": wave * npw * 1024,
                     "ringend": RING * cfg.STAGE, "maskuntil": maskuntil, "scale2x2": float(scale2), "role": role})
    wg.run(order)
    dV = np.zeros((WGKEYS, D), np.float32)
    dK = np.zeros((WGKEYS, D), np.float32)
    for w in wg.waves:
        assert not w.lds_q and not w.vm_q, "memory operations left in flight"
        pair, role = w.id & 1, w.id >> 1
        out = dV if role == 0 else dK
        for kb_ in range(2):
            keys = 64 * pair + 32 * kb_ + np.arange(32)
            for db in range(ndb):
                for r in range(16):
                    for h in range(2):
                        dcol = 32 * db + ROWMAP[r][h]
                        out[keys, dcol] = w.a[16 * (2 * db + kb_) + r].view(np.float32)[32 * h:32 * h + 32] * (scale if role else np.float32(1.0))
    return dV, dK, wg


def check(R=96, C=128, cfg=None, causal=False, seed=0, cblk=0, Dr=None, **kw):
    cfg = cfg or Cfg()
    f16 = cfg.dtype == "f16"
    Dr = Dr or cfg.D
    rng = np.random.default_rng(seed)
    q, k, v = (rand16((n, Dr), rng, f16=f16) for n in (R, C, C))
    do = rand16((R, Dr), rng, f16=cfg.gdtype == "f16")
    L, Dt, dv, dk = reference(q, k, v, do, f16, causal, g16=cfg.gdtype == "f16")
    Ls = load_prec(store_prec(L, cfg.lprec)[0], cfg.lprec)
    Ds = load_prec(store_prec(Dt, cfg.dprec)[0], cfg.dprec)
    dV, dK, wg = run_block(q, k, v, do, Ls, Ds, cblk, cfg, causal, Dr=Dr, **kw)
    n = min(WGKEYS, C - cblk * WGKEYS)
    sl = slice(cblk * WGKEYS, cblk * WGKEYS + n)
    return (np.abs(dV[:n, :Dr] - dv[sl]).max(), np.abs(dK[:n, :Dr] - dk[sl]).max(), np.abs(dv[sl]).max(), np.abs(dk[sl]).max(), wg)


if __name__ == "__main__":
    import time
    for D in (256, 192, 160):
        t0 = time.time()
        ev, ek, mv, mk, wg = check(cfg=Cfg("bf16", "f16", "bf16", D=D))
        print("D=%d  max |ddV| %.3e (|dV| max %.2f)   max |ddK| %.3e (|dK| max %.2f)   %.1f s" % (D, ev, mv, ek, mk, time.time() - t0))
