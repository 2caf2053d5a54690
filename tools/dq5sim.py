#!/usr/bin/env python3
"""CPU harness for the instruction stream of attn_dq16_p5 (tools/dq5gen.py) on the lane-exact model of tools/p4sim.py: one
workgroup = two wave pairs x 64 query rows (waves 0, 1: S-role; waves 2, 3: P-role), all key blocks.  The C++ part of the kernel
(attn_dq16_p5.h: Q' / dO fragments, the D term, the operands of the asm statement, the epilogue's dQ scale) is restated in
`run_block`."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dkv4sim import LOG2E, rand16, to_f32  # noqa: E402
from dq4sim import reference  # noqa: E402
from dq5gen import KRING, VRING, XPAR, Cfg, Stream  # noqa: E402
from p4sim import ROWMAP, Workgroup, f32_to_h16  # noqa: E402

WGROWS = 128


def run_block(q, k, v, do, L, Dt, rblk=0, cfg=None, causal=False, dma_mode="late", order=(0, 1, 2, 3), stream=None):
    """q, do [R][Dr], k, v [C][Dr] uint16 bit patterns (Dr <= cfg.D); L (base-2 log-sum-exp), Dt (sum dO o O, unscaled) float [R].
    Returns dQ [128][cfg.D] float32 of row block `rblk`."""
    cfg = cfg or Cfg()
    f16 = cfg.dtype == "f16"
    R, C, D = q.shape[0], k.shape[0], cfg.D
    Dr = q.shape[1]
    assert Dr <= D and Dr % 8 == 0
    nks, npw, TI = cfg.nks, cfg.NPW, cfg.TI
    instrs = stream if stream is not None else Stream(cfg).build()
    wg = Workgroup(instrs, dma_mode)
    ld2 = Dr * 2
    kb_, vb_ = k.reshape(-1).view(np.uint8), v.reshape(-1).view(np.uint8)
    scale = np.float32(1.0) / np.sqrt(np.float32(Dr))
    scale2 = np.float32(LOG2E) * scale
    coff = C - R
    nt_total = (C + 31) // 32
    nt = nt_total
    if causal:
        last_row = min(R, (rblk + 1) * WGROWS) - 1
        nt = min(nt_total, (last_row + coff) // 32 + 1)
    ragged = (C % 32 != 0) and nt == nt_total
    qs = q
    if not cfg.exact:
        qs = f32_to_h16((to_f32(q, f16) * scale2).astype(np.float32).reshape(-1), f16).astype(np.uint16).reshape(q.shape)
    lane = np.arange(64)
    qq, hi, n16 = lane & 31, lane >> 5, lane & 15
    for w in wg.waves:
        wave = w.id
        pair, role = wave & 1, wave >> 1
        r0 = rblk * WGROWS + pair * 64
        src = qs if role == 0 else do
        for b in range(2):
            for s_ in range(nks):
                for l in range(64):
                    row = r0 + b * 32 + int(qq[l])
                    d0 = 16 * s_ + 8 * int(hi[l])
                    chunk = src[row, d0:d0 + 8].view(np.uint32) if (row < R and d0 < Dr) else np.zeros(4, np.uint32)
                    for t in range(4):
                        w.a[128 + 4 * (16 * b + s_) + t][l] = chunk[t]
        koff = []
        for i in range(4):
            p = (wave * npw + i) * 64 + lane
            db, key, slot = p >> 7, (p >> 2) & 31, p & 3
            chunk = db * 4 + (slot ^ ((key >> 2) & 3))
            koff.append(np.where((i < npw) & (chunk * 8 < Dr), key * ld2 + chunk * 16, 0xFFFFFF00).astype(np.uint32))
        trow = (n16 >> 2) + 4 * hi
        tchunk = 2 * ((lane >> 4) & 1) + ((n16 & 3) >> 1)
        thalf = (n16 & 3) & 1
        rbase = 0 if role == 0 else cfg.VR0            # row reads: K ring (S-role) or V ring (P-role)
        tback = (2 if role == 0 else 1) * TI           # transposing reads start two / one K blocks behind
        ra0 = rbase + qq * 64 + ((hi ^ ((qq >> 2) & 3)) * 16)
        for b in range(2):
            rows = r0 + b * 32 + qq
            ok = rows < R
            lrow = np.where(ok, L[np.minimum(rows, R - 1)], 0.0).astype(np.float32)
            drow = np.where(ok, Dt[np.minimum(rows, R - 1)], 0.0).astype(np.float32)
            term = -(lrow / scale2 if cfg.exact else lrow) if role == 0 else -drow
            w.vn["negt%d" % b] = term.astype(np.float32).view(np.uint32)
            lim = np.minimum(C - 1, rows + coff) if causal else np.full(64, C - 1)
            w.vn["lim%d" % b] = (lim - 4 * hi).astype(np.int64).astype(np.uint32)
        w.vn.update({
            "ra0": ra0.astype(np.uint32), "ra1": (ra0 ^ 32).astype(np.uint32),
            "ta0": ((trow * 64 + ((tchunk ^ (hi & 3)) * 16) + thalf * 8 - tback) & 0xFFFFFFFF).astype(np.uint32),
            "ta1": (((trow + 8) * 64 + ((tchunk ^ ((hi + 2) & 3)) * 16) + thalf * 8 - tback) & 0xFFFFFFFF).astype(np.uint32),
            "xp": (cfg.XP0 + pair * 2 * XPAR + 16 * lane).astype(np.uint32),
            "xs": (cfg.XS0 + pair * 2 * XPAR + 16 * lane).astype(np.uint32),
        })
        for i in range(4):
            w.vn["koff%d" % i], w.vn["voff%d" % i] = koff[i].copy(), koff[i].copy()
        minlim = min(C - 1, rblk * WGROWS + coff) if causal else C - 1       # first row of the WORKGROUP (both waves of a pair mask alike)
        minlim = min(C - 1, r0 + coff) if causal else C - 1
        maskfrom = (minlim + 1) // 32 if (causal or ragged) else nt
        w.sn.update({"kres": (kb_, C * ld2), "vres": (vb_, C * ld2), "nt": nt, "kinc": 32 * ld2, "vinc": 32 * ld2,
                     "wrk0": wave * npw * 1024, "wrv0": cfg.VR0 + wave * npw * 1024, "kend": KRING * TI, "vend": cfg.VR0 + VRING * TI,
                     "maskfrom": maskfrom, "scale2x2": float(scale2), "role": role})
    wg.run(order)
    dQ = np.zeros((WGROWS, D), np.float32)
    for w in wg.waves:
        assert not w.lds_q and not w.vm_q, "memory operations left in flight"
        pair, role = w.id & 1, w.id >> 1
        for b in range(2):
            for dbl in range(cfg.share(role)):
                db = dbl if role == 0 else cfg.ndbs + dbl
                for r in range(16):
                    reg = w.a[16 * (4 * b + dbl) + r].view(np.float32)
                    for h in range(2):
                        dQ[pair * 64 + b * 32 + np.arange(32), 32 * db + ROWMAP[r][h]] = reg[32 * h:32 * h + 32] * scale
    return dQ, wg


def check(R=128, C=96, cfg=None, causal=False, seed=0, rblk=0, Dr=None, **kw):
    cfg = cfg or Cfg()
    f16 = cfg.dtype == "f16"
    Dr = Dr or cfg.D
    rng = np.random.default_rng(seed)
    q, k, v, do = (rand16((n, Dr), rng, f16=f16) for n in (R, C, C, R))
    L, Dt, dq = reference(q, k, v, do, f16, causal)
    dQ, wg = run_block(q, k, v, do, L, Dt, rblk, cfg, causal, **kw)
    n = min(WGROWS, R - rblk * WGROWS)
    sl = slice(rblk * WGROWS, rblk * WGROWS + n)
    return np.abs(dQ[:n, :Dr] - dq[sl]).max(), np.abs(dq[sl]).max(), wg


if __name__ == "__main__":
    import time
    for D in (256, 192, 160):
        t0 = time.time()
        e, m, wg = check(cfg=Cfg("bf16", D=D))
        print("D=%d  max |ddQ| %.3e (|dQ| max %.2f)  %.1f s" % (D, e, m, time.time() - t0))
