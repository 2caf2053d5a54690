#!/usr/bin/env python3
"""CPU harness for the instruction stream of attn_dq16_p4 (tools/dq4gen.py) on the lane-exact model of tools/p4sim.py:
one workgroup = 256 query rows, all key tiles.  The C++ part of the kernel (attn_dq16_p4.h: Q' and dO fragments, the
D term, the operands of the asm statement, the epilogue's dQ scale) is restated in `run_block`."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dq4gen import Cfg, RING, STAGE, Stream  # noqa: E402
from dkv4sim import LOG2E, rand16, reference as _ref_kv, to_f32  # noqa: E402
from p4sim import ROWMAP, Workgroup, f32_to_h16  # noqa: E402


def reference(q, k, v, do, f16=False, causal=False):
    """float64 backward pass: L (base-2), D (unscaled), dQ"""
    qf, kf, vf, gf = (to_f32(x, f16).astype(np.float64) for x in (q, k, v, do))
    R, C = qf.shape[0], kf.shape[0]
    scale = 1.0 / np.sqrt(qf.shape[1])
    s = qf @ kf.T * scale
    if causal:
        s = np.where(np.arange(C)[None, :] > (np.arange(R)[:, None] + (C - R)), -np.inf, s)
    mx = s.max(axis=1, keepdims=True)
    p = np.exp(s - mx)
    lsum = p.sum(axis=1, keepdims=True)
    p /= lsum
    L = (mx[:, 0] + np.log(lsum[:, 0])) * LOG2E
    dterm = (gf * (p @ vf)).sum(axis=1)
    ds = p * (gf @ vf.T - dterm[:, None]) * scale
    return L, dterm, ds @ kf


def run_block(q, k, v, do, L, Dt, rblk=0, cfg=None, causal=False, dma_mode="late", order=(0, 1, 2, 3), stream=None):
    """q, do [R][128], k, v [C][128] uint16 bit patterns; L (base-2 log-sum-exp), Dt (sum dO o O, unscaled) float [R].
    Returns dQ [256][128] float32 of row block `rblk`."""
    cfg = cfg or Cfg()
    f16 = cfg.dtype == "f16"
    R, C, D = q.shape[0], k.shape[0], cfg.D
    assert q.shape[1] == D
    nks, ndb, pw = D // 16, D // 32, D // 32
    instrs = stream if stream is not None else Stream(cfg).build()
    wg = Workgroup(instrs, dma_mode)
    ld2 = D * 2
    kb_, vb_ = k.reshape(-1).view(np.uint8), v.reshape(-1).view(np.uint8)
    tr = bool(getattr(cfg, "tr", 0))
    if tr:   # K^T, V^T: [D][C] in memory
        assert C % 64 == 0 and D == 128
        kb_, vb_ = np.ascontiguousarray(k.T).reshape(-1).view(np.uint8), np.ascontiguousarray(v.T).reshape(-1).view(np.uint8)
    ldt2 = C * 2
    scale = np.float32(1.0) / np.sqrt(np.float32(D))
    scale2 = np.float32(LOG2E) * scale
    coff = C - R
    nt_total = (C + 63) // 64
    nt = nt_total
    if causal:
        last_row = min(R, (rblk + 1) * 256) - 1
        nt = min(nt_total, (last_row + coff) // 64 + 1)
    ragged = (C % 64 != 0) and nt == nt_total
    qs = q
    if not cfg.exact:
        qs = f32_to_h16((to_f32(q, f16) * scale2).astype(np.float32).reshape(-1), f16).astype(np.uint16).reshape(q.shape)
    lane = np.arange(64)
    qq, hi, n16 = lane & 31, lane >> 5, lane & 15
    for w in wg.waves:
        wave = w.id
        r0 = rblk * 256 + wave * 64
        for base, src in ((128, qs), (192, do)):
            for b in range(2):
                for s_ in range(nks):
                    for l in range(64):
                        row = r0 + b * 32 + int(qq[l])
                        d0 = 16 * s_ + 8 * int(hi[l])
                        if tr and row < R:   # elements 4 hi + {0..3, 8..11} of the step: the order the transposing reads return
                            d0 = 16 * s_ + 4 * int(hi[l])
                            chunk = np.concatenate([src[row, d0:d0 + 4], src[row, d0 + 8:d0 + 12]]).view(np.uint32)
                        else:
                            chunk = src[row, d0:d0 + 8].view(np.uint32) if row < R else np.zeros(4, np.uint32)
                        for t in range(4):
                            w.a[base + 4 * (b * 8 + s_) + t][l] = chunk[t]
        koff, voff = [], []
        for i in range(pw):
            p = (wave * pw + i) * 64 + lane
            db, key, slot = p >> 8, (p >> 2) & 63, p & 3
            chunk = db * 4 + (slot ^ ((key >> 2) & 3))
            if tr:   # image [2 blocks of 32 keys][128 elements][4 chunks of 8 keys ^ (element >> 2) & 3]
                d_ = (p >> 2) & 127
                src_off = d_ * ldt2 + ((p >> 9) * 32 + ((p & 3) ^ ((d_ >> 2) & 3)) * 8) * 2
                koff.append(src_off.astype(np.uint32))
                voff.append(src_off.astype(np.uint32))
                continue
            koff.append((key * ld2 + chunk * 16).astype(np.uint32))
            voff.append((key * ld2 + chunk * 16).astype(np.uint32))
        trow = (n16 >> 2) + 4 * hi
        tchunk = 2 * ((lane >> 4) & 1) + ((n16 & 3) >> 1)
        thalf = (n16 & 3) & 1
        ka0 = qq * 64 + ((hi ^ ((qq >> 2) & 3)) * 16)
        for b in range(2):
            rows = r0 + b * 32 + qq
            ok = rows < R
            lrow = np.where(ok, L[np.minimum(rows, R - 1)], 0.0).astype(np.float32)
            drow = np.where(ok, Dt[np.minimum(rows, R - 1)], 0.0).astype(np.float32)
            w.vn["negl%d" % b] = (-(lrow / scale2 if cfg.exact else lrow)).astype(np.float32).view(np.uint32)
            w.vn["negd%d" % b] = (-drow).astype(np.float32).view(np.uint32)
            lim = np.minimum(C - 1, rows + coff) if causal else np.full(64, C - 1)
            w.vn["lim%d" % b] = (lim - 4 * hi).astype(np.int64).astype(np.uint32)
        w.vn.update({
            "ka0": ka0.astype(np.uint32), "ka1": (ka0 ^ 32).astype(np.uint32),
            "ta0": (trow * 64 + ((tchunk ^ (hi & 3)) * 16) + thalf * 8).astype(np.uint32),
            "ta1": ((trow + 8) * 64 + ((tchunk ^ ((hi + 2) & 3)) * 16) + thalf * 8).astype(np.uint32),
        })
        if tr:
            ta = [(qq * 64 + ((c ^ ((qq >> 2) & 3)) * 16) + 8 * hi).astype(np.uint32) for c in range(4)]
            w.vn.update({"ka0": w.vn["ta0"], "ka1": w.vn["ta1"]})          # rows + 0 / + 8 of a 16-element step (transposing reads)
            w.vn.update({"ta%d" % c: ta[c] for c in range(4)})
        for i in range(4):
            oob = np.full(64, 0xFFFFFF00, np.uint32)
            w.vn["koff%d" % i], w.vn["voff%d" % i] = (koff[i], voff[i]) if i < pw else (oob, oob)
        minlim = min(C - 1, r0 + coff) if causal else C - 1
        maskfrom = (minlim + 1) // 64 if (causal or ragged) else nt
        wnt = nt
        if causal:
            wlast = min(R, r0 + 64) - 1
            wnt = max(1, min(nt, (wlast + coff) // 64 + 1)) if wlast >= r0 else 1
        w.sn.update({"kres": (kb_, C * ld2), "vres": (vb_, C * ld2), "nt": nt, "wnt": wnt,
                     "kinc": 128 if tr else 64 * ld2, "vinc": 128 if tr else 64 * ld2,
                     "wr0": wave * pw * 1024, "ringend": RING * STAGE, "maskfrom": maskfrom, "scale2x2": float(scale2)})
    wg.run(order)
    dQ = np.zeros((256, D), np.float32)
    for w in wg.waves:
        assert not w.lds_q and not w.vm_q, "memory operations left in flight"
        for b in range(2):
            for db in range(ndb):
                for r in range(16):
                    reg = w.a[16 * (4 * b + db) + r].view(np.float32)
                    for h in range(2):
                        dQ[w.id * 64 + b * 32 + np.arange(32), 32 * db + ROWMAP[r][h]] = reg[32 * h:32 * h + 32] * scale
    return dQ, wg


def check(R=256, C=192, cfg=None, causal=False, seed=0, rblk=0, **kw):
    cfg = cfg or Cfg()
    f16 = cfg.dtype == "f16"
    rng = np.random.default_rng(seed)
    q, k, v, do = (rand16((n, cfg.D), rng, f16=f16) for n in (R, C, C, R))
    L, Dt, dq = reference(q, k, v, do, f16, causal)
    dQ, wg = run_block(q, k, v, do, L, Dt, rblk, cfg, causal, **kw)
    n = min(256, R - rblk * 256)
    sl = slice(rblk * 256, rblk * 256 + n)
    return np.abs(dQ[:n] - dq[sl]).max(), np.abs(dq[sl]).max(), wg


if __name__ == "__main__":
    e, m, wg = check()
    print("max |ddQ| %.3e (|dQ| max %.2f)" % (e, m))
    print({k_: v_ for k_, v_ in sorted(wg.waves[0].count.items())})
