#!/usr/bin/env python3
"""CPU harness for the instruction stream of attn_fwd16_p5 (tools/f256gen.py) on the lane-exact model of
tools/p4sim.py: one workgroup = 256 query rows, D = 256, all 32-key steps.  The C++ part of the kernel
(attn_fwd16_p5.h: the Q' fragments parked in LDS, the operands of the asm statement, the epilogue) is restated here."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from f256gen import Cfg, RING, STAGE, Stream  # noqa: E402
from p4sim import FLT_MAX, ROWMAP, Workgroup, f32_to_h16, h16_to_f32, rand_bf16, reference  # noqa: E402

def run_block(q, k, v, rblk, cfg=None, causal=False, dma_mode="late", order=(0, 1, 2, 3), stream=None):
    """q [R][D], k / v [C][D] as uint16 bit patterns (D = cfg.D).  Returns O [256][D] f32, L [256]."""
    cfg = cfg or Cfg()
    D, nks, ndb, pw = cfg.D, cfg.nks, cfg.ndb, cfg.pw
    f16 = cfg.dtype == "f16"
    R, C = q.shape[0], k.shape[0]
    instrs = stream if stream is not None else Stream(cfg).build()
    wg = Workgroup(instrs, dma_mode)
    kb, vb = k.reshape(-1).view(np.uint8), v.reshape(-1).view(np.uint8)
    kt, vt = bool(getattr(cfg, "tr", 0) & 1), bool(getattr(cfg, "tr", 0) & 2)
    # K^T / V^T: [D][C] in memory, DENSE rows here.  The kernel takes whole steps only (C % 32 == 0); on dense rows a ragged last
    # step also comes out right with the same streams -- what follows the sequence in a row is the next row's finite data (zeros
    # behind the last row), K^T's tail is replaced by the edge mask and P = 0 meets finite values in V^T's -- which
    # tests/test_f256_stream.py pins; padded rows (ld > C) may hold anything and would need the last step's offsets of tools/p4gen.py.
    if kt:
        kb = np.ascontiguousarray(k.T).reshape(-1).view(np.uint8)
    if vt:
        vb = np.ascontiguousarray(v.T).reshape(-1).view(np.uint8)
    ldt2 = C * 2
    ld2 = D * 2
    nt_total = (C + 31) // 32
    coff = C - R
    nt = nt_total
    if causal:
        last_row = min(R, (rblk + 1) * 256) - 1
        nt = min(nt_total, (last_row + coff) // 32 + 1)
    ragged = (C % 32 != 0) and nt == nt_total
    scale2 = float(np.float32(1.44269504089) / np.sqrt(np.float32(D)))
    if cfg.fold:
        qf = h16_to_f32(q.astype(np.uint32).reshape(-1), f16).reshape(q.shape)
        q = f32_to_h16((qf * np.float32(scale2)).astype(np.float32).reshape(-1), f16).astype(np.uint16).reshape(q.shape)
    lane = np.arange(64)
    qq, hi, n16 = lane & 31, lane >> 5, lane & 15
    for w in wg.waves:
        wave = w.id
        r0 = rblk * 256 + wave * 64
        back = wave * 32768
        for i in range(2 * nks):
            rb, ks = divmod(i, nks)
            data = np.zeros((64, 16), np.uint8)
            for l in range(64):
                row = r0 + 32 * rb + int(qq[l])
                d0 = 16 * ks + 8 * int(hi[l])
                if row < R and kt:   # elements 4 hi + {0..3, 8..11}: the order the transposing reads of K^T return
                    d0 = 16 * ks + 4 * int(hi[l])
                    data[l] = np.concatenate([q[row, d0:d0 + 4], q[row, d0 + 8:d0 + 12]]).view(np.uint8)
                elif row < R:
                    data[l] = q[row, d0:d0 + 8].view(np.uint8)
            wg.lds_write16(back + i * 1024 + 16 * lane, data)
        koff, koff_t = [], []     # lane offsets of the pieces of a row-major / a transposed tile
        for i in range(pw):
            p = (wave * pw + i) * 64 + lane
            db, key, slot = p >> 7, (p >> 2) & 31, p & 3
            chunk = db * 4 + (slot ^ ((key >> 2) & 3))
            d_ = p >> 2               # transposed: image [D elements][4 chunks of 8 keys ^ (element >> 2) & 3]
            koff_t.append(np.where(d_ < D, d_ * ldt2 + ((p & 3) ^ ((d_ >> 2) & 3)) * 16, 0xFFFFFF00).astype(np.uint32))
            koff.append(np.where(chunk * 8 < D, key * ld2 + chunk * 16, 0xFFFFFF00).astype(np.uint32))
        while len(koff) < 4:
            koff.append(np.full(64, 0xFFFFFF00, np.uint32))
            koff_t.append(np.full(64, 0xFFFFFF00, np.uint32))
        trow = (n16 >> 2) + 4 * hi
        tchunk = 2 * ((lane >> 4) & 1) + ((n16 & 3) >> 1)
        thalf = (n16 & 3) & 1
        ka0 = qq * 64 + ((hi ^ ((qq >> 2) & 3)) * 16)
        m_init = np.full(64, 0.0 if cfg.fold else -FLT_MAX, np.float32).view(np.uint32)
        w.vn.update({
            "m0": m_init.copy(), "m1": m_init.copy(), "l0": np.zeros(64, np.uint32), "l1": np.zeros(64, np.uint32),
            "onesw": np.where(lane < 32, 0xBC00BC00 if f16 else 0xBF80BF80, 0).astype(np.uint32),
            "ka0": ka0.astype(np.uint32), "ka1": (ka0 ^ 32).astype(np.uint32),
            "ta0": ((RING - 1) * STAGE + trow * 64 + ((tchunk ^ (hi & 3)) * 16) + thalf * 8).astype(np.uint32),
            "ta1": ((RING - 1) * STAGE + (trow + 8) * 64 + ((tchunk ^ ((hi + 2) & 3)) * 16) + thalf * 8).astype(np.uint32),
            "qback": (back + 16 * lane).astype(np.uint32),
        })
        vstart = (RING - 1) * STAGE          # the V addresses start one stage behind, like ta0 / ta1 above
        if kt:     # (the lane terms of a transposing read: what ta0 / ta1 are for a row-major V)
            w.vn.update({"ka0": (w.vn["ta0"] - vstart).astype(np.uint32), "ka1": (w.vn["ta1"] - vstart).astype(np.uint32)})
        if vt:
            w.vn.update({"ta%d" % c: (vstart + qq * 64 + ((c ^ ((qq >> 2) & 3)) * 16) + 8 * hi).astype(np.uint32) for c in range(4)})
        for i in range(4):
            w.vn["koff%d" % i], w.vn["voff%d" % i] = (koff_t if kt else koff)[i].copy(), (koff_t if vt else koff)[i].copy()
        for b in range(2):
            row = r0 + b * 32 + qq
            lim = np.minimum(C - 1, row + coff) if causal else np.full(64, C - 1)
            w.vn["lim%d" % b] = (lim - 4 * hi).astype(np.int64).astype(np.uint32)
        minlim = min(C - 1, r0 + coff) if causal else C - 1
        maskfrom = (minlim + 1) // 32 if (causal or ragged) else nt
        wnt = nt
        if causal:
            wlast = min(R, r0 + 64) - 1
            wnt = max(1, min(nt, (wlast + coff) // 32 + 1)) if wlast >= r0 else 1
        w.sn.update({"kres": (kb, C * ld2), "vres": (vb, C * ld2), "nt": nt, "wnt": wnt, "scale2": scale2, "kinc": 64 if kt else 32 * ld2,
                     "vinc": 64 if vt else 32 * ld2, "wr0": wave * pw * 1024, "ringend": RING * STAGE, "maskfrom": maskfrom})
    wg.run(order)
    O = np.zeros((256, D), np.float32)
    L = np.zeros(256, np.float32)
    for w in wg.waves:
        assert not w.lds_q and not w.vm_q, "memory operations left in flight"
        for b in range(2):
            l = w.vn["l%d" % b].view(np.float32)
            mm = w.vn["m%d" % b].view(np.float32)
            ltot = l[:32] + l[32:] + np.float32(1.401298464e-45)
            for db in range(ndb):
                for r in range(16):
                    reg = w.a[16 * (8 * b + db) + r].view(np.float32)
                    for h in range(2):
                        O[w.id * 64 + b * 32 + np.arange(32), 32 * db + ROWMAP[r][h]] = reg[32 * h:32 * h + 32] / ltot
            L[w.id * 64 + b * 32 + np.arange(32)] = mm[:32] + np.log2(ltot)
    return O, L, wg


def check(R=256, C=96, rblk=0, cfg=None, causal=False, seed=0, spike=None, **kw):
    cfg = cfg or Cfg()
    f16 = cfg.dtype == "f16"
    D = cfg.D
    rng = np.random.default_rng(seed)
    q, k, v = (rand_bf16(s, rng, f16=f16) for s in ((R, D), (C, D), (C, D)))
    if spike is not None:
        qrow, krow, gain = spike
        qf = h16_to_f32(q[qrow].astype(np.uint32), f16)
        k[krow] = f32_to_h16((qf * gain).astype(np.float32), f16).astype(np.uint16)
    O, L, wg = run_block(q, k, v, rblk, cfg=cfg, causal=causal, **kw)
    Oref, Lref = reference(q, k, v, causal=causal, f16=f16)
    rows = np.arange(rblk * 256, min(R, rblk * 256 + 256))
    return np.abs(O[: len(rows)] - Oref[rows]).max(), np.abs(L[: len(rows)] - Lref[rows]).max(), wg


if __name__ == "__main__":
    dO, dL, wg = check()
    print("max |dO| %.3e  max |dL| %.3e" % (dO, dL))
    print({k_: v_ for k_, v_ in sorted(wg.waves[0].count.items())})
