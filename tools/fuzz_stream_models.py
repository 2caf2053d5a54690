#!/usr/bin/env python3
"""Random problems for the lane-exact CPU models of the hand-placed instruction streams (no GPU): every compiled stream family --
forward D <= 128 (tools/p4gen.py, with the transposed families), the persistent forward stream (p4pgen), forward 128 < D <= 256
(f256gen, with the transposed and the model-only one-operand streams), backwardQuery (dq4gen) and backwardKeyValue (dkv4gen), with
their transposed streams, and the role-split backward streams of the 160 / 192 / 256 buckets (dq5gen, dkv5gen) -- on random shapes, row / column blocks, causal or dense, LDS-DMA landing as early or as late as the
waits allow, waves in a random order.  Each case goes through the `_check` helper of that family's test file (same float64
reference, same tolerances as tests/test_*_stream.py); the seeded test matrices pin chosen corners, this walks between them.

    python tools/fuzz_stream_models.py --family p4 --cases 40 --seed 1
"""
import argparse
import os
import sys
import time
import traceback

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", "tests"))


def modes(rng):
    order = tuple(int(x) for x in rng.permutation(4))
    return dict(dma_mode=str(rng.choice(["early", "late"])), order=order)


def shape(rng, rmax, cmax, cmul=1, rmul=1, causal=False):
    R = int(rng.integers(1, rmax // rmul + 1)) * rmul
    C = int(rng.integers(1, cmax // cmul + 1)) * cmul
    if causal and C < R:      # (the causal mask of the extension is anchored at the last key: rows need C >= R to see any)
        C = ((R + cmul - 1) // cmul) * cmul
    return R, C


def fuzz_p4(rng):
    import p4gen
    import test_p4_stream as t
    names = [n for n in p4gen.PRODUCT_STREAMS] + list(p4gen.TR_STREAMS)
    name = str(rng.choice(names))
    cfg = p4gen.VARIANTS[name]
    causal = bool(rng.integers(2))
    tr = bool(getattr(cfg, "tr", 0))
    R, C = shape(rng, 600, 640, cmul=8 if tr else 1, causal=causal)
    rblk = int(rng.integers((R + 255) // 256))
    kw = dict(tr_pad=int(rng.choice([0, 8, 24]))) if tr else {}
    t._check(R, C, rblk=rblk, causal=causal, cfg=cfg, seed=int(rng.integers(1 << 30)), tol_o=6e-3, **modes(rng), **kw)
    return name, R, C, rblk, causal


def fuzz_p4p(rng):
    import p4pgen
    import test_p4p_stream as t
    name = str(rng.choice(sorted(p4pgen.PRODUCT_STREAMS)))
    cfg = p4pgen.VARIANTS[name]
    H = int(rng.integers(1, 3))
    R, C = shape(rng, 700, 640, causal=bool(cfg.causal))
    m = modes(rng)
    # causal + BF16: the first rows of the first block see one or a few keys -- the 8-bit roundings of P (and, in the FOLD
    # streams, of Q' = Q log2(e) / sqrt(D): scores off by 2^-9 |s|) are not averaged out, |dO| reaches 4-5.3e-3 where the seeded
    # matrix's 4e-3 holds from a few dozen keys on (found by this tool: seed 3 cases 72, 287, 365, all BF16_FOLD_L16_CAUSAL, R = C)
    tol_o = 1e-2 if (cfg.causal and cfg.dtype == "bf16" and not cfg.o16) else None
    if tol_o is None and cfg.fold and cfg.dtype == "bf16" and not cfg.o16:
        tol_o = 8e-3   # dense, folded scale: Q' rounded to 8 bits moves peaked rows by up to 5.3e-3 (seed 8 cases 61, 275; seeded matrix: 4e-3)
    D = int(rng.choice([128, 128, 128, 120, 104, 96, 80, 72]))     # head dimensions below the bucket: chunks beyond D fetched out of range
    kw = dict(D=D, ld=128) if D < 128 else {}
    t._check(H, R, C, cfg=cfg, seed=int(rng.integers(1 << 30)), stores=str(rng.choice(["early", "late"])), tol_o=tol_o, **m, **kw)
    return name, H, R, C, D


def fuzz_p5(rng):
    import f256gen
    import test_f256_stream as t
    pool = {n: c for n, c in f256gen.VARIANTS.items() if not c.prof and c.D > 128}
    pool.update(f256gen.TR_VARIANTS)
    pool.update(f256gen.MODEL_ONLY_VARIANTS)
    name = str(rng.choice(sorted(pool)))
    cfg = pool[name]
    causal = bool(rng.integers(2))
    R, C = shape(rng, 520, 352, cmul=32 if cfg.tr else 1, causal=causal)
    rblk = int(rng.integers((R + 255) // 256))
    t._check(R, C, rblk=rblk, causal=causal, cfg=cfg, seed=int(rng.integers(1 << 30)), tol_o=6e-3, **modes(rng))
    return name, R, C, rblk, causal


def fuzz_dq4(rng):
    import dq4gen
    import test_dq4_stream as t
    pool = {n: c for n, c in dq4gen.VARIANTS.items() if not c.prof and not c.abl}
    pool.update(dq4gen.TR_VARIANTS)
    name = str(rng.choice(sorted(pool)))
    cfg = pool[name]
    causal = bool(rng.integers(2))
    R, C = shape(rng, 520, 640, cmul=64 if cfg.tr else 1, causal=causal)
    rblk = int(rng.integers((R + 255) // 256))
    t._check(R, C, rblk=rblk, causal=causal, cfg=cfg, seed=int(rng.integers(1 << 30)), **modes(rng))
    return name, R, C, rblk, causal


def fuzz_dkv4(rng):
    import dkv4gen
    import test_dkv4_stream as t
    pool = {n: c for n, c in dkv4gen.VARIANTS.items() if not c.prof and not c.abl}
    pool.update(dkv4gen.TR_VARIANTS)
    name = str(rng.choice(sorted(pool)))
    cfg = pool[name]
    causal = bool(rng.integers(2))
    R, C = shape(rng, 420, 600, rmul=32 if cfg.tr else 1, causal=causal)
    cblk = int(rng.integers((C + 255) // 256))
    import dkv4sim
    ev, ek, mv, mk, _ = dkv4sim.check(R=R, C=C, cblk=cblk, causal=causal, cfg=cfg, seed=int(rng.integers(1 << 30)), **modes(rng))
    # as tests/test_dkv4_stream.py, except that FP16 streams with the reference's mixed storage (L in FP16, D in BF16) get the BF16
    # bound: D's 8 bits of mantissa enter dS' = P (dP - D) whatever the type of the operands, and the short rows of a causal block do
    # not average it out (seed 3 case 397: the row-major and the transposed stream differ from float64 by the same 4.7e-3 |dK|_max)
    rel = 2.5e-3 if (cfg.dtype == "f16" and not cfg.mix and cfg.dprec == "f32") else 1.2e-2
    assert ev < rel * max(1.0, mv) and ek < rel * max(1.0, mk), (ev, mv, ek, mk)
    return name, R, C, cblk, causal


def fuzz_dq5(rng):
    """role-split backwardQuery streams (buckets 160 / 192 / 256): every loop exit (1 .. 14 key blocks), row blocks, head dimensions
    inside the bucket"""
    import dq5gen
    import test_dq5_stream as t
    pool = {n: c for n, c in dq5gen.VARIANTS.items() if not c.prof}
    name = str(rng.choice(sorted(pool)))
    cfg = pool[name]
    causal = bool(rng.integers(2))
    R, C = shape(rng, 400, 448, causal=causal)
    rblk = int(rng.integers((R + 127) // 128))
    Dr = int(rng.choice([cfg.D, cfg.D, cfg.D - 8, cfg.D - 24]))
    t._check(R, C, rblk=rblk, causal=causal, cfg=cfg, seed=int(rng.integers(1 << 30)), Dr=Dr, **modes(rng))
    return name, R, C, rblk, causal, Dr


def fuzz_dkv5(rng):
    """role-split backwardKeyValue streams (buckets 160 / 192 / 256)"""
    import dkv5gen
    import test_dkv5_stream as t
    pool = {n: c for n, c in dkv5gen.VARIANTS.items() if not c.prof}
    name = str(rng.choice(sorted(pool)))
    cfg = pool[name]
    causal = bool(rng.integers(2))
    R, C = shape(rng, 448, 400, causal=causal)
    cblk = int(rng.integers((C + 127) // 128))
    Dr = int(rng.choice([cfg.D, cfg.D, cfg.D - 8, cfg.D - 24]))
    t._check(R, C, cblk=cblk, causal=causal, cfg=cfg, seed=int(rng.integers(1 << 30)), Dr=Dr, **modes(rng))
    return name, R, C, cblk, causal, Dr


FAMILIES = {"p4": fuzz_p4, "p4p": fuzz_p4p, "p5": fuzz_p5, "dq4": fuzz_dq4, "dkv4": fuzz_dkv4, "dq5": fuzz_dq5, "dkv5": fuzz_dkv5}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--family", choices=sorted(FAMILIES), required=True)
    ap.add_argument("--cases", type=int, default=20)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    bad, t0 = 0, time.time()
    for i in range(a.cases):
        rng = np.random.default_rng([a.seed, i])
        try:
            what = FAMILIES[a.family](rng)
            print("ok  ", a.family, i, what, flush=True)
        except AssertionError as e:
            bad += 1
            print("FAIL", a.family, i, "seed", (a.seed, i), "".join(traceback.format_exception_only(type(e), e)).strip()[:400], flush=True)
            traceback.print_exc(limit=3)
    print("%s: %d cases, %d outside tolerance, %.0f s (seed %d)" % (a.family, a.cases, bad, time.time() - t0, a.seed), flush=True)
    sys.exit(1 if bad else 0)
