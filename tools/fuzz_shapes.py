#!/usr/bin/env python3
"""developer check: random (R, C, D, causal, precision mode, 16-bit type) problems through all three kernels against the oracle
with the reference's mixed tolerances (and a tighter gradient bound); prints every failure, exits non-zero if there is one.

  python tools/fuzz_shapes.py [cases] [seed] [--transposed [--backward] [--workspace]] [--fp32]

--fp32: FP32 descriptors (no low-precision flag) with D a multiple of 4 up to 128 and two odd ones: the FP32 production kernels
(csrc/attn_f32.h) and the general kernels, the reference's FP32 tolerance (2e-5; 5e-5 beyond its 777-long sequences).

--transposed: forward only, a random non-empty pattern of transposed (Q, K, V, O) per problem (transposeState), sequence lengths
that are multiples of 64 / of 8 / odd in equal parts -- the hand-placed stream on K^T + V^T, the 8 x 32 kernel's in-place code
objects with aligned rows, and their gather path.
--backward (with --transposed): all three kernels on the transposed operands (the gradients follow their operands), without a
workspace unless --workspace -- the general kernels in the product library, the in-place backward kernels where the developer
library has them (MFA_LIBRARY=.../libmfa_hip_dev.so), the re-layout path with --workspace.
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import harness
from test_attention_gpu import make_desc, round_inputs, TOL_MIXED, TOL_MIXED_SHORT
from metal_flash_attention_amd import GEMMOperandPrecision as P
from oracle import Network, NetworkDescriptor

fp32 = "--fp32" in sys.argv
transposed = "--transposed" in sys.argv
backward = not transposed or "--backward" in sys.argv
with_workspace = "--workspace" in sys.argv
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
cases = int(argv[0]) if len(argv) > 0 else 120
rng = np.random.default_rng(int(argv[1]) if len(argv) > 1 else 0)
bad = 0
seen = {}
for i in range(cases):
    D = int(rng.choice([8, 40, 64, 72, 96, 104, 112, 120, 128, 136, 152, 160, 176, 192, 200, 232, 256, 272, 320, 344, 384]))
    causal = bool(rng.integers(2))
    R = int(rng.integers(1, 700))
    C = int(rng.integers(R if causal else 1, 900))
    if with_workspace and not transposed and i % 2:   # long traversals: several pieces per row / key block (round 6: every bucket cuts them itself)
        C = int(rng.integers(max(R, 1000), 4000))
        R = C if i % 4 == 1 else R
    low_mid = bool(rng.integers(2))
    in_type = P.BF16 if rng.integers(2) else P.FP16
    if fp32:
        D = int(rng.choice([4, 8, 20, 36, 48, 60, 64, 68, 72, 84, 96, 100, 112, 124, 128, 30, 66]))
        low_mid = False
    tr = (False,) * 4
    if transposed:
        tr = tuple(bool(b) for b in rng.integers(2, size=4))
        if not any(tr):
            tr = (True, True, True, True)
        g = (64, 8, 1)[i % 3]
        R, C = max(g, R // g * g), max(g, C // g * g)
        if causal and C < R:
            C = R
    net = Network(NetworkDescriptor(R, C, D), seed=1000 + i)
    desc = make_desc(R, C, D, low_in=not fp32, low_mid=low_mid, in_type=in_type, tr=tr)
    if transposed:
        desc.lowPrecisionOutputs = bool(rng.integers(2))
    run = harness.DeviceRun(desc, net, causal=causal, run_backward=backward)
    got = run.execute(with_workspace=with_workspace)
    if not fp32:
        round_inputs(net, desc)
    ref = net.run(causal=causal, backward=backward)
    tol = TOL_MIXED_SHORT if C <= 20 else TOL_MIXED
    if fp32:
        tol = {k: (2e-5 if max(R, C) <= 777 else 5e-5) for k in TOL_MIXED}
    if not fp32 and low_mid and backward:
        # D is STORED in BF16 in the reference's mixed mode (AttentionDescriptor+Precisions.swift:82-83), truncated like its software bfloat: one
        # unit in the last place is up to 2^-7 |D| -- above the reference's absolute 1e-1 once |D| > 13 (a causal row that sees ONE key at a wide
        # head: |dO . V| ~ sqrt(D)); the bound follows the storage format there
        tol = dict(tol, D=max(tol["D"], 2.0 ** -7 * float(np.abs(ref["D"]).max())))
    if not backward:
        tol = {k: v for k, v in tol.items() if k in ("O", "L")}
    failures, report = harness.compare(ref, got, tol)
    variants = [k.launchForm(run.buffers, row=R, column=C, causal=causal).split(" (")[0] if (transposed or fp32) else k.variant for k in run.kernels.values()]
    for v in variants:
        seen[v] = seen.get(v, 0) + 1
    outs = ("O", "dQ", "dK", "dV") if backward else ("O",)
    tails = {k: v for k, v in run.tails_ok.items() if backward or k in ("O", "L")}
    ok = not failures and all(tails.values()) and all(np.isfinite(got[n]).all() for n in outs)
    if not ok:
        bad += 1
        print("FAIL", (R, C, D), "causal" if causal else "dense", "mixed" if low_mid else "fp32mid", in_type.name, tr, failures, tails, variants)
print(f"{cases - bad} of {cases} random problems within the reference's {'FP32' if fp32 else 'mixed'} tolerances")
for v, n in sorted(seen.items()):
    print(f"  {n:4d} x {v}")
sys.exit(1 if bad else 0)
