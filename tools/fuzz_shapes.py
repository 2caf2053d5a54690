#!/usr/bin/env python3
"""developer check: random (R, C, D, causal, precision mode, 16-bit type) problems through all three kernels against the oracle
with the reference's mixed tolerances (and a tighter gradient bound); prints every failure, exits non-zero if there is one.

  python tools/fuzz_shapes.py [cases] [seed]
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

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
seen = {}
for i in range(cases):
    D = int(rng.choice([8, 40, 64, 72, 96, 104, 112, 120, 128, 136, 152, 160, 176, 192, 200, 232, 256]))
    causal = bool(rng.integers(2))
    R = int(rng.integers(1, 700))
    C = int(rng.integers(R if causal else 1, 900))
    low_mid = bool(rng.integers(2))
    in_type = P.BF16 if rng.integers(2) else P.FP16
    net = Network(NetworkDescriptor(R, C, D), seed=1000 + i)
    desc = make_desc(R, C, D, low_in=True, low_mid=low_mid, in_type=in_type)
    run = harness.DeviceRun(desc, net, causal=causal)
    got = run.execute()
    round_inputs(net, desc)
    ref = net.run(causal=causal)
    failures, report = harness.compare(ref, got, TOL_MIXED_SHORT if C <= 20 else TOL_MIXED)
    variants = [k.variant for k in run.kernels.values()]
    for v in variants:
        seen[v] = seen.get(v, 0) + 1
    ok = not failures and all(run.tails_ok.values()) and all(np.isfinite(got[n]).all() for n in ("O", "dQ", "dK", "dV"))
    if not ok:
        bad += 1
        print("FAIL", (R, C, D), "causal" if causal else "dense", "mixed" if low_mid else "fp32mid", in_type.name, failures, run.tails_ok, variants)
print(f"{cases - bad} of {cases} random problems within the reference's mixed tolerances")
for v, n in sorted(seen.items()):
    print(f"  {n:4d} x {v}")
sys.exit(1 if bad else 0)
