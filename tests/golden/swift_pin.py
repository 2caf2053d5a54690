#!/usr/bin/env python3
"""Closes "parity unpinned" on a machine that has a Swift toolchain (this image has none): feeds the seeded inputs of
tests/golden/network_golden.npz to the REFERENCE's own `Network` (Tests/FlashAttentionTests/Utilities/Network.swift:70-402) and
compares what it returns with the committed oracle outputs.

  python tests/golden/swift_pin.py export <dir>     writes <dir>/network_golden_inputs.bin
  (on the Swift machine) add the XCTest of INTEGRATION.md section 8 to the reference's test target, run it with
      MFA_GOLDEN_DIR=<dir> swift test --filter OraclePinTest      -> <dir>/network_golden_swift.bin
  python tests/golden/swift_pin.py check <dir>      compares (FP32 tolerances of the reference's own tests, 2e-5)

File format (little endian): int32 ncases, then per case int32 seed, R, C, D followed by float32 arrays -- inputs file: Q[R*D], K[C*D],
V[C*D], dO[R*D]; outputs file: O[R*D], L[R], D[R], dV[C*D], dK[C*D], dQ[R*D].  L is Network.createLTerm (natural log), D is
Network.createDTerm (unscaled), exactly what the npz stores.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def export(out_dir):
    g = np.load(os.path.join(HERE, "network_golden.npz"))
    with open(os.path.join(out_dir, "network_golden_inputs.bin"), "wb") as f:
        cases = g["cases"]
        f.write(np.int32(len(cases)).tobytes())
        for seed, R, C, D in cases:
            f.write(np.array([seed, R, C, D], np.int32).tobytes())
            for name in ("Q", "K", "V", "dO"):
                f.write(np.ascontiguousarray(g[f"s{seed}_{name}"], np.float32).tobytes())
    print("wrote", os.path.join(out_dir, "network_golden_inputs.bin"))


def check(out_dir, tol=2e-5):
    g = np.load(os.path.join(HERE, "network_golden.npz"))
    raw = open(os.path.join(out_dir, "network_golden_swift.bin"), "rb").read()
    pos = 4
    n = int(np.frombuffer(raw[:4], np.int32)[0])
    worst = 0.0
    for _ in range(n):
        seed, R, C, D = (int(x) for x in np.frombuffer(raw[pos:pos + 16], np.int32))
        pos += 16
        for name, count in (("O", R * D), ("L", R), ("D", R), ("dV", C * D), ("dK", C * D), ("dQ", R * D)):
            got = np.frombuffer(raw[pos:pos + 4 * count], np.float32)
            pos += 4 * count
            ref = np.asarray(g[f"s{seed}_{name}"], np.float32).reshape(-1)
            err = float(np.abs(got - ref).max())
            worst = max(worst, err)
            assert err <= tol * max(1.0, float(np.abs(ref).max())), (seed, name, err)
    print(f"the reference's Network reproduces the committed oracle outputs: {n} cases, worst |diff| {worst:.2e}")


if __name__ == "__main__":
    {"export": export, "check": check}[sys.argv[1]](sys.argv[2])
