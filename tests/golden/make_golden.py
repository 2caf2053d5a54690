#!/usr/bin/env python3
"""Regenerates tests/golden/network_golden.npz from the C oracle (oracle/network.c).

The reference holds NO golden vectors for this path (its Network is seeded from the system RNG,
Network.swift:115-129) and cannot be run here (no Swift toolchain), so these
fixtures pin OUR restatement: seeded inputs (seed, R, C, D) -> O, L, D, dV, dK, dQ in fp32.  tests/test_oracle.py checks that the oracle still reproduces them bit-for-bit
and that the independent numpy fp64 restatement agrees.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import Network, NetworkDescriptor  # noqa: E402

CASES = [  # (seed, R, C, D): small members of SquareAttentionTest.swift:6-25 + rectangular ones
    (0, 10, 10, 3), (1, 8, 8, 2), (2, 23, 23, 2), (3, 4, 4, 1), (4, 32, 32, 64), (5, 64, 64, 40),
    (6, 7, 33, 5), (7, 40, 9, 17), (8, 1, 128, 16), (9, 48, 31, 32),
    (10, 128, 128, 64),   # BASELINE config 1: forward single-head N=128 D=64 fp32 (CPU plumbing)
]


def main():
    out = {"cases": np.array(CASES, np.int64)}
    for seed, R, C, D in CASES:
        net = Network(NetworkDescriptor(R, C, D), seed=seed, threads=1)
        res = net.run(backward=True)
        key = f"s{seed}"
        out[f"{key}_Q"] = net.Q
        out[f"{key}_K"] = net.K
        out[f"{key}_V"] = net.V
        out[f"{key}_dO"] = net.dO
        for name, val in res.items():
            out[f"{key}_{name}"] = val
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "network_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
