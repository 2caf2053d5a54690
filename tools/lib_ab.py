#!/usr/bin/env python3
"""developer A/B of two builds of the library on bench.py workloads: bench.py is run once per (library, workload, fill) in
interleaved rounds (a process per run: the library is chosen at import), the medians of `roofline.frac` are compared.

  python tools/lib_ab.py --a metal_flash_attention_amd/libmfa_hip_prev.so --b metal_flash_attention_amd/libmfa_hip.so \
      --workloads dq_bf16_d256,dkv_bf16_d256 [--fills normal,zero] [--rounds 2]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(lib, workload, fill, steps):
    env = dict(os.environ, MFA_LIBRARY=os.path.join(ROOT, lib))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--fill", fill, "--steps", str(steps),
                          "--warmup", "5", "--no-cpu-baseline"], env=env, capture_output=True, text=True)
    for line in out.stdout.splitlines():
        if line.startswith("{"):
            j = json.loads(line)
            return j["roofline"]["frac"], j["ms_per_step"]
    raise RuntimeError(out.stdout[-2000:] + out.stderr[-2000:])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--a", default="metal_flash_attention_amd/libmfa_hip_prev.so")
    ap.add_argument("--b", default="metal_flash_attention_amd/libmfa_hip.so")
    ap.add_argument("--workloads", default="dq_bf16_d256,dkv_bf16_d256,dq_bf16_d128,dkv_bf16_d128,fwd_bf16_d256")
    ap.add_argument("--fills", default="normal,zero")
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    print("A = %s\nB = %s" % (args.a, args.b))
    for w in args.workloads.split(","):
        for fill in args.fills.split(","):
            res = {"A": [], "B": []}
            for _ in range(args.rounds):
                for tag, lib in (("A", args.a), ("B", args.b)):
                    res[tag].append(run(lib, w, fill, args.steps))
            med = {t: sorted(r)[len(r) // 2] for t, r in res.items()}
            print("%-22s %-6s  A frac %.4f (%.3f ms)   B frac %.4f (%.3f ms)   B/A %.3f   all A %s  B %s" % (
                w, fill, med["A"][0], med["A"][1], med["B"][0], med["B"][1], med["B"][0] / med["A"][0],
                " ".join("%.4f" % f for f, _ in res["A"]), " ".join("%.4f" % f for f, _ in res["B"])))
            sys.stdout.flush()


if __name__ == "__main__":
    main()
