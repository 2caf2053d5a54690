#!/usr/bin/env python3
"""Calibration of the power-limited ceiling on THIS box (round 6): the vendor GEMM (hipBLASLt behind torch.matmul,
bf16, n = 8192, A B^T) on N(0,1) and on all-zero operands, timed with HIP events on the launch stream.  Run plain for TF/s; under
`rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_BUSY_CYCLES` (tools/vendor_calib.sh) for its matrix-pipe
busy fraction and effective clock -- the numbers the attention kernel's own are read against.

  python tools/vendor_calib.py [--n 8192] [--iters 50] [--fills normal,zero]
"""
import argparse


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=8192)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--fills", default="normal,zero")
    args = ap.parse_args()
    import torch
    n = args.n
    flops = 2.0 * n * n * n
    for fill in args.fills.split(","):
        g = torch.Generator(device="cuda")
        g.manual_seed(0)
        if fill == "zero":
            a = torch.zeros((n, n), device="cuda", dtype=torch.bfloat16)
            b = torch.zeros((n, n), device="cuda", dtype=torch.bfloat16)
        else:
            a = torch.randn((n, n), generator=g, device="cuda").to(torch.bfloat16)
            b = torch.randn((n, n), generator=g, device="cuda").to(torch.bfloat16)
        c = torch.empty((n, n), device="cuda", dtype=torch.bfloat16)
        for _ in range(10):
            torch.matmul(a, b.t(), out=c)
        torch.cuda.synchronize()
        times = []
        for _ in range(args.rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                torch.matmul(a, b.t(), out=c)
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1) / args.iters)
        times.sort()
        med = times[len(times) // 2]
        print(f"vendor_gemm bf16 n={n} A*B^T fill={fill:6s} med {med:8.4f} ms {flops / med / 1e9:8.1f} TF frac {flops / med / 2.5e12:6.4f} best {flops / times[0] / 2.5e12:6.4f}", flush=True)


if __name__ == "__main__":
    main()
