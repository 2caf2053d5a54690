#!/usr/bin/env python3
"""GEMM operator throughput in the style of Tests/FlashAttentionTests/GEMM/LaplacianTest.swift:45-117
(testPerformance: n-1, n, n+1 for all four transpose states), with larger n and the roofline beside it.
  python tools/bench_gemm.py [--sizes 4096,8192] [--dtypes bf16,f32]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="4096,8192")
    ap.add_argument("--dtypes", default="bf16,f32")
    ap.add_argument("--odd", action="store_true", help="also n-1 and n+1 (misaligned problem sizes)")
    ap.add_argument("--vendor", action="store_true",
                    help="also time torch.matmul (hipBLASLt) on the same operands: a calibration of what a tuned "
                         "library reaches on this box with random data, not part of the product")
    args = ap.parse_args()
    import torch
    from metal_flash_attention_amd import GEMMDescriptor, GEMMKernel, GEMMKernelDescriptor, GEMMOperandPrecision as P
    peak = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}
    sustained = {"bf16": 1594.1, "f16": 1594.1, "f32": None}   # the vendor GEMM on N(0,1) operands on this chip (profiles/r06_vendor_gemm_calibration.txt)
    stream = torch.cuda.current_stream().cuda_stream
    for dt in args.dtypes.split(","):
        prec = {"bf16": P.BF16, "f16": P.FP16, "f32": P.FP32}[dt]
        tdt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[dt]
        for n0 in (int(x) for x in args.sizes.split(",")):
            for n in ((n0 - 1, n0, n0 + 1) if args.odd else (n0,)):
                for tA, tB in ((False, False), (False, True), (True, False), (True, True)):
                    g = torch.Generator(device="cuda").manual_seed(n)
                    a = torch.randn((n, n), generator=g, device="cuda").to(tdt)
                    b = torch.randn((n, n), generator=g, device="cuda").to(tdt)
                    c = torch.empty((n, n), device="cuda", dtype=torch.float32)
                    d = GEMMDescriptor()
                    d.matrixDimensions = (n, n, n)
                    d.memoryPrecisions = (prec, prec, P.FP32)
                    d.transposeState = (tA, tB)
                    k = GEMMKernel(GEMMKernelDescriptor(descriptor=d))
                    iters = 10 if dt != "f32" else 3
                    best = min(k.time(a, b, c, descriptor=d, stream=stream, warmup=1, iterations=iters) / iters for _ in range(3))
                    tf = 2.0 * n ** 3 / best / 1e9
                    ref = (a.float().T if tA else a.float()) @ (b.float().T if tB else b.float())
                    err = (c - ref).abs().max().item() / ref.abs().max().item()
                    extra = f"  {tf / sustained[dt] * 100:5.1f} % of the vendor GEMM on N(0,1)" if sustained[dt] else ""
                    print(f"{dt:4s} n={n:5d} {'A^T' if tA else 'A  '} {'B^T' if tB else 'B  '} {k.variant:32s} {best:8.3f} ms "
                          f"{tf:8.1f} TFLOP/s  {tf / peak[dt] * 100:5.1f} % of spec{extra}  rel err {err:.1e}")
                    if args.vendor:
                        am, bm = (a.T if tA else a), (b.T if tB else b)
                        out = torch.empty((n, n), device="cuda", dtype=tdt)
                        for _ in range(2):
                            torch.matmul(am, bm, out=out)
                        vbest = float("inf")
                        for _ in range(3):
                            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                            e0.record()
                            for _ in range(iters):
                                torch.matmul(am, bm, out=out)
                            e1.record()
                            torch.cuda.synchronize()
                            vbest = min(vbest, e0.elapsed_time(e1) / iters)
                        print(f"{dt:4s} n={n:5d} {'A^T' if tA else 'A  '} {'B^T' if tB else 'B  '} {'torch.matmul (hipBLASLt), ' + dt + ' C':32s} "
                              f"{vbest:8.3f} ms {2.0 * n ** 3 / vbest / 1e9:8.1f} TFLOP/s")


if __name__ == "__main__":
    main()
