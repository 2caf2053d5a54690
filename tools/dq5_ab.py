#!/usr/bin/env python3
"""developer A/B (libmfa_hip_dev.so): attn_dq16_p5's timing-only ablation streams (tools/dq5gen.py DEV_ABLATIONS) against the
product stream -- D = 256, bf16, mixed mode, interleaved rounds in ONE process, on all-zero operands (full clock: cycles) and
optionally on N(0,1).  The ablated streams compute garbage; only their launch time means anything.

  python tools/dq5_ab.py [--streams ABL_ROWRD,...] [--fills zero,normal]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("MFA_LIBRARY", os.path.join(ROOT, "metal_flash_attention_amd", "libmfa_hip_dev.so"))


def main():
    import dq5gen
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--heads", type=int, default=64)
    ap.add_argument("--streams", default=",".join(dq5gen.DEV_ABLATIONS))
    ap.add_argument("--fills", default="zero")
    ap.add_argument("--rounds", type=int, default=3)
    args = ap.parse_args()
    import torch
    from metal_flash_attention_amd import (AttentionDescriptor, AttentionKernel, AttentionKernelType as T,
                                           AttentionOperand as Op, GEMMOperandPrecision as P)
    N, D, H = args.N, 256, args.heads
    desc = AttentionDescriptor()
    desc.lowPrecisionInputs = desc.lowPrecisionIntermediates = True
    desc.lowPrecisionInputType = P.BF16
    desc.matrixDimensions = (N, N, D)
    desc.transposeState = (False,) * 4
    mem = desc.memoryPrecisions
    tp = {P.FP32: torch.float32, P.FP16: torch.float16, P.BF16: torch.bfloat16}
    stream = torch.cuda.current_stream().cuda_stream
    names = ["product"] + ["D256_BF16_FOLD_" + s for s in args.streams.split(",") if s]
    flops = 6.0 * N * N * D * H
    for fill in args.fills.split(","):
        g = torch.Generator(device="cuda")
        g.manual_seed(0)
        if fill == "zero":
            bufs = {op: torch.zeros((H, N, D), device="cuda", dtype=torch.bfloat16) for op in (Op.Q, Op.K, Op.V, Op.dO)}
        else:
            bufs = {op: torch.randn((H, N, D), generator=g, device="cuda").to(torch.bfloat16) for op in (Op.Q, Op.K, Op.V, Op.dO)}
        bufs[Op.O] = torch.zeros((H, N, D), device="cuda", dtype=tp[mem[Op.O]])
        bufs[Op.L] = torch.zeros((H, N), device="cuda", dtype=tp[mem[Op.L]])
        bufs[Op.D] = torch.zeros((H, N), device="cuda", dtype=tp[mem[Op.D]])
        bufs[Op.dQ] = torch.zeros((H, N, D), device="cuda")
        hs = {op: (N if op in (Op.L, Op.D) else N * D) for op in bufs}
        os.environ.pop("MFA_DQ5_DEV_STREAM", None)
        AttentionKernel(desc.kernelDescriptor(T.forward)).dispatch(bufs, row=N, column=N, heads=H, headStrides=hs, stream=stream)   # a real L
        kernels = {}
        for name in names:
            os.environ.pop("MFA_DQ5_DEV_STREAM", None)
            if name != "product":
                os.environ["MFA_DQ5_DEV_STREAM"] = name
            kernels[name] = AttentionKernel(desc.kernelDescriptor(T.backwardQuery))
        os.environ.pop("MFA_DQ5_DEV_STREAM", None)
        times = {n: [] for n in names}
        for _ in range(args.rounds):
            for name in names:
                times[name].append(kernels[name].time(bufs, row=N, column=N, heads=H, headStrides=hs, stream=stream, warmup=2, iterations=10) / 10)
        print("## fill = %s   (dQ, N = %d, D = %d, %d heads; 2.5 PF roof)" % (fill, N, D, H))
        base = sorted(times["product"])[len(times["product"]) // 2]
        for name in names:
            t = sorted(times[name])
            med = t[len(t) // 2]
            print(f"{kernels[name].variant:44s} med {med:8.4f} ms  frac {flops / med / 1e9 / 2.5e3:6.4f}  vs product {med / base:6.3f}")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
