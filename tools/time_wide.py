#!/usr/bin/env python3
"""Forward attention at 256 < D <= 384 with 16-bit inputs (attn_fwd16_wide, round 6) against the fp32-arithmetic kernel that served these
head dimensions until round 5 (FP32 inputs of the same shape: the same code path those launches took), N(0,1) operands.

  python tools/time_wide.py [--N 4096 --heads 32]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--heads", type=int, default=32)
    args = ap.parse_args()
    import torch
    from metal_flash_attention_amd import (AttentionDescriptor, AttentionKernel, AttentionKernelType,
                                           AttentionOperand as Op, GEMMOperandPrecision as P)
    N, H = args.N, args.heads
    stream = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda")
    g.manual_seed(0)
    for D in (320, 384):
        for causal in (False, True):
            for low in (True, False):
                desc = AttentionDescriptor()
                desc.lowPrecisionInputs = low
                desc.lowPrecisionInputType = P.BF16
                desc.matrixDimensions = (N, N, D)
                desc.transposeState = (False,) * 4
                k = AttentionKernel(desc.kernelDescriptor(AttentionKernelType.forward))
                dt = torch.bfloat16 if low else torch.float32
                bufs = {op: torch.randn((H, N, D), generator=g, device="cuda").to(dt) for op in (Op.Q, Op.K, Op.V)}
                bufs[Op.O] = torch.zeros((H, N, D), device="cuda")
                bufs[Op.L] = torch.zeros((H, N), device="cuda")
                hs = {Op.Q: N * D, Op.K: N * D, Op.V: N * D, Op.O: N * D, Op.L: N}
                kw = dict(row=N, column=N, heads=H, headStrides=hs, stream=stream, causal=causal)
                ms = k.time(bufs, warmup=3, iterations=10, **kw) / 10
                work = (N + 1) / (2.0 * N) if causal else 1.0
                tf = 4.0 * N * N * D * H * work / (ms * 1e-3) / 1e12
                print(f"D={D} causal={int(causal)} {'bf16' if low else 'fp32'} inputs  {k.variant:44s} {ms:8.3f} ms  {tf:7.1f} TF  {tf / 2500:6.3f} of the bf16 roof", flush=True)
                del bufs


if __name__ == "__main__":
    main()
