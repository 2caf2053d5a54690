#!/usr/bin/env python3
"""Attention at 256 < D <= 384 with 16-bit inputs (attn_fwd16_wide, attn_dq16 with 32-key tiles, attn_dkv16_wide: round 6) against the
fp32-arithmetic kernels that served these head dimensions until round 5 (FP32 inputs of the same shape: the same code path those launches
took), N(0,1) operands.

  python tools/time_wide.py [--N 4096 --heads 32] [--backward]
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
    ap.add_argument("--backward", action="store_true")
    args = ap.parse_args()
    import torch
    from metal_flash_attention_amd import (AttentionDescriptor, AttentionKernel, AttentionKernelType,
                                           AttentionOperand as Op, GEMMOperandPrecision as P)
    N, H = args.N, args.heads
    stream = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device="cuda")
    g.manual_seed(0)
    T = AttentionKernelType
    FLOPS = {T.forward: 4.0, T.backwardQuery: 6.0, T.backwardKeyValue: 8.0}
    for D in (320, 384):
        for causal in (False, True):
            for low in (True, False):
                desc = AttentionDescriptor()
                desc.lowPrecisionInputs = low
                desc.lowPrecisionInputType = P.BF16
                desc.matrixDimensions = (N, N, D)
                desc.transposeState = (False,) * 4
                dt = torch.bfloat16 if low else torch.float32
                prec = desc.memoryPrecisions
                tdt = {P.FP32: torch.float32, P.FP16: torch.float16, P.BF16: torch.bfloat16}
                bufs = {op: (torch.randn((H, N, D), generator=g, device="cuda") * (0.1 if op == Op.dO else 1)).to(tdt[prec[op]])
                        for op in ((Op.Q, Op.K, Op.V, Op.dO) if args.backward else (Op.Q, Op.K, Op.V))}
                for op in (Op.O, Op.dQ, Op.dK, Op.dV) if args.backward else (Op.O,):
                    bufs[op] = torch.zeros((H, N, D), device="cuda", dtype=tdt[prec[op]])
                bufs[Op.L] = torch.zeros((H, N), device="cuda", dtype=tdt[prec[Op.L]])
                bufs[Op.D] = torch.zeros((H, N), device="cuda", dtype=tdt[prec[Op.D]])
                hs = {op: N * D for op in (Op.Q, Op.K, Op.V, Op.O, Op.dO, Op.dQ, Op.dK, Op.dV)}
                hs[Op.L] = hs[Op.D] = N
                kw = dict(row=N, column=N, heads=H, headStrides=hs, stream=stream, causal=causal)
                for t in (T.forward, T.backwardQuery, T.backwardKeyValue) if args.backward else (T.forward,):
                    if t != T.forward and not low and N * H > 4096 * 8:
                        continue   # (the fp32-arithmetic backward kernels at this size take seconds: time them with --heads 8)
                    k = AttentionKernel(desc.kernelDescriptor(t))
                    k.dispatch(bufs, **kw)    # (real L and D for the backward kernels)
                    ms = k.time(bufs, warmup=2, iterations=5, **kw) / 5
                    work = (N + 1) / (2.0 * N) if causal else 1.0
                    tf = FLOPS[t] * N * N * D * H * work / (ms * 1e-3) / 1e12
                    print(f"D={D} causal={int(causal)} {'bf16' if low else 'fp32'} inputs  {t.name:17s} {k.variant:44s} {ms:8.3f} ms  {tf:7.1f} TF  {tf / 2500:6.3f} of the bf16 roof", flush=True)
                del bufs


if __name__ == "__main__":
    main()
