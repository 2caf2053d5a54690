#!/usr/bin/env python3
"""Device time (HIP events around back-to-back launches inside the C library) of single-head launches of the three kernels
(the reference's own benchmark shape, SquareAttentionTest.swift:159-165), as launched without a workspace vs. traversal-parallel
through one.  Developer tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metal_flash_attention_amd import (AttentionDescriptor, AttentionKernel, AttentionKernelType as KT,
                                       AttentionOperand as Op, GEMMOperandPrecision as P)
FLOPS = {KT.forward: 4.0, KT.backwardQuery: 6.0, KT.backwardKeyValue: 8.0}
mixed = "--mixed" in sys.argv
for N, D in ((4096, 64), (4096, 128), (8192, 128), (16384, 128), (16384, 64), (8192, 256), (16384, 256)):
    desc = AttentionDescriptor(); desc.lowPrecisionInputs = True; desc.lowPrecisionInputType = P.BF16
    desc.lowPrecisionIntermediates = mixed
    desc.matrixDimensions = (N, N, D); desc.transposeState = (False,) * 4
    prec = desc.memoryPrecisions
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    b = {op: (torch.randn((N, D), generator=g, device="cuda") * (0.1 if op == Op.dO else 1)).to(torch.bfloat16) for op in (Op.Q, Op.K, Op.V, Op.dO)}
    def out(op, shape):
        p = prec[op]
        return torch.empty(shape, device="cuda", dtype=torch.float32 if p == P.FP32 else torch.float16 if p == P.FP16 else torch.bfloat16)
    for op in (Op.O, Op.dQ, Op.dK, Op.dV):
        b[op] = out(op, (N, D))
    b[Op.L], b[Op.D] = out(Op.L, (N,)), out(Op.D, (N,))
    s = torch.cuda.current_stream().cuda_stream
    line = f"N={N:6d} D={D:4d}"
    for t in (KT.forward, KT.backwardQuery, KT.backwardKeyValue):
        k = AttentionKernel(desc.kernelDescriptor(t))
        need = k.workspaceSize(row=N, column=N)
        ws = torch.empty(max(need, 16), dtype=torch.uint8, device="cuda")
        k.dispatch(b, row=N, column=N, stream=s)   # (real L and D for the backward kernels)
        t0 = min(k.time(b, row=N, column=N, stream=s, warmup=3, iterations=20) for _ in range(3)) / 20
        t1 = min(k.time(b, row=N, column=N, stream=s, warmup=3, iterations=20, workspace=ws) for _ in range(3)) / 20
        fl = FLOPS[t] * N * N * D
        form = k.launchForm(b, row=N, column=N, workspace=ws)
        line += f"\n    {t.name:17s} alone {t0*1e3:8.1f} us ({fl/t0/1e9:7.1f} TF)   with workspace {t1*1e3:8.1f} us ({fl/t1/1e9:7.1f} TF)  {need/2**20:6.1f} MiB  {form}"
    print(line)
