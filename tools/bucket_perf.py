#!/usr/bin/env python3
"""developer probe: forward / dQ / dK-dV launch times per head dimension (N = 4096, 64 heads, bf16)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from metal_flash_attention_amd import (AttentionDescriptor, AttentionKernel, AttentionKernelType as T, AttentionOperand as Op,
                                       GEMMOperandPrecision as P)
N, H = 4096, 64
dims = [int(x) for x in sys.argv[1:]] or [128, 160, 192, 256]
for D in dims:
    desc = AttentionDescriptor()
    desc.lowPrecisionInputs = True
    desc.lowPrecisionIntermediates = os.environ.get("MIXED", "0") == "1"
    desc.lowPrecisionInputType = P.BF16
    desc.matrixDimensions = (N, N, D)
    desc.transposeState = (False,) * 4
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    bufs = {op: torch.randn((H, N, D), generator=g, device="cuda").to(torch.bfloat16) for op in (Op.Q, Op.K, Op.V, Op.dO)}
    mem = desc.memoryPrecisions
    tp = {P.FP32: torch.float32, P.FP16: torch.float16, P.BF16: torch.bfloat16}
    bufs[Op.O] = torch.zeros((H, N, D), device="cuda"); bufs[Op.L] = torch.zeros((H, N), device="cuda", dtype=tp[mem[Op.L]]); bufs[Op.D] = torch.zeros((H, N), device="cuda", dtype=tp[mem[Op.D]])
    for op in (Op.dQ, Op.dK, Op.dV):
        bufs[op] = torch.zeros((H, N, D), device="cuda")
    hs = {op: (N if op in (Op.L, Op.D) else N * D) for op in bufs}
    stream = torch.cuda.current_stream().cuda_stream
    line = [f"D={D:3d}"]
    for t, gemms in ((T.forward, 4), (T.backwardQuery, 6), (T.backwardKeyValue, 8)):
        k = AttentionKernel(desc.kernelDescriptor(t))
        for _ in range(20):
            k.dispatch(bufs, row=N, column=N, heads=H, headStrides=hs, stream=stream)
        ms = k.time(bufs, row=N, column=N, heads=H, headStrides=hs, stream=stream, warmup=5, iterations=20) / 20
        line.append(f"{k.variant}: {ms:.3f} ms {gemms * N * N * D * H / ms / 1e9:.0f} TF")
    print("  ".join(line))
