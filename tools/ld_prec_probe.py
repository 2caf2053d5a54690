#!/usr/bin/env python3
"""developer probe: dK/dV launch time against the storage types of L and D (role-split kernel, D = 160)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from metal_flash_attention_amd import (AttentionDescriptor, AttentionKernel, AttentionKernelType as T, AttentionOperand as Op,
                                       GEMMOperandPrecision as P)
N, H, D = 4096, 64, int(sys.argv[1]) if len(sys.argv) > 1 else 160
tp = {P.FP32: torch.float32, P.FP16: torch.float16, P.BF16: torch.bfloat16}
g = torch.Generator(device="cuda"); g.manual_seed(0)
base = {op: torch.randn((H, N, D), generator=g, device="cuda").to(torch.bfloat16) for op in (Op.Q, Op.K, Op.V, Op.dO)}
for lp, dp in ((P.FP32, P.FP32), (P.FP16, P.BF16), (P.FP16, P.FP32), (P.FP32, P.BF16)):
    desc = AttentionDescriptor()
    desc.lowPrecisionInputs = True
    desc.lowPrecisionInputType = P.BF16
    desc.matrixDimensions = (N, N, D)
    desc.transposeState = (False,) * 4
    kd = desc.kernelDescriptor(T.backwardKeyValue)
    mp = dict(kd.memoryPrecisions); mp[Op.L] = lp; mp[Op.D] = dp
    kd.memoryPrecisions = mp
    k = AttentionKernel(kd)
    bufs = dict(base)
    bufs[Op.L] = torch.full((H, N), 8.0, device="cuda").to(tp[lp]); bufs[Op.D] = torch.zeros((H, N), device="cuda", dtype=tp[dp])
    for op in (Op.dK, Op.dV):
        bufs[op] = torch.zeros((H, N, D), device="cuda")
    hs = {op: (N if op in (Op.L, Op.D) else N * D) for op in bufs}
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(20):
        k.dispatch(bufs, row=N, column=N, heads=H, headStrides=hs, stream=stream)
    ms = k.time(bufs, row=N, column=N, heads=H, headStrides=hs, stream=stream, warmup=5, iterations=20) / 20
    print(f"L {lp.name} D {dp.name}: {k.variant} {ms:.3f} ms")
