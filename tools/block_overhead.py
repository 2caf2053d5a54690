#!/usr/bin/env python3
"""developer probe: per-workgroup fixed cost of the forward kernel = launch time with C keys, for growing C"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from metal_flash_attention_amd import (AttentionDescriptor, AttentionKernel, AttentionKernelType, AttentionOperand as Op,
                                       GEMMOperandPrecision as P)
R, D, H = 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 128, 256
g = torch.Generator(device="cuda"); g.manual_seed(0)
q = torch.randn((H, R, D), generator=g, device="cuda").to(torch.bfloat16)
for C in (64, 128, 256, 512, 1024, 2048, 4096):
    desc = AttentionDescriptor()
    desc.lowPrecisionInputs = desc.lowPrecisionIntermediates = True
    desc.lowPrecisionInputType = P.BF16
    desc.matrixDimensions = (R, C, D)
    desc.transposeState = (False,) * 4
    k = AttentionKernel(desc.kernelDescriptor(AttentionKernelType.forward))
    kv = torch.randn((H, C, D), generator=g, device="cuda").to(torch.bfloat16)
    bufs = {Op.Q: q, Op.K: kv, Op.V: kv.clone(), Op.O: torch.empty((H, R, D), device="cuda"), Op.L: torch.empty((H, R), device="cuda", dtype=torch.float16)}
    hs = {Op.Q: R * D, Op.K: C * D, Op.V: C * D, Op.O: R * D, Op.L: R}
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(30):
        k.dispatch(bufs, row=R, column=C, heads=H, headStrides=hs, stream=stream)
    ms = k.time(bufs, row=R, column=C, heads=H, headStrides=hs, stream=stream, warmup=5, iterations=20) / 20
    blocks_per_cu = (R // 256) * H / 256
    print(f"{k.variant} C={C:5d} tiles={C // 64:3d}: {ms * 1e3:8.1f} us/launch, {ms * 1e3 / blocks_per_cu:7.2f} us per workgroup")
