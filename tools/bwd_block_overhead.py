#!/usr/bin/env python3
"""developer probe: per-workgroup fixed cost of the backward kernels = launch time against the length of the traversed dimension
(dK/dV walks the ROWS of a 256-key block, dQ the KEYS of a 256-row block), the parallel dimension fixed at 4096."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from metal_flash_attention_amd import (AttentionDescriptor, AttentionKernel, AttentionKernelType as KT, AttentionOperand as Op,
                                       GEMMOperandPrecision as P)
D, H, PAR = int(sys.argv[1]) if len(sys.argv) > 1 else 128, 64, 4096
g = torch.Generator(device="cuda"); g.manual_seed(0)
rnd = lambda *s: torch.randn(s, generator=g, device="cuda").to(torch.bfloat16)
for kind in (KT.backwardKeyValue, KT.backwardQuery):
    prev = None
    for T in (64, 128, 256, 512, 1024, 2048, 4096):
        R, C = (T, PAR) if kind == KT.backwardKeyValue else (PAR, T)
        desc = AttentionDescriptor()
        desc.lowPrecisionInputs = desc.lowPrecisionIntermediates = True
        desc.lowPrecisionInputType = P.BF16
        desc.matrixDimensions = (R, C, D)
        desc.transposeState = (False,) * 4
        ks = {t: AttentionKernel(desc.kernelDescriptor(t)) for t in KT}
        bufs = {Op.Q: rnd(H, R, D), Op.K: rnd(H, C, D), Op.V: rnd(H, C, D), Op.dO: rnd(H, R, D),
                Op.O: torch.empty((H, R, D), device="cuda"), Op.L: torch.empty((H, R), device="cuda", dtype=torch.float16),
                Op.D: torch.empty((H, R), device="cuda", dtype=torch.bfloat16), Op.dQ: torch.empty((H, R, D), device="cuda"),
                Op.dK: torch.empty((H, C, D), device="cuda"), Op.dV: torch.empty((H, C, D), device="cuda")}
        hs = {op: (R if op in (Op.L, Op.D) else (R if op in (Op.Q, Op.O, Op.dO, Op.dQ) else C) * D) for op in bufs}
        stream = torch.cuda.current_stream().cuda_stream
        for t in KT:
            ks[t].dispatch(bufs, row=R, column=C, heads=H, headStrides=hs, stream=stream)
        for _ in range(20):
            ks[kind].dispatch(bufs, row=R, column=C, heads=H, headStrides=hs, stream=stream)
        ms = ks[kind].time(bufs, row=R, column=C, heads=H, headStrides=hs, stream=stream, warmup=5, iterations=20) / 20
        per_cu = (PAR // 256) * H / 256
        us = ms * 1e3 / per_cu
        slope = "" if prev is None else f"  (+{(us - prev[1]) / (T - prev[0]) * 64:6.2f} us per 64 of the traversal)"
        print(f"{ks[kind].variant:34s} traversal {T:5d}: {ms * 1e3:8.1f} us/launch, {us:7.2f} us per workgroup{slope}")
        prev = (T, us)
