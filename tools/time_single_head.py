#!/usr/bin/env python3
"""Device time (HIP events around back-to-back launches inside the C library) of single-head forward
launches, row-parallel vs column-parallel (workspace).  Developer tool."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metal_flash_attention_amd import (AttentionDescriptor, AttentionKernel, AttentionKernelType,
                                       AttentionOperand as Op, GEMMOperandPrecision as P)
for N, D in ((4096, 64), (4096, 128), (8192, 128), (16384, 128), (16384, 64)):
    desc = AttentionDescriptor(); desc.lowPrecisionInputs = True; desc.lowPrecisionInputType = P.BF16
    desc.matrixDimensions = (N, N, D); desc.transposeState = (False,) * 4
    k = AttentionKernel(desc.kernelDescriptor(AttentionKernelType.forward))
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    b = {op: torch.randn((N, D), generator=g, device="cuda").to(torch.bfloat16) for op in (Op.Q, Op.K, Op.V)}
    b[Op.O] = torch.empty((N, D), device="cuda"); b[Op.L] = torch.empty(N, device="cuda")
    need = k.workspaceSize(row=N, column=N)
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    t0 = min(k.time(b, row=N, column=N, stream=s, warmup=3, iterations=20) for _ in range(3)) / 20
    t1 = min(k.time(b, row=N, column=N, stream=s, warmup=3, iterations=20, workspace=ws) for _ in range(3)) / 20
    fl = 4.0 * N * N * D
    print(f"N={N:6d} D={D:4d}  row-parallel {t0*1e3:8.1f} us ({fl/t0/1e9:7.1f} TF)   column-parallel {t1*1e3:8.1f} us ({fl/t1/1e9:7.1f} TF)  workspace {need/2**20:.1f} MiB")
