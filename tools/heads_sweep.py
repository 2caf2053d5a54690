import os, sys
sys.path.insert(0, "/root/repo")
import torch
from metal_flash_attention_amd import (AttentionDescriptor, AttentionKernel, AttentionKernelType as T, AttentionOperand as Op, GEMMOperandPrecision as P)
N = 4096
for D in (64, 128):
    desc = AttentionDescriptor(); desc.lowPrecisionInputs = True; desc.lowPrecisionIntermediates = True
    desc.lowPrecisionInputType = P.BF16; desc.matrixDimensions = (N, N, D); desc.transposeState = (False,)*4
    k = AttentionKernel(desc.kernelDescriptor(T.forward))
    stream = torch.cuda.current_stream().cuda_stream
    for H in (16, 32, 64, 128, 256, 512):
        g = torch.Generator(device="cuda"); g.manual_seed(0)
        bufs = {op: torch.randn((H, N, D), generator=g, device="cuda").to(torch.bfloat16) for op in (Op.Q, Op.K, Op.V)}
        bufs[Op.O] = torch.zeros((H, N, D), device="cuda"); bufs[Op.L] = torch.zeros((H, N), device="cuda", dtype=torch.float16)
        hs = {Op.Q: N*D, Op.K: N*D, Op.V: N*D, Op.O: N*D, Op.L: N}
        kw = dict(row=N, column=N, heads=H, headStrides=hs)
        for _ in range(20): k.dispatch(bufs, stream=stream, **kw)
        ms = min(k.time(bufs, stream=stream, warmup=3, iterations=20, **kw) / 20 for _ in range(3))
        print(f"D={D} heads={H:4d} blocks/CU={H*16/256:5.1f} {ms*1000:8.1f} us  per block-round {ms*1000/(H*16/256):7.2f} us  frac {4*N*N*D*H/ms/1e9/2500:.3f}  {k.launchForm(bufs, **kw)[:40]}")
