#!/usr/bin/env python3
"""Block-sparse speed-up of the three bf16 kernels: dense vs a banded mask (causal sliding window of 1024 keys)
and a random mask of 50 % density.  N = 8192, D = 128, 32 heads.  Developer tool."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from metal_flash_attention_amd import (AttentionDescriptor, AttentionKernel, AttentionKernelType as KT, AttentionOperand as Op,
                                       GEMMOperandPrecision as P)

N, D, H = 8192, 128, 32
desc = AttentionDescriptor()
desc.lowPrecisionInputs = True
desc.lowPrecisionInputType = P.BF16
desc.matrixDimensions = (N, N, D)
desc.transposeState = (False,) * 4
g = torch.Generator(device="cuda").manual_seed(0)
mk = lambda *s: torch.randn(*s, generator=g, device="cuda")
bufs = {Op.Q: mk(H, N, D).bfloat16(), Op.K: mk(H, N, D).bfloat16(), Op.V: mk(H, N, D).bfloat16(), Op.dO: mk(H, N, D).bfloat16(),
        Op.O: torch.empty(H, N, D, device="cuda"), Op.L: torch.empty(H, N, device="cuda"), Op.D: torch.empty(H, N, device="cuda"),
        Op.dQ: torch.empty(H, N, D, device="cuda"), Op.dK: torch.empty(H, N, D, device="cuda"), Op.dV: torch.empty(H, N, D, device="cuda")}
hs = {op: (N if op in (Op.L, Op.D) else N * D) for op in bufs}
rb, cb = N // 256, N // 128
words = (cb + 31) // 32


def pack(bits):
    out = np.zeros((rb, words), np.uint32)
    for i in range(rb):
        for j in np.nonzero(bits[i])[0]:
            out[i, j // 32] |= np.uint32(1) << np.uint32(j % 32)
    return torch.from_numpy(out.view(np.int32)).cuda()


rows = np.arange(rb)[:, None] * 256
cols = np.arange(cb)[None, :] * 128
band = (cols <= rows + 255) & (cols + 127 >= rows - 1024)          # causal window of ~1024 keys, block granularity
rng = np.random.default_rng(0)
rand = rng.random((rb, cb)) < 0.5
stream = torch.cuda.current_stream().cuda_stream
kernels = {t: AttentionKernel(desc.kernelDescriptor(t)) for t in KT}
for t in KT:   # L and D must exist before the backward kernels are timed
    kernels[t].dispatch(bufs, row=N, column=N, heads=H, headStrides=hs, stream=stream)
for name, bits, causal in (("dense", None, False), ("dense causal", None, True), ("band 1024 (causal)", band, True), ("random 50 %", rand, False)):
    m = pack(bits) if bits is not None else None
    density = 1.0 if bits is None else float(bits.mean())
    line = f"{name:22s} density {density:5.2f} "
    for t in KT:
        kw = dict(row=N, column=N, heads=H, headStrides=hs, stream=stream, causal=causal)
        if m is not None:
            kw.update(blockMask=m, blockMaskWords=words)
        arr, params, keep = kernels[t]._marshal(bufs, N, N, H, 1, None, hs, None, None, causal, None, None, m, words if m is not None else 0, (0, 0))
        import ctypes
        from metal_flash_attention_amd._abi import lib, check
        ms = ctypes.c_float()
        check(lib().mfa_attention_kernel_time(kernels[t]._handle, ctypes.byref(arr), ctypes.byref(params), ctypes.c_void_p(stream), 2, 5, ctypes.byref(ms)))
        line += f"| {t.name:16s} {ms.value / 5:7.3f} ms "
    print(line)
