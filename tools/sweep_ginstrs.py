#!/usr/bin/env python3
"""FWD / dQ / dK-dV GINSTR/s in the format of the reference's tables (README.md:108-175,
Documentation/FlashAttention Variants.xlsx "Generalization"): per kernel, per head dimension, at N = 8192 and
16384; single head (the reference's benchmark shape; all three kernels get a workspace and split their traversal)
and 32 heads.
  GINSTR = (2D+5) N^2 forward, (3D+5) N^2 backward-dQ, (4D+5) N^2 backward-dK/dV  (README.md:108-124)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from metal_flash_attention_amd import (AttentionDescriptor, AttentionKernel, AttentionKernelType as KT, AttentionOperand as Op,
                                       GEMMOperandPrecision as P)
from metal_flash_attention_amd._abi import check, lib

OPS = {KT.forward: lambda D: 2 * D + 5, KT.backwardQuery: lambda D: 3 * D + 5, KT.backwardKeyValue: lambda D: 4 * D + 5}
stream = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda").manual_seed(0)
print(f"{'dtype':5s} {'N':>6s} {'D':>4s} {'heads':>5s} | {'FWD':>9s} {'dQ':>9s} {'dK/dV':>9s}  GINSTR/s | kernels")
import argparse
_ap = argparse.ArgumentParser(); _ap.add_argument("--dtypes", default="bf16,f32"); _args = _ap.parse_args()
for dtype in _args.dtypes.split(","):
    for N in (8192, 16384):
        for D in (64, 128, 256):
            for H in (1, 32):
                if dtype == "f32" and (H == 32 and N == 16384):
                    continue   # 30+ s of fp32 work: skipped in the sweep
                desc = AttentionDescriptor()
                desc.lowPrecisionInputs = dtype == "bf16"
                desc.lowPrecisionInputType = P.BF16
                desc.matrixDimensions = (N, N, D)
                desc.transposeState = (False,) * 4
                tdt = torch.bfloat16 if dtype == "bf16" else torch.float32
                mk = lambda *s: torch.randn(*s, generator=g, device="cuda").to(tdt)
                bufs = {Op.Q: mk(H, N, D), Op.K: mk(H, N, D), Op.V: mk(H, N, D), Op.dO: mk(H, N, D),
                        Op.O: torch.empty(H, N, D, device="cuda"), Op.L: torch.empty(H, N, device="cuda"),
                        Op.D: torch.empty(H, N, device="cuda"), Op.dQ: torch.empty(H, N, D, device="cuda"),
                        Op.dK: torch.empty(H, N, D, device="cuda"), Op.dV: torch.empty(H, N, D, device="cuda")}
                hs = {op: (N if op in (Op.L, Op.D) else N * D) for op in bufs}
                out, names = [], []
                for t in KT:
                    k = AttentionKernel(desc.kernelDescriptor(t))
                    need = k.workspaceSize(row=N, column=N, heads=H)   # traversal-parallel launch when the grid is small
                    ws = torch.empty(need, dtype=torch.uint8, device="cuda") if need else None
                    k.dispatch(bufs, row=N, column=N, heads=H, headStrides=hs, stream=stream, workspace=ws)   # L, D exist before the backward kernels
                    iters = 3 if dtype == "f32" else 5
                    arr, params, keep = k._marshal(bufs, N, N, H, 1, None, hs, None, ws, False)
                    best = 1e9
                    for _ in range(3):
                        ms = ctypes.c_float()
                        check(lib().mfa_attention_kernel_time(k._handle, ctypes.byref(arr), ctypes.byref(params), ctypes.c_void_p(stream), 1, iters, ctypes.byref(ms)))
                        best = min(best, ms.value / iters)
                    out.append(OPS[t](D) * N * N * H / (best * 1e-3) / 1e9)
                    names.append(k.variant.replace("attn_", ""))
                print(f"{dtype:5s} {N:6d} {D:4d} {H:5d} | {out[0]:9.0f} {out[1]:9.0f} {out[2]:9.0f}           | {' '.join(names)}")
                del bufs
                torch.cuda.empty_cache()
