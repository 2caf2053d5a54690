#!/usr/bin/env python3
"""Segment timing of the persistent forward kernel from its PROF stream (developer build): shader-clock sums per segment of the
block loop, per wave, left in O[first row of the wave's last block][0:16].

  python tools/p4p_prof.py [--N 4096 --heads 256]      (needs make -C metal_flash_attention_amd/csrc DEV=1)
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("MFA_LIBRARY", os.path.join(ROOT, "metal_flash_attention_amd", "libmfa_hip_dev.so"))
os.environ["MFA_P4P_PROF"] = "1"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--heads", type=int, default=256)
    args = ap.parse_args()
    import torch
    import p4pgen
    from metal_flash_attention_amd import (AttentionDescriptor, AttentionKernel, AttentionKernelType,
                                           AttentionOperand as Op, GEMMOperandPrecision as P)
    N, D, H = args.N, 128, args.heads
    desc = AttentionDescriptor()
    desc.lowPrecisionInputs = True
    desc.lowPrecisionInputType = P.BF16
    desc.matrixDimensions = (N, N, D)
    desc.transposeState = (False,) * 4
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    bufs = {op: torch.randn((H, N, D), generator=g, device="cuda").to(torch.bfloat16) for op in (Op.Q, Op.K, Op.V)}
    hs = {Op.Q: N * D, Op.K: N * D, Op.V: N * D, Op.O: N * D, Op.L: N}
    k = AttentionKernel(desc.kernelDescriptor(AttentionKernelType.forward))
    o = torch.zeros((H, N, D), device="cuda")
    bufs[Op.O], bufs[Op.L] = o, torch.zeros((H, N), device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        k.dispatch(bufs, row=N, column=N, heads=H, headStrides=hs, stream=stream)
    torch.cuda.synchronize()
    ms = k.time(bufs, row=N, column=N, heads=H, headStrides=hs, stream=stream, warmup=3, iterations=10) / 10
    o.zero_()
    k.dispatch(bufs, row=N, column=N, heads=H, headStrides=hs, stream=stream)
    torch.cuda.synchronize()
    c = o[:, ::64, 0:16].contiguous().view(torch.int32).reshape(-1, 16)
    nb = p4pgen.PROF_NAMES.index("blocks")
    c = c[c[:, 15] == p4pgen.PROF_MAGIC].double()      # the waves that ended a workgroup
    blocks = c[:, nb].mean().item()
    nt = (N + 63) // 64
    print(f"{k.variant} (persistent PROF stream) N={N} heads={H}: {ms:.4f} ms/launch, {c.shape[0]} waves reported, {blocks:.1f} blocks per workgroup")
    print("shader clocks per block (mean over reporting waves; loop_* are sums over the block's %d loop tiles):" % (nt + (nt & 1) - 1))
    tot = 0.0
    for i, name in enumerate(p4pgen.PROF_NAMES[:nb]):
        v = (c[:, i] / c[:, nb]).mean().item()
        if name != "tile1_wait":
            tot += v
        print(f"  {name:12s} {v:10.0f}")
    print(f"  {'sum':12s} {tot:10.0f}   (launch / blocks per workgroup = {ms * 1e3 / blocks:.1f} us)")


if __name__ == "__main__":
    main()
