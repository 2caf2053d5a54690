#!/usr/bin/env python3
"""A/B timing of forward-kernel variants in ONE process, interleaved rounds (guide rule 24), with a
cross-check of every variant's O/L against the first variant.  Developer tool, not product.

  python tools/ab_fwd16.py --impls v1,v2:0,v2:1 [--N 4096 --D 128 --heads 256 --rounds 5 --iters 5]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the MFA_*_IMPL knobs exist only in the developer build of the library (make -C metal_flash_attention_amd/csrc DEV=1)
os.environ.setdefault("MFA_LIBRARY", os.path.join(ROOT, "metal_flash_attention_amd", "libmfa_hip_dev.so"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--impls", default="v1,v2:0")
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--D", type=int, default=128)
    ap.add_argument("--heads", type=int, default=256)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=5)
    args = ap.parse_args()
    import torch
    from metal_flash_attention_amd import (AttentionDescriptor, AttentionKernel, AttentionKernelType,
                                           AttentionOperand as Op, GEMMOperandPrecision as P)
    N, D, H = args.N, args.D, args.heads
    desc = AttentionDescriptor()
    desc.lowPrecisionInputs = True
    desc.lowPrecisionInputType = P.BF16 if args.dtype == "bf16" else P.FP16
    desc.matrixDimensions = (N, N, D)
    desc.transposeState = (False,) * 4
    tdt = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    bufs = {op: torch.randn((H, N, D), generator=g, device="cuda").to(tdt) for op in (Op.Q, Op.K, Op.V)}
    hs = {Op.Q: N * D, Op.K: N * D, Op.V: N * D, Op.O: N * D, Op.L: N}
    kernels, outs = {}, {}
    stream = torch.cuda.current_stream().cuda_stream
    for impl in args.impls.split(","):
        os.environ["MFA_FWD16_IMPL"] = impl
        k = AttentionKernel(desc.kernelDescriptor(AttentionKernelType.forward))
        o = torch.full((H, N, D), float("nan"), device="cuda")
        l = torch.zeros((H, N), device="cuda")
        b = dict(bufs); b[Op.O] = o; b[Op.L] = l
        k.dispatch(b, row=N, column=N, heads=H, headStrides=hs, stream=stream)
        torch.cuda.synchronize()
        kernels[impl] = (k, b)
        outs[impl] = (o, l)
    first = args.impls.split(",")[0]
    times = {impl: [] for impl in kernels}
    for r in range(args.rounds):
        for impl, (k, b) in kernels.items():
            ms = k.time(b, row=N, column=N, heads=H, headStrides=hs, stream=stream, warmup=1, iterations=args.iters)
            times[impl].append(ms / args.iters)
    flops = 4.0 * N * N * D * H
    for impl, (k, b) in kernels.items():
        t = sorted(times[impl])
        med, best = t[len(t) // 2], t[0]
        do = (outs[impl][0] - outs[first][0]).abs().max().item()
        dl = (outs[impl][1] - outs[first][1]).abs().max().item()
        nan = torch.isnan(outs[impl][0]).any().item()
        print(f"{impl:8s} {k.variant:42s} med {med:8.4f} ms  {flops / med / 1e9:8.1f} TF  best {flops / best / 1e9:8.1f} TF "
              f" |dO| {do:.2e} |dL| {dl:.2e} nan={nan}")


if __name__ == "__main__":
    main()
