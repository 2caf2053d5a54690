#!/usr/bin/env python3
"""developer A/B (libmfa_hip_dev.so): the one-block-per-workgroup forward kernel against its persistent form, and the
persistent form's start stagger, interleaved rounds in ONE process.

  python tools/p4p_ab.py [--N 4096 --heads 256 --mixed 1 --staggers 0,1,2,4]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MFA_LIBRARY", os.path.join(ROOT, "metal_flash_attention_amd", "libmfa_hip_dev.so"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--heads", type=int, default=256)
    ap.add_argument("--mixed", type=int, default=1)
    ap.add_argument("--staggers", default="0,1,2,4")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    args = ap.parse_args()
    import torch
    from metal_flash_attention_amd import (AttentionDescriptor, AttentionKernel, AttentionKernelType,
                                           AttentionOperand as Op, GEMMOperandPrecision as P)
    N, D, H = args.N, 128, args.heads
    desc = AttentionDescriptor()
    desc.lowPrecisionInputs = True
    desc.lowPrecisionIntermediates = bool(args.mixed)
    desc.lowPrecisionInputType = P.BF16
    desc.matrixDimensions = (N, N, D)
    desc.transposeState = (False,) * 4
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    bufs = {op: torch.randn((H, N, D), generator=g, device="cuda").to(torch.bfloat16) for op in (Op.Q, Op.K, Op.V)}
    hs = {Op.Q: N * D, Op.K: N * D, Op.V: N * D, Op.O: N * D, Op.L: N}
    k = AttentionKernel(desc.kernelDescriptor(AttentionKernelType.forward))
    bufs[Op.O] = torch.zeros((H, N, D), device="cuda")
    bufs[Op.L] = torch.zeros((H, N), device="cuda", dtype=torch.float16 if args.mixed else torch.float32)
    stream = torch.cuda.current_stream().cuda_stream
    configs = [("one block / workgroup", {"MFA_P4_NO_PERSISTENT": "1"})] + [
        ("persistent, stagger %s" % s, {"MFA_P4P_STAGGER": s}) for s in args.staggers.split(",")]
    outs, times = {}, {name: [] for name, _ in configs}

    def setenv(env):
        for key in ("MFA_P4_NO_PERSISTENT", "MFA_P4P_STAGGER"):
            os.environ.pop(key, None)
        os.environ.update(env)

    for name, env in configs:
        setenv(env)
        bufs[Op.O].zero_()
        k.dispatch(bufs, row=N, column=N, heads=H, headStrides=hs, stream=stream)
        torch.cuda.synchronize()
        outs[name] = (bufs[Op.O].clone(), bufs[Op.L].float().clone())
    for _ in range(40):
        k.dispatch(bufs, row=N, column=N, heads=H, headStrides=hs, stream=stream)
    for r in range(args.rounds):
        for name, env in configs:
            setenv(env)
            times[name].append(k.time(bufs, row=N, column=N, heads=H, headStrides=hs, stream=stream, warmup=2, iterations=args.iters) / args.iters)
    flops = 4.0 * N * N * D * H
    first = configs[0][0]
    for name, _ in configs:
        t = sorted(times[name])
        med = t[len(t) // 2]
        do = (outs[name][0] - outs[first][0]).abs().max().item()
        dl = (outs[name][1] - outs[first][1]).abs().max().item()
        print(f"{name:28s} med {med:8.4f} ms {flops / med / 1e9:8.1f} TF  best {flops / t[0] / 1e9:8.1f} TF  |dO| {do:.2e} |dL| {dl:.2e} vs {first}")


if __name__ == "__main__":
    main()
