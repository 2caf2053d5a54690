#!/usr/bin/env python3
"""developer check + A/B of the D <= 64 persistent forward kernel (attn_fwd16_p6): the launch against the C oracle on a few heads,
then interleaved timing rounds against the eight-wave kernel (table row | 64 | 256 | 32 | 64 |) -- N(0,1) and all-zero operands.

  python tools/p6_ab.py [--N 4096 --D 64 --heads 256] [--dev STREAM,...]   (--dev: developer streams, libmfa_hip_dev.so)
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--D", type=int, default=64)
    ap.add_argument("--heads", type=int, default=256)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--dev", default="")
    ap.add_argument("--fills", default="normal,zero")
    ap.add_argument("--mixed", type=int, default=1, help="0: lowPrecisionInputs only (exact-scale streams, FP32 L)")
    args = ap.parse_args()
    if args.dev:
        os.environ.setdefault("MFA_LIBRARY", os.path.join(ROOT, "metal_flash_attention_amd", "libmfa_hip_dev.so"))
    import numpy as np
    import torch
    import metal_flash_attention_amd as mfa
    from metal_flash_attention_amd import (AttentionDescriptor, AttentionKernel, AttentionKernelType as T, AttentionOperand as Op,
                                           GEMMOperandPrecision as P)
    N, D, H = args.N, args.D, args.heads
    desc = AttentionDescriptor()
    desc.lowPrecisionInputs = True
    desc.lowPrecisionIntermediates = bool(args.mixed)
    desc.lowPrecisionInputType = P.BF16
    desc.matrixDimensions = (N, N, D)
    desc.transposeState = (False,) * 4
    mfa.setParameterFile(T.forward, True, "| 32 | 256 | 64 | 64 | Q, O |\n| 64 | 256 | 64 | 64 | Q, O |\n| 128 | 256 | 64 | 128 | Q, O |\n")
    k6 = AttentionKernel(desc.kernelDescriptor(T.forward))
    mfa.setParameterFile(T.forward, True, "| 32 | 128 | 32 | 32 | Q, O |\n| 64 | 256 | 32 | 64 | Q, O |\n| 128 | 256 | 64 | 128 | Q, O |\n")
    k3 = AttentionKernel(desc.kernelDescriptor(T.forward))
    mfa.resetParameterFiles()
    print("# variants:", k6.variant, "|", k3.variant)
    hs = {Op.Q: N * D, Op.K: N * D, Op.V: N * D, Op.O: N * D, Op.L: N}
    stream = torch.cuda.current_stream().cuda_stream
    flops = 4.0 * N * N * D * H
    kw = dict(row=N, column=N, heads=H, headStrides=hs)
    for fill in args.fills.split(","):
        g = torch.Generator(device="cuda")
        g.manual_seed(0)
        if fill == "zero":
            bufs = {op: torch.zeros((H, N, D), device="cuda", dtype=torch.bfloat16) for op in (Op.Q, Op.K, Op.V)}
        else:
            bufs = {op: torch.randn((H, N, D), generator=g, device="cuda").to(torch.bfloat16) for op in (Op.Q, Op.K, Op.V)}
        bufs[Op.O] = torch.zeros((H, N, D), device="cuda")
        bufs[Op.L] = torch.zeros((H, N), device="cuda", dtype=torch.float16 if args.mixed else torch.float32)
        print("# launch form:", k6.launchForm(bufs, **kw))
        runs = [("p6", k6, None), ("v3 (8 x 32)", k3, None)] + [("p6:" + s, k6, s) for s in args.dev.split(",") if s]

        def setenv(s):
            os.environ.pop("MFA_P6_DEV_STREAM", None)
            if s:
                os.environ["MFA_P6_DEV_STREAM"] = s

        outs = {}
        for name, k, s in runs:
            setenv(s)
            bufs[Op.O].zero_()
            bufs[Op.L].zero_()
            k.dispatch(bufs, stream=stream, **kw)
            torch.cuda.synchronize()
            outs[name] = (bufs[Op.O].clone(), bufs[Op.L].float().clone())
        if fill == "normal":   # the C oracle on two heads (fp64 accumulation of the same rounded inputs)
            from oracle import network_np
            for h in (0, H - 1):
                q, kk, v = (bufs[op][h].float().cpu().numpy().astype(np.float64) for op in (Op.Q, Op.K, Op.V))
                s = q @ kk.T / np.sqrt(D)
                m = s.max(axis=1, keepdims=True)
                p = np.exp(s - m)
                l = p.sum(axis=1, keepdims=True)
                oref = (p @ v) / l
                lref = (m[:, 0] * 1.44269504089 + np.log2(l[:, 0]))
                for name in outs:
                    do = np.abs(outs[name][0][h].cpu().numpy() - oref).max()
                    dl = np.abs(outs[name][1][h].cpu().numpy() - lref).max()
                    print(f"#   head {h:3d} {name:24s} |O - fp64| {do:.2e}   |L - fp64| {dl:.2e}")
        times = {name: [] for name, _, _ in runs}
        for _ in range(30):
            k6.dispatch(bufs, stream=stream, **kw)
        for r in range(args.rounds):
            for name, k, s in runs:
                setenv(s)
                times[name].append(k.time(bufs, stream=stream, warmup=2, iterations=args.iters, **kw) / args.iters)
        print("## fill = %s   (N = %d, D = %d, %d heads, %s; 2.5 PF roof)" % (fill, N, D, H, "mixed mode" if args.mixed else "fp32 intermediates"))
        for name, _, _ in runs:
            t = sorted(times[name])
            med = t[len(t) // 2]
            do = (outs[name][0] - outs["p6"][0]).abs().max().item()
            print(f"{name:28s} med {med:8.4f} ms {flops / med / 1e9:8.1f} TF frac {flops / med / 2.5e12:6.4f}  best {flops / t[0] / 2.5e12:6.4f}  |dO vs p6| {do:.2e}")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
