#!/usr/bin/env python3
"""developer A/B (libmfa_hip_dev.so): the persistent forward kernel's developer streams (tools/p4pgen.py: slot-balanced schedules,
code-placement pads, timing-only ablations) against the product stream -- headline shape, mixed mode, interleaved rounds in ONE
process, on N(0,1) and on all-zero operands (full clock: cycle efficiency).

  python tools/p4p_streams_ab.py [--streams A,B,...] [--fills normal,zero]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("MFA_LIBRARY", os.path.join(ROOT, "metal_flash_attention_amd", "libmfa_hip_dev.so"))


def main():
    import p4pgen
    V = p4pgen.VARIANTS
    dev = [n for n in V if n not in p4pgen.PRODUCT_STREAMS and V[n].fold and V[n].l16 and not V[n].o16 and not V[n].causal
           and V[n].dtype == "bf16" and not V[n].merge and not V[n].fuse]
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--heads", type=int, default=256)
    ap.add_argument("--streams", default=",".join(dev))
    ap.add_argument("--fills", default="normal,zero")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--mixed", type=int, default=1, help="0: the fp32-intermediates mode (exact-scale streams, FP32 L)")
    ap.add_argument("--causal", type=int, default=0, help="1: causal launches (the causal developer streams)")
    ap.add_argument("--heat", type=float, default=0.0, help="seconds of back-to-back product launches in front of the timed rounds (a GPU "
                    "under sustained load sits at its power / thermal limit: the clock the streams are then granted differs from a cold start)")
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
    k = AttentionKernel(desc.kernelDescriptor(AttentionKernelType.forward))
    hs = {Op.Q: N * D, Op.K: N * D, Op.V: N * D, Op.O: N * D, Op.L: N}
    stream = torch.cuda.current_stream().cuda_stream
    names = ["product"] + [s for s in args.streams.split(",") if s]
    flops = 4.0 * N * N * D * H * ((N + 1) / (2.0 * N) if args.causal else 1.0)
    causal = bool(args.causal)
    for fill in args.fills.split(","):
        g = torch.Generator(device="cuda")
        g.manual_seed(0)
        if fill == "zero":
            bufs = {op: torch.zeros((H, N, D), device="cuda", dtype=torch.bfloat16) for op in (Op.Q, Op.K, Op.V)}
        else:
            bufs = {op: torch.randn((H, N, D), generator=g, device="cuda").to(torch.bfloat16) for op in (Op.Q, Op.K, Op.V)}
        bufs[Op.O] = torch.zeros((H, N, D), device="cuda")
        bufs[Op.L] = torch.zeros((H, N), device="cuda", dtype=torch.float16 if args.mixed else torch.float32)

        def setenv(name):
            os.environ.pop("MFA_P4P_DEV_STREAM", None)
            if name != "product":
                os.environ["MFA_P4P_DEV_STREAM"] = name

        outs, times = {}, {n: [] for n in names}
        for name in names:
            setenv(name)
            bufs[Op.O].zero_()
            k.dispatch(bufs, row=N, column=N, heads=H, headStrides=hs, stream=stream, causal=causal)
            torch.cuda.synchronize()
            outs[name] = (bufs[Op.O].clone(), bufs[Op.L].float().clone())
        setenv("product")
        for _ in range(40):
            k.dispatch(bufs, row=N, column=N, heads=H, headStrides=hs, stream=stream, causal=causal)
        import time as _time
        t_heat = _time.perf_counter()
        while _time.perf_counter() - t_heat < args.heat:
            for _ in range(50):
                k.dispatch(bufs, row=N, column=N, heads=H, headStrides=hs, stream=stream, causal=causal)
            torch.cuda.synchronize()
        for r in range(args.rounds):
            for name in names:
                setenv(name)
                times[name].append(k.time(bufs, row=N, column=N, heads=H, headStrides=hs, stream=stream, warmup=2, iterations=args.iters, causal=causal) / args.iters)
        print("## fill = %s   (N = %d, D = %d, %d heads, %s; 2.5 PF roof)" % (fill, N, D, H, "mixed mode" if args.mixed else "fp32 intermediates"))
        for name in names:
            t = sorted(times[name])
            med = t[len(t) // 2]
            do = (outs[name][0] - outs["product"][0]).abs().max().item()
            dl = (outs[name][1] - outs["product"][1]).abs().max().item()
            print(f"{name:28s} med {med:8.4f} ms {flops / med / 1e9:8.1f} TF frac {flops / med / 2.5e12:6.4f}  best {flops / t[0] / 2.5e12:6.4f}  |dO| {do:.2e} |dL| {dl:.2e}")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
