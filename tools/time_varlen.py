#!/usr/bin/env python3
"""Forward launches with per-batch lengths at D = 128 (bf16, mixed mode): the persistent kernels' geometry streams (round 6) against the kernels
that served them until round 5 (developer library knobs: MFA_P4_NO_PERSISTENT=1 -> attn_fwd16_p4, one block per workgroup; MFA_P6_OFF=1 ->
the eight-wave attn_fwd16_v3 at D <= 64).

  python tools/time_varlen.py [--N 4096 --batches 8 --heads 32]
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
    ap.add_argument("--batches", type=int, default=8)
    ap.add_argument("--heads", type=int, default=32)
    ap.add_argument("--D", type=int, default=128, help="128: attn_fwd16_p4p; 64: attn_fwd16_p6")
    args = ap.parse_args()
    import numpy as np
    import torch
    from metal_flash_attention_amd import (AttentionDescriptor, AttentionKernel, AttentionKernelType,
                                           AttentionOperand as Op, GEMMOperandPrecision as P)
    N, B, H, D = args.N, args.batches, args.heads, args.D
    knob = "MFA_P4_NO_PERSISTENT" if D > 64 else "MFA_P6_OFF"
    desc = AttentionDescriptor()
    desc.lowPrecisionInputs = desc.lowPrecisionIntermediates = True
    desc.lowPrecisionInputType = P.BF16
    desc.matrixDimensions = (N, N, D)
    desc.transposeState = (False,) * 4
    k = AttentionKernel(desc.kernelDescriptor(AttentionKernelType.forward))
    g = torch.Generator(device="cuda")
    g.manual_seed(0)
    bufs = {op: torch.randn((B, H, N, D), generator=g, device="cuda").to(torch.bfloat16) for op in (Op.Q, Op.K, Op.V)}
    bufs[Op.O] = torch.zeros((B, H, N, D), device="cuda")
    bufs[Op.L] = torch.zeros((B, H, N), device="cuda", dtype=torch.float16)
    hs = {Op.Q: N * D, Op.K: N * D, Op.V: N * D, Op.O: N * D, Op.L: N}
    bs = {op: v * H for op, v in hs.items()}
    stream = torch.cuda.current_stream().cuda_stream
    rng = np.random.default_rng(1)
    import time as _time
    arms = (("persistent (round 6)", None), ("round-5 kernel (knob)", "1"))

    def setenv(env):
        os.environ.pop(knob, None)
        os.environ.pop("MFA_P4P_LENGTHS", None)
        if env:
            os.environ[knob] = env
        elif D > 64:
            os.environ["MFA_P4P_LENGTHS"] = "1"    # (D = 128: the developer library's switch for routing lengths to the persistent kernel)

    for name, lens in (("full length", [N] * B), ("uniform 25..100 %", [int(x) for x in rng.integers(N // 4, N + 1, B)])):
        rl = torch.tensor(lens, dtype=torch.int32, device="cuda")
        for causal in (False, True):
            flops = sum(4.0 * L * L * D * H * ((L + 1) / (2.0 * L) if causal else 1.0) for L in lens)
            kw = dict(row=N, column=N, heads=H, batches=B, headStrides=hs, batchStrides=bs, stream=stream, causal=causal, rowLengths=rl, columnLengths=rl)
            # spin-up (the first launches of a process run at a clock the later ones do not get: an un-warmed first arm lost 6 % in the
            # first version of this tool), then interleaved rounds, medians
            t0 = _time.perf_counter()
            while _time.perf_counter() - t0 < 1.0:
                for _ in range(20):
                    k.dispatch(bufs, **kw)
                torch.cuda.synchronize()
            times, forms = {a[0]: [] for a in arms}, {}
            for r in range(5):
                for label, env in arms:
                    setenv(env)
                    forms[label] = k.launchForm(bufs, **{x: y for x, y in kw.items() if x != "stream"})
                    for _ in range(3):
                        k.dispatch(bufs, **kw)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(10):
                        k.dispatch(bufs, **kw)
                    e1.record()
                    torch.cuda.synchronize()
                    times[label].append(e0.elapsed_time(e1) / 10)
            for label, _ in arms:
                ms = sorted(times[label])[2]
                print(f"{name:18s} causal={int(causal)} {label:24s} {ms:8.4f} ms {flops / ms / 1e9:8.1f} TF {flops / ms / 2.5e12:6.3f}   {forms[label][:60]}", flush=True)
    setenv("1")
    os.environ.pop(knob, None)


if __name__ == "__main__":
    main()
