#!/usr/bin/env python3
"""developer probe: the FP32 path (BASELINE config 3) kernel by kernel -- forward / dQ / dK-dV launch times, N = 4096, 32 heads,
FP32 operands.   python tools/f32_perf.py [--causal] [--fill zero] [--heads H] [--n N] [D ...]
Each cell: launch form: ms, TFLOP/s, fraction of the 157.3 TF fp32 matrix roof."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from metal_flash_attention_amd import AttentionDescriptor, AttentionKernel, AttentionKernelType as T, AttentionOperand as Op  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("dims", nargs="*", type=int)
ap.add_argument("--causal", action="store_true")
ap.add_argument("--fill", default="normal", choices=("normal", "zero"))
ap.add_argument("--heads", type=int, default=32)
ap.add_argument("--n", type=int, default=4096)
args = ap.parse_args()
N, H = args.n, args.heads
print(f"# tools/f32_perf.py: N={N} heads={H} fp32 causal={int(args.causal)} fill={args.fill}")
for D in args.dims or [64, 128]:
    desc = AttentionDescriptor()
    desc.lowPrecisionInputs = desc.lowPrecisionIntermediates = False
    desc.matrixDimensions = (N, N, D)
    desc.transposeState = (False,) * 4
    g = torch.Generator(device="cuda")
    g.manual_seed(0)
    bufs = {op: torch.randn((H, N, D), generator=g, device="cuda") for op in (Op.Q, Op.K, Op.V, Op.dO)}
    if args.fill == "zero":
        for t in bufs.values():
            t.zero_()
    for op in (Op.O, Op.dQ, Op.dK, Op.dV):
        bufs[op] = torch.zeros((H, N, D), device="cuda")
    bufs[Op.L] = torch.zeros((H, N), device="cuda")
    bufs[Op.D] = torch.zeros((H, N), device="cuda")
    hs = {op: (N if op in (Op.L, Op.D) else N * D) for op in bufs}
    stream = torch.cuda.current_stream().cuda_stream
    work = (N + 1) / (2.0 * N) if args.causal else 1.0
    line = [f"D={D:3d}"]
    for t, gemms in ((T.forward, 4), (T.backwardQuery, 6), (T.backwardKeyValue, 8)):
        k = AttentionKernel(desc.kernelDescriptor(t))
        kw = dict(row=N, column=N, heads=H, headStrides=hs, causal=args.causal)
        for _ in range(5):
            k.dispatch(bufs, stream=stream, **kw)
        ms = k.time(bufs, stream=stream, warmup=2, iterations=10, **kw) / 10
        tf = gemms * N * N * D * H * work / ms / 1e9
        line.append(f"{k.launchForm(bufs, **kw).split(' ')[0]}: {ms:.3f} ms {tf:.1f} TF {tf / 157.3:.3f}")
    print("  ".join(line))
    if os.environ.get("MFA_F32_PROF") and D == 128:   # developer library: dQ[row 0 of a block][0..4] = clocks per tile of the phases
        dq = bufs[Op.dQ].cpu()
        names = ("first products", "barrier", "LDS-DMA issue", "softmax arithmetic", "second product")
        for blk in (0, 7, 31):
            v = dq[0, 128 * blk, :5].tolist()
            print(f"    dQ phase clocks per tile, head 0 row block {blk}: " + ", ".join(f"{n} {x:.0f}" for n, x in zip(names, v)) + f"; sum {sum(v):.0f}")
