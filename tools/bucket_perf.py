#!/usr/bin/env python3
"""developer probe: forward / dQ / dK-dV launch times per head dimension (N = 4096, 64 heads, bf16), product library.
   python tools/bucket_perf.py [--mixed] [--transposed] [--fill zero] [--heads H] [--n N] [D ...]
--mixed: lowPrecisionIntermediates (FOLD / pre-scaled streams, FP16 L, BF16 D); --transposed: every operand stored [D][N], no
workspace (the in-place kernels); --fill zero: all-zero operands (clock experiment, MI355X_MICROARCH.md DVFS note).
Each cell: variant (launch form when it differs): ms, TFLOP/s, fraction of the 2.5 PF bf16 roof."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from metal_flash_attention_amd import (AttentionDescriptor, AttentionKernel, AttentionKernelType as T, AttentionOperand as Op,  # noqa: E402
                                       GEMMOperandPrecision as P)

ap = argparse.ArgumentParser()
ap.add_argument("dims", nargs="*", type=int)
ap.add_argument("--mixed", action="store_true")
ap.add_argument("--transposed", action="store_true")
ap.add_argument("--causal", action="store_true")
ap.add_argument("--fill", default="normal", choices=("normal", "zero"))
ap.add_argument("--dkv32", action="store_true", help="A/B: the 32-key role-split pairs (attn_dkv16_rs.h) at D > 128 instead of attn_dkv16_p5")
ap.add_argument("--f16", action="store_true", help="FP16 Q / K / V (the reference's own low-precision type; dO stays BF16 as its descriptors say)")
ap.add_argument("--heads", type=int, default=64)
ap.add_argument("--n", type=int, default=4096)
args = ap.parse_args()
if os.environ.get("MIXED", "0") == "1":   # (the round-2/3 spelling)
    args.mixed = True
N, H = args.n, args.heads
if args.dkv32:
    import metal_flash_attention_amd as mfa
    mfa.setParameterFile(T.backwardKeyValue, True, "| 64 | 256 | 32 | 64 | K, V, dV, dK |\n| 128 | 256 | 32 | 128 | K, V, dV, dK |\n| 160 | 64 | 32 | 160 | K, V, dV, dK |\n"
                         "| 192 | 64 | 32 | 192 | K, V, dV, dK |\n| 256 | 64 | 32 | 256 | K, V, dV, dK |\n")
print(f"# tools/bucket_perf.py: N={N} heads={H} {'f16 (dO bf16)' if args.f16 else 'bf16'} mixed={int(args.mixed)} transposed={int(args.transposed)} causal={int(args.causal)} fill={args.fill}")
for D in args.dims or [64, 128, 160, 192, 256]:
    desc = AttentionDescriptor()
    desc.lowPrecisionInputs = True
    desc.lowPrecisionIntermediates = args.mixed
    desc.lowPrecisionInputType = P.FP16 if args.f16 else P.BF16
    desc.matrixDimensions = (N, N, D)
    desc.transposeState = (args.transposed,) * 4
    g = torch.Generator(device="cuda")
    g.manual_seed(0)
    tin = torch.float16 if args.f16 else torch.bfloat16
    bufs = {op: torch.randn((H, N, D), generator=g, device="cuda").to(torch.bfloat16 if op == Op.dO else tin) for op in (Op.Q, Op.K, Op.V, Op.dO)}
    if args.fill == "zero":
        for t in bufs.values():
            t.zero_()
    mem = desc.memoryPrecisions
    tp = {P.FP32: torch.float32, P.FP16: torch.float16, P.BF16: torch.bfloat16}
    bufs[Op.O] = torch.zeros((H, N, D), device="cuda")
    bufs[Op.L] = torch.zeros((H, N), device="cuda", dtype=tp[mem[Op.L]])
    bufs[Op.D] = torch.zeros((H, N), device="cuda", dtype=tp[mem[Op.D]])
    for op in (Op.dQ, Op.dK, Op.dV):
        bufs[op] = torch.zeros((H, N, D), device="cuda")
    hs = {op: (N if op in (Op.L, Op.D) else N * D) for op in bufs}
    stream = torch.cuda.current_stream().cuda_stream
    work = (N + 1) / (2.0 * N) if args.causal else 1.0
    line = [f"D={D:3d}"]
    for t, gemms in ((T.forward, 4), (T.backwardQuery, 6), (T.backwardKeyValue, 8)):
        k = AttentionKernel(desc.kernelDescriptor(t))
        kw = dict(row=N, column=N, heads=H, headStrides=hs, causal=args.causal)
        for _ in range(20):
            k.dispatch(bufs, stream=stream, **kw)
        ms = k.time(bufs, stream=stream, warmup=5, iterations=20, **kw) / 20
        tf = gemms * N * N * D * H * work / ms / 1e9
        form = k.launchForm(bufs, **kw)
        name = k.variant if form.startswith(k.variant) else f"{k.variant} [{form.split(' ')[0]}]"
        line.append(f"{name}: {ms:.3f} ms {tf:.0f} TF {tf / 2500:.3f}")
    print("  ".join(line))
