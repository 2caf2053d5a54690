#!/usr/bin/env python3
"""Phase timing of attn_dkv16_p4 / attn_dq16_p4 from their PROF streams (developer tool, libmfa_hip_dev.so): the streams
stamp the shader clock at the phase seams of every step and every wave leaves the sums in the first four words of its
first output row (dV / dQ, which are garbage there in such a run).

  python tools/bwd4_prof.py [--kernel dkv|dq] [--N 4096 --heads 64]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("MFA_LIBRARY", os.path.join(ROOT, "metal_flash_attention_amd", "libmfa_hip_dev.so"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", default="dkv", choices=("dkv", "dq"))
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--heads", type=int, default=64)
    ap.add_argument("--prof", type=int, default=1, help="0: time the product stream instead")
    ap.add_argument("--stream", default="", help="name of a PROF / ablation stream of the kernel's generator")
    args = ap.parse_args()
    import torch
    from metal_flash_attention_amd import (AttentionDescriptor, AttentionKernel, AttentionKernelType as T,
                                           AttentionOperand as Op, GEMMOperandPrecision as P)
    import dkv4gen
    import dq4gen
    N, D, H = args.N, 128, args.heads
    desc = AttentionDescriptor()
    desc.lowPrecisionInputs = desc.lowPrecisionIntermediates = True
    desc.lowPrecisionInputType = P.BF16
    desc.matrixDimensions = (N, N, D)
    desc.transposeState = (False,) * 4
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    bufs = {op: torch.randn((H, N, D), generator=g, device="cuda").to(torch.bfloat16) for op in (Op.Q, Op.K, Op.V, Op.dO)}
    mem = desc.memoryPrecisions
    tp = {P.FP32: torch.float32, P.FP16: torch.float16, P.BF16: torch.bfloat16}
    bufs[Op.O] = torch.zeros((H, N, D), device="cuda", dtype=tp[mem[Op.O]])
    bufs[Op.L] = torch.zeros((H, N), device="cuda", dtype=tp[mem[Op.L]])
    bufs[Op.D] = torch.zeros((H, N), device="cuda", dtype=tp[mem[Op.D]])
    for op in (Op.dQ, Op.dK, Op.dV):
        bufs[op] = torch.zeros((H, N, D), device="cuda")
    hs = {op: (N if op in (Op.L, Op.D) else N * D) for op in bufs}
    stream = torch.cuda.current_stream().cuda_stream
    for t in (T.forward, T.backwardQuery):      # real L and D
        AttentionKernel(desc.kernelDescriptor(t)).dispatch(bufs, row=N, column=N, heads=H, headStrides=hs, stream=stream)
    if args.prof:
        if args.kernel == "dkv":
            os.environ["MFA_DKV16_IMPL"] = "p4:%d" % (1000 + list(dkv4gen.VARIANTS).index("BF16_MIXED_PROF"))
        else:
            os.environ["MFA_DQ16_IMPL"] = "p4:%d" % (1000 + list(dq4gen.VARIANTS).index(args.stream or "BF16_FOLD_PROF"))
    kt = T.backwardKeyValue if args.kernel == "dkv" else T.backwardQuery
    k = AttentionKernel(desc.kernelDescriptor(kt))
    for _ in range(3):
        k.dispatch(bufs, row=N, column=N, heads=H, headStrides=hs, stream=stream)
    torch.cuda.synchronize()
    ms = k.time(bufs, row=N, column=N, heads=H, headStrides=hs, stream=stream, warmup=1, iterations=5) / 5
    flops = (8.0 if args.kernel == "dkv" else 6.0) * N * N * D * H
    print(f"{k.variant}: {ms:.4f} ms/launch, {flops / ms / 1e9:.1f} TF")
    if not args.prof:
        return
    out = bufs[Op.dV] if args.kernel == "dkv" else bufs[Op.dQ]
    c = out[:, ::64, 0:4].contiguous().view(torch.int32).double()      # [H][N/64 waves][4 sums]
    if args.kernel == "dkv":
        steps, names, n_mfma = N // 32, ("S   (18 MFMA)", "dP  (18 MFMA)", "dV  (16 MFMA)", "dK  (16 MFMA) + seam"), 68
        per = c / steps
        units_per_cu = steps * (N // 256) * H / 256
    else:
        steps, names, n_mfma = N // 64, ("S(0) P(0) (32 MFMA)", "S(1) P(1) (32 MFMA)", "Q(0) Q(1) (32 MFMA) + seam"), 96
        per = c[..., :3] / steps
        units_per_cu = steps * (N // 256) * H / 256
    print("shader-clock cycles per step and wave (mean / min / max over all waves):")
    for i, name in enumerate(names):
        print(f"  {name:28s} {per[..., i].mean():8.1f} {per[..., i].min():8.1f} {per[..., i].max():8.1f}")
    tot = per.sum(-1)
    print(f"  {'step total':28s} {tot.mean():8.1f} {tot.min():8.1f} {tot.max():8.1f}   ({n_mfma} MFMA = {32 * n_mfma} cycles of matrix pipe)")
    print(f"  implied clock if the loop were the whole launch: {tot.mean() * units_per_cu / (ms * 1e-3) / 1e9:.2f} GHz")


if __name__ == "__main__":
    main()
