#!/usr/bin/env python3
"""Developer check of one forward variant on a small problem against a float64 attention (which rows / which output)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MFA_LIBRARY", os.path.join(ROOT, "metal_flash_attention_amd", "libmfa_hip_dev.so"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--impl", default="p4:10")
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--R", type=int, default=256)
    ap.add_argument("--C", type=int, default=256)
    ap.add_argument("--zero", default="", help="q / k: zero that operand (scores all equal)")
    ap.add_argument("--qscale", type=float, default=1.0)
    args = ap.parse_args()
    import numpy as np
    import torch
    from metal_flash_attention_amd import (AttentionDescriptor, AttentionKernel, AttentionKernelType,
                                           AttentionOperand as Op, GEMMOperandPrecision as P)
    R, C, D = args.R, args.C, 128
    desc = AttentionDescriptor()
    desc.lowPrecisionInputs = True
    desc.lowPrecisionInputType = P.BF16 if args.dtype == "bf16" else P.FP16
    desc.matrixDimensions = (R, C, D)
    desc.transposeState = (False,) * 4
    tdt = torch.bfloat16 if args.dtype == "bf16" else torch.float16
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    q, k, v = (torch.randn((n, D), generator=g, device="cuda").to(tdt) for n in (R, C, C))
    if args.zero == "q":
        q.zero_()
    if args.zero == "k":
        k.zero_()
    q = (q.float() * args.qscale).to(tdt)
    os.environ["MFA_FWD16_IMPL"] = args.impl
    kern = AttentionKernel(desc.kernelDescriptor(AttentionKernelType.forward))
    o = torch.full((R, D), float("nan"), device="cuda")
    l = torch.zeros((R,), device="cuda")
    kern.dispatch({Op.Q: q, Op.K: k, Op.V: v, Op.O: o, Op.L: l}, row=R, column=C, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    qd, kd, vd = (x.double().cpu().numpy() for x in (q, k, v))
    s = qd @ kd.T / np.sqrt(D)
    mx = s.max(1, keepdims=True)
    p = np.exp(s - mx)
    oref = (p @ vd) / p.sum(1, keepdims=True)
    lref = (mx[:, 0] + np.log(p.sum(1))) * 1.44269504089
    do = np.abs(o.cpu().numpy() - oref).max(1)
    dl = l.cpu().numpy() - lref
    print(kern.variant)
    for w in range(0, R, 32):
        print(f"rows {w:4d}-{w + 31:4d}: max|dO| {do[w:w + 32].max():.3e}  dL mean {dl[w:w + 32].mean():+.4f} min {dl[w:w + 32].min():+.4f} max {dl[w:w + 32].max():+.4f}")
    print("L got", l[:4].cpu().numpy(), "ref", lref[:4])


if __name__ == "__main__":
    main()
