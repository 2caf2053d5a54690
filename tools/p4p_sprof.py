#!/usr/bin/env python3
"""Per-phase shader clocks of the PRODUCT schedule of the persistent forward kernel (round 6): the developer stream
BF16_FOLD_L16_SPROF is the headline stream plus s_memtime stamps into scalar accumulators (tools/p4pgen.py, pprof = 2) --
phase A | wait for this wave's LDS-DMA pieces (vmcnt) | barrier | phase B, summed over a block's loop tiles, and everything
outside the tile loop ("rest": table, Q fragments, tile 0, tail, epilogue).  Lane 0 of every wave leaves its sums in
O[first row of its last block][0:16].  Each stamp flushes the LDS queue (s_memtime returns through lgkmcnt): the stamped stream
runs a few per cent slower than the product; the SHARES are what it is read for.

  python tools/p4p_sprof.py [--N 4096 --heads 256 --fill normal|zero]     (needs make -C metal_flash_attention_amd/csrc DEV=1)
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
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--heads", type=int, default=256)
    ap.add_argument("--fill", default="normal")
    ap.add_argument("--stream", default="BF16_FOLD_L16_SPROF")
    args = ap.parse_args()
    import torch
    import p4pgen
    from metal_flash_attention_amd import (AttentionDescriptor, AttentionKernel, AttentionKernelType,
                                           AttentionOperand as Op, GEMMOperandPrecision as P)
    N, D, H = args.N, 128, args.heads
    desc = AttentionDescriptor()
    desc.lowPrecisionInputs = True
    desc.lowPrecisionIntermediates = True
    desc.lowPrecisionInputType = P.BF16
    desc.matrixDimensions = (N, N, D)
    desc.transposeState = (False,) * 4
    g = torch.Generator(device="cuda")
    g.manual_seed(0)
    if args.fill == "zero":
        bufs = {op: torch.zeros((H, N, D), device="cuda", dtype=torch.bfloat16) for op in (Op.Q, Op.K, Op.V)}
    else:
        bufs = {op: torch.randn((H, N, D), generator=g, device="cuda").to(torch.bfloat16) for op in (Op.Q, Op.K, Op.V)}
    hs = {Op.Q: N * D, Op.K: N * D, Op.V: N * D, Op.O: N * D, Op.L: N}
    k = AttentionKernel(desc.kernelDescriptor(AttentionKernelType.forward))
    o = torch.zeros((H, N, D), device="cuda")
    bufs[Op.O], bufs[Op.L] = o, torch.zeros((H, N), device="cuda", dtype=torch.float16)
    stream = torch.cuda.current_stream().cuda_stream
    kw = dict(row=N, column=N, heads=H, headStrides=hs, stream=stream)
    ms_product = k.time(bufs, warmup=5, iterations=20, **kw) / 20
    os.environ["MFA_P4P_DEV_STREAM"] = args.stream
    ms = k.time(bufs, warmup=5, iterations=20, **kw) / 20
    o.zero_()
    k.dispatch(bufs, **kw)
    torch.cuda.synchronize()
    c = o[:, ::64, 0:16].contiguous().view(torch.int32).reshape(-1, 16)
    c = c[c[:, 15] == p4pgen.PROF_MAGIC].double()      # the waves that ended a workgroup
    names = p4pgen.SPROF_NAMES
    nb = names.index("blocks")
    blocks = c[:, nb].mean().item()
    nt = (N + 63) // 64
    nt += nt & 1
    print(f"{k.variant} stream {args.stream} N={N} heads={H} fill={args.fill}: {ms:.4f} ms/launch stamped, {ms_product:.4f} ms product; "
          f"{c.shape[0]} waves reported, {blocks:.1f} blocks per workgroup, {nt - 1} loop tiles per block")
    tot = 0.0
    per = {}
    for i, name in enumerate(names[:nb]):
        per[name] = (c[:, i] / c[:, nb]).mean().item()
        tot += per[name]
    print("shader clocks per block (mean over the reporting waves) | per loop tile | share")
    for name in names[:nb]:
        v = per[name]
        tile = "%8.1f" % (v / (nt - 1)) if name.startswith("loop") else "        "
        print(f"  {name:10s} {v:10.0f}  {tile}  {100 * v / tot:5.1f} %")
    print(f"  {'sum':10s} {tot:10.0f}   (launch / blocks per workgroup = {ms * 1e3 / blocks:.1f} us; matrix time of a block = {nt * 2048} clocks)")
    # spread over the waves of the per-tile barrier time (who waits for whom)
    bar = (c[:, names.index("loop_bar")] / c[:, nb] / (nt - 1))
    vm = (c[:, names.index("loop_vm")] / c[:, nb] / (nt - 1))
    print(f"  barrier clocks per tile over the reporting waves: min {bar.min().item():.0f} median {bar.median().item():.0f} max {bar.max().item():.0f};"
          f" vmcnt wait: min {vm.min().item():.0f} median {vm.median().item():.0f} max {vm.max().item():.0f}")


if __name__ == "__main__":
    main()
