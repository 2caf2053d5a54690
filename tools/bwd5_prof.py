#!/usr/bin/env python3
"""Phase timing of attn_dq16_p5 / attn_dkv16_p5 from their PROF streams (developer tool, libmfa_hip_dev.so): the streams stamp the
shader clock three times per full iteration and every wave leaves the sums in the first words of its first output row (dQ / dV /
dK, which are garbage there in such a run):  pa = from behind the barrier to the end of phase A, pb = phase B up to the seam,
pc = the seam's waits + barrier.  Waves 0, 1 of a workgroup: S-role / V-role; waves 2, 3: P-role / K-role.

  python tools/bwd5_prof.py [--D 256|160] [--N 4096 --heads 64]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MFA_LIBRARY", os.path.join(ROOT, "metal_flash_attention_amd", "libmfa_hip_dev.so"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--D", type=int, default=256)
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--heads", type=int, default=64)
    ap.add_argument("--fill", default="normal", choices=("normal", "zero"))
    args = ap.parse_args()
    import numpy as np
    import torch
    from metal_flash_attention_amd import (AttentionDescriptor, AttentionKernel, AttentionKernelType as T,
                                           AttentionOperand as Op, GEMMOperandPrecision as P)
    N, D, H = args.N, args.D, args.heads
    desc = AttentionDescriptor()
    desc.lowPrecisionInputs = desc.lowPrecisionIntermediates = True
    desc.lowPrecisionInputType = P.BF16
    desc.matrixDimensions = (N, N, D)
    desc.transposeState = (False,) * 4
    g = torch.Generator(device="cuda")
    g.manual_seed(0)
    if args.fill == "zero":
        bufs = {op: torch.zeros((H, N, D), device="cuda", dtype=torch.bfloat16) for op in (Op.Q, Op.K, Op.V, Op.dO)}
    else:
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
    torch.cuda.synchronize()
    nblocks = N // 32
    for kind, kt, outs in (("dq", T.backwardQuery, (Op.dQ,)), ("dkv", T.backwardKeyValue, (Op.dV, Op.dK))):
        os.environ.pop("MFA_BWD5_PROF", None)
        k0 = AttentionKernel(desc.kernelDescriptor(kt))
        ms0 = k0.time(bufs, row=N, column=N, heads=H, headStrides=hs, stream=stream, warmup=3, iterations=10) / 10
        os.environ["MFA_BWD5_PROF"] = "1"
        k = AttentionKernel(desc.kernelDescriptor(kt))
        ms = k.time(bufs, row=N, column=N, heads=H, headStrides=hs, stream=stream, warmup=3, iterations=10) / 10
        torch.cuda.synchronize()
        print(f"{k0.variant}: {ms0:.3f} ms    {k.variant}: {ms:.3f} ms   ({nblocks} blocks of 32 per workgroup)")
        # rows 0 (pair 0) and 64 (pair 1) of every 128-row / 128-key workgroup hold the first role's sums; + 32 (dq: rows of the P-role
        # wave) -- dkv: the K-role wave writes dK, the V-role wave dV, both at the pair's first key
        for role in (0, 1):
            if kind == "dq":
                rows = bufs[Op.dQ].view(torch.int32).view(H, N, D)[:, 32 * role::64, :4]
            else:
                rows = bufs[outs[role]].view(torch.int32).view(H, N, D)[:, 0::64, :4]
            x = rows.reshape(-1, 4).cpu().numpy().astype(np.float64)
            n = x[:, 3].mean()
            per = x[:, :3].mean(axis=0) / max(n - (3 if kind == "dq" else 2), 1)     # full iterations per traversal
            name = {("dq", 0): "S-role", ("dq", 1): "P-role", ("dkv", 0): "V-role", ("dkv", 1): "K-role"}[(kind, role)]
            print(f"   {name}: per full iteration  tail+phase A {per[0]:7.0f}   phase B {per[1]:7.0f}   seam wait+barrier {per[2]:7.0f}   "
                  f"sum {per.sum():7.0f} shader clocks   (blocks {n:.0f})")
            if kind == "dq":
                words = bufs[Op.dQ].view(torch.int32).view(H, N, D)[:, 32 * role::128, :16].reshape(-1, 16).cpu().numpy().astype(np.int64) & 0xFFFFFFFF
                timeline(words)


def timeline(rows):
    """rows: int64 array [waves][11] of the dQ PROF words (one wave per row).  Per compute unit (XCC_ID, HW_ID's se / sh / cu
    fields) the workgroups in start order: prologue, traversal, epilogue, and the gap from a workgroup's end to the next one's
    start on the same compute unit, in microseconds of the 100 MHz wall clock."""
    import numpy as np
    t_in = rows[:, 4] + (rows[:, 5] << 32)
    pro, trav, epi = rows[:, 6], rows[:, 7] - rows[:, 6], rows[:, 8] - rows[:, 7]
    cu = ((rows[:, 10] & 0xF) << 8) | ((rows[:, 9] >> 8) & 0xFF)
    gaps, per_cu = [], []
    for c in np.unique(cu):
        idx = np.nonzero(cu == c)[0]
        idx = idx[np.argsort(t_in[idx])]
        per_cu.append(len(idx))
        for a, b in zip(idx[:-1], idx[1:]):
            gaps.append(t_in[b] - (t_in[a] + rows[a, 8]))
    gaps = np.array(gaps, dtype=np.float64)
    span = (t_in + rows[:, 8]).max() - t_in.min()
    print("   timeline (one wave of this role per workgroup, 10 ns ticks -> us): %d compute units, %.1f workgroups each, launch span %.1f us" % (
        len(per_cu), np.mean(per_cu), span / 100.0))
    print("   entry -> arguments decoded %.2f us -> fragments cached %.2f us -> statement %.2f us;   statement end -> workgroup barrier %.2f us" % (
        rows[:, 11].mean() / 100.0, (rows[:, 12] - rows[:, 11]).mean() / 100.0, (rows[:, 6] - rows[:, 12]).mean() / 100.0,
        (rows[:, 13] - rows[:, 7]).mean() / 100.0))
    print("   first batch of loads (S-role: its only one): issued at %.2f us, all data at %.2f us after entry" % (
        rows[:, 14].mean() / 100.0, rows[:, 15].mean() / 100.0))
    print("   prologue %.2f us   traversal %.2f us   epilogue + stores %.2f us   end -> next start on the same CU: mean %.2f us, "
          "median %.2f us, p90 %.2f us" % (pro.mean() / 100.0, trav.mean() / 100.0, epi.mean() / 100.0, gaps.mean() / 100.0,
                                          np.median(gaps) / 100.0, np.percentile(gaps, 90) / 100.0))


if __name__ == "__main__":
    main()
