#!/usr/bin/env python3
"""Phase timing of attn_fwd16_p4 from its PROF stream (developer tool): the stream stamps the shader clock at the
end of phase A, behind the barrier and at the end of phase B of every steady-state tile and every wave leaves the
three sums in O[first row of the wave][0:4] (which this tool reads back; the O of such a run is garbage there).

  python tools/p4_prof.py [--N 4096 --heads 256]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the MFA_*_IMPL knobs exist only in the developer build of the library (make -C metal_flash_attention_amd/csrc DEV=1)
os.environ.setdefault("MFA_LIBRARY", os.path.join(ROOT, "metal_flash_attention_amd", "libmfa_hip_dev.so"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=4096)
    ap.add_argument("--heads", type=int, default=256)
    ap.add_argument("--impl", default="BF16_FOLD_PROF", help="p4:<n>, or the name of a PROF stream of tools/p4gen.py")
    ap.add_argument("--causal", action="store_true")
    args = ap.parse_args()
    if not args.impl.startswith("p4:"):
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import p4gen
        args.impl = "p4:%d" % (1000 + list(p4gen.VARIANTS).index(args.impl))
    import torch
    from metal_flash_attention_amd import (AttentionDescriptor, AttentionKernel, AttentionKernelType,
                                           AttentionOperand as Op, GEMMOperandPrecision as P)
    N, D, H = args.N, 128, args.heads
    desc = AttentionDescriptor()
    desc.lowPrecisionInputs = True
    desc.lowPrecisionInputType = P.BF16
    desc.matrixDimensions = (N, N, D)
    desc.transposeState = (False,) * 4
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    bufs = {op: torch.randn((H, N, D), generator=g, device="cuda").to(torch.bfloat16) for op in (Op.Q, Op.K, Op.V)}
    hs = {Op.Q: N * D, Op.K: N * D, Op.V: N * D, Op.O: N * D, Op.L: N}
    os.environ["MFA_FWD16_IMPL"] = args.impl
    k = AttentionKernel(desc.kernelDescriptor(AttentionKernelType.forward))
    o = torch.zeros((H, N, D), device="cuda")
    bufs[Op.O], bufs[Op.L] = o, torch.zeros((H, N), device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        k.dispatch(bufs, row=N, column=N, heads=H, headStrides=hs, stream=stream, causal=args.causal)
    torch.cuda.synchronize()
    ms = k.time(bufs, row=N, column=N, heads=H, headStrides=hs, stream=stream, warmup=3, iterations=10, causal=args.causal) / 10
    if args.causal:
        c = o[:, ::64, 0:4].contiguous().view(torch.int32).double()      # [H][N/64 waves][pa, pw, pb, nt]
        nblk = N // 256
        c = c.view(H, nblk, 4, 4).mean(0)                                 # [block][wave][4]
        print(f"{k.variant} causal: {ms:.4f} ms/launch")
        print("block  nt | per wave: steady tiles it computed (from the clock sums / mean tile), loop cycles (pa+pw+pb)")
        tile = (c[-1, 3, :3].sum() / (c[-1, 3, 3] - 1)).item()
        for b in (0, 1, nblk // 2, nblk - 2, nblk - 1):
            row = "  ".join(f"w{w}: {c[b, w, :3].sum().item():9.0f} cyc ({c[b, w, :3].sum().item() / tile:5.1f} t)" for w in range(4))
            print(f"{b:4d} {int(c[b, 3, 3].item()):4d} | {row}")
        tot_pair = [(c[i, 3, :3].sum() + c[nblk - 1 - i, 3, :3].sum()).item() for i in range(nblk // 2)]
        print("longest wave's loop cycles per pair of blocks:", " ".join(f"{x / tile:.1f}t" for x in tot_pair), f"  (tile = {tile:.0f} cycles)")
        per_cu_pairs = (nblk // 2) * H / 256
        print(f"launch time per pair on a CU: {ms * 1e-3 / per_cu_pairs * 1e6:.1f} us")
        return
    extra = o[:, ::64, 7:9].contiguous().view(torch.int32).double()
    print("kernel entry -> Q loads issued %.0f, -> Q loads returned %.0f cycles" % (extra[..., 0].mean(), extra[..., 1].mean()))
    more = o[:, ::64, 9:12].contiguous().view(torch.int32).double()
    import numpy as _np
    for nm, x in (("Q issue - descriptors", (extra[..., 0] - more[..., 2])), ("Q back - Q issue", extra[..., 1] - extra[..., 0]),
                  ("stores complete", o[:, ::64, 6].contiguous().view(torch.int32).double())):
        q = _np.percentile(x.cpu().numpy().ravel(), [0, 10, 25, 50, 75, 90, 100])
        print("  %-24s percentiles 0/10/25/50/75/90/100: %s" % (nm, " ".join("%.0f" % v for v in q)))
    print("kernel entry -> block decoded %.0f, -> lengths known %.0f, -> descriptors built %.0f cycles (min over waves %.0f / %.0f / %.0f)" % (
        more[..., 0].mean(), more[..., 1].mean(), more[..., 2].mean(), more[..., 0].min(), more[..., 1].min(), more[..., 2].min()))
    fixed = o[:, ::64, 4:7].contiguous().view(torch.int32).double()   # [H][waves][entry -> statement, statement, statement -> stores issued]
    c = o[:, ::64, 0:4].contiguous().view(torch.int32).double()      # [H][N/64 waves][pa, pw, pb, nt]
    steady = c[..., 3] - 1
    per = c[..., :3] / steady[..., None]
    print(f"{k.variant}: {ms:.4f} ms/launch, {4.0 * N * N * D * H / ms / 1e9:.1f} TF")
    print("shader-clock cycles per steady tile and wave (mean / min / max over all waves):")
    for i, name in enumerate(("phase A (32 MFMA)", "waits + barrier", "phase B (32 MFMA)")):
        print(f"  {name:20s} {per[..., i].mean():8.1f} {per[..., i].min():8.1f} {per[..., i].max():8.1f}")
    tot = per.sum(-1)
    print(f"  {'tile total':20s} {tot.mean():8.1f} {tot.min():8.1f} {tot.max():8.1f}   (64 MFMA = 2048 cycles of matrix pipe)")
    loop = (c[..., :3].sum(-1))
    print("per workgroup (mean over waves): kernel entry -> statement %.0f, statement %.0f (of which the steady tiles %.0f: prologue + first tile + "
          "drain %.0f), statement -> stores complete %.0f shader-clock cycles" % (fixed[..., 0].mean(), fixed[..., 1].mean(), loop.mean(),
          (fixed[..., 1] - loop).mean(), fixed[..., 2].mean()))
    per_wg_us = ms * 1e3 / ((N // 256) * H / 256)
    print("launch time per workgroup on a CU: %.1f us" % per_wg_us)
    tiles_per_cu = (N // 64) * (N // 256) * H / 256
    print(f"  implied clock if the loop were the whole launch: {tot.mean() * tiles_per_cu / (ms * 1e-3) / 1e9:.2f} GHz")


if __name__ == "__main__":
    main()
