#!/usr/bin/env python3
"""Developer library only (MFA_LIBRARY=.../libmfa_hip_dev.so): the two backward kernels on transposed operands ([D][sequence]) --
what the product library does (no workspace: the general kernels; with a workspace: copy to row-major, hand-placed kernel, copy
back) against the developer kernels that read the operands in place (attn_dq16_p4_tr.h, attn_dkv16_p4_tr.h; the developer
library's choice for launches without a workspace, MFA_BWD16_TR=0 turns them off).
Same buffers, torch events around back-to-back launches on the current stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metal_flash_attention_amd import (AttentionDescriptor, AttentionKernel, AttentionKernelType as KT,
                                       AttentionOperand as Op, GEMMOperandPrecision as P)
tdt = {P.FP32: torch.float32, P.FP16: torch.float16, P.BF16: torch.bfloat16}
FLOPS = {KT.backwardQuery: 6.0, KT.backwardKeyValue: 8.0}
H = 8
for N, D, mixed in ((8192, 128, False), (8192, 128, True), (4096, 128, True)):
    desc = AttentionDescriptor(); desc.lowPrecisionInputs = True; desc.lowPrecisionInputType = P.BF16
    desc.lowPrecisionIntermediates = mixed
    desc.matrixDimensions = (N, N, D); desc.transposeState = (True,) * 4
    prec = desc.memoryPrecisions
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    b = {op: (torch.randn((H, D, N), generator=g, device="cuda") * (0.1 if op == Op.dO else 1)).to(tdt[prec[op]]) for op in (Op.Q, Op.K, Op.V, Op.dO)}
    for op in (Op.O, Op.dQ, Op.dK, Op.dV):
        b[op] = torch.empty((H, D, N), device="cuda", dtype=tdt[prec[op]])
    b[Op.L], b[Op.D] = torch.empty((H, N), device="cuda", dtype=tdt[prec[Op.L]]), torch.empty((H, N), device="cuda", dtype=tdt[prec[Op.D]])
    hs = {op: (N if op in (Op.L, Op.D) else N * D) for op in b}
    args = dict(row=N, column=N, heads=H, batches=1, headStrides=hs, batchStrides={op: v * H for op, v in hs.items()})
    s = torch.cuda.current_stream().cuda_stream
    ks = {t: AttentionKernel(desc.kernelDescriptor(t)) for t in KT}
    ks[KT.forward].dispatch(b, stream=s, **args)

    def timed(k, iters, **kw):
        for _ in range(2): k.dispatch(b, stream=s, **args, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): k.dispatch(b, stream=s, **args, **kw)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters
    for t, outs in ((KT.backwardQuery, (Op.dQ,)), (KT.backwardKeyValue, (Op.dK, Op.dV))):
        k = ks[t]
        need = k.workspaceSize(row=N, column=N, heads=H)
        ws = torch.empty(max(need, 256), dtype=torch.uint8, device="cuda")
        fl = FLOPS[t] * N * N * D * H
        res, out = {}, {}
        os.environ["MFA_BWD16_TR"] = "0"           # the developer library's A/B knob: what the product library launches
        res["general"] = timed(k, 2)
        os.environ.pop("MFA_BWD16_TR", None)
        res["workspace"] = timed(k, 10, workspace=ws)
        out["workspace"] = [b[o].float().clone() for o in outs]
        res["in place"] = timed(k, 10)
        out["in place"] = [b[o].float().clone() for o in outs]
        diff = max((x - y).abs().max().item() for x, y in zip(out["workspace"], out["in place"]))
        print(f"N={N} D={D} H={H} {'mixed' if mixed else 'fp32 '} {t.name:17s}: " +
              "   ".join(f"{n} {v:8.3f} ms ({fl/v/1e9:6.1f} TF)" for n, v in res.items()) +
              f"   workspace {need/2**20:.0f} MiB   max |d| workspace vs in place {diff:.2e}", flush=True)
