#!/usr/bin/env python3
"""Developer tool: device time of one-head launches through a workspace as a function of the number of pieces (MFA_SPLITS in the
developer library overrides the heuristic's count; `choose_splits` of mfa_kernel.hip still caps it at 256 traversal elements per piece).
One process per count (the knob is read when a launch is planned, but workspaceSize and dispatch must agree)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import torch
    from metal_flash_attention_amd import (AttentionDescriptor, AttentionKernel, AttentionKernelType as KT,
                                           AttentionOperand as Op, GEMMOperandPrecision as P)
    FLOPS = {KT.forward: 4.0, KT.backwardQuery: 6.0, KT.backwardKeyValue: 8.0}
    shapes = ((4096, 64), (4096, 128), (8192, 128), (16384, 128), (8192, 256), (16384, 256))
    if os.environ.get("SWEEP_SHAPES"):   # "N:D,N:D"
        shapes = tuple(tuple(int(x) for x in sh.split(":")) for sh in os.environ["SWEEP_SHAPES"].split(","))
    for N, D in shapes:
        desc = AttentionDescriptor(); desc.lowPrecisionInputs = True; desc.lowPrecisionInputType = P.BF16
        desc.lowPrecisionIntermediates = True
        desc.matrixDimensions = (N, N, D); desc.transposeState = (False,) * 4
        prec = desc.memoryPrecisions
        g = torch.Generator(device="cuda"); g.manual_seed(0)
        b = {op: (torch.randn((N, D), generator=g, device="cuda") * (0.1 if op == Op.dO else 1)).to(torch.bfloat16) for op in (Op.Q, Op.K, Op.V, Op.dO)}
        def out(op, shape):
            p = prec[op]
            return torch.empty(shape, device="cuda", dtype=torch.float32 if p == P.FP32 else torch.float16 if p == P.FP16 else torch.bfloat16)
        for op in (Op.O, Op.dQ, Op.dK, Op.dV):
            b[op] = out(op, (N, D))
        b[Op.L], b[Op.D] = out(Op.L, (N,)), out(Op.D, (N,))
        s = torch.cuda.current_stream().cuda_stream
        line = f"N={N:6d} D={D:4d} MFA_SPLITS={os.environ.get('MFA_SPLITS', '-'):>3s}"
        for t in (KT.forward, KT.backwardQuery, KT.backwardKeyValue):
            k = AttentionKernel(desc.kernelDescriptor(t))
            need = k.workspaceSize(row=N, column=N)
            ws = torch.empty(max(need, 16), dtype=torch.uint8, device="cuda")
            k.dispatch(b, row=N, column=N, stream=s)
            t1 = min(k.time(b, row=N, column=N, stream=s, warmup=3, iterations=20, workspace=ws) for _ in range(4)) / 20
            form = k.launchForm(b, row=N, column=N, workspace=ws)
            x = form.split("column-parallel x")[1].split()[0] if "column-parallel x" in form else "1"
            line += f"   {t.name[:9]:9s} x{x:>2s} {t1*1e3:7.1f} us {FLOPS[t]*N*N*D/t1/1e9/2500:5.3f}"
        print(line, flush=True)
else:
    env = dict(os.environ, MFA_LIBRARY=os.path.join(ROOT, "metal_flash_attention_amd", "libmfa_hip_dev.so"))
    for sp in (None,) + tuple(int(x) for x in os.environ.get("SWEEP_COUNTS", "2,4,8,16,32,64").split(",") if x):
        e = dict(env)
        if sp:
            e["MFA_SPLITS"] = str(sp)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=e)
