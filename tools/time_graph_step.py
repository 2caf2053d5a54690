#!/usr/bin/env python3
"""A single-head training step (forward + dQ + dK/dV, column-parallel through a workspace: 3 kernels + 4 combine passes) launched
eagerly from Python vs. captured once into a hipGraph and replayed: wall time per step over many steps.  Developer tool."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metal_flash_attention_amd import (AttentionDescriptor, AttentionKernel, AttentionKernelType as KT,
                                       AttentionOperand as Op, GEMMOperandPrecision as P)
STEPS = 200
for N, D in ((1024, 64), (4096, 64), (4096, 128), (8192, 128)):
    desc = AttentionDescriptor(); desc.lowPrecisionInputs = True; desc.lowPrecisionInputType = P.BF16
    desc.lowPrecisionIntermediates = True
    desc.matrixDimensions = (N, N, D); desc.transposeState = (False,) * 4
    prec = desc.memoryPrecisions
    tdt = {P.FP32: torch.float32, P.FP16: torch.float16, P.BF16: torch.bfloat16}
    g = torch.Generator(device="cuda"); g.manual_seed(0)
    b = {op: (torch.randn((N, D), generator=g, device="cuda") * (0.1 if op == Op.dO else 1)).to(torch.bfloat16) for op in (Op.Q, Op.K, Op.V, Op.dO)}
    for op in (Op.O, Op.dQ, Op.dK, Op.dV):
        b[op] = torch.empty((N, D), device="cuda", dtype=tdt[prec[op]])
    b[Op.L], b[Op.D] = torch.empty(N, device="cuda", dtype=tdt[prec[Op.L]]), torch.empty(N, device="cuda", dtype=tdt[prec[Op.D]])
    ks = {t: AttentionKernel(desc.kernelDescriptor(t)) for t in KT}
    ws = {t: torch.empty(max(k.workspaceSize(row=N, column=N), 256), dtype=torch.uint8, device="cuda") for t, k in ks.items()}

    def step(stream):
        for t in (KT.forward, KT.backwardQuery, KT.backwardKeyValue):
            ks[t].dispatch(b, row=N, column=N, stream=stream, workspace=ws[t])
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):
            step(side.cuda_stream)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        step(torch.cuda.current_stream().cuda_stream)
    s = torch.cuda.current_stream().cuda_stream
    res = {}
    for name, fn in (("eager", lambda: step(s)), ("graph", graph.replay)):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(STEPS):
            fn()
        torch.cuda.synchronize()
        res[name] = (time.perf_counter() - t0) / STEPS * 1e6
    dev = sum(ks[t].time(b, row=N, column=N, stream=s, warmup=2, iterations=20, workspace=ws[t]) / 20 for t in KT) * 1e3
    print(f"one head N={N:5d} D={D:3d}: eager {res['eager']:7.1f} us/step   hipGraph replay {res['graph']:7.1f} us/step   (device time of the launches alone: {dev:6.1f} us)")
