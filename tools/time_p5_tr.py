#!/usr/bin/env python3
"""Developer library only (MFA_LIBRARY=.../libmfa_hip_dev.so): forward on K^T / V^T in place at the head-dimension buckets
160 / 192 / 256 -- the 8 x 32 kernel's transposed code object (what the product library launches) against the hand-placed
stream attn_fwd16_p5_tr (the developer library's choice; MFA_FWD16_P5_TR=0 turns it off), same buffers, torch events around back-to-back launches on the current stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from metal_flash_attention_amd import (AttentionDescriptor, AttentionKernel, AttentionKernelType as KT,
                                       AttentionOperand as Op, GEMMOperandPrecision as P)
H, ITER = (int(sys.argv[1]) if len(sys.argv) > 1 else 8), 10
CAUSAL = "--causal" in sys.argv
tdt = {P.FP32: torch.float32, P.FP16: torch.float16, P.BF16: torch.bfloat16}
for N, D in ((8192, 256), (8192, 192), (8192, 160)):
    for mixed in (False, True):
        desc = AttentionDescriptor(); desc.lowPrecisionInputs = True; desc.lowPrecisionInputType = P.BF16
        desc.lowPrecisionIntermediates = mixed
        desc.matrixDimensions = (N, N, D); desc.transposeState = (False, True, True, False)
        k = AttentionKernel(desc.kernelDescriptor(KT.forward))
        g = torch.Generator(device="cuda"); g.manual_seed(0)
        b = {Op.Q: torch.randn((H, N, D), generator=g, device="cuda").to(torch.bfloat16),
             Op.K: torch.randn((H, D, N), generator=g, device="cuda").to(torch.bfloat16),
             Op.V: torch.randn((H, D, N), generator=g, device="cuda").to(torch.bfloat16),
             Op.O: torch.empty((H, N, D), device="cuda", dtype=tdt[desc.memoryPrecisions[Op.O]]),
             Op.L: torch.empty((H, N), device="cuda", dtype=tdt[desc.memoryPrecisions[Op.L]])}
        hs = {Op.Q: N * D, Op.K: N * D, Op.V: N * D, Op.O: N * D, Op.L: N}
        args = dict(row=N, column=N, heads=H, batches=1, causal=CAUSAL, headStrides=hs, batchStrides={op: v * H for op, v in hs.items()})
        s = torch.cuda.current_stream().cuda_stream
        out, res = {}, {}
        for name in ("8x32", "stream"):
            if name == "stream": os.environ.pop("MFA_FWD16_P5_TR", None)
            else: os.environ["MFA_FWD16_P5_TR"] = "0"    # the developer library's A/B knob: keep the 8 x 32 code object
            for _ in range(2): k.dispatch(b, stream=s, **args)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(ITER): k.dispatch(b, stream=s, **args)
            e1.record(); torch.cuda.synchronize()
            res[name] = e0.elapsed_time(e1) / ITER
            out[name] = b[Op.O].float().clone()
        fl = 4.0 * N * N * D * H * (0.5 if CAUSAL else 1.0)
        diff = (out["8x32"] - out["stream"]).abs().max().item()
        print(f"N={N} D={D} H={H}{' causal' if CAUSAL else ''} {'mixed' if mixed else 'fp32 '}: 8 x 32 TR {res['8x32']:7.3f} ms ({fl/res['8x32']/1e9:6.1f} TF)   "
              f"p5_tr stream {res['stream']:7.3f} ms ({fl/res['stream']/1e9:6.1f} TF)   max |dO| between them {diff:.2e}", flush=True)
