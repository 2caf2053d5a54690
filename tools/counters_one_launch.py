import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MFA_LIBRARY", os.path.join(ROOT, "metal_flash_attention_amd", "libmfa_hip_dev.so"))
mode = sys.argv[1]   # dense | geom | causal
import torch
from metal_flash_attention_amd import (AttentionDescriptor, AttentionKernel, AttentionKernelType, AttentionOperand as Op, GEMMOperandPrecision as P)
N, B, H, D = 4096, 8, 32, 128
desc = AttentionDescriptor(); desc.lowPrecisionInputs = desc.lowPrecisionIntermediates = True; desc.lowPrecisionInputType = P.BF16
desc.matrixDimensions = (N, N, D); desc.transposeState = (False,) * 4
k = AttentionKernel(desc.kernelDescriptor(AttentionKernelType.forward))
g = torch.Generator(device="cuda"); g.manual_seed(0)
bufs = {op: torch.randn((B, H, N, D), generator=g, device="cuda").to(torch.bfloat16) for op in (Op.Q, Op.K, Op.V)}
bufs[Op.O] = torch.zeros((B, H, N, D), device="cuda"); bufs[Op.L] = torch.zeros((B, H, N), device="cuda", dtype=torch.float16)
hs = {Op.Q: N * D, Op.K: N * D, Op.V: N * D, Op.O: N * D, Op.L: N}; bs = {op: v * H for op, v in hs.items()}
kw = dict(row=N, column=N, heads=H, batches=B, headStrides=hs, batchStrides=bs, stream=torch.cuda.current_stream().cuda_stream)
if mode == "geom":
    os.environ["MFA_P4P_LENGTHS"] = "1"
    rl = torch.tensor([N] * B, dtype=torch.int32, device="cuda"); kw.update(rowLengths=rl, columnLengths=rl)
if mode == "causal":
    kw.update(causal=True)
print(mode, k.launchForm(bufs, **{x: y for x, y in kw.items() if x != "stream"}))
for _ in range(30):
    k.dispatch(bufs, **kw)
torch.cuda.synchronize()
