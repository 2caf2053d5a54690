"""PyTorch-ROCm binding of the three attention kernels (the "caller" layer the
reference does not have -- only its tests call the kernels, Tests/FlashAttentionTests/Attention/*.swift).

    from metal_flash_attention_amd.torch_binding import flash_attention
    o = flash_attention(q, k, v, causal=False)      # q [B, H, R, D], k / v [B, H, C, D]; bf16, fp16 or fp32
    o.sum().backward()                              # dQ, dK, dV through backwardQuery / backwardKeyValue

    from metal_flash_attention_amd.torch_binding import flash_attention_op   # the same through torch.library ops
    f = torch.compile(lambda q, k, v: flash_attention_op(q, k, v, causal=True), fullgraph=True)

forward  = AttentionKernelType.forward           -> O (the inputs' dtype, fused cast), L (fp32), both saved
backward = AttentionKernelType.backwardQuery     -> D, dQ      (needs O, dO, L)
           AttentionKernelType.backwardKeyValue  -> dK, dV     (needs L, D)
exactly the dispatch order of the reference's test (SquareAttentionTest.swift:355-368).  torch owns the
device memory and the stream; all arithmetic happens in libmfa_hip.so (no eager fallback: a missing
library or a CPU tensor raises).
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from .attention import (AttentionDescriptor, AttentionKernel, AttentionKernelType, AttentionOperand as Op,
                        GEMMOperandPrecision as P)

_KERNELS: Dict[Tuple, AttentionKernel] = {}


def _kernel(dtype: torch.dtype, R: int, C: int, D: int, kind: AttentionKernelType, fast_scale: bool = False) -> AttentionKernel:
    # a kernel object depends on (precisions, head dimension, type) only -- the sequence lengths are launch parameters
    # (mfa_launch_params.row / .column), so the cache is bounded by the handful of head dimensions a model uses
    fast_scale = bool(fast_scale) and dtype != torch.float32
    key = (dtype, D, kind, fast_scale)
    k = _KERNELS.get(key)
    if k is None:
        desc = AttentionDescriptor()
        desc.lowPrecisionInputs = dtype != torch.float32
        if dtype != torch.float32:
            desc.lowPrecisionInputType = P.BF16 if dtype == torch.bfloat16 else P.FP16
            desc.lowPrecisionOutputs = True     # O, dQ, dK, dV leave the kernels already in the inputs' type
        # fast_scale = the reference's mixed-precision mode (lowPrecisionIntermediates): the streams that fold the softmax scale
        # into the 16-bit operand (|dL| ~ 2e-3 with BF16), L stored in FP16 and D in BF16 (+Precisions.swift:82-83)
        desc.lowPrecisionIntermediates = fast_scale
        desc.matrixDimensions = (R, C, D)
        desc.transposeState = (False, False, False, False)
        k = _KERNELS[key] = AttentionKernel(desc.kernelDescriptor(kind))
    return k


def _check(q, k, v):
    if not (q.is_cuda and k.is_cuda and v.is_cuda):
        raise RuntimeError("flash_attention: tensors must live on the GPU (there is no CPU path)")
    if q.dtype not in (torch.bfloat16, torch.float16, torch.float32) or k.dtype != q.dtype or v.dtype != q.dtype:
        raise TypeError("flash_attention: q, k, v must share one of bfloat16 / float16 / float32")
    if q.dim() != 4 or k.dim() != 4 or v.shape != k.shape or q.shape[:2] != k.shape[:2] or q.shape[3] != k.shape[3]:
        raise ValueError("flash_attention: expected q [B, H, R, D] and k, v [B, H, C, D]")
    if k.device != q.device or v.device != q.device:
        raise RuntimeError(f"flash_attention: q, k, v must live on one device (got {q.device}, {k.device}, {v.device})")


def _check_block_mask(block_mask, R, C):
    """bitmap rows cover every 256-row block, words cover every 128-key block (the kernels index it unchecked)"""
    if block_mask.dim() != 2:
        raise ValueError("flash_attention: block_mask must be [ceil(R / 256)][words] (pack_block_mask)")
    rows_needed, words_needed = (R + 255) // 256, ((C + 127) // 128 + 31) // 32
    if block_mask.shape[0] < rows_needed or block_mask.shape[1] < words_needed:
        raise ValueError(f"flash_attention: block_mask is {tuple(block_mask.shape)}, needs at least "
                         f"({rows_needed}, {words_needed}) for R={R}, C={C}")


def _strides(B, H, R, C, D):
    hs = {Op.Q: R * D, Op.K: C * D, Op.V: C * D, Op.O: R * D, Op.L: R, Op.D: R,
          Op.dO: R * D, Op.dV: C * D, Op.dK: C * D, Op.dQ: R * D}
    return hs, {op: s * H for op, s in hs.items()}


def _as_strided_operand(t):
    """A [B, H, N, D] VIEW whose last dimension is contiguous (e.g. a permuted [B, N, H, D] tensor, a slice of a fused QKV
    projection) is handed to the kernels as it is: the C ABI takes a leading dimension and head / batch strides per operand
    (mfa_launch_params), so no copy is made.  Anything else is made contiguous first."""
    if t.stride(-1) != 1 or any(st < 0 for st in t.stride()) or (t.shape[2] > 1 and t.stride(2) < t.shape[3]):
        t = t.contiguous()
    return t, int(t.stride(2)) if t.shape[2] > 1 else int(t.shape[3]), int(t.stride(1)), int(t.stride(0))


def _apply_layouts(hs, bs, lds, **operands):
    """overwrite the packed strides of the given operands (name -> tensor) with their real ones"""
    out = {}
    for name, t in operands.items():
        t, ld, head, batch = _as_strided_operand(t)
        op = getattr(Op, name)
        lds[op], hs[op], bs[op] = ld, head, batch
        out[name] = t
    return out


def _run_forward(q, k, v, causal, q_lengths, k_lengths, block_mask, fast_scale):
    """forward dispatch -> (o, l, the (possibly strided) views handed to the kernel, lengths, mask arguments, fast_scale)"""
    _check(q, k, v)
    B, H, R, D = q.shape
    C = k.shape[2]
    hs, bs = _strides(B, H, R, C, D)
    lds = {}
    views = _apply_layouts(hs, bs, lds, Q=q, K=k, V=v)
    q, k, v = views["Q"], views["K"], views["V"]
    o = torch.empty((B, H, R, D), dtype=q.dtype, device=q.device)      # fused output cast: no fp32 copy of O
    fast_scale = bool(fast_scale) and q.dtype != torch.float32
    l = torch.empty((B, H, R), dtype=torch.float16 if fast_scale else torch.float32, device=q.device)
    kernel = _kernel(q.dtype, R, C, D, AttentionKernelType.forward, fast_scale)
    need = kernel.workspaceSize(row=R, column=C, heads=H, batches=B)
    lengths = q_lengths is not None or k_lengths is not None
    mask_kw = {}
    if block_mask is not None:   # int32 [ceil(R / 256)][words]: bit b of word w = column block 32 w + b (128 keys each)
        _check_block_mask(block_mask, R, C)
        block_mask = block_mask.to(device=q.device, dtype=torch.int32).contiguous()
        mask_kw = dict(blockMask=block_mask, blockMaskWords=int(block_mask.shape[-1]))
    if lengths:   # padding rows of the outputs are not written by the kernels: define them as zero
        o.zero_()
        l.zero_()
        q_lengths = None if q_lengths is None else q_lengths.to(device=q.device, dtype=torch.int32).contiguous()
        k_lengths = None if k_lengths is None else k_lengths.to(device=q.device, dtype=torch.int32).contiguous()
    ws = torch.empty(need, dtype=torch.uint8, device=q.device) if need and not causal and not lengths and not mask_kw else None
    # the C side launches on the CURRENT device (hipGetDevice) and this stream: make both the tensors' device
    with torch.cuda.device(q.device):
        kernel.dispatch({Op.Q: q, Op.K: k, Op.V: v, Op.O: o, Op.L: l}, row=R, column=C, heads=H, batches=B,
                        headStrides=hs, batchStrides=bs, leadingDimensions=lds,
                        stream=torch.cuda.current_stream(q.device).cuda_stream, workspace=ws, causal=causal, rowLengths=q_lengths, columnLengths=k_lengths, **mask_kw)
    return o, l, (q, k, v), (q_lengths, k_lengths), mask_kw, fast_scale


def _run_backward(q, k, v, o, l, grad_out, causal, lengths, mask_kw, fast_scale):
    """backwardQuery (writes D, dQ) then backwardKeyValue (dK, dV), the dispatch order of SquareAttentionTest.swift:355-368"""
    B, H, R, D = q.shape
    C = k.shape[2]
    # dO in the kernels' gradient storage type (AttentionDescriptor+Precisions.swift:13-17): BF16 whenever
    # the inputs are 16-bit (also next to FP16 Q/K/V, the reference's mix), FP32 with FP32 inputs
    do = grad_out.to(torch.float32 if q.dtype == torch.float32 else torch.bfloat16)   # no copy when it already is
    alloc = torch.zeros if lengths != (None, None) else torch.empty   # padding gets zero gradients
    dq = alloc((B, H, R, D), dtype=q.dtype, device=q.device)
    dk = alloc((B, H, C, D), dtype=q.dtype, device=q.device)
    dv = alloc((B, H, C, D), dtype=q.dtype, device=q.device)
    dterm = alloc((B, H, R), dtype=torch.bfloat16 if fast_scale else torch.float32, device=q.device)
    bufs = {Op.Q: q, Op.K: k, Op.V: v, Op.O: o, Op.L: l, Op.D: dterm, Op.dO: do, Op.dQ: dq, Op.dK: dk, Op.dV: dv}
    hs, bs = _strides(B, H, R, C, D)
    lds = {}
    # saved as the (possibly strided) views the forward used; a strided grad_out (e.g. the gradient of a permuted view)
    # is passed with its own leading dimension / head / batch strides instead of being copied
    # every operand is the tensor its strides describe: an input that had to be made contiguous (torch.library path: the raw inputs
    # are what was saved) must be passed as that copy, not as the original pointer
    views = _apply_layouts(hs, bs, lds, Q=q, K=k, V=v, dO=do)
    bufs[Op.Q], bufs[Op.K], bufs[Op.V], bufs[Op.dO] = views["Q"], views["K"], views["V"], views["dO"]
    with torch.cuda.device(q.device):
        stream = torch.cuda.current_stream(q.device).cuda_stream
        for kind in (AttentionKernelType.backwardQuery, AttentionKernelType.backwardKeyValue):   # dQ writes D first
            _kernel(q.dtype, R, C, D, kind, fast_scale).dispatch(bufs, row=R, column=C, heads=H, batches=B, headStrides=hs,
                                                                  batchStrides=bs, leadingDimensions=lds, stream=stream, causal=causal,
                                                                  rowLengths=lengths[0], columnLengths=lengths[1], **mask_kw)
    return dq, dk, dv


class _FlashAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, causal: bool, q_lengths=None, k_lengths=None, block_mask=None, fast_scale=False):
        o, l, (q, k, v), lengths, mask_kw, fast_scale = _run_forward(q, k, v, causal, q_lengths, k_lengths, block_mask, fast_scale)
        ctx.save_for_backward(q, k, v, o, l)
        ctx.causal = causal
        ctx.fast_scale = fast_scale
        ctx.lengths = lengths
        ctx.mask_kw = mask_kw
        return o

    @staticmethod
    def backward(ctx, grad_out):
        q, k, v, o, l = ctx.saved_tensors
        dq, dk, dv = _run_backward(q, k, v, o, l, grad_out, ctx.causal, ctx.lengths, ctx.mask_kw, ctx.fast_scale)
        return dq, dk, dv, None, None, None, None, None


# ---- the same two steps as torch.library custom ops: opaque to the tracer but with shape functions and an autograd formula, so a
# function that calls flash_attention_op compiles with torch.compile(fullgraph=True) (the autograd.Function above is a graph break).
# Dense / causal only; per-batch lengths and block masks stay on flash_attention().
def _register_ops():
    """defines mfa::attention_forward / mfa::attention_backward once per process (a module reload finds them already defined);
    torch < 2.4 has no torch.library.custom_op: flash_attention() keeps working there, flash_attention_op() raises."""
    if not hasattr(torch.library, "custom_op"):
        return False
    try:
        torch.ops.mfa.attention_forward  # noqa: B018 -- AttributeError when the op is not defined yet
        torch.ops.mfa.attention_backward  # noqa: B018
        return True
    except (AttributeError, RuntimeError):
        pass

    @torch.library.custom_op("mfa::attention_forward", mutates_args=(), device_types="cuda")
    def _op_forward(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, causal: bool, fast_scale: bool) -> Tuple[torch.Tensor, torch.Tensor]:
        o, l, _views, _lengths, _mask, _fast = _run_forward(q, k, v, causal, None, None, None, fast_scale)
        return o, l


    @_op_forward.register_fake
    def _op_forward_fake(q, k, v, causal, fast_scale):
        B, H, R, D = q.shape
        fast = bool(fast_scale) and q.dtype != torch.float32
        return q.new_empty((B, H, R, D)), q.new_empty((B, H, R), dtype=torch.float16 if fast else torch.float32)


    @torch.library.custom_op("mfa::attention_backward", mutates_args=(), device_types="cuda")
    def _op_backward(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, o: torch.Tensor, l: torch.Tensor, grad_out: torch.Tensor,
                     causal: bool, fast_scale: bool) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        _check(q, k, v)
        fast = bool(fast_scale) and q.dtype != torch.float32
        return _run_backward(q, k, v, o, l, grad_out, causal, (None, None), {}, fast)


    @_op_backward.register_fake
    def _op_backward_fake(q, k, v, o, l, grad_out, causal, fast_scale):
        return torch.empty_like(q, memory_format=torch.contiguous_format), torch.empty_like(k, memory_format=torch.contiguous_format), \
            torch.empty_like(v, memory_format=torch.contiguous_format)


    def _op_setup_context(ctx, inputs, output):
        q, k, v, causal, fast_scale = inputs
        o, l = output
        ctx.save_for_backward(q, k, v, o, l)
        ctx.causal, ctx.fast_scale = causal, fast_scale


    def _op_autograd(ctx, grad_o, grad_l):
        q, k, v, o, l = ctx.saved_tensors
        dq, dk, dv = torch.ops.mfa.attention_backward(q, k, v, o, l, grad_o, ctx.causal, ctx.fast_scale)
        return dq, dk, dv, None, None


    _op_forward.register_autograd(_op_autograd, setup_context=_op_setup_context)
    return True


_HAVE_OPS = _register_ops()


def flash_attention_op(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, causal: bool = False, fast_scale: bool = False) -> torch.Tensor:
    """flash_attention(q, k, v, causal, fast_scale=...) through the torch.library ops `mfa::attention_forward` /
    `mfa::attention_backward`: traceable by torch.compile (fullgraph) and torch.export; dense or causal."""
    if not _HAVE_OPS:
        raise RuntimeError("flash_attention_op needs torch.library.custom_op (torch >= 2.4); use flash_attention() on this torch")
    return torch.ops.mfa.attention_forward(q, k, v, causal, fast_scale)[0]


def pack_block_mask(bits: torch.Tensor) -> torch.Tensor:
    """bool [row blocks of 256][column blocks of 128] -> the int32 bitmap the kernels read."""
    rb, cb = bits.shape
    words = (cb + 31) // 32
    padded = torch.zeros((rb, words * 32), dtype=torch.int64, device=bits.device)
    padded[:, :cb] = bits.to(torch.int64)
    weights = (1 << torch.arange(32, dtype=torch.int64, device=bits.device))
    packed = (padded.view(rb, words, 32) * weights).sum(-1)
    return torch.where(packed >= 2 ** 31, packed - 2 ** 32, packed).to(torch.int32)


def flash_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, causal: bool = False,
                    q_lengths: torch.Tensor = None, k_lengths: torch.Tensor = None,
                    block_mask: torch.Tensor = None, fast_scale: bool = False) -> torch.Tensor:
    """softmax(q k^T / sqrt(D)) v per (batch, head); causal: row r sees column c iff c <= r + (C - R).
    q_lengths / k_lengths ([B] integers, optional): batch entry b uses only its first q_lengths[b] rows and
    k_lengths[b] keys (padded batches); padding rows of the output and of the gradients are zero.
    block_mask (optional, from pack_block_mask): blocks of 256 rows x 128 keys that are attended at all.
    fast_scale (16-bit inputs only): the reference's mixed-precision mode (lowPrecisionIntermediates) -- the softmax scale is
    folded into the 16-bit operand once instead of being applied in fp32 per score (2-4 % faster; |dL| ~ 2e-3 with bf16),
    L is kept in FP16 and D in BF16 between forward and backward."""
    return _FlashAttention.apply(q, k, v, causal, q_lengths, k_lengths, block_mask, fast_scale)
