"""PyTorch autograd binding (metal_flash_attention_amd/torch_binding.py): forward and gradients against a
plain fp32 torch implementation of the same function on the same (rounded) inputs."""
import math

import pytest

torch = pytest.importorskip("torch")


def reference(q, k, v, causal):
    q, k, v = (t.detach().float().requires_grad_(True) for t in (q, k, v))
    s = q @ k.transpose(-1, -2) / math.sqrt(q.shape[-1])
    if causal:
        R, C = s.shape[-2:]
        mask = torch.arange(C, device=s.device)[None, :] <= torch.arange(R, device=s.device)[:, None] + (C - R)
        s = s.masked_fill(~mask, float("-inf"))
    o = torch.softmax(s, dim=-1) @ v
    return q, k, v, o


def test_cpu_tensors_are_refused():
    from metal_flash_attention_amd.torch_binding import flash_attention
    q = torch.zeros(1, 1, 8, 16)
    with pytest.raises(RuntimeError, match="GPU"):
        flash_attention(q, q, q)


def test_library_ops_have_shape_functions_without_a_gpu():
    """mfa::attention_forward / mfa::attention_backward (torch.library): meta tensors flow through the registered shape functions
    -- what torch.compile / torch.export need to trace a graph that calls the kernels"""
    from metal_flash_attention_amd import torch_binding as tb  # noqa: F401  (registers the ops)
    q = torch.empty(2, 3, 64, 32, device="meta", dtype=torch.bfloat16)
    k = torch.empty(2, 3, 80, 32, device="meta", dtype=torch.bfloat16)
    o, l = torch.ops.mfa.attention_forward(q, k, k, False, True)
    assert o.shape == q.shape and o.dtype == torch.bfloat16 and l.shape == (2, 3, 64) and l.dtype == torch.float16
    o, l = torch.ops.mfa.attention_forward(q.float(), k.float(), k.float(), True, True)      # fast_scale is a 16-bit notion
    assert o.dtype == torch.float32 and l.dtype == torch.float32
    dq, dk, dv = torch.ops.mfa.attention_backward(q, k, k, q, l, q, False, False)
    assert dq.shape == q.shape and dk.shape == k.shape and dv.shape == k.shape


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_library_op_compiles_fullgraph_and_matches(dtype):
    """flash_attention_op inside torch.compile(fullgraph=True): no graph break (the autograd.Function form is one), same forward and
    gradients as the plain fp32 torch reference; torch.library.opcheck validates schema, fake implementation and autograd wiring"""
    from metal_flash_attention_amd import torch_binding as tb
    B, H, R, C, D = 2, 2, 200, 264, 128
    g = torch.Generator(device="cuda").manual_seed(21)
    q, k, v = (torch.randn(B, H, n, D, generator=g, device="cuda").to(dtype).requires_grad_(True) for n in (R, C, C))

    def f(q, k, v):
        return tb.flash_attention_op(q * 1.0, k, v, causal=True).float().square().sum()

    loss = torch.compile(f, backend="aot_eager", fullgraph=True)(q, k, v)
    loss.backward()
    qr, kr, vr, orf = reference(q, k, v, True)
    ref_loss = orf.square().sum()
    ref_loss.backward()
    tol = 2e-4 if dtype == torch.float32 else 6e-2
    assert abs(loss.item() - ref_loss.item()) < tol * max(1.0, abs(ref_loss.item()))
    for got, ref, name in ((q.grad, qr.grad, "dQ"), (k.grad, kr.grad, "dK"), (v.grad, vr.grad, "dV")):
        assert (got.float() - ref).abs().max().item() < (5e-4 if dtype == torch.float32 else 8e-2), name
    torch.library.opcheck(torch.ops.mfa.attention_forward, (q.detach(), k.detach(), v.detach(), False, False),
                          test_utils=("test_schema", "test_faketensor"))


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_library_op_backward_uses_the_laid_out_copies_of_non_contiguous_inputs(dtype):
    """inputs whose LAST dimension is not contiguous (x.transpose(-1, -2)) have to be copied; the torch.library path saves the raw
    inputs, so the backward op must hand the kernels the copy its strides describe, not the original pointer"""
    from metal_flash_attention_amd import torch_binding as tb
    B, H, R, C, D = 1, 2, 136, 200, 64
    g = torch.Generator(device="cuda").manual_seed(5)
    qt, kt, vt = (torch.randn(B, H, D, n, generator=g, device="cuda").to(dtype).requires_grad_(True) for n in (R, C, C))

    def f(qt, kt, vt):
        return tb.flash_attention_op(qt.transpose(-1, -2), kt.transpose(-1, -2), vt.transpose(-1, -2), causal=False).float().square().sum()

    loss = torch.compile(f, backend="aot_eager", fullgraph=True)(qt, kt, vt)
    loss.backward()
    qr, kr, vr, orf = reference(qt.transpose(-1, -2), kt.transpose(-1, -2), vt.transpose(-1, -2), False)
    orf.square().sum().backward()
    for got, ref, name in ((qt.grad, qr.grad, "dQ"), (kt.grad, kr.grad, "dK"), (vt.grad, vr.grad, "dV")):
        assert (got.float().transpose(-1, -2) - ref).abs().max().item() < (5e-4 if dtype == torch.float32 else 8e-2), name


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.bfloat16, 3e-2), (torch.float16, 3e-2)])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("shape", [(2, 3, 200, 200, 64), (1, 2, 129, 300, 128), (1, 1, 64, 64, 40), (1, 2, 150, 260, 320), (2, 1, 200, 200, 384)])
def test_forward_and_gradients_match_torch(shape, causal, dtype, tol):
    from metal_flash_attention_amd.torch_binding import flash_attention
    B, H, R, C, D = shape
    g = torch.Generator(device="cuda").manual_seed(R + C)
    q = torch.randn(B, H, R, D, generator=g, device="cuda").to(dtype).requires_grad_(True)
    k = torch.randn(B, H, C, D, generator=g, device="cuda").to(dtype).requires_grad_(True)
    v = torch.randn(B, H, C, D, generator=g, device="cuda").to(dtype).requires_grad_(True)
    w = torch.randn(B, H, R, D, generator=g, device="cuda").to(dtype)          # upstream gradient
    o = flash_attention(q, k, v, causal=causal)
    assert o.dtype == dtype and o.shape == (B, H, R, D)
    o.backward(w)
    qr, kr, vr, orf = reference(q, k, v, causal)
    orf.backward(w.float())
    assert (o.float() - orf).abs().max().item() < tol
    for got, ref, name in ((q.grad, qr.grad, "dQ"), (k.grad, kr.grad, "dK"), (v.grad, vr.grad, "dV")):
        assert got.dtype == dtype
        assert (got.float() - ref).abs().max().item() < max(tol, 5e-2 if dtype != torch.float32 else 0), name


@pytest.mark.gpu
def test_long_single_head_uses_split_kv_workspace():
    from metal_flash_attention_amd.torch_binding import flash_attention
    g = torch.Generator(device="cuda").manual_seed(1)
    q = torch.randn(1, 1, 256, 64, generator=g, device="cuda").bfloat16()
    k = torch.randn(1, 1, 8192, 64, generator=g, device="cuda").bfloat16()
    v = torch.randn(1, 1, 8192, 64, generator=g, device="cuda").bfloat16()
    o = flash_attention(q, k, v)
    _, _, _, ref = reference(q, k, v, False)
    assert (o.float() - ref).abs().max().item() < 2e-2


@pytest.mark.gpu
@pytest.mark.parametrize("causal", [False, True])
def test_padded_batch_with_lengths(causal):
    from metal_flash_attention_amd.torch_binding import flash_attention
    B, H, N, D = 3, 2, 160, 64
    lens = torch.tensor([160, 40, 97])
    g = torch.Generator(device="cuda").manual_seed(9)
    q, k, v = (torch.randn(B, H, N, D, generator=g, device="cuda").bfloat16().requires_grad_(True) for _ in range(3))
    w = torch.randn(B, H, N, D, generator=g, device="cuda").bfloat16()
    o = flash_attention(q, k, v, causal=causal, q_lengths=lens, k_lengths=lens)
    o.backward(w)
    for b, n in enumerate(lens.tolist()):
        qr, kr, vr, orf = reference(q[b:b + 1, :, :n], k[b:b + 1, :, :n], v[b:b + 1, :, :n], causal)
        orf.backward(w[b:b + 1, :, :n].float())
        assert (o[b, :, :n].float() - orf[0]).abs().max().item() < 3e-2
        assert (o[b, :, n:] == 0).all() and (q.grad[b, :, n:] == 0).all() and (k.grad[b, :, n:] == 0).all()
        for got, ref in ((q.grad, qr.grad), (k.grad, kr.grad), (v.grad, vr.grad)):
            assert (got[b, :, :n].float() - ref[0]).abs().max().item() < 5e-2


@pytest.mark.gpu
def test_block_mask_through_the_binding():
    from metal_flash_attention_amd.torch_binding import flash_attention, pack_block_mask
    B, H, N, D = 1, 2, 1024, 64
    g = torch.Generator(device="cuda").manual_seed(4)
    q, k, v = (torch.randn(B, H, N, D, generator=g, device="cuda").bfloat16().requires_grad_(True) for _ in range(3))
    bits = torch.rand(N // 256, N // 128, generator=g, device="cuda") < 0.5
    bits[:, 0] = True                                  # every row block sees something
    o = flash_attention(q, k, v, block_mask=pack_block_mask(bits))
    o.float().sum().backward()
    dense = bits.repeat_interleave(256, 0).repeat_interleave(128, 1)
    qr, kr, vr = (t.detach().float().requires_grad_(True) for t in (q, k, v))
    s = (qr @ kr.transpose(-1, -2) / math.sqrt(D)).masked_fill(~dense, float("-inf"))
    ref = torch.softmax(s, dim=-1) @ vr
    ref.sum().backward()
    assert (o.float() - ref).abs().max().item() < 3e-2
    for got, want in ((q.grad, qr.grad), (k.grad, kr.grad), (v.grad, vr.grad)):
        assert (got.float() - want).abs().max().item() < 6e-2


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_strided_views_are_passed_without_a_copy(dtype, monkeypatch):
    """[B, N, H, D] storage viewed as [B, H, N, D] and slices of one fused QKV tensor: the binding hands the kernels the real
    leading dimension / head / batch strides instead of calling .contiguous()"""
    from metal_flash_attention_amd import torch_binding as tb
    B, H, N, D = 2, 4, 200, 64
    g = torch.Generator(device="cuda").manual_seed(3)
    qkv = torch.randn(B, N, 3, H, D, generator=g, device="cuda").to(dtype)          # fused projection output
    q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3).requires_grad_(False) for i in range(3))
    assert not q.is_contiguous()
    calls = []
    real = torch.Tensor.contiguous
    monkeypatch.setattr(torch.Tensor, "contiguous", lambda self, *a, **kw: (calls.append(tuple(self.shape)), real(self, *a, **kw))[1])
    o = tb.flash_attention(q, k, v, causal=True)
    monkeypatch.undo()
    assert (B, H, N, D) not in calls, "a [B, H, N, D] view with a contiguous last dimension must not be copied"
    _, _, _, ref = reference(q, k, v, True)
    assert (o.float() - ref).abs().max().item() < (3e-2 if dtype != torch.float32 else 2e-5)
    qg, kg, vg = (t.detach().clone().requires_grad_(True) for t in (q, k, v))   # gradients through strided saved views
    q2 = qkv.detach().clone().requires_grad_(True)
    o2 = tb.flash_attention(*(q2[:, :, i].permute(0, 2, 1, 3) for i in range(3)), causal=False)
    o2.sum().backward()
    qr, kr, vr, orf = reference(q, k, v, False)
    orf.sum().backward()
    got = q2.grad[:, :, 0].permute(0, 2, 1, 3).float()
    assert (got - qr.grad).abs().max().item() < (6e-2 if dtype != torch.float32 else 5e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_fast_scale_selects_the_mixed_precision_streams(dtype):
    """fast_scale=True = the reference's mixed-precision mode (lowPrecisionIntermediates): FOLD / pre-scaled streams, FP16 L and
    BF16 D between forward and backward; same function, same tolerances.  The upstream gradient arrives as a strided view."""
    from metal_flash_attention_amd import torch_binding as tb
    from metal_flash_attention_amd import AttentionKernelType
    B, H, R, C, D = 2, 3, 300, 449, 128
    g = torch.Generator(device="cuda").manual_seed(11)
    q, k, v = (torch.randn(B, H, n, D, generator=g, device="cuda").to(dtype).requires_grad_(True) for n in (R, C, C))
    w = torch.randn(B, R, H, D, generator=g, device="cuda").to(torch.bfloat16).permute(0, 2, 1, 3)   # [B, H, R, D] view
    o = tb.flash_attention(q, k, v, fast_scale=True)
    assert tb._kernel(dtype, R, C, D, AttentionKernelType.forward, True).variant.endswith("_fold")
    assert not tb._kernel(dtype, R, C, D, AttentionKernelType.forward, False).variant.endswith("_fold")
    o.backward(w)
    qr, kr, vr, orf = reference(q, k, v, False)
    orf.backward(w.float())
    assert (o.float() - orf).abs().max().item() < 3e-2
    for got, ref, name in ((q.grad, qr.grad, "dQ"), (k.grad, kr.grad, "dK"), (v.grad, vr.grad, "dV")):
        assert (got.float() - ref).abs().max().item() < 5e-2, name
