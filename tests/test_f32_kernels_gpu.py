"""The FP32 production kernels (csrc/attn_f32.h: BASELINE config 3's code objects) against the oracle, and the hand-over rule:
the general kernels' variants of the 64 / 128 head blocks launch them when every operand is FP32, row-major, rows 16-byte
aligned and D % 4 == 0 (mfa_attention_kernel_launch_form names the code object); every other launch stays on the general kernels."""
import numpy as np
import pytest

import harness
from harness import TOL_FP32
from metal_flash_attention_amd import (AttentionDescriptor, AttentionKernel, AttentionKernelType, AttentionOperand as Op,
                                       GEMMOperandPrecision as P)
from oracle import Network, NetworkDescriptor

pytestmark = pytest.mark.gpu

T = AttentionKernelType
NAMES = {T.forward: "attn_f32_fwd", T.backwardQuery: "attn_f32_dq", T.backwardKeyValue: "attn_f32_dkv"}


def make_desc(R, C, D, tr=(False,) * 4):
    d = AttentionDescriptor()
    d.lowPrecisionInputs = d.lowPrecisionIntermediates = False
    d.matrixDimensions = (R, C, D)
    d.transposeState = tuple(tr)
    return d


# ragged rows / columns (edge tiles, empty waves), D below the head block (zero-filled chunks), one row, one tile, many tiles
SHAPES = [(128, 128, 128), (300, 555, 64), (257, 600, 128), (129, 131, 100), (64, 64, 4), (1, 50, 8), (1000, 900, 72),
          (160, 96, 128), (33, 1, 64), (640, 640, 60), (96, 1024, 128)]


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("shape", SHAPES)
def test_fp32_kernels_match_the_oracle(shape, causal):
    R, C, D = shape
    if causal and C < R:
        pytest.skip("causal launches need column >= row (mfa.h)")
    net = Network(NetworkDescriptor(R, C, D), seed=R * 7 + C)
    run = harness.DeviceRun(make_desc(R, C, D), net, causal=causal)
    for t, k in run.kernels.items():
        form = k.launchForm(run.buffers, row=R, column=C, causal=causal)
        if D <= 32:   # the 32 head block has no FP32 code object of its own
            assert form == k.variant and "generic" in form, form
        else:
            assert form.startswith(NAMES[t] + ("_d64" if D <= 64 else "_d128")), (t, form, k.variant)
    got = run.execute()
    ref = net.run(causal=causal)
    failures, report = harness.compare(ref, got, TOL_FP32 if max(R, C) <= 777 else {k: 5e-5 for k in TOL_FP32})
    assert not failures, (failures, report)
    assert all(run.tails_ok.values()), run.tails_ok
    assert not np.isnan(got["O"]).any()


@pytest.mark.parametrize("what", ["D % 4", "transposed", "16-bit", "leading dimension", "head block 256"])
def test_launches_the_fp32_kernels_do_not_take(what):
    """they stay on the general kernels (and still match the oracle)"""
    R, C, D = 150, 200, 64
    tr, kw = (False,) * 4, {}
    if what == "D % 4":
        D = 30
    elif what == "transposed":
        tr = (False, True, False, False)
    elif what == "head block 256":
        D = 200
    desc = make_desc(R, C, D, tr)
    if what == "16-bit":
        desc.lowPrecisionInputs = True
    net = Network(NetworkDescriptor(R, C, D), seed=5)
    run = harness.DeviceRun(desc, net)
    for t, k in run.kernels.items():
        if what == "leading dimension":   # rows of 65 floats: not 16-byte aligned
            form = k.launchForm(run.buffers, row=R, column=C, leadingDimensions={Op.K: D + 1})
        else:
            form = k.launchForm(run.buffers, row=R, column=C)
        assert "attn_f32_" not in form, (what, form)
    if what != "leading dimension":
        got = run.execute()
        if what == "16-bit":
            from test_attention_gpu import round_inputs
            round_inputs(net, desc)
        failures, report = harness.compare(net.run(), got, harness.TOL_MIXED if what == "16-bit" else TOL_FP32)
        assert not failures, failures


def test_fp32_kernels_batched_heads_with_strides():
    """B x H problems behind head / batch strides, ragged R (the L / D vectors of a head then start on a 4-byte boundary only)"""
    import torch
    B, H, R, C, D = 2, 3, 150, 210, 128
    rng = np.random.default_rng(3)
    host = {n: rng.standard_normal((B, H, R if n in ("Q", "dO") else C, D)).astype(np.float32) for n in ("Q", "K", "V", "dO")}
    bufs = {Op.Q: torch.from_numpy(host["Q"]).cuda(), Op.K: torch.from_numpy(host["K"]).cuda(), Op.V: torch.from_numpy(host["V"]).cuda(),
            Op.dO: torch.from_numpy(host["dO"]).cuda()}
    nan = float("nan")
    bufs[Op.O] = torch.full((B, H, R, D), nan, device="cuda")
    bufs[Op.dQ] = torch.full((B, H, R, D), nan, device="cuda")
    bufs[Op.dK] = torch.full((B, H, C, D), nan, device="cuda")
    bufs[Op.dV] = torch.full((B, H, C, D), nan, device="cuda")
    Lpad = R + 1   # head stride of L and D: 151 floats
    bufs[Op.L] = torch.full((B, H, Lpad), nan, device="cuda")
    bufs[Op.D] = torch.full((B, H, Lpad), nan, device="cuda")
    hs = {Op.Q: R * D, Op.K: C * D, Op.V: C * D, Op.O: R * D, Op.L: Lpad, Op.D: Lpad, Op.dO: R * D, Op.dV: C * D, Op.dK: C * D, Op.dQ: R * D}
    bs = {op: s * H for op, s in hs.items()}
    desc = make_desc(R, C, D)
    stream = torch.cuda.current_stream().cuda_stream
    kw = dict(row=R, column=C, heads=H, batches=B, headStrides=hs, batchStrides=bs)
    for t in (T.forward, T.backwardQuery, T.backwardKeyValue):
        k = AttentionKernel(desc.kernelDescriptor(t))
        assert k.launchForm(bufs, **kw).startswith(NAMES[t]), k.launchForm(bufs, **kw)
        k.dispatch(bufs, stream=stream, **kw)
    torch.cuda.synchronize()
    out = {op: bufs[op].cpu().numpy() for op in (Op.O, Op.L, Op.D, Op.dQ, Op.dK, Op.dV)}
    for b in range(B):
        for h in range(H):
            net = Network(NetworkDescriptor(R, C, D), seed=0)
            net.Q, net.K, net.V, net.dO = (np.ascontiguousarray(host[n][b, h]) for n in ("Q", "K", "V", "dO"))
            net.invalidate()
            ref = net.run()
            got = dict(O=out[Op.O][b, h], L=out[Op.L][b, h, :R] / np.float32(harness.LOG2E), D=out[Op.D][b, h, :R] * np.sqrt(np.float32(D)),
                       dQ=out[Op.dQ][b, h], dK=out[Op.dK][b, h], dV=out[Op.dV][b, h])
            failures, report = harness.compare(ref, got, TOL_FP32)
            assert not failures, (b, h, failures)
            assert np.isnan(out[Op.L][b, h, R:]).all() and np.isnan(out[Op.D][b, h, R:]).all()
