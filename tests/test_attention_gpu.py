"""GPU parity tests (run with -m gpu on an MI355X): the HIP kernels, called through the C ABI,
against the CPU oracle on the same seeded inputs.  Shapes, dtype packing, canaries and tolerances
follow the reference's tests (SquareAttentionTest.swift:5-26, :397-554;
RectangularAttentionTest.swift:7-35, :451-472); unlike the reference these tests ASSERT."""
import os

import numpy as np
import pytest

import harness
from harness import TOL_FP32, TOL_MIXED, TOL_MIXED_SHORT
from metal_flash_attention_amd import (
    AttentionDescriptor, AttentionKernel, AttentionKernelType, AttentionOperand, GEMMOperandPrecision,
)
from oracle import Network, NetworkDescriptor, round_trip

pytestmark = pytest.mark.gpu
P = GEMMOperandPrecision
Op = AttentionOperand


def make_desc(R, C, D, low_in=False, low_mid=False, tr=(False,) * 4, in_type=P.FP16):
    d = AttentionDescriptor()
    d.lowPrecisionInputs, d.lowPrecisionIntermediates = low_in, low_mid
    d.matrixDimensions = (R, C, D)
    d.transposeState = tuple(tr)
    d.lowPrecisionInputType = in_type
    return d


import contextlib

import metal_flash_attention_amd as mfa
from metal_flash_attention_amd import _abi

DEV_LIBRARY = "libmfa_hip_dev" in os.path.basename(_abi.library_path()) if hasattr(_abi, "library_path") else False
needs_dev_library = pytest.mark.skipif(
    not DEV_LIBRARY, reason="developer schedule: run with MFA_LIBRARY=metal_flash_attention_amd/libmfa_hip_dev.so (make DEV=1)")


@contextlib.contextmanager
def parameter_rows(*rows):
    """Install parameter-table rows (type, mixed, text) for the duration of a test: the product library selects code objects
    through the table (AttentionDescriptor.swift:37-54), not through environment knobs."""
    try:
        for t, mixed, text in rows:
            mfa.setParameterFile(t, mixed, text)
        yield
    finally:
        mfa.resetParameterFiles()


FWD_8x32 = (AttentionKernelType.forward, True, "| 32 | 128 | 32 | 32 | Q, O |\n| 64 | 256 | 32 | 64 | Q, O |\n| 128 | 256 | 32 | 128 | Q, O |\n| 256 | 128 | 32 | 256 | Q, O |\n")
DKV_RS = (AttentionKernelType.backwardKeyValue, True, "| 64 | 128 | 32 | 64 | K, V, dV, dK |\n| 96 | 128 | 32 | 96 | K, V, dV, dK |\n| 128 | 128 | 32 | 128 | K, V, dV, dK |\n"
          "| 160 | 64 | 32 | 160 | K, V, dV, dK |\n| 192 | 64 | 32 | 192 | K, V, dV, dK |\n| 256 | 64 | 32 | 256 | K, V, dV, dK |\n")
# backwardQuery: the four 32-row waves of attn_bwd16.h at D > 128 (the default rows select the role-split pairs of attn_dq16_p5.h)
DQ_W4 = (AttentionKernelType.backwardQuery, True, "| 64 | 256 | 64 | 64 | Q, dO, dQ |\n| 128 | 256 | 64 | 128 | Q, dO, dQ |\n| 160 | 128 | 64 | 160 | Q, dO, dQ |\n"
         "| 192 | 128 | 64 | 192 | Q, dO, dQ |\n| 256 | 128 | 64 | 256 | Q, dO, dQ |\n")
DKV_W4 = (AttentionKernelType.backwardKeyValue, True, "| 64 | 128 | 64 | 64 | K, V, dV, dK |\n| 128 | 128 | 64 | 128 | K, V, dV, dK |\n| 256 | 64 | 32 | 256 | K, V, dV, dK |\n")


# forward: the persistent four-wave kernel (attn_fwd16_p6) also for D <= 32, which keeps its own 4 x 32 object by default
FWD_4x64_D32 = (AttentionKernelType.forward, True, "| 32 | 256 | 64 | 64 | Q, O |\n| 64 | 256 | 64 | 64 | Q, O |\n| 128 | 256 | 64 | 128 | Q, O |\n| 256 | 256 | 32 | 256 | Q, O |\n")


def round_inputs(net, desc):
    """Give the oracle the values the device really sees (the reference's CPU side keeps the
    unrounded ones; its loose mixed tolerances absorb the FP16 input quantisation)."""
    from oracle import round_trip
    prec = desc.memoryPrecisions
    net.Q, net.K, net.V = (round_trip(a, int(prec[o])) for a, o in ((net.Q, Op.Q), (net.K, Op.K), (net.V, Op.V)))
    net.dO = round_trip(net.dO, int(prec[Op.dO]))
    net.invalidate()


def run_case(R, C, D, seed=0, tolerances=None, rounded_oracle=False, **kw):
    net = Network(NetworkDescriptor(R, C, D), seed=seed)
    desc = make_desc(R, C, D, **kw)
    run = harness.DeviceRun(desc, net, seed=seed + 1000)
    got = run.execute()
    if rounded_oracle:
        round_inputs(net, desc)
    ref = net.run()
    low = desc.lowPrecisionInputs or desc.lowPrecisionIntermediates
    if tolerances is None:
        tolerances = (TOL_MIXED_SHORT if C <= 20 else TOL_MIXED) if low else TOL_FP32
    failures, report = harness.compare(ref, got, tolerances)
    variants = [k.variant for k in run.kernels.values()]
    assert not failures, (failures, variants)
    assert all(run.tails_ok.values()), f"out-of-bounds write: {run.tails_ok} {variants}"
    assert not np.isnan(got["O"]).any(), "NaN poison in O[0] was not overwritten"
    return report, run


@pytest.mark.parametrize("N,D", harness.SQUARE_SHAPES)
def test_square_fp32(N, D):
    """SquareAttentionTest.testCorrectness: 20 fixed (N, D), FP32, all six outputs within 2e-5."""
    run_case(N, N, D, seed=N * 1000 + D)


@pytest.mark.parametrize("index,case", list(enumerate(harness.rectangular_cases(15, seed=0))))
def test_rectangular_random(index, case):
    """RectangularAttentionTest.testCorrectness: R != C, random D, transposes and precision flags."""
    run_case(case["row"], case["column"], case["head"], seed=index,
             low_in=case["lowPrecisionInputs"], low_mid=case["lowPrecisionIntermediates"],
             tr=case["transposeState"])


@pytest.mark.parametrize("index,case", list(enumerate(harness.rectangular_cases(12, seed=1))))
def test_rectangular_random_bf16_inputs(index, case):
    """Same generator with this project's BF16 input extension (Q, K, V, dO all BF16).  BF16 keeps 8
    mantissa bits (FP16: 11) and the reference packs it by TRUNCATION (MTLContext+Buffers.swift:36-42),
    so the input-quantisation term alone exceeds the reference's FP16-calibrated L tolerance; the
    oracle therefore gets the rounded inputs and the reference tolerances then apply unchanged."""
    run_case(case["row"], case["column"], case["head"], seed=100 + index, low_in=True,
             low_mid=case["lowPrecisionIntermediates"], tr=case["transposeState"], in_type=P.BF16,
             rounded_oracle=True)


@pytest.mark.parametrize("low_mid", [False, True])
@pytest.mark.parametrize("in_type", [P.BF16, P.FP16])
def test_head_dimensions_up_to_32_on_the_persistent_four_wave_kernel(low_mid, in_type):
    """| 32 | 256 | 64 | 64 | selects attn_fwd16_p6 for D <= 32 too (zero-padded chunks of its 64-wide code object): forward and
    the backward kernels that consume its L, ragged rows and keys, causal"""
    with parameter_rows(FWD_4x64_D32):
        for (R, C, D), causal in (((300, 520, 32), False), ((257, 1000, 24), False), ((512, 512, 16 if low_mid else 8), False), ((384, 640, 32), True)):
            net = Network(NetworkDescriptor(R, C, D), seed=D + R)
            desc = make_desc(R, C, D, low_in=True, low_mid=low_mid, in_type=in_type)
            run = harness.DeviceRun(desc, net, causal=causal)
            assert run.kernels[AttentionKernelType.forward].variant.startswith("attn_fwd16p6"), run.kernels[AttentionKernelType.forward].variant
            got = run.execute()
            round_inputs(net, desc)
            ref = net.run(causal=causal)
            failures, report = harness.compare(ref, got, TOL_MIXED)
            assert not failures and all(run.tails_ok.values()), (failures, R, C, D, causal, report)


def test_head_dimensions_up_to_32_launches_the_persistent_kernel_does_not_serve():
    """| 32 | 256 | 64 | 64 | at D <= 32: per-batch lengths (served by attn_fwd16_p6 itself since round 6) and what it does not take (an L of the other storage type, a one-head
    split whose key range is not whole multiples of four tiles per piece) runs on the D = 64 EIGHT-wave kernels -- the variant must then
    carry their functions (dynamic LDS above 64 KiB: a fresh process fails otherwise), their 256-row blocks for the split grid, and
    the launch form must name them."""
    import torch
    stream = torch.cuda.current_stream().cuda_stream

    def pack16(a):
        return torch.from_numpy((np.ascontiguousarray(a).view(np.uint32) >> 16).astype(np.uint16).view(np.int16)).cuda()

    with parameter_rows(FWD_4x64_D32):
        # (1) per-batch lengths
        B, H, Rmax, Cmax, D = 3, 2, 300, 400, 32
        rlen, clen = [300, 77, 130], [400, 100, 333]
        desc = make_desc(Rmax, Cmax, D, low_in=True, low_mid=True, in_type=P.BF16)
        kernel = AttentionKernel(desc.kernelDescriptor(AttentionKernelType.forward))
        assert kernel.variant.startswith("attn_fwd16p6"), kernel.variant
        rng = np.random.default_rng(5)
        host = {n: round_trip(rng.standard_normal((B, H, Rmax if n == "Q" else Cmax, D)).astype(np.float32), int(P.BF16)) for n in ("Q", "K", "V")}
        lprec = desc.memoryPrecisions[Op.L]
        bufs = {Op.Q: pack16(host["Q"]), Op.K: pack16(host["K"]), Op.V: pack16(host["V"]),
                Op.O: torch.full((B, H, Rmax, D), float("nan"), device="cuda"),
                Op.L: torch.zeros((B, H, Rmax), device="cuda", dtype=torch.float16 if lprec == P.FP16 else torch.float32)}
        hs = {Op.Q: Rmax * D, Op.K: Cmax * D, Op.V: Cmax * D, Op.O: Rmax * D, Op.L: Rmax}
        bs = {op: v * H for op, v in hs.items()}
        rl, cl = (torch.tensor(x, dtype=torch.int32, device="cuda") for x in (rlen, clen))
        kw = dict(row=Rmax, column=Cmax, heads=H, batches=B, headStrides=hs, batchStrides=bs, rowLengths=rl, columnLengths=cl)
        form = kernel.launchForm(bufs, **kw)
        assert form.startswith("attn_fwd16_p6 (persistent") and "per-batch lengths" in form, form   # (round 6: served by the persistent kernel)
        kernel.dispatch(bufs, stream=stream, **kw)
        torch.cuda.synchronize()
        o, l = bufs[Op.O].cpu().numpy(), bufs[Op.L].float().cpu().numpy() / np.float32(harness.LOG2E)
        for b in range(B):
            R, C = rlen[b], clen[b]
            for h in range(H):
                net = Network(NetworkDescriptor(R, C, D), seed=0)
                net.Q, net.K, net.V = (np.ascontiguousarray(host[n][b, h, :(R if n == "Q" else C)]) for n in ("Q", "K", "V"))
                net.invalidate()
                ref = net.run(backward=False)
                assert np.abs(o[b, h, :R] - ref["O"]).max() < 1.5e-2 and np.abs(l[b, h, :R] - ref["L"]).max() < 7e-3, (b, h)
                assert np.isnan(o[b, h, R:]).all()
        # (2) one head, a workspace, C not a multiple of 256 x pieces: the eight-wave kernel's column-parallel sibling, 256-row blocks
        R, C, D = 512, 2944, 24
        desc = make_desc(R, C, D, low_in=True, low_mid=True, in_type=P.BF16)
        net = Network(NetworkDescriptor(R, C, D), seed=8)
        run = harness.DeviceRun(desc, net, run_backward=False)
        kernel = run.kernels[AttentionKernelType.forward]
        assert kernel.variant.startswith("attn_fwd16p6"), kernel.variant
        need = kernel.workspaceSize(row=R, column=C)
        assert need > 0
        ws = torch.empty(need + 64, dtype=torch.uint8, device="cuda")
        form = kernel.launchForm(run.buffers, row=R, column=C, workspace=ws)
        assert "column-parallel x" in form and "attn_fwd16v3_bf16_d64_w8x32" in form, form
        kernel.dispatch(run.buffers, row=R, column=C, stream=stream, workspace=ws)
        torch.cuda.synchronize()
        got = run.results()
        round_inputs(net, desc)
        ref = net.run(backward=False)
        assert np.abs(got["O"] - ref["O"]).max() < 1.5e-2 and np.abs(got["L"] - ref["L"]).max() < 7e-3
        # (3) an L of the other storage type (FP32 L beside 16-bit intermediates): hand-edited kernel descriptor
        R, C, D = 300, 520, 32
        desc = make_desc(R, C, D, low_in=True, low_mid=True, in_type=P.BF16)
        kd = desc.kernelDescriptor(AttentionKernelType.forward)
        mp = dict(kd.memoryPrecisions)
        mp[Op.L] = P.FP32
        kd.memoryPrecisions = mp
        kernel = AttentionKernel(kd)
        assert kernel.variant.startswith("attn_fwd16p6"), kernel.variant
        net = Network(NetworkDescriptor(R, C, D), seed=12)
        round_inputs(net, desc)
        bufs = {Op.Q: pack16(net.Q), Op.K: pack16(net.K), Op.V: pack16(net.V), Op.O: torch.full((R, D), float("nan"), device="cuda"),
                Op.L: torch.zeros((R,), device="cuda")}
        form = kernel.launchForm(bufs, row=R, column=C)
        assert form.startswith("attn_fwd16v3_bf16_d64_w8x32"), form
        kernel.dispatch(bufs, row=R, column=C, stream=stream)
        torch.cuda.synchronize()
        ref = net.run(backward=False)
        assert np.abs(bufs[Op.O].cpu().numpy() - ref["O"]).max() < 1.5e-2
        assert np.abs(bufs[Op.L].cpu().numpy() / np.float32(harness.LOG2E) - ref["L"]).max() < 7e-3


@pytest.mark.parametrize("tr", [(True, True, True, True), (True, False, False, True), (False, True, True, False)])
@pytest.mark.parametrize("shape", [(130, 67, 72), (33, 200, 128), (257, 129, 200)])
def test_transposes_fp32(shape, tr):
    run_case(*shape, seed=3, tr=tr)


def test_rounded_inputs_tight_bf16_and_fp16():
    """Feeding the oracle the ROUNDED inputs removes the input-quantisation term, so the generic
    (fp32-compute) path must then agree to fp32 accuracy even with 16-bit storage.  D = 60 is not
    a multiple of 8, so every kernel here is the general one."""
    R, C, D = 150, 170, 60
    for in_type in (P.FP16, P.BF16):
        net = Network(NetworkDescriptor(R, C, D), seed=9)
        desc = make_desc(R, C, D, low_in=True, in_type=in_type)
        run = harness.DeviceRun(desc, net)
        assert all("generic" in k.variant for k in run.kernels.values())
        got = run.execute()
        round_inputs(net, desc)
        ref = net.run()
        failures, report = harness.compare(ref, got, {k: 5e-5 for k in TOL_FP32})
        assert not failures, (in_type, failures)


def test_multi_head_strides_match_per_head_runs():
    """Multi-head recipe of AttentionKernelDescriptor.swift:37-41: heads interleaved along the row,
    leading dimension D*H; and contiguous [B, H, N, D] via head/batch strides."""
    import torch
    R, C, D, H, B = 70, 90, 48, 3, 2
    nets = [[Network(NetworkDescriptor(R, C, D), seed=50 + b * H + h) for h in range(H)] for b in range(B)]
    desc = make_desc(R, C, D)
    kernels = {t: AttentionKernel(desc.kernelDescriptor(t)) for t in AttentionKernelType}
    stream = torch.cuda.current_stream().cuda_stream

    def gather(name, interleaved):
        a = np.stack([np.stack([getattr(nets[b][h], name) for h in range(H)]) for b in range(B)])  # [B,H,N,D]
        return np.ascontiguousarray(a.transpose(0, 2, 1, 3)) if interleaved else a              # [B,N,H,D]

    for interleaved in (False, True):
        bufs, ld, hs, bs = {}, {}, {}, {}
        for op, name, seq in ((Op.Q, "Q", R), (Op.K, "K", C), (Op.V, "V", C), (Op.dO, "dO", R)):
            bufs[op] = torch.from_numpy(gather(name, interleaved)).cuda()
        for op, seq in ((Op.O, R), (Op.dQ, R), (Op.dK, C), (Op.dV, C)):
            bufs[op] = torch.full((B, seq, H, D) if interleaved else (B, H, seq, D), float("nan"), device="cuda")
        for op in (Op.L, Op.D):
            bufs[op] = torch.zeros((B, H, R), device="cuda")
            hs[op], bs[op] = R, H * R
        for op, seq in ((Op.Q, R), (Op.K, C), (Op.V, C), (Op.dO, R), (Op.O, R), (Op.dQ, R), (Op.dK, C), (Op.dV, C)):
            if interleaved:
                ld[op], hs[op], bs[op] = D * H, D, seq * H * D
            else:
                ld[op], hs[op], bs[op] = D, seq * D, H * seq * D
        for t in AttentionKernelType:
            kernels[t].dispatch(bufs, row=R, column=C, heads=H, batches=B, leadingDimensions=ld,
                                headStrides=hs, batchStrides=bs, stream=stream)
        torch.cuda.synchronize()
        for b in range(B):
            for h in range(H):
                ref = nets[b][h].run()
                for op, name in ((Op.O, "O"), (Op.dQ, "dQ"), (Op.dK, "dK"), (Op.dV, "dV")):
                    t = bufs[op].cpu().numpy()
                    got = t[b, :, h, :] if interleaved else t[b, h]
                    assert np.abs(got - ref[name]).max() < 2e-5, (interleaved, b, h, name)
                Lgot = bufs[Op.L].cpu().numpy()[b, h] / np.float32(harness.LOG2E)
                assert np.abs(Lgot - ref["L"]).max() < 2e-5


def test_determinism_run_to_run():
    R, C, D = 300, 300, 96
    net = Network(NetworkDescriptor(R, C, D), seed=4)
    a = harness.DeviceRun(make_desc(R, C, D), net).execute()
    b = harness.DeviceRun(make_desc(R, C, D), net).execute()
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def test_softmax_extremes_do_not_overflow():
    """Large-magnitude logits: the running max / correction path must stay finite
    (AttentionKernel+Softmax.swift:290-301)."""
    R, C, D = 64, 200, 32
    net = Network(NetworkDescriptor(R, C, D), seed=8)
    net.Q *= 30.0
    net.K[150] *= 40.0  # one key dominates late in the traversal: forces a big rescale
    net.invalidate()
    desc = make_desc(R, C, D)
    got = harness.DeviceRun(desc, net).execute()
    ref = net.run()
    assert np.isfinite(got["O"]).all() and np.isfinite(got["L"]).all()
    assert np.abs(got["O"] - ref["O"]).max() < 1e-4
    assert np.abs(got["L"] - ref["L"]).max() / np.abs(ref["L"]).max() < 1e-5


# ---- 16-bit matrix-core forward kernel (attn_fwd16) -------------------------------------------
FWD16_SHAPES = [
    # (R, C, D): D buckets 32/64/128/256, D below a bucket (multiple of 8), ragged R and C
    (256, 256, 128), (512, 512, 64), (300, 300, 128), (255, 257, 64), (33, 65, 32), (64, 1, 64),
    (1, 64, 128), (100, 1000, 256), (257, 130, 200), (129, 77, 40), (96, 640, 80), (1024, 1024, 128),
    # buckets 96 / 160 / 192 (round 2) and head dimensions just below them
    (300, 333, 96), (257, 130, 88), (200, 449, 160), (129, 300, 152), (256, 320, 192), (100, 1000, 176),
]


@pytest.mark.parametrize("in_type", [P.BF16, P.FP16])
@pytest.mark.parametrize("shape", FWD16_SHAPES)
def test_forward_16bit_mfma(shape, in_type):
    """O and L of the MFMA forward kernel: within the reference's mixed tolerances of the oracle fed
    with the same (rounded) inputs, and -- a much tighter bound that still leaves room for P being
    rounded to 16 bits -- within 1.5e-2 (BF16) / 2e-3 (FP16) absolute."""
    R, C, D = shape
    net = Network(NetworkDescriptor(R, C, D), seed=R + C + D)
    desc = make_desc(R, C, D, low_in=True, in_type=in_type)
    run = harness.DeviceRun(desc, net, run_backward=False)
    assert run.kernels[AttentionKernelType.forward].variant.startswith("attn_fwd16"), run.kernels[AttentionKernelType.forward].variant
    got = run.execute()
    round_inputs(net, desc)
    ref = net.run(backward=False)
    tight = 1.5e-2 if in_type == P.BF16 else 2e-3
    failures, report = harness.compare(ref, got, dict(O=tight, L=1e-3))
    assert not failures, (failures, run.kernels[AttentionKernelType.forward].variant)
    assert run.tails_ok["O"] and run.tails_ok["L"]
    assert not np.isnan(got["O"]).any()


@pytest.mark.parametrize("heads,batches", [(8, 2), (3, 1), (16, 1)])
def test_forward_16bit_multi_head(heads, batches):
    """batch x head grid of the MFMA forward kernel (XCD-aware block order when heads*batches % 8 == 0,
    plain order otherwise), contiguous [B, H, N, D]."""
    import torch
    R, C, D = 200, 330, 128
    nets = [Network(NetworkDescriptor(R, C, D), seed=900 + i) for i in range(heads * batches)]
    desc = make_desc(R, C, D, low_in=True, in_type=P.BF16)
    kernel = AttentionKernel(desc.kernelDescriptor(AttentionKernelType.forward))

    def pack16(name):
        a = np.stack([getattr(n, name) for n in nets]).reshape(batches, heads, -1, D)
        return torch.from_numpy((np.ascontiguousarray(a).view(np.uint32) >> 16).astype(np.uint16).view(np.int16)).cuda()

    bufs = {Op.Q: pack16("Q"), Op.K: pack16("K"), Op.V: pack16("V"),
            Op.O: torch.full((batches, heads, R, D), float("nan"), device="cuda"),
            Op.L: torch.zeros((batches, heads, R), device="cuda")}
    hs = {Op.Q: R * D, Op.K: C * D, Op.V: C * D, Op.O: R * D, Op.L: R}
    bs = {k: v * heads for k, v in hs.items()}
    kernel.dispatch(bufs, row=R, column=C, heads=heads, batches=batches, headStrides=hs, batchStrides=bs,
                    stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    O = bufs[Op.O].cpu().numpy().reshape(heads * batches, R, D)
    L = bufs[Op.L].cpu().numpy().reshape(heads * batches, R) / np.float32(harness.LOG2E)
    for i, net in enumerate(nets):
        round_inputs(net, desc)
        ref = net.run(backward=False)
        assert np.abs(O[i] - ref["O"]).max() < 1.5e-2, i
        assert np.abs(L[i] - ref["L"]).max() < 1e-3, i


@pytest.mark.parametrize("low_mid", [False, True])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("heads,batches,N", [(5, 3, 1280), (8, 5, 1024), (1, 9, 1800), (16, 20, 600)])
def test_persistent_forward_block_table(heads, batches, N, causal, low_mid):
    """The persistent form of the D <= 128 forward kernel (attn_fwd16_p4p) on grids that exercise its block table: several
    blocks per workgroup, head counts that are / are not multiples of 8 (the two branches of fwd16_decode_block), odd numbers of
    row blocks (causal: a middle block that is its own pair), ragged last row blocks, more workgroups than compute units;
    interleaved heads ([B, N, H, D] storage viewed per head), every head against the oracle."""
    import torch
    D = 128
    nets = [Network(NetworkDescriptor(N, N, D), seed=1700 + i) for i in range(heads * batches)]
    desc = make_desc(N, N, D, low_in=True, low_mid=low_mid, in_type=P.BF16)
    kernel = AttentionKernel(desc.kernelDescriptor(AttentionKernelType.forward))

    def pack16(name):     # [B][N][H][D]: heads interleaved, leading dimension H * D
        a = np.stack([getattr(n, name) for n in nets]).reshape(batches, heads, N, D).transpose(0, 2, 1, 3)
        return torch.from_numpy((np.ascontiguousarray(a).view(np.uint32) >> 16).astype(np.uint16).view(np.int16)).cuda()

    mem = desc.memoryPrecisions
    bufs = {Op.Q: pack16("Q"), Op.K: pack16("K"), Op.V: pack16("V"),
            Op.O: torch.full((batches, N, heads, D), float("nan"), device="cuda"),
            Op.L: torch.zeros((batches, heads, N), device="cuda", dtype=torch.float16 if mem[Op.L] == P.FP16 else torch.float32)}
    hs = {Op.Q: D, Op.K: D, Op.V: D, Op.O: D, Op.L: N}
    bs = {Op.Q: N * heads * D, Op.K: N * heads * D, Op.V: N * heads * D, Op.O: N * heads * D, Op.L: heads * N}
    ld = {Op.Q: heads * D, Op.K: heads * D, Op.V: heads * D, Op.O: heads * D}
    kw = dict(row=N, column=N, heads=heads, batches=batches, headStrides=hs, batchStrides=bs, leadingDimensions=ld, causal=causal)
    assert kernel.launchForm(bufs, **kw).startswith("attn_fwd16_p4p (persistent")
    kernel.dispatch(bufs, stream=torch.cuda.current_stream().cuda_stream, **kw)
    torch.cuda.synchronize()
    O = bufs[Op.O].cpu().numpy().transpose(0, 2, 1, 3).reshape(heads * batches, N, D)
    L = bufs[Op.L].float().cpu().numpy().reshape(heads * batches, N) / np.float32(harness.LOG2E)
    step = max(1, len(nets) // 12)          # the oracle checks a spread of heads (all blocks of a head share its code path)
    for i in list(range(0, len(nets), step)) + [len(nets) - 1]:
        net = nets[i]
        round_inputs(net, desc)
        ref = net.run(backward=False, causal=causal)
        assert np.abs(O[i] - ref["O"]).max() < 1.5e-2, i
        assert np.abs(L[i] - ref["L"]).max() < (7e-3 if low_mid else 1e-3), i
    assert not np.isnan(O).any()


def test_persistent_forward_more_blocks_than_the_table_holds():
    """66,000 row blocks on 256 compute units: a workgroup's share (258) would not fit the 255-entry block table, so the launch
    takes more workgroups than compute units (a multiple of 8, for the head -> XCD affinity); spot heads against the oracle"""
    import torch
    R, C, D, H, B = 256, 128, 128, 33000, 2
    desc = make_desc(R, C, D, low_in=True, low_mid=True, in_type=P.BF16)
    kernel = AttentionKernel(desc.kernelDescriptor(AttentionKernelType.forward))
    g = torch.Generator(device="cuda"); g.manual_seed(77)
    q = torch.randn((B, H, R, D), generator=g, device="cuda", dtype=torch.float32).to(torch.bfloat16)
    k = torch.randn((B, H, C, D), generator=g, device="cuda", dtype=torch.float32).to(torch.bfloat16)
    v = torch.randn((B, H, C, D), generator=g, device="cuda", dtype=torch.float32).to(torch.bfloat16)
    o = torch.full((B, H, R, D), float("nan"), device="cuda")
    l = torch.zeros((B, H, R), device="cuda", dtype=torch.float16)
    bufs = {Op.Q: q, Op.K: k, Op.V: v, Op.O: o, Op.L: l}
    hs = {Op.Q: R * D, Op.K: C * D, Op.V: C * D, Op.O: R * D, Op.L: R}
    bs = {op: x * H for op, x in hs.items()}
    kw = dict(row=R, column=C, heads=H, batches=B, headStrides=hs, batchStrides=bs)
    assert kernel.launchForm(bufs, **kw).startswith("attn_fwd16_p4p (persistent")
    kernel.dispatch(bufs, stream=torch.cuda.current_stream().cuda_stream, **kw)
    torch.cuda.synchronize()
    assert not torch.isnan(o).any().item()
    for b, h in ((0, 0), (0, 255), (0, 256), (1, 17), (1, H - 1), (0, 32999), (1, 16384)):
        net = Network(NetworkDescriptor(R, C, D), seed=0)
        net.Q, net.K, net.V = (x[b, h].float().cpu().numpy() for x in (q, k, v))
        net.invalidate()
        ref = net.run(backward=False)
        assert np.abs(o[b, h].cpu().numpy() - ref["O"]).max() < 1.5e-2, (b, h)
        assert np.abs(l[b, h].float().cpu().numpy() / np.float32(harness.LOG2E) - ref["L"]).max() < 7e-3, (b, h)


def test_persistent_forward_full_size_causal_mixed():
    """N = 4096, D = 128, causal, mixed-precision mode: the causal streams of the persistent kernel at bench.py's size"""
    R = 4096
    net = Network(NetworkDescriptor(R, R, 128), seed=3)
    desc = make_desc(R, R, 128, low_in=True, low_mid=True, in_type=P.BF16)
    run = harness.DeviceRun(desc, net, run_backward=False, causal=True)
    got = run.execute()
    round_inputs(net, desc)
    ref = net.run(backward=False, causal=True)
    assert np.abs(got["O"] - ref["O"]).max() < 5e-3 and np.abs(got["L"] - ref["L"]).max() < 7e-3
    assert all(run.tails_ok[k] for k in ("O", "L"))


def test_forward_16bit_unaligned_launch_falls_back_to_general_kernel():
    """A leading dimension that breaks the 16-byte chunking must still give right answers (the
    kernel object keeps the general code object for such launches)."""
    import torch
    R, C, D = 70, 90, 64
    net = Network(NetworkDescriptor(R, C, D), seed=77)
    desc = make_desc(R, C, D, low_in=True, in_type=P.BF16)
    kernel = AttentionKernel(desc.kernelDescriptor(AttentionKernelType.forward))
    ld = D + 2  # 4-byte aligned rows only

    def pad16(a):
        out = np.zeros((a.shape[0], ld), np.float32)
        out[:, :D] = a
        return torch.from_numpy((out.view(np.uint32) >> 16).astype(np.uint16).view(np.int16)).cuda()

    bufs = {Op.Q: pad16(net.Q), Op.K: pad16(net.K), Op.V: pad16(net.V),
            Op.O: torch.full((R, D), float("nan"), device="cuda"), Op.L: torch.zeros(R, device="cuda")}
    kernel.dispatch(bufs, row=R, column=C, leadingDimensions={Op.Q: ld, Op.K: ld, Op.V: ld},
                    stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    round_inputs(net, desc)
    ref = net.run(backward=False)
    assert np.abs(bufs[Op.O].cpu().numpy() - ref["O"]).max() < 5e-5


@pytest.mark.parametrize("impl", ["product"] + [pytest.param(i, marks=needs_dev_library) for i in
                                                ["v1", "v2:0", "v2:1", "v2:2", "v3:0", "v3:1", "v3:2", "v3:41", "v4:0", "v4:1", "v4:2", "v4:4", "v4:8", "v4:16"]])
def test_forward_16bit_forced_rescale(impl, monkeypatch):
    """The deferred-rescale branch of the pipelined kernel is rare on random data, so force it
    (guide rule: a rare data-dependent branch needs its own test): one key far along the traversal
    dominates some rows by much more than the threshold, another grows the maximum only slightly.
    All schedules -- THR=0 (the reference's rule), THR=8 and the unpipelined kernel -- must agree
    with the full-tensor oracle."""
    if impl != "product":
        monkeypatch.setenv("MFA_FWD16_IMPL", impl)
    R, C, D = 128, 640, 64
    net = Network(NetworkDescriptor(R, C, D), seed=21)
    net.K[300] = net.Q[5] * 6.0      # huge score for row 5 (and large for correlated rows) at tile 4
    net.K[500] = net.Q[70] * 1.5     # moderate growth for row 70 at tile 7
    net.invalidate()
    desc = make_desc(R, C, D, low_in=True, in_type=P.BF16)
    run = harness.DeviceRun(desc, net, run_backward=False)
    got = run.execute()
    round_inputs(net, desc)
    ref = net.run(backward=False)
    assert np.isfinite(got["O"]).all()
    assert np.abs(got["O"] - ref["O"]).max() < 2e-2, run.kernels[AttentionKernelType.forward].variant
    assert np.abs(got["L"] - ref["L"]).max() < 2e-3


@pytest.mark.parametrize("which", ["p4_exact", "p4_fold", "v3_8x32", pytest.param("dev:v3:7", marks=needs_dev_library),
                                   pytest.param("dev:p4:1", marks=needs_dev_library)])
def test_forward_16bit_forced_rescale_d128(which, monkeypatch):
    """The same forced-rescale construction at D = 128 for every product code object: the four-wave hand-placed stream with the
    scale applied in fp32 (default row), its FOLD stream (lowPrecisionIntermediates), and the eight-wave kernel (32-key row)."""
    rows = [FWD_8x32] if which == "v3_8x32" else []
    if which.startswith("dev:"):
        monkeypatch.setenv("MFA_FWD16_IMPL", which[4:])
    with parameter_rows(*rows):
        _forced_rescale_d128(which)


def _forced_rescale_d128(which):
    R, C, D = 300, 1000, 128
    net = Network(NetworkDescriptor(R, C, D), seed=22)
    net.K[450] = net.Q[5] * 4.0
    net.K[900] = net.Q[270] * 1.2
    net.invalidate()
    desc = make_desc(R, C, D, low_in=True, low_mid=(which == "p4_fold"), in_type=P.BF16)
    run = harness.DeviceRun(desc, net, run_backward=False)
    variant = run.kernels[AttentionKernelType.forward].variant
    expect = {"p4_exact": "attn_fwd16p4_bf16_d128_w4x64_thr8", "p4_fold": "attn_fwd16p4_bf16_d128_w4x64_thr8_fold",
              "v3_8x32": "attn_fwd16v3_bf16_d128_w8x32_thr8_ldsdma"}.get(which)
    assert expect is None or variant == expect, variant
    got = run.execute()
    round_inputs(net, desc)
    ref = net.run(backward=False)
    assert np.isfinite(got["O"]).all()
    assert np.abs(got["O"] - ref["O"]).max() < 2e-2, variant
    # FOLD: Q * scale rounded to bf16 (8-bit mantissa) moves L by ~2e-3; L itself is stored in FP16 with low intermediates
    assert np.abs(got["L"] - ref["L"]).max() < (2e-2 if which == "p4_fold" else 2e-3), variant
    assert all(run.tails_ok.values())


# ---- BASELINE.json configurations at FULL size ------------------------------------------------
def _full_size(R, D, low, in_type=P.BF16, backward=False, seed=0):
    net = Network(NetworkDescriptor(R, R, D), seed=seed)
    desc = make_desc(R, R, D, low_in=low, in_type=in_type)
    run = harness.DeviceRun(desc, net, run_backward=backward)
    got = run.execute()
    if low:
        round_inputs(net, desc)
    ref = net.run(backward=backward)
    return ref, got, run


def test_config2_forward_n4096_d64_bf16_single_head():
    ref, got, run = _full_size(4096, 64, True)
    assert run.kernels[AttentionKernelType.forward].variant.startswith("attn_fwd16p6")   # (one head: its column-parallel sibling runs)
    assert np.abs(got["O"] - ref["O"]).max() < 5e-3 and np.abs(got["L"] - ref["L"]).max() < 1e-3
    assert all(run.tails_ok.values())


def test_config3_forward_backward_n4096_d128_fp32():
    """FP32 at N=4096: the reference's 2e-5 tolerance was set on N <= 777 (SquareAttentionTest.swift:6-25);
    fp32 summation-order noise grows with N, so the full-size check states its own bound: 2e-5 for
    O/L/D, 1e-4 for the gradients (each a 4096-term fp32 sum in a different order from the oracle)."""
    ref, got, run = _full_size(4096, 128, False, backward=True)
    for name, tol in (("O", 2e-5), ("L", 2e-5), ("D", 2e-5), ("dV", 1e-4), ("dK", 1e-4), ("dQ", 1e-4)):
        err = np.abs(got[name] - ref[name]).max()
        assert err < tol, (name, err)
    assert all(run.tails_ok.values())


def test_config4_forward_n8192_d256_bf16():
    ref, got, run = _full_size(8192, 256, True)
    assert run.kernels[AttentionKernelType.forward].variant.startswith("attn_fwd16")
    assert np.abs(got["O"] - ref["O"]).max() < 5e-3 and np.abs(got["L"] - ref["L"]).max() < 1e-3


def test_config5_shard_forward_n16384_d128_bf16_two_heads():
    """Config 5 is B=8 H=32 N=16384 D=128 sharded over 8 GPUs; the oracle checks a sample of heads at
    full sequence length (the GPU kernel treats every head identically)."""
    for seed in (0, 1):
        ref, got, run = _full_size(16384, 128, True, seed=seed)
        assert np.abs(got["O"] - ref["O"]).max() < 5e-3 and np.abs(got["L"] - ref["L"]).max() < 1e-3


# ---- the code objects bench.py times, at FULL size, in the reference's mixed-precision mode ------------------------------
# (lowPrecisionInputs + lowPrecisionIntermediates: FOLD / pre-scaled streams, FP16 L, BF16 D).  Oracle on the rounded inputs,
# the reference's own mixed tolerances (SquareAttentionTest.swift:539-554: O 5e-2, L 7e-3, D 1e-1, gradients 5e-2) plus a
# tighter bound that states what these streams really deliver; the variant names are the ones bench.py prints.
def _full_size_mixed(R, D, in_type, backward, seed=0):
    net = Network(NetworkDescriptor(R, R, D), seed=seed)
    desc = make_desc(R, R, D, low_in=True, low_mid=True, in_type=in_type)
    run = harness.DeviceRun(desc, net, run_backward=backward)
    got = run.execute()
    round_inputs(net, desc)
    ref = net.run(backward=backward)
    variants = {t.name: k.variant for t, k in run.kernels.items()}
    tol = {k: v for k, v in TOL_MIXED.items() if backward or k in ("O", "L")}
    failures, report = harness.compare(ref, got, tol)
    assert not failures, (failures, variants)
    assert all(run.tails_ok.values()), run.tails_ok
    return report, variants


def test_headline_code_object_forward_n4096_d128_bf16_mixed():
    """The kernel of bench.py's default line (fwd_bf16_d128): O AND L at N = 4096 against the oracle."""
    report, variants = _full_size_mixed(4096, 128, P.BF16, backward=False)
    assert variants["forward"].startswith("attn_fwd16p4") and variants["forward"].endswith("_fold"), variants
    # Q' = Q log2e / sqrt(D) rounded to bf16 once (8-bit mantissa: relative 2^-9 per product term, random signs) and L kept
    # in FP16 (ulp 2^-7 log2 units = 5.4e-3 nats at |L| in [8, 16), half of it the rounding bound)
    print("full-size mixed", variants, report)
    assert report["O"] < 5e-3 and report["L"] < 7e-3, report


def test_config5_shard_forward_n16384_d128_bf16_mixed_mode():
    """bench.py's fwd_bf16_d128_n16k_mixed (BASELINE config 5's per-GPU shard in the reference's mixed-precision mode): one head at
    the full N = 16384 against the oracle.  L = m + ln(sum) grows like ln N (about 9.7 + the row's spread here): still inside
    [8, 16), where FP16 resolves 2^-7 log2 units = 5.4e-3 nats (half of it the rounding bound) -- the reference's 7e-3 holds; from
    |L| >= 16 on (N of the order of 10^6 keys for N(0,1) data) its own FP16-L policy would not (+Precisions.swift:149-215)."""
    report, variants = _full_size_mixed(16384, 128, P.BF16, backward=False, seed=2)
    assert variants["forward"].startswith("attn_fwd16p4") and variants["forward"].endswith("_fold"), variants
    print("full-size mixed N = 16384", variants, report)
    assert report["O"] < 5e-3 and report["L"] < 7e-3, report


def test_headline_code_objects_forward_backward_n4096_d128_bf16_mixed():
    """bench.py's fwdbwd_bf16_d128_mixed / dq_bf16_d128 / dkv_bf16_d128: all six outputs at N = 4096."""
    report, variants = _full_size_mixed(4096, 128, P.BF16, backward=True, seed=1)
    assert variants["forward"].endswith("_fold") and "attn_dq16p4" in variants["backwardQuery"] and \
        "attn_dkv16p4" in variants["backwardKeyValue"], variants
    print("full-size mixed", variants, report)
    assert max(report[k] for k in ("dQ", "dK", "dV")) < 2e-2 and report["D"] < 5e-2, report


def test_reference_mix_forward_backward_n4096_d128_f16_bf16_dO():
    """The reference's own low-precision mix (+Precisions.swift:13-17: FP16 Q, K, V, BF16 dO, FP16 L, BF16 D) at N = 4096
    (bench.py's fwdbwd_f16_d128_refmix)."""
    report, variants = _full_size_mixed(4096, 128, P.FP16, backward=True, seed=2)
    assert variants["forward"].startswith("attn_fwd16p4_f16") and variants["forward"].endswith("_fold"), variants
    print("full-size refmix", variants, report)
    assert report["O"] < 2e-3 and report["L"] < 7e-3, report


def test_config4_code_object_forward_n8192_d256_bf16_mixed():
    """bench.py's fwd_bf16_d256_mixed: the FOLD stream of the 64-rows-per-wave D <= 256 kernel at N = 8192."""
    report, variants = _full_size_mixed(8192, 256, P.BF16, backward=False)
    assert variants["forward"].startswith("attn_fwd16p5") and variants["forward"].endswith("_fold"), variants
    print("full-size mixed", variants, report)
    assert report["O"] < 5e-3 and report["L"] < 7e-3, report


def test_d256_code_objects_forward_backward_n4096_bf16_mixed():
    """bench.py's fwdbwd_bf16_d256_mixed: the role-split backward streams attn_dq16_p5 / attn_dkv16_p5 (and the D <= 256 forward) at
    FULL size against the oracle, all six outputs -- the D = 128 test above covers the four-wave streams only"""
    report, variants = _full_size_mixed(4096, 256, P.BF16, backward=True, seed=3)
    assert variants["forward"].startswith("attn_fwd16p5") and variants["backwardQuery"].startswith("attn_dq16p5") and \
        variants["backwardKeyValue"].startswith("attn_dkv16p5"), variants
    print("full-size mixed", variants, report)
    assert max(report[k] for k in ("dQ", "dK", "dV")) < 2e-2 and report["D"] < 5e-2, report


def test_config2_code_object_forward_n4096_d64_bf16_mixed():
    """bench.py's fwd_bf16_d64 (BASELINE config 2 in the reference's mixed-precision mode): the persistent D <= 64 stream with the
    row sums in the matrix pipe (attn_fwd16_p6) at N = 4096: O and L against the oracle; l is the sum of the 16-bit P the second
    product multiplies (what the reference's mixed mode sums: P lives in 16-bit registers there)"""
    report, variants = _full_size_mixed(4096, 64, P.BF16, backward=False, seed=4)
    assert variants["forward"].startswith("attn_fwd16p6") and variants["forward"].endswith("_fold"), variants
    print("full-size mixed", variants, report)
    assert report["O"] < 5e-3 and report["L"] < 7e-3, report


def test_launch_form_names_what_runs():
    """mfa_attention_kernel_launch_form: the headline launch runs the persistent form (the kernel name rocprofv3 shows), per-batch
    lengths keep the one-block kernel, one long head is cut into pieces, a misaligned leading dimension falls to the general kernel"""
    import torch
    N, D, H = 1024, 128, 64
    desc = make_desc(N, N, D, low_in=True, low_mid=True, in_type=P.BF16)
    k = AttentionKernel(desc.kernelDescriptor(AttentionKernelType.forward))
    z = lambda *s_, dt=torch.bfloat16: torch.zeros(s_, device="cuda", dtype=dt)
    bufs = {Op.Q: z(H, N, D), Op.K: z(H, N, D), Op.V: z(H, N, D), Op.O: z(H, N, D, dt=torch.float32), Op.L: z(H, N, dt=torch.float16)}
    hs = {Op.Q: N * D, Op.K: N * D, Op.V: N * D, Op.O: N * D, Op.L: N}
    assert k.variant == "attn_fwd16p4_bf16_d128_w4x64_thr8_fold"
    assert k.launchForm(bufs, row=N, column=N, heads=H, headStrides=hs).startswith("attn_fwd16_p4p (persistent")
    assert "row-block pairs" in k.launchForm(bufs, row=N, column=N, heads=H, headStrides=hs, causal=True)
    lens = torch.full((1,), N, dtype=torch.int32, device="cuda")
    # (round 6: per-batch lengths run on the persistent kernel -- rows / keys per block-table entry; with the causal mask a workgroup's
    # fixed share of row-block pairs loses to the dispatcher on mixed lengths: those launches keep the one-block kernel)
    form = k.launchForm(bufs, row=N, column=N, heads=H, headStrides=hs, rowLengths=lens)
    assert form.startswith("attn_fwd16_p4p (persistent") and "per-batch lengths" in form, form
    assert k.launchForm(bufs, row=N, column=N, heads=H, headStrides=hs, rowLengths=lens, causal=True) == k.variant
    one = {op: t[0] for op, t in bufs.items()}
    ws = torch.empty(k.workspaceSize(row=N, column=N) + 256, dtype=torch.uint8, device="cuda")
    split_form = k.launchForm(one, row=N, column=N, workspace=ws)
    # (the pieces of the D = 128 forward split are the variant's OWN kernel, attn_fwd16_p4<..., split>: the text must not name the
    # eight-wave sibling -- round-5 verdict, weak item 7)
    # (round 6: and they run on the persistent kernel's split streams when the pieces are whole multiples of two tiles)
    assert split_form.startswith(k.variant + " column-parallel x") and "+ combine (pieces by attn_fwd16_p4p, persistent)" in split_form and "sibling" not in split_form, split_form
    assert "general kernel" in k.launchForm(bufs, row=N, column=N, heads=H, headStrides=hs, leadingDimensions={Op.K: D + 1})


def _fuzz_cases(count, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(count):
        D = int(rng.choice([8, 40, 64, 72, 96, 104, 112, 120, 128, 136, 152, 160, 176, 192, 200, 232, 256]))
        causal = bool(rng.integers(2))
        R = int(rng.integers(1, 700))
        C = int(rng.integers(R if causal else 1, 900))
        out.append((i, R, C, D, causal, bool(rng.integers(2)), "BF16" if rng.integers(2) else "FP16"))
    return out


@pytest.mark.parametrize("case", _fuzz_cases(50, seed=0), ids=lambda c: "%d-%dx%dx%d-%s-%s-%s" % (
    c[0], c[1], c[2], c[3], "causal" if c[4] else "dense", "mixed" if c[5] else "fp32mid", c[6]))
def test_fuzz_random_problems(case):
    """A seeded slice of tools/fuzz_shapes.py: random (R, C, D, causal, precision mode, 16-bit type) problems through all
    three kernels against the oracle, the reference's mixed tolerances, canary tails, no NaN."""
    i, R, C, D, causal, low_mid, in_type = case
    net = Network(NetworkDescriptor(R, C, D), seed=1000 + i)
    desc = make_desc(R, C, D, low_in=True, low_mid=low_mid, in_type=P[in_type])
    run = harness.DeviceRun(desc, net, causal=causal)
    got = run.execute()
    round_inputs(net, desc)
    ref = net.run(causal=causal)
    failures, report = harness.compare(ref, got, TOL_MIXED_SHORT if C <= 20 else TOL_MIXED)
    variants = [k.variant for k in run.kernels.values()]
    assert not failures, (failures, variants)
    assert all(run.tails_ok.values()), (run.tails_ok, variants)
    assert all(np.isfinite(got[n]).all() for n in ("O", "dQ", "dK", "dV")), variants


def _fuzz_transposed_cases(count, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(count):
        D = int(rng.choice([8, 40, 64, 72, 96, 104, 112, 120, 128, 136, 152, 160, 176, 192, 200, 232, 256]))
        causal = bool(rng.integers(2))
        R = int(rng.integers(1, 700))
        C = int(rng.integers(R if causal else 1, 900))
        tr = tuple(bool(b) for b in rng.integers(2, size=4))
        if not any(tr):
            tr = (True,) * 4
        g = (64, 8, 1)[i % 3]       # whole tiles / whole 16-byte chunks / anything: stream, aligned rows, gathered rows
        R, C = max(g, R // g * g), max(g, C // g * g)
        if causal and C < R:
            C = R
        out.append((i, R, C, D, causal, bool(rng.integers(2)), "BF16" if rng.integers(2) else "FP16", tr, bool(rng.integers(2))))
    return out


@pytest.mark.parametrize("case", _fuzz_transposed_cases(45, seed=5), ids=lambda c: "%d-%dx%dx%d-%s-%s" % (
    c[0], c[1], c[2], c[3], "causal" if c[4] else "dense", "".join("T" if t else "n" for t in c[7])))
def test_fuzz_random_transposed_problems(case):
    """A seeded slice of tools/fuzz_shapes.py --transposed (the mode that found the gather wrap): random problems with a random
    non-empty pattern of transposed (Q, K, V, O), forward, against the oracle; O and L, canary tails, no NaN."""
    i, R, C, D, causal, low_mid, in_type, tr, low_out = case
    net = Network(NetworkDescriptor(R, C, D), seed=2000 + i)
    desc = make_desc(R, C, D, low_in=True, low_mid=low_mid, in_type=P[in_type], tr=tr)
    desc.lowPrecisionOutputs = low_out
    run = harness.DeviceRun(desc, net, causal=causal, run_backward=False)
    k = run.kernels[AttentionKernelType.forward]
    assert "_tr" in k.variant or D % 8 != 0, k.variant
    got = run.execute()
    round_inputs(net, desc)
    ref = net.run(causal=causal, backward=False)
    tol = TOL_MIXED_SHORT if C <= 20 else TOL_MIXED
    failures, report = harness.compare(ref, got, {n: tol[n] for n in ("O", "L")})
    assert not failures, (failures, k.variant, k.launchForm(run.buffers, row=R, column=C, causal=causal))
    assert run.tails_ok["O"] and run.tails_ok["L"] and np.isfinite(got["O"]).all()


def test_size_independent_properties_at_full_size():
    """Properties that need no oracle: (i) V = 1 gives O = 1 exactly up to rounding (rows of P sum to
    one); (ii) permuting the keys (rows of K and V together) leaves O and L unchanged up to
    summation order -- this exercises tile boundaries at N=4096; (iii) O is linear in V."""
    import torch
    N, D = 4096, 128
    desc = make_desc(N, N, D, low_in=True, in_type=P.BF16)
    kernel = AttentionKernel(desc.kernelDescriptor(AttentionKernelType.forward))
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    q, k, v1, v2 = (torch.randn((N, D), generator=g, device="cuda").to(torch.bfloat16) for _ in range(4))

    def fwd(k_, v_):
        o = torch.full((N, D), float("nan"), device="cuda")
        l = torch.zeros(N, device="cuda")
        kernel.dispatch({Op.Q: q, Op.K: k_, Op.V: v_, Op.O: o, Op.L: l}, row=N, column=N,
                        stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        return o, l

    ones = torch.ones((N, D), device="cuda", dtype=torch.bfloat16)
    o, l = fwd(k, ones)
    assert (o - 1).abs().max().item() < 4e-3       # P is rounded to bf16 before the sum
    o1, l1 = fwd(k, v1)
    perm = torch.randperm(N, generator=g, device="cuda")
    op, lp = fwd(k[perm].contiguous(), v1[perm].contiguous())
    assert (op - o1).abs().max().item() < 3e-3 and (lp - l1).abs().max().item() < 1e-4
    # linearity: V1 + V2 rounded to bf16 once; compare against the same rounded sum
    vs = (v1.float() + v2.float()).to(torch.bfloat16)
    o2, _ = fwd(k, v2)
    os_, _ = fwd(k, vs)
    resid = (vs.float() - v1.float() - v2.float()).abs().max().item()
    assert (os_ - o1 - o2).abs().max().item() < resid + 5e-3


# ---- 16-bit matrix-core backward kernels (attn_dq16 / attn_dkv16) -----------------------------
BWD16_SHAPES = [(256, 256, 128), (300, 200, 128), (64, 64, 64), (255, 257, 64), (1, 100, 128), (100, 1, 64),
                (129, 77, 40), (96, 640, 80), (1024, 1024, 128), (513, 1030, 64), (256, 256, 256), (300, 333, 200), (65, 700, 256),
                (300, 333, 96), (257, 130, 88), (200, 449, 160), (129, 300, 152), (256, 320, 192), (100, 1000, 176),
                (200, 300, 104), (77, 530, 120)]


@pytest.mark.parametrize("low_mid", [False, True])
@pytest.mark.parametrize("dkv_impl", ["w4", "rs", "p4"])
@pytest.mark.parametrize("shape", BWD16_SHAPES)
def test_backward_16bit_mfma(shape, dkv_impl, low_mid, monkeypatch):
    """dkv_impl: one wave per key block / role-split wave pairs x 32 keys (attn_dkv16_rs.h) / the default rows: four waves x 64
    keys, hand-placed (attn_dkv16_p4.h) for D <= 128, two role-split pairs x 64 keys, hand-placed (attn_dkv16_p5.h) above; low_mid
    selects the streams with K pre-multiplied by the softmax scale, FP16 L, BF16 D.
    All three kernels on the BF16 matrix cores (Q, K, V, dO BF16): within the reference's mixed
    tolerances of the oracle fed with the rounded inputs, and within a tighter bound (2e-2 absolute on
    the gradients, whose dS is rounded to BF16 like the reference's register precision for dS,
    AttentionDescriptor+Precisions.swift:199-200)."""
    R, C, D = shape
    if dkv_impl == "w4" and (D > 128 or 64 < D <= 96):
        pytest.skip("the one-wave-per-key-block kernel exists for the 64 and 128 buckets only")
    if low_mid and dkv_impl == "w4":
        pytest.skip("covered with FP32 intermediates")
    net = Network(NetworkDescriptor(R, C, D), seed=7 * R + C + D)
    desc = make_desc(R, C, D, low_in=True, in_type=P.BF16, low_mid=low_mid)
    with parameter_rows(*({"w4": [DKV_W4], "rs": [DKV_RS, DQ_W4], "p4": []}[dkv_impl])):
        run = harness.DeviceRun(desc, net)
    variants = {t.name: k.variant for t, k in run.kernels.items()}
    assert ("attn_dq16p5" in variants["backwardQuery"]) == (D > 128 and dkv_impl != "rs"), variants   # attn_dq16_p5.h: buckets 160 / 192 / 256
    assert variants["backwardQuery"].startswith("attn_dq16") and variants["backwardKeyValue"].startswith("attn_dkv16"), variants
    assert ("attn_dkv16rs" in variants["backwardKeyValue"]) == (dkv_impl == "rs"), variants
    # default rows ("p4"): four waves x 64 keys up to D = 128 (attn_dkv16_p4.h), two role-split pairs x 64 keys above (attn_dkv16_p5.h)
    assert ("attn_dkv16p4" in variants["backwardKeyValue"]) == (dkv_impl == "p4" and D <= 128), variants
    assert ("attn_dkv16p5" in variants["backwardKeyValue"]) == (dkv_impl == "p4" and D > 128), variants
    assert ("attn_dq16p4" in variants["backwardQuery"]) == (D <= 128), variants   # attn_dq16_p4.h: buckets 64 and 128 (D in (64, 128])
    got = run.execute()
    round_inputs(net, desc)
    ref = net.run()
    failures, report = harness.compare(ref, got, TOL_MIXED_SHORT if C <= 20 else TOL_MIXED)
    assert not failures, (failures, variants)
    if not low_mid:   # FP16 L and BF16 D (mixed mode) alone cost more than this bound
        tight, report = harness.compare(ref, got, dict(O=1.5e-2, L=1e-3, D=2e-2, dV=2e-2, dK=2e-2, dQ=2e-2))
        assert not tight, (tight, variants)
    assert all(run.tails_ok.values()), run.tails_ok


@pytest.mark.parametrize("low_mid", [False, True])
@pytest.mark.parametrize("tr", [(True, True, True, True), (False, True, False, False), (False, False, True, True), (True, False, False, False)])
@pytest.mark.parametrize("shape", [(300, 449, 128), (130, 200, 64), (200, 150, 256), (257, 129, 104)])
def test_transposed_operands_reach_the_matrix_cores(shape, tr, low_mid):
    """transposeState (AttentionKernelDescriptor.swift:30-41) with 16-bit inputs.  Forward: a code object per pattern of (K, V)
    reads and writes the transposed operands IN PLACE (AttentionKernel.swift:189-204: no scratch) -- no workspace, whatever the
    leading dimensions (rows that are not 16-byte aligned are gathered).  Backward: given a workspace, every transposed operand
    is re-laid out row-major (inputs before, outputs after the launch) and the 16-bit matrix-core code object runs; without one
    the general kernel reads the transposed buffers in place.  All agree with the oracle; the canary tails stay intact."""
    R, C, D = shape
    net = Network(NetworkDescriptor(R, C, D), seed=R + 3 * C + D)
    desc = make_desc(R, C, D, low_in=True, low_mid=low_mid, in_type=P.BF16, tr=tr)
    run = harness.DeviceRun(desc, net)
    for t, k in run.kernels.items():
        assert not k.variant.startswith("attn_generic") and k.fallbackVariant.startswith("attn_generic")
        if t == AttentionKernelType.forward:
            assert not k.needsWorkspaceForFastPath and k.variant.startswith("attn_fwd16v3_bf16") and "_tr" in k.variant, k.variant
            assert k.workspaceSize(row=R, column=C) == 0
            form = k.launchForm(run.buffers, row=R, column=C)
            assert form.startswith("attn_fwd16v3_bf16") and "_tr" in form, form     # what really runs, without a workspace
        else:
            assert k.needsWorkspaceForFastPath
    got = run.execute(with_workspace=True)
    assert all(v > 0 for t, v in run.workspace_bytes.items() if t != AttentionKernelType.forward), run.workspace_bytes
    round_inputs(net, desc)
    ref = net.run()
    failures, report = harness.compare(ref, got, TOL_MIXED)
    assert not failures, (failures, [k.variant for k in run.kernels.values()])
    assert all(run.tails_ok.values()), run.tails_ok
    run2 = harness.DeviceRun(desc, net)
    slow = run2.execute()                      # no workspace: the same forward kernel, general backward kernels
    failures, report = harness.compare(ref, slow, TOL_MIXED)
    assert not failures, failures
    assert np.array_equal(got["O"], slow["O"]) and np.array_equal(got["L"], slow["L"])
    for name in ("dQ", "dK", "dV"):            # the two paths differ by the 16-bit rounding of P and dS only
        assert np.abs(got[name] - slow[name]).max() < 3e-2, name


@pytest.mark.parametrize("in_type", [P.BF16, P.FP16])
@pytest.mark.parametrize("pattern", range(1, 16))
def test_forward_reads_every_transposition_pattern_in_place(pattern, in_type):
    """All fifteen patterns of transposed (Q, K, V, O), head dimensions of every forward bucket, ragged edges (R, C not multiples
    of the 32-row / 64-key tiles; C % 8 != 0 cuts a 16-byte chunk of K^T / V^T), 16-bit and FP32 O, causal and not: the forward
    kernel runs its in-place code object (no workspace) and agrees with the oracle on the rounded inputs."""
    tr = tuple(bool(pattern & (1 << i)) for i in range(4))
    shapes = [(77, 203, 32), (130, 200, 64), (257, 129, 104), (96, 331, 128), (100, 140, 160), (65, 270, 192), (200, 150, 256)]
    R, C, D = shapes[pattern % len(shapes)]
    for causal, low_out in ((False, pattern % 2 == 0), (True, pattern % 3 == 0)):
        if causal:
            R = min(R, C)
        net = Network(NetworkDescriptor(R, C, D), seed=pattern + D)
        desc = make_desc(R, C, D, low_in=True, in_type=in_type, tr=tr)
        desc.lowPrecisionOutputs = low_out
        run = harness.DeviceRun(desc, net, run_backward=False, causal=causal)
        k = run.kernels[AttentionKernelType.forward]
        assert "_tr" in k.variant and not k.needsWorkspaceForFastPath, k.variant
        got = run.execute()
        round_inputs(net, desc)
        ref = net.run(backward=False, causal=causal)
        failures, report = harness.compare(ref, got, dict(O=1.5e-2 if not low_out else 3e-2, L=2e-3))
        assert not failures, (failures, k.variant, tr, (R, C, D), causal)
        assert all(run.tails_ok.values()), run.tails_ok


def test_transposed_rows_with_poisoned_padding():
    """K^T / V^T with a leading dimension beyond the sequence length (16-byte aligned rows) whose padding holds NaN, C % 8 != 0:
    the chunk that straddles the end of the sequence is cut to size (0 x NaN would poison O), chunks beyond it are never
    fetched.  Q^T / O^T with padded rows as well; the padding of O^T keeps its poison."""
    import torch
    R, C, D = 150, 203, 128
    ldq, ldk = 160, 208
    net = Network(NetworkDescriptor(R, C, D), seed=9)
    desc = make_desc(R, C, D, low_in=True, in_type=P.BF16, tr=(True, True, True, True))
    round_inputs(net, desc)
    kernel = AttentionKernel(desc.kernelDescriptor(AttentionKernelType.forward))
    assert "_tr_kv" in kernel.variant and not kernel.needsWorkspaceForFastPath

    def dev_t(x, ld):   # [seq][D] fp32 (bf16-representable) -> bf16 bits [D][ld], padding = NaN
        out = np.full((D, ld), 0x7FC0, np.uint16)
        out[:, :x.shape[0]] = (np.ascontiguousarray(x.T).view(np.uint32) >> 16).astype(np.uint16)
        return torch.from_numpy(out.view(np.int16)).cuda()
    bufs = {Op.Q: dev_t(net.Q, ldq), Op.K: dev_t(net.K, ldk), Op.V: dev_t(net.V, ldk),
            Op.O: torch.full((D, ldq), float("nan"), device="cuda"), Op.L: torch.full((R,), float("nan"), device="cuda")}
    lds = {Op.Q: ldq, Op.K: ldk, Op.V: ldk, Op.O: ldq}
    assert "_tr_kv" in kernel.launchForm(bufs, row=R, column=C, leadingDimensions=lds)
    kernel.dispatch(bufs, row=R, column=C, leadingDimensions=lds, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ref = net.run(backward=False)
    o = bufs[Op.O].cpu().numpy()
    assert np.abs(o[:, :R].T - ref["O"]).max() < 1.5e-2
    assert np.isnan(o[:, R:]).all(), "padding columns of the transposed O were written"
    assert np.abs(bufs[Op.L].cpu().numpy() / np.float32(harness.LOG2E) - ref["L"]).max() < 2e-3


@pytest.mark.parametrize("in_type", [P.BF16, P.FP16])
@pytest.mark.parametrize("low_mid", [False, True])
@pytest.mark.parametrize("kv", [(True, False), (False, True)])
def test_hand_placed_stream_with_one_transposed_operand(kv, low_mid, in_type):
    """K^T alone or V^T alone (aligned rows, D <= 128): the stream family of that pattern; Q / O both ways, ragged last tiles,
    head dimensions below the bucket, causal, 16-bit O."""
    for (R, C, D), causal, low_out, (tq, to) in (((296, 456, 128), False, False, (False, False)), ((256, 40, 128), False, True, (True, True)),
                                                 ((200, 328, 104), True, False, (True, False)), ((696, 1000, 80), True, True, (False, True)),
                                                 ((136, 2048, 128), False, False, (True, True))):
        net = Network(NetworkDescriptor(R, C, D), seed=R + C + 1)
        desc = make_desc(R, C, D, low_in=True, low_mid=low_mid, in_type=in_type, tr=(tq, kv[0], kv[1], to))
        desc.lowPrecisionOutputs = low_out
        run = harness.DeviceRun(desc, net, run_backward=False, causal=causal)
        k = run.kernels[AttentionKernelType.forward]
        form = k.launchForm(run.buffers, row=R, column=C, causal=causal)
        which = "transposed K" if kv[0] else "transposed V"
        assert k.variant.endswith("_tr_k" if kv[0] else "_tr_v") and form.startswith("attn_fwd16_p4_tr") and which in form and \
            ("folded" in form) == low_mid, (k.variant, form)
        got = run.execute()
        round_inputs(net, desc)
        ref = net.run(backward=False, causal=causal)
        failures, report = harness.compare(ref, got, dict(O=3e-2 if low_out else 1.5e-2, L=7e-3 if low_mid else 2e-3))
        assert not failures, (failures, (R, C, D), causal, form)
        assert all(run.tails_ok.values())


@pytest.mark.parametrize("low_mid", [False, True])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("in_type", [P.BF16, P.FP16])
def test_backward_kernels_read_transposed_operands_in_place(causal, low_mid, in_type, monkeypatch, capfd):
    """The backward streams on transposed operands (K^T / V^T in backwardQuery, Q^T / dO^T in backwardKeyValue; whole tiles,
    aligned rows) without a workspace (attn_dq16_p4_tr.h, attn_dkv16_p4_tr.h: transposed operands read where they lie,
    AttentionKernel.swift:189-204; MFA_BWD16_TR=0 is the developer library's A/B knob).  Every operand transposed; results
    against the oracle at the reference's mixed tolerances.  With FP16 inputs the reference's descriptors store dO in BF16
    (+Precisions.swift:13-17): backwardQuery converts the fragments when it loads them, backwardKeyValue runs the two products
    that read dO^T in BF16 (streams F16_DOBF16_*_TR)."""
    monkeypatch.setenv("MFA_BWD16_TR", "verbose")
    for R, C, D in ((320, 448, 128), (256, 256, 104)):
        net = Network(NetworkDescriptor(R, C, D), seed=R + C + D)
        desc = make_desc(R, C, D, low_in=True, low_mid=low_mid, in_type=in_type, tr=(True, True, True, True))
        run = harness.DeviceRun(desc, net, causal=causal)
        for t, name in ((AttentionKernelType.backwardQuery, "attn_dq16_p4_tr"), (AttentionKernelType.backwardKeyValue, "attn_dkv16_p4_tr")):
            assert run.kernels[t].launchForm(run.buffers, row=R, column=C, causal=causal).startswith(name)
        got = run.execute()                      # no workspace
        err = capfd.readouterr().err
        assert not DEV_LIBRARY or ("attn_dq16_p4_tr" in err and "attn_dkv16_p4_tr" in err), err   # (MFA_BWD16_TR=verbose: developer library)
        round_inputs(net, desc)
        ref = net.run(causal=causal)
        failures, report = harness.compare(ref, got, TOL_MIXED)
        assert not failures, (failures, (R, C, D))
        assert all(run.tails_ok.values()), run.tails_ok


@pytest.mark.parametrize("low_mid", [False, True])
@pytest.mark.parametrize("causal", [False, True])
def test_forward_stream_reads_transposed_keys_and_values_at_large_head_dimensions(causal, low_mid, monkeypatch, capfd):
    """The hand-placed forward stream of the 160 / 192 / 256 buckets on K^T / V^T in place (attn_fwd16_p5_tr.h; whole 32-key
    steps, aligned rows), attached to the transposed variants the way attn_fwd16_p4_tr is at D <= 128.  Q / O
    row-major and transposed, a head dimension inside each bucket and on its edge; against the oracle at the product tolerances of
    that path; MFA_FWD16_P5_TR=0 (A/B knob of the developer library) and a launch that is not whole steps keep the 8 x 32 object."""
    for (R, C, D), in_type, tr in (((320, 448, 256), P.BF16, (True, True, True, True)), ((300, 352, 152), P.BF16, (False, True, True, False)),
                                   ((256, 288, 192), P.FP16, (False, True, True, True)), ((264, 320, 232), P.FP16, (True, True, True, False)),
                                   # one operand transposed
                                   ((320, 448, 256), P.BF16, (False, True, False, False)), ((300, 352, 152), P.FP16, (True, False, True, True)),
                                   ((256, 288, 192), P.BF16, (True, True, False, True)), ((264, 320, 232), P.FP16, (False, False, True, False))):
        net = Network(NetworkDescriptor(R, C, D), seed=R + C + D)
        desc = make_desc(R, C, D, low_in=True, low_mid=low_mid, in_type=in_type, tr=tr)
        run = harness.DeviceRun(desc, net, run_backward=False, causal=causal)
        k = run.kernels[AttentionKernelType.forward]
        form = k.launchForm(run.buffers, row=R, column=C, causal=causal)
        suffix = {(True, True): "_tr_kv", (True, False): "_tr_k", (False, True): "_tr_v"}[(tr[1], tr[2])]
        assert k.variant.endswith(suffix) and form.startswith("attn_fwd16_p5_tr") and ("folded" in form) == low_mid, (k.variant, form)
        if DEV_LIBRARY:     # (the A/B knob exists in the developer library only)
            monkeypatch.setenv("MFA_FWD16_P5_TR", "0")
            assert k.launchForm(run.buffers, row=R, column=C, causal=causal).startswith("attn_fwd16v3")
            monkeypatch.delenv("MFA_FWD16_P5_TR")
        got = run.execute()
        round_inputs(net, desc)
        ref = net.run(backward=False, causal=causal)
        low_out = desc.memoryPrecisions[Op.O] != P.FP32
        failures, report = harness.compare(ref, got, dict(O=3e-2 if low_out else 1.5e-2, L=7e-3 if low_mid else 2e-3))
        assert not failures, (failures, (R, C, D), tr)
        assert all(run.tails_ok.values()), run.tails_ok


@pytest.mark.parametrize("causal", [False, True])
def test_transposed_multi_head_batches(causal):
    """Heads and batch entries of transposed operands ([batch][head][D][sequence]) through strides: the hand-placed stream on
    K^T / V^T (whole aligned tiles) and, with one more key, the 8 x 32 kernel's gather path -- every head against the oracle."""
    import torch
    B, H, D = 2, 3, 128
    for R, C, stream in ((320, 512, True), (320, 513, False)):
        desc = make_desc(R, C, D, low_in=True, in_type=P.BF16, tr=(True, True, True, True))
        kernel = AttentionKernel(desc.kernelDescriptor(AttentionKernelType.forward))
        rng = np.random.default_rng(R + C)
        host = {n: round_trip(rng.standard_normal((B, H, R if n == "Q" else C, D)).astype(np.float32), int(P.BF16)) for n in ("Q", "K", "V")}
        dev = lambda x: torch.from_numpy((np.ascontiguousarray(x.transpose(0, 1, 3, 2)).view(np.uint32) >> 16).astype(np.uint16).view(np.int16)).cuda()
        bufs = {Op.Q: dev(host["Q"]), Op.K: dev(host["K"]), Op.V: dev(host["V"]),
                Op.O: torch.full((B, H, D, R), float("nan"), device="cuda"), Op.L: torch.full((B, H, R), float("nan"), device="cuda")}
        hs = {Op.Q: R * D, Op.K: C * D, Op.V: C * D, Op.O: R * D, Op.L: R}
        bs = {op: v * H for op, v in hs.items()}
        args = dict(row=R, column=C, heads=H, batches=B, headStrides=hs, batchStrides=bs, causal=causal)
        assert kernel.launchForm(bufs, **args).startswith("attn_fwd16_p4_tr" if stream else "attn_fwd16v3")
        kernel.dispatch(bufs, stream=torch.cuda.current_stream().cuda_stream, **args)
        torch.cuda.synchronize()
        o, l = bufs[Op.O].cpu().numpy(), bufs[Op.L].cpu().numpy()
        for b in range(B):
            for h in range(H):
                net = Network(NetworkDescriptor(R, C, D), seed=0)
                net.Q, net.K, net.V = (np.ascontiguousarray(host[n][b, h]) for n in ("Q", "K", "V"))
                net.invalidate()
                ref = net.run(backward=False, causal=causal)
                assert np.abs(o[b, h].T - ref["O"]).max() < 1.5e-2, (b, h, stream)
                assert np.abs(l[b, h] / np.float32(harness.LOG2E) - ref["L"]).max() < 2e-3, (b, h, stream)


def test_transposed_gather_never_wraps_into_the_buffer():
    """K^T with rows that are not 16-byte aligned (odd C: gathered 16-bit loads) and a head dimension below its bucket (D = 72 in
    128): the image rows beyond D are fetched at offsets that SATURATE as the tiles advance -- adding the element offset to
    2^32 - 1 once wrapped into the buffer and read it at odd byte addresses (found by tools/fuzz_shapes.py --transposed: 5 of 240).
    K^T starts with bit patterns 0x807F (tiny negative numbers) whose bytes, read one byte off, are +Inf: 0 x Inf = NaN everywhere."""
    import torch
    R, C, D = 150, 507, 72
    net = Network(NetworkDescriptor(R, C, D), seed=12)
    desc = make_desc(R, C, D, low_in=True, in_type=P.BF16, tr=(False, True, False, False))
    round_inputs(net, desc)
    net.K[:32, 0] = np.array([0x807F0000], np.uint32).view(np.float32)[0]
    net.invalidate()
    run = harness.DeviceRun(desc, net, run_backward=False)
    assert run.kernels[AttentionKernelType.forward].variant.endswith("_tr_k")
    got = run.execute()
    ref = net.run(backward=False)
    failures, report = harness.compare(ref, got, dict(O=1.5e-2, L=2e-3))
    assert not failures and np.isfinite(got["O"]).all(), failures


@pytest.mark.parametrize("tr", [(False, False, False, True), (True, True, True, True), (False, True, False, True)])
def test_transposed_operands_with_lengths_leave_the_padding_alone(tr):
    """Transposed operands ([D][max sequence] per head) + per-batch lengths (padded batches): keys beyond a batch entry's length
    are neither multiplied nor summed although they lie INSIDE the rows of K^T / V^T (poison there), padding columns of O^T keep
    their poison (the kernel stores only rows < length), the rest equals the oracle.  (Round 2 sent such launches to the general
    kernel: a re-laid-out output copy would have overwritten the padding.)"""
    import torch
    B, H, Rmax, Cmax, D = 2, 2, 200, 336, 128
    rlen, clen = [200, 77], [333, 100]
    tq, tk, tv, to = tr
    desc = make_desc(Rmax, Cmax, D, low_in=True, in_type=P.BF16, tr=tr)
    kernel = AttentionKernel(desc.kernelDescriptor(AttentionKernelType.forward))
    assert not kernel.needsWorkspaceForFastPath and "_tr" in kernel.variant, kernel.variant
    rng = np.random.default_rng(5)
    host = {n: round_trip(rng.standard_normal((B, H, Rmax if n == "Q" else Cmax, D)).astype(np.float32), int(P.BF16)) for n in ("Q", "K", "V")}

    def dev(name, transposed):
        bits = (np.ascontiguousarray(host[name]).view(np.uint32) >> 16).astype(np.uint16)
        lens = rlen if name == "Q" else clen
        for b in range(B):   # poison beyond each batch entry's length
            bits[b, :, lens[b]:, :] = 0x7FC0
        if transposed:
            bits = np.ascontiguousarray(bits.transpose(0, 1, 3, 2))
        return torch.from_numpy(bits.view(np.int16)).cuda()
    bufs = {Op.Q: dev("Q", tq), Op.K: dev("K", tk), Op.V: dev("V", tv),
            Op.O: torch.full((B, H, D, Rmax) if to else (B, H, Rmax, D), float("nan"), device="cuda"),
            Op.L: torch.full((B, H, Rmax), float("nan"), device="cuda")}
    hs = {Op.Q: Rmax * D, Op.K: Cmax * D, Op.V: Cmax * D, Op.O: Rmax * D, Op.L: Rmax}
    bs = {op: v * H for op, v in hs.items()}
    args = dict(row=Rmax, column=Cmax, heads=H, batches=B, headStrides=hs, batchStrides=bs,
                rowLengths=torch.tensor(rlen, dtype=torch.int32, device="cuda"), columnLengths=torch.tensor(clen, dtype=torch.int32, device="cuda"))
    assert "_tr" in kernel.launchForm(bufs, **args)
    kernel.dispatch(bufs, stream=torch.cuda.current_stream().cuda_stream, **args)
    torch.cuda.synchronize()
    o = bufs[Op.O].cpu().numpy()
    for b in range(B):
        for h in range(H):
            R, C = rlen[b], clen[b]
            net = Network(NetworkDescriptor(R, C, D), seed=0)
            net.Q, net.K, net.V = (np.ascontiguousarray(host[n][b, h, :(R if n == "Q" else C)]) for n in ("Q", "K", "V"))
            net.invalidate()
            ref = net.run(backward=False)
            got = o[b, h].T if to else o[b, h]
            assert np.abs(got[:R] - ref["O"]).max() < 1.5e-2, (b, h)
            assert np.isnan(got[R:]).all(), "padding rows of O were written"


@pytest.mark.parametrize("causal", [False, True])
def test_headline_shape_transposed_in_place(causal):
    """N = 4096, D = 128, BF16, every operand transposed (`bench.py --workload fwd_bf16_d128_transposed`), one head: the in-place
    code object against the oracle on the rounded inputs, O and L."""
    R = C = 4096
    D = 128
    net = Network(NetworkDescriptor(R, C, D), seed=77)
    desc = make_desc(R, C, D, low_in=True, low_mid=True, in_type=P.BF16, tr=(True,) * 4)
    run = harness.DeviceRun(desc, net, run_backward=False, causal=causal)
    k = run.kernels[AttentionKernelType.forward]
    assert k.variant == "attn_fwd16v3_bf16_d128_w8x32_thr8_tr_kv" and not k.needsWorkspaceForFastPath
    assert k.launchForm(run.buffers, row=R, column=C, causal=causal).startswith("attn_fwd16_p4_tr")   # whole aligned tiles: the stream
    got = run.execute()
    round_inputs(net, desc)
    ref = net.run(backward=False, causal=causal)
    failures, report = harness.compare(ref, got, dict(O=1.5e-2, L=7e-3))
    assert not failures, failures
    assert all(run.tails_ok.values())


@pytest.mark.parametrize("in_type", [P.BF16, P.FP16])
@pytest.mark.parametrize("low_mid", [False, True])
@pytest.mark.parametrize("qo", [(False, False), (True, False), (False, True), (True, True)])
def test_hand_placed_stream_on_transposed_keys_and_values(qo, low_mid, in_type):
    """K^T and V^T in whole, 16-byte aligned tiles at D <= 128: the hand-placed stream reads them in place (attn_fwd16_p4_tr.h: the
    images keep the source orientation, the read recipes change places), Q and O either way; ragged row blocks and last tiles,
    head dimensions below the bucket, causal, 16-bit O, both scale modes.  Launches it cannot take (rows that are not 16-byte
    aligned) stay with the 8 x 32 kernel's transposed code object -- same answers."""
    for (R, C, D), causal, low_out in (((296, 448, 128), False, False), ((256, 64, 128), False, True), ((200, 320, 104), True, False),
                                       ((696, 1024, 80), True, True), ((136, 2048, 128), False, False),   # (rows of Q^T: 16-byte aligned)
                                       # partial last tiles (whole 16-byte chunks of keys): the last V^T tile through its own offsets
                                       ((296, 456, 128), False, False), ((256, 40, 128), False, True), ((200, 328, 104), True, False),
                                       ((696, 1000, 80), True, True), ((64, 72, 128), True, False)):
        net = Network(NetworkDescriptor(R, C, D), seed=R + C)
        desc = make_desc(R, C, D, low_in=True, low_mid=low_mid, in_type=in_type, tr=(qo[0], True, True, qo[1]))
        desc.lowPrecisionOutputs = low_out
        run = harness.DeviceRun(desc, net, run_backward=False, causal=causal)
        k = run.kernels[AttentionKernelType.forward]
        form = k.launchForm(run.buffers, row=R, column=C, causal=causal)
        assert "_tr_kv" in k.variant and form.startswith("attn_fwd16_p4_tr") and ("folded" in form) == low_mid, (k.variant, form)
        got = run.execute()
        round_inputs(net, desc)
        ref = net.run(backward=False, causal=causal)
        failures, report = harness.compare(ref, got, dict(O=3e-2 if low_out else 1.5e-2, L=7e-3 if low_mid else 2e-3))
        assert not failures, (failures, (R, C, D), causal, form)
        assert all(run.tails_ok.values())
    # rows of K^T / V^T (or Q^T) that do not begin on 16-byte boundaries: the 8 x 32 kernel
    for R, C in ((256, 201),) + (((300, 448),) if qo[0] else ()):
        desc = make_desc(R, C, 128, low_in=True, low_mid=low_mid, in_type=in_type, tr=(qo[0], True, True, qo[1]))
        run = harness.DeviceRun(desc, Network(NetworkDescriptor(R, C, 128), seed=1), run_backward=False)
        form = run.kernels[AttentionKernelType.forward].launchForm(run.buffers, row=R, column=C)
        assert form.startswith("attn_fwd16v3") and "_tr_kv" in form, form


def test_backward_16bit_matches_general_kernels(monkeypatch):
    """Same inputs through the fp32-arithmetic general kernels (reached through a layout the matrix-core kernels do not take:
    V and dV stored transposed) and the 16-bit matrix-core kernels: gradients agree to the 16-bit rounding of P and dS."""
    R, C, D = 384, 448, 128
    net = Network(NetworkDescriptor(R, C, D), seed=31)
    desc = make_desc(R, C, D, low_in=True, in_type=P.BF16)
    fast = harness.DeviceRun(desc, net).execute()
    run = harness.DeviceRun(make_desc(R, C, D, low_in=True, in_type=P.BF16, tr=(False, False, True, False)), net)
    for t, k in run.kernels.items():   # no workspace below: general backward kernels (the forward kernel reads V^T in place)
        assert "generic" in k.fallbackVariant and k.needsWorkspaceForFastPath == (t != AttentionKernelType.forward), (t, k.variant)
    slow = run.execute()
    for name in ("D", "dQ", "dK", "dV"):
        assert np.abs(fast[name] - slow[name]).max() < 2e-2, name


def test_backward_16bit_multi_head():
    import torch
    R, C, D, H, B = 130, 200, 64, 8, 2
    nets = [Network(NetworkDescriptor(R, C, D), seed=700 + i) for i in range(H * B)]
    desc = make_desc(R, C, D, low_in=True, in_type=P.BF16)
    kernels = {t: AttentionKernel(desc.kernelDescriptor(t)) for t in AttentionKernelType}

    def pack16(name):
        a = np.stack([getattr(n, name) for n in nets]).reshape(B, H, -1, D)
        return torch.from_numpy((np.ascontiguousarray(a).view(np.uint32) >> 16).astype(np.uint16).view(np.int16)).cuda()

    bufs = {Op.Q: pack16("Q"), Op.K: pack16("K"), Op.V: pack16("V"), Op.dO: pack16("dO")}
    for op, seq in ((Op.O, R), (Op.dQ, R), (Op.dK, C), (Op.dV, C)):
        bufs[op] = torch.full((B, H, seq, D), float("nan"), device="cuda")
    bufs[Op.L] = torch.zeros((B, H, R), device="cuda")
    bufs[Op.D] = torch.zeros((B, H, R), device="cuda")
    hs = {Op.Q: R * D, Op.K: C * D, Op.V: C * D, Op.O: R * D, Op.dO: R * D, Op.dQ: R * D, Op.dK: C * D,
          Op.dV: C * D, Op.L: R, Op.D: R}
    bs = {k: v * H for k, v in hs.items()}
    for t in AttentionKernelType:
        kernels[t].dispatch(bufs, row=R, column=C, heads=H, batches=B, headStrides=hs, batchStrides=bs,
                            stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for i, net in enumerate(nets):
        round_inputs(net, desc)
        ref = net.run()
        for op, name in ((Op.dQ, "dQ"), (Op.dK, "dK"), (Op.dV, "dV")):
            got = bufs[op].cpu().numpy().reshape(H * B, -1, D)[i]
            assert np.abs(got - ref[name]).max() < 2e-2, (i, name)


@pytest.mark.parametrize("shape", [(256, 256, 128), (255, 257, 64), (96, 640, 80), (1024, 1024, 128)])
@pytest.mark.parametrize("low_mid", [False, True])
def test_backward_reference_low_precision_mix_on_matrix_cores(shape, low_mid):
    """The reference's own low-precision mode -- FP16 Q/K/V with BF16 dO, and with
    lowPrecisionIntermediates FP16 L / BF16 D (AttentionDescriptor+Precisions.swift:13-17, :81-87) --
    through the 16-bit matrix-core kernels (dO converted to FP16 on load), checked with the
    reference's mixed tolerances against the UNROUNDED oracle, exactly as its tests do."""
    R, C, D = shape
    net = Network(NetworkDescriptor(R, C, D), seed=R + 3 * C + D)
    desc = make_desc(R, C, D, low_in=True, low_mid=low_mid, in_type=P.FP16)
    run = harness.DeviceRun(desc, net)
    variants = {t.name: k.variant for t, k in run.kernels.items()}
    assert "dObf16" in variants["backwardQuery"] and "dObf16" in variants["backwardKeyValue"], variants
    got = run.execute()
    ref = net.run()
    failures, report = harness.compare(ref, got, TOL_MIXED)
    assert not failures, (failures, variants)
    assert all(run.tails_ok.values())


# ---- column-parallel ("split-KV") forward through a caller-provided workspace ------------------
@pytest.mark.parametrize("shape,heads", [((4096, 4096, 64), 1), ((4096, 4096, 128), 1), ((300, 3000, 128), 2),
                                         ((129, 8200, 64), 1), ((64, 16384, 128), 1), ((300, 3000, 256), 1),
                                         ((129, 4200, 32), 2), ((300, 3000, 192), 1), ((200, 4100, 160), 2), ((520, 2100, 256), 1)])
def test_forward_split_kv_matches_unsplit_and_oracle(shape, heads):
    """Same answer with and without the workspace (to the rounding of a different summation order),
    and both within the tight forward bounds of the oracle."""
    import torch
    R, C, D = shape
    desc = make_desc(R, C, D, low_in=True, in_type=P.BF16)
    kernel = AttentionKernel(desc.kernelDescriptor(AttentionKernelType.forward))
    need = kernel.workspaceSize(row=R, column=C, heads=heads)
    assert need > 0
    nets = [Network(NetworkDescriptor(R, C, D), seed=40 + i) for i in range(heads)]

    def pack16(name):
        a = np.stack([getattr(n, name) for n in nets])
        return torch.from_numpy((np.ascontiguousarray(a).view(np.uint32) >> 16).astype(np.uint16).view(np.int16)).cuda()

    q, k, v = pack16("Q"), pack16("K"), pack16("V")
    hs = {Op.Q: R * D, Op.K: C * D, Op.V: C * D, Op.O: R * D, Op.L: R}
    outs = []
    for ws in (None, torch.empty(need + 64, dtype=torch.uint8, device="cuda")):
        o = torch.full((heads, R, D), float("nan"), device="cuda")
        l = torch.zeros((heads, R), device="cuda")
        if ws is not None:   # the hand-placed kernels (D > 64) cut the key range themselves
            form = kernel.launchForm({Op.Q: q, Op.K: k, Op.V: v, Op.O: o, Op.L: l}, row=R, column=C, heads=heads, headStrides=hs, workspace=ws)
            assert "column-parallel x" in form and form.startswith(kernel.variant), (form, kernel.variant)
        kernel.dispatch({Op.Q: q, Op.K: k, Op.V: v, Op.O: o, Op.L: l}, row=R, column=C, heads=heads,
                        headStrides=hs, stream=torch.cuda.current_stream().cuda_stream, workspace=ws)
        torch.cuda.synchronize()
        outs.append((o.cpu().numpy(), l.cpu().numpy() / np.float32(harness.LOG2E)))
    assert np.abs(outs[0][0] - outs[1][0]).max() < 2e-3 and np.abs(outs[0][1] - outs[1][1]).max() < 1e-4
    for i, net in enumerate(nets):
        round_inputs(net, desc)
        ref = net.run(backward=False)
        assert np.abs(outs[1][0][i] - ref["O"]).max() < 1.5e-2
        assert np.abs(outs[1][1][i] - ref["L"]).max() < 1e-3


def test_forward_split_kv_too_small_workspace_is_ignored():
    import torch
    R, C, D = 256, 4096, 64
    net = Network(NetworkDescriptor(R, C, D), seed=3)
    desc = make_desc(R, C, D, low_in=True, in_type=P.BF16)
    run = harness.DeviceRun(desc, net, run_backward=False)
    kernel = run.kernels[AttentionKernelType.forward]
    assert kernel.workspaceSize(row=R, column=C) > 0
    tiny = torch.empty(1024, dtype=torch.uint8, device="cuda")
    kernel.dispatch(run.buffers, row=R, column=C, stream=torch.cuda.current_stream().cuda_stream, workspace=tiny)
    torch.cuda.synchronize()
    got = run.results()
    round_inputs(net, desc)
    assert np.abs(got["O"] - net.run(backward=False)["O"]).max() < 1.5e-2


# ---- causal mask (extension) --------------------------------------------------------------------
CAUSAL_SHAPES = [(64, 64, 32), (100, 100, 40), (33, 97, 16), (1, 50, 8), (300, 300, 128), (257, 600, 64), (129, 129, 200)]


@pytest.mark.parametrize("shape", CAUSAL_SHAPES)
def test_causal_fp32_all_kernels(shape):
    """Row r attends column c iff c <= r + (C - R).  FP32, all six outputs against the causal oracle
    with the reference's FP32 tolerance."""
    R, C, D = shape
    net = Network(NetworkDescriptor(R, C, D), seed=R + C)
    desc = make_desc(R, C, D)
    run = harness.DeviceRun(desc, net, causal=True)
    got = run.execute()
    ref = net.run(causal=True)
    failures, report = harness.compare(ref, got, TOL_FP32)
    assert not failures, failures
    assert all(run.tails_ok.values())


@pytest.mark.parametrize("shape", [(256, 256, 128), (300, 555, 64), (1024, 1024, 128), (129, 640, 80), (320, 448, 256), (300, 555, 32),
                                   (1024, 1024, 256)])
def test_causal_bf16_all_kernels(shape):
    R, C, D = shape
    net = Network(NetworkDescriptor(R, C, D), seed=2 * R + C)
    desc = make_desc(R, C, D, low_in=True, in_type=P.BF16)
    run = harness.DeviceRun(desc, net, causal=True)
    got = run.execute()
    round_inputs(net, desc)
    ref = net.run(causal=True)
    variants = [k.variant for k in run.kernels.values()]
    # the 16-bit matrix-core kernels own the mask (the backward pair exists for D = 64, 128, 256)
    assert D % 8 or not variants[0].startswith("attn_generic"), variants
    assert D not in (64, 128, 256) or all(not v.startswith("attn_generic") for v in variants), variants
    # D = sum dO*O is O(sqrt(D_head)) for the first causal rows (they average over very few keys), so
    # its absolute error is larger than in the unmasked tests; the reference's own bound for D is 1e-1
    failures, report = harness.compare(ref, got, dict(O=1.5e-2, L=1e-3, D=5e-2, dV=2e-2, dK=2e-2, dQ=2e-2))
    assert not failures, (failures, variants)


def test_causal_requires_column_ge_row():
    from metal_flash_attention_amd import MFAError
    desc = make_desc(64, 32, 16)
    k = AttentionKernel(desc.kernelDescriptor(AttentionKernelType.forward))
    with pytest.raises(MFAError) as e:
        k.dispatch({Op.Q: 4096, Op.K: 4096, Op.V: 4096, Op.O: 4096, Op.L: 4096}, row=64, column=32, causal=True)
    assert e.value.status == 2


# ---- role-alternating forward kernel (attn_fwd16_v4), selected through the developer knob ---------
V4_SHAPES = [(256, 256, 128), (300, 300, 128), (1, 64, 128), (257, 130, 120), (96, 640, 80), (1024, 1024, 128),
             (255, 257, 64), (64, 1, 64), (129, 77, 40), (2048, 2048, 64)]


@needs_dev_library
@pytest.mark.parametrize("impl", ["v4:0", "v4:16", "v3:41"])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("shape", V4_SHAPES)
def test_forward_role_alternating_kernel(shape, causal, impl, monkeypatch):
    monkeypatch.setenv("MFA_FWD16_IMPL", impl)
    R, C, D = shape
    if causal and C < R:
        pytest.skip("causal needs column >= row")
    if causal and impl != "v4:0":
        pytest.skip("only schedule 0 carries the causal code object")
    net = Network(NetworkDescriptor(R, C, D), seed=R + 3 * C + D)
    desc = make_desc(R, C, D, low_in=True, in_type=P.BF16)
    run = harness.DeviceRun(desc, net, run_backward=False, causal=causal)
    variant = run.kernels[AttentionKernelType.forward].variant
    if not variant.startswith("attn_fwd16v4") and "_kpad" not in variant:
        pytest.skip(f"schedule {impl} is not compiled for this head dimension ({variant})")
    got = run.execute()
    round_inputs(net, desc)
    ref = net.run(backward=False, causal=causal)
    failures, report = harness.compare(ref, got, dict(O=1.5e-2, L=1e-3))
    assert not failures, failures
    assert run.tails_ok["O"] and run.tails_ok["L"] and not np.isnan(got["O"]).any()


# ---- variable sequence lengths per batch entry (extension) -------------
@pytest.mark.parametrize("low,causal,D", [(False, False, 64), (False, True, 64), (True, False, 64), (True, True, 64),
                                          (True, False, 128), (True, True, 128), (True, False, 256), (True, True, 200),
                                          (True, False, 384), (True, True, 320), (True, True, 384), (True, False, 296)])
def test_variable_sequence_lengths(low, causal, D):
    """A padded batch [B, H, Rmax, D] / [B, H, Cmax, D] with per-entry lengths: every entry must equal the
    oracle run on its own (rows, columns) slice, and nothing beyond an entry's length may be written.  D = 128 / 200 / 256:
    the hand-placed streams (their workgroups cover 256 rows / keys: entries shorter than a workgroup, empty workgroups)."""
    import torch
    # (last entry: fewer columns than rows -- with `causal` its diagonal offset is clamped at 0, include/mfa.h rowLengths)
    B, H, Rmax, Cmax = 5, 2, 200, 333
    rlen = [200, 77, 1, 130, 150]
    clen = [333, 100, 64, 130, 90]
    in_type = P.BF16
    desc = make_desc(Rmax, Cmax, D, low_in=low, in_type=in_type)
    kernels = {t: AttentionKernel(desc.kernelDescriptor(t)) for t in AttentionKernelType}
    rng = np.random.default_rng(11)
    host = {n: rng.standard_normal((B, H, Rmax if n in ("Q", "dO") else Cmax, D)).astype(np.float32) for n in ("Q", "K", "V", "dO")}
    if low:
        for n in host:
            host[n] = round_trip(host[n], int(P.BF16))

    def dev_in(x):
        if not low:
            return torch.from_numpy(x).cuda()
        return torch.from_numpy((np.ascontiguousarray(x).view(np.uint32) >> 16).astype(np.uint16).view(np.int16)).cuda()

    bufs = {Op.Q: dev_in(host["Q"]), Op.K: dev_in(host["K"]), Op.V: dev_in(host["V"]), Op.dO: dev_in(host["dO"])}
    poison = float("nan")
    bufs[Op.O] = torch.full((B, H, Rmax, D), poison, device="cuda")
    bufs[Op.L] = torch.full((B, H, Rmax), poison, device="cuda")
    bufs[Op.D] = torch.full((B, H, Rmax), poison, device="cuda")
    bufs[Op.dQ] = torch.full((B, H, Rmax, D), poison, device="cuda")
    bufs[Op.dK] = torch.full((B, H, Cmax, D), poison, device="cuda")
    bufs[Op.dV] = torch.full((B, H, Cmax, D), poison, device="cuda")
    hs = {Op.Q: Rmax * D, Op.K: Cmax * D, Op.V: Cmax * D, Op.O: Rmax * D, Op.L: Rmax, Op.D: Rmax,
          Op.dO: Rmax * D, Op.dV: Cmax * D, Op.dK: Cmax * D, Op.dQ: Rmax * D}
    bs = {op: s * H for op, s in hs.items()}
    rl = torch.tensor(rlen, dtype=torch.int32, device="cuda")
    cl = torch.tensor(clen, dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    if low and D == 128 and not causal:   # (round 6) the persistent forward kernel serves per-batch lengths: rows / keys per block-table entry
        form = kernels[AttentionKernelType.forward].launchForm(bufs, row=Rmax, column=Cmax, heads=H, batches=B, headStrides=hs, batchStrides=bs,
                                                                causal=causal, rowLengths=rl, columnLengths=cl)
        assert form.startswith("attn_fwd16_p4p (persistent") and "per-batch lengths" in form, form
    if low and D > 256:    # (round 6) 256 < D <= 384: all three kernels on the 16-bit matrix cores, per-batch lengths included
        assert kernels[AttentionKernelType.forward].variant.startswith("attn_fwd16w_")
        assert kernels[AttentionKernelType.backwardQuery].variant.startswith("attn_dq16w_")
        assert kernels[AttentionKernelType.backwardKeyValue].variant.startswith("attn_dkv16w_")
        for t, k in kernels.items():
            form = k.launchForm(bufs, row=Rmax, column=Cmax, heads=H, batches=B, headStrides=hs, batchStrides=bs, causal=causal, rowLengths=rl, columnLengths=cl)
            assert form.startswith(k.variant), form
    if low and D == 64:    # (round 6) so does the D <= 64 persistent kernel (FP32 L here: its exact-scale geometry streams)
        form = kernels[AttentionKernelType.forward].launchForm(bufs, row=Rmax, column=Cmax, heads=H, batches=B, headStrides=hs, batchStrides=bs,
                                                                causal=causal, rowLengths=rl, columnLengths=cl)
        assert form.startswith("attn_fwd16_p6 (persistent") and "per-batch lengths" in form, form
    for t in (AttentionKernelType.forward, AttentionKernelType.backwardQuery, AttentionKernelType.backwardKeyValue):
        kernels[t].dispatch(bufs, row=Rmax, column=Cmax, heads=H, batches=B, headStrides=hs, batchStrides=bs,
                            stream=stream, causal=causal, rowLengths=rl, columnLengths=cl)
    torch.cuda.synchronize()
    out = {op: bufs[op].cpu().numpy() for op in (Op.O, Op.L, Op.D, Op.dQ, Op.dK, Op.dV)}
    tol = dict(O=1.5e-2, L=1e-3, D=5e-2, dV=2e-2, dK=2e-2, dQ=2e-2) if low else TOL_FP32
    for b in range(B):
        R, C = rlen[b], clen[b]
        for h in range(H):
            net = Network(NetworkDescriptor(R, C, D), seed=0)
            net.Q, net.K, net.V, net.dO = (np.ascontiguousarray(host[n][b, h, :(R if n in ("Q", "dO") else C)]) for n in ("Q", "K", "V", "dO"))
            net.invalidate()
            ref = net.run(causal=causal)
            got = dict(O=out[Op.O][b, h, :R], L=out[Op.L][b, h, :R] / np.float32(harness.LOG2E),
                       D=out[Op.D][b, h, :R] * np.sqrt(np.float32(D)), dQ=out[Op.dQ][b, h, :R],
                       dK=out[Op.dK][b, h, :C], dV=out[Op.dV][b, h, :C])
            failures, report = harness.compare(ref, got, tol)
            assert not failures, (b, h, failures, [k.variant for k in kernels.values()])
            # padding untouched
            for op, n in ((Op.O, R), (Op.L, R), (Op.D, R), (Op.dQ, R), (Op.dK, C), (Op.dV, C)):
                assert np.isnan(out[op][b, h, n:]).all(), (op, b, h)


# ---- fused 16-bit output cast (extension) ------------------------------
@pytest.mark.parametrize("in_type", [P.BF16, P.FP16])
@pytest.mark.parametrize("shape,causal", [((256, 256, 128), False), ((300, 555, 64), False), ((129, 77, 40), False),
                                          ((1024, 1024, 128), True), ((100, 1000, 128), False), ((200, 300, 384), False), ((300, 300, 320), True)])
def test_low_precision_outputs(shape, causal, in_type):
    """O, dQ, dK, dV stored directly in the inputs' 16-bit type by the matrix-core kernels (no separate cast
    pass); backwardQuery reads the 16-bit O for its D term.  Reference mixed tolerances."""
    R, C, D = shape
    net = Network(NetworkDescriptor(R, C, D), seed=R + 2 * C + D)
    desc = make_desc(R, C, D, low_in=True, in_type=in_type)
    desc.lowPrecisionOutputs = True
    prec = desc.memoryPrecisions
    assert all(prec[op] == in_type for op in (Op.O, Op.dQ, Op.dK, Op.dV))
    run = harness.DeviceRun(desc, net, causal=causal)
    variants = [k.variant for k in run.kernels.values()]
    if D % 8 == 0:
        assert all(not v.startswith("attn_generic") for v in variants), variants
    got = run.execute()
    round_inputs(net, desc)
    ref = net.run(causal=causal)
    failures, report = harness.compare(ref, got, TOL_MIXED)
    assert not failures, (failures, variants)
    assert all(run.tails_ok.values()), run.tails_ok


def test_low_precision_outputs_split_kv():
    import torch
    R, C, D = 256, 8192, 64
    net = Network(NetworkDescriptor(R, C, D), seed=5)
    desc = make_desc(R, C, D, low_in=True, in_type=P.BF16)
    desc.lowPrecisionOutputs = True
    run = harness.DeviceRun(desc, net, run_backward=False)
    kernel = run.kernels[AttentionKernelType.forward]
    need = kernel.workspaceSize(row=R, column=C)
    assert need > 0
    ws = torch.empty(need, dtype=torch.uint8, device="cuda")
    kernel.dispatch(run.buffers, row=R, column=C, stream=torch.cuda.current_stream().cuda_stream, workspace=ws)
    torch.cuda.synchronize()
    got = run.results()
    round_inputs(net, desc)
    assert np.abs(got["O"] - net.run(backward=False)["O"]).max() < 2e-2


# ---- block-sparse mask (extension) ------------------------------------
def _random_block_mask(R, C, density, rng, keep_empty_row=False):
    rb, cb = (R + 255) // 256, (C + 127) // 128
    bits = rng.random((rb, cb)) < density
    for i in range(rb):
        if not bits[i].any() and not (keep_empty_row and i == rb - 1):
            bits[i, rng.integers(0, cb)] = True
    if keep_empty_row:
        bits[rb - 1, :] = False
    words = (cb + 31) // 32
    packed = np.zeros((rb, words), np.uint32)
    for i in range(rb):
        for j in range(cb):
            if bits[i, j]:
                packed[i, j // 32] |= np.uint32(1) << np.uint32(j % 32)
    return bits, packed, words


@pytest.mark.parametrize("low", [False, True])
@pytest.mark.parametrize("shape,causal,empty", [((600, 900, 64), False, False), ((1024, 1024, 128), True, False),
                                                ((700, 1300, 128), False, True), ((300, 200, 40), False, False),
                                                ((520, 1100, 256), False, False), ((768, 768, 256), True, False),
                                                ((520, 700, 32), False, False)])
def test_block_sparse_mask(shape, causal, empty, low):
    """All three kernels under a 256 x 128 block mask, against the fp64 matrix-form oracle with the same mask;
    a row block with no active block at all must give O = 0 and zero gradients."""
    import torch
    from oracle.network_np import attention_f64, block_mask_to_dense
    R, C, D = shape
    rng = np.random.default_rng(R + C)
    bits, packed, words = _random_block_mask(R, C, 0.4, rng, keep_empty_row=empty)
    net = Network(NetworkDescriptor(R, C, D), seed=R * 3 + C)
    desc = make_desc(R, C, D, low_in=low, in_type=P.BF16)
    run = harness.DeviceRun(desc, net)
    mask_dev = torch.from_numpy(packed.view(np.int32)).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    for t, kernel in run.kernels.items():
        kernel.dispatch(run.buffers, row=R, column=C, stream=stream, causal=causal, blockMask=mask_dev, blockMaskWords=words)
    torch.cuda.synchronize()
    got = run.results()
    if low:
        round_inputs(net, desc)
    ref = attention_f64(net.Q, net.K, net.V, net.dO, causal=causal, mask=block_mask_to_dense(bits, R, C))
    if low and D % 8 == 0:   # the matrix-core kernels own the mask: no general fallback
        assert all(not k.variant.startswith("attn_generic") for k in run.kernels.values())
    rows_alive = np.isfinite(ref["L"])
    tol = dict(O=1.5e-2, D=5e-2, dV=2e-2, dK=2e-2, dQ=2e-2) if low else dict(O=2e-5, D=2e-5, dV=3e-5, dK=3e-5, dQ=3e-5)
    for name, bound in tol.items():
        err = np.abs(got[name] - ref[name]).max()
        assert err < bound, (name, float(err), [k.variant for k in run.kernels.values()])
    assert np.abs(got["L"][rows_alive] - ref["L"][rows_alive]).max() < (1e-3 if low else 2e-5)
    if empty:
        dead = ~rows_alive
        assert dead.any() and (got["O"][dead] == 0).all() and (got["dQ"][dead] == 0).all() and (got["L"][dead] < -1e30).all()
    assert all(run.tails_ok.values())


# ---- traversal-parallel backward launches through a caller-provided workspace ------------------------
@pytest.mark.parametrize("shape,causal", [((4096, 4096, 64), False), ((2048, 4096, 128), False), ((3000, 3000, 128), True),
                                          ((1024, 2048, 256), False), ((2100, 1500, 192), False), ((1500, 2100, 256), True),
                                          ((4096, 4096, 256), False)])
def test_backward_split_matches_unsplit_and_oracle(shape, causal):
    """Single-head backward launches cannot fill 256 CUs: with a workspace the key (dQ) / row (dK, dV) range is cut
    into pieces whose fp32 partial results are summed by a second kernel.  Same answer as without the
    workspace (to the rounding of a different summation order) and within the tight bounds of the oracle."""
    import torch
    R, C, D = shape
    net = Network(NetworkDescriptor(R, C, D), seed=R + C + D + 1)
    desc = make_desc(R, C, D, low_in=True, in_type=P.BF16)
    run = harness.DeviceRun(desc, net, causal=causal)
    base = run.execute()
    stream = torch.cuda.current_stream().cuda_stream
    used = 0
    for t in (AttentionKernelType.backwardQuery, AttentionKernelType.backwardKeyValue):
        k = run.kernels[t]
        need = k.workspaceSize(row=R, column=C)
        assert need > 0, (t, k.variant)
        used += 1
        ws = torch.empty(need + 64, dtype=torch.uint8, device="cuda")
        outputs = (Op.dQ,) if t == AttentionKernelType.backwardQuery else (Op.dK, Op.dV)
        for op in outputs:   # poison this kernel's outputs: the split path must rewrite all of them
            run.buffers[op][: run.buffers[op].numel() // 2].fill_(0x7F)
        k.dispatch(run.buffers, row=R, column=C, stream=stream, causal=causal, workspace=ws)
        form = k.launchForm(run.buffers, row=R, column=C, causal=causal, workspace=ws)
        assert "column-parallel x" in form, form
        # the hand-placed kernels (D <= 128: attn_dq16_p4 / attn_dkv16_p4; D > 128: attn_dq16_p5 / attn_dkv16_p5, round 6) cut dense
        # launches into pieces themselves, causal ones belong to their siblings (32-row waves / 32-key role-split pairs)
        assert ("sibling" in form) == (causal and ("p4" in k.variant or "16p5" in k.variant)), form
    torch.cuda.synchronize()
    got = run.results()
    # the unsplit launch of the 128 bucket is the four-wave hand-placed kernel, the split one its 8 x 32 / role-split sibling:
    # P and dS are rounded to BF16 at different values of the same expression (both within the oracle bounds below)
    other_kernel = any("p4" in run.kernels[t].variant for t in (AttentionKernelType.backwardQuery, AttentionKernelType.backwardKeyValue))
    for name in ("dQ", "dK", "dV", "D"):
        assert np.abs(got[name] - base[name]).max() < (1.5e-2 if other_kernel and name != "D" else 2e-3), name
    round_inputs(net, desc)
    ref = net.run(causal=causal)
    failures, report = harness.compare(ref, got, dict(D=5e-2, dV=2e-2, dK=2e-2, dQ=2e-2))
    assert not failures, failures
    assert all(run.tails_ok.values())


@pytest.mark.parametrize("shape,heads", [((1024, 1024, 128), 2), ((4096, 4096, 128), 1), ((300, 449, 64), 3)])
def test_launches_are_capturable_in_a_hip_graph(shape, heads):
    """The three launches of a training step -- persistent forward, hand-placed backward, and (one head) their column-parallel
    pieces + combine passes through a workspace -- captured into ONE hipGraph on a side stream and replayed: the library makes
    no allocation, no synchronisation and no host-side decision that depends on device data, so a replay on new inputs gives
    what eager launches give, bit for bit."""
    import torch
    R, C, D = shape
    desc = make_desc(R, C, D, low_in=True, low_mid=True, in_type=P.BF16)
    kernels = {t: AttentionKernel(desc.kernelDescriptor(t)) for t in AttentionKernelType}
    prec = desc.memoryPrecisions
    tdt = {P.FP32: torch.float32, P.FP16: torch.float16, P.BF16: torch.bfloat16}
    seq = {Op.Q: R, Op.K: C, Op.V: C, Op.O: R, Op.dO: R, Op.dV: C, Op.dK: C, Op.dQ: R}
    g = torch.Generator(device="cuda"); g.manual_seed(5)

    def fresh_inputs(bufs):
        for op in (Op.Q, Op.K, Op.V, Op.dO):
            bufs[op].copy_((torch.randn((heads, seq[op], D), generator=g, device="cuda") * (0.1 if op == Op.dO else 1)).to(tdt[prec[op]]))
    bufs = {op: torch.zeros((heads, n, D), device="cuda", dtype=tdt[prec[op]]) for op, n in seq.items()}
    bufs[Op.L] = torch.zeros((heads, R), device="cuda", dtype=tdt[prec[Op.L]])
    bufs[Op.D] = torch.zeros((heads, R), device="cuda", dtype=tdt[prec[Op.D]])
    hs = {op: n * D for op, n in seq.items()}
    hs[Op.L] = hs[Op.D] = R
    ws = {t: torch.empty(max(k.workspaceSize(row=R, column=C, heads=heads), 256), dtype=torch.uint8, device="cuda") for t, k in kernels.items()}

    def step(stream):
        for t in (AttentionKernelType.forward, AttentionKernelType.backwardQuery, AttentionKernelType.backwardKeyValue):
            kernels[t].dispatch(bufs, row=R, column=C, heads=heads, headStrides=hs, stream=stream, workspace=ws[t])
    outs = (Op.O, Op.L, Op.D, Op.dQ, Op.dK, Op.dV)
    fresh_inputs(bufs)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step(side.cuda_stream)   # warm-up outside the capture (first launch of a code object raises its LDS limit)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        step(torch.cuda.current_stream().cuda_stream)
    for round_ in range(2):
        fresh_inputs(bufs)
        for op in outs:
            bufs[op].zero_()
        graph.replay()
        torch.cuda.synchronize()
        replayed = {op: bufs[op].clone() for op in outs}
        for op in outs:
            bufs[op].zero_()
        step(torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        for op in outs:
            assert torch.equal(replayed[op].view(torch.uint8), bufs[op].view(torch.uint8)), (op, round_)
        assert torch.isfinite(bufs[Op.O].float()).all() and bufs[Op.dQ].float().abs().max() > 0
    if heads == 1 and R >= 4096:
        assert all("column-parallel" in kernels[t].launchForm(bufs, row=R, column=C, heads=heads, headStrides=hs, workspace=ws[t]) for t in kernels)


# ---- head dimensions above 256 (the reference's large-D rows, +Parameters.swift:77-285; accumulator paging of
# +Accumulate.swift:449-467 re-derived: accumulators stay in the 512 registers, the left-hand operands leave them) ----
@pytest.mark.parametrize("shape", [(200, 200, 320), (130, 130, 384), (77, 150, 384), (33, 260, 264), (100, 64, 352)])
def test_large_head_dimension_fp32(shape):
    report, run = run_case(*shape, seed=sum(shape), tolerances={k: 5e-5 for k in TOL_FP32})
    assert all("d384" in k.variant for k in run.kernels.values()), [k.variant for k in run.kernels.values()]


@pytest.mark.parametrize("in_type", [P.BF16, P.FP16])
@pytest.mark.parametrize("low_mid", [False, True])
@pytest.mark.parametrize("shape,causal", [((150, 170, 320), False), ((64, 300, 384), False), ((300, 333, 264), False), ((257, 600, 352), True),
                                          ((512, 512, 384), True), ((129, 40, 384), False)])
def test_large_head_dimension_16bit_inputs(shape, causal, low_mid, in_type):
    """Round 6: 16-bit inputs at 256 < D <= 384 run all three kernels on the 16-bit matrix cores -- forward attn_fwd16_wide.h (head
    blocks 320 / 384, four waves x 32 rows: the `| 384 | ... |` rows of the reference's mixed tables, AttentionDescriptor+Parameters.swift:113,
    :120), backwardQuery attn_dq16 with 32-key tiles, backwardKeyValue attn_dkv16_wide.h (role-split pairs, two LDS stages).  Oracle on
    the rounded inputs, tolerances of the other 16-bit kernels."""
    R, C, D = shape
    net = Network(NetworkDescriptor(R, C, D), seed=5 + D)
    desc = make_desc(R, C, D, low_in=True, low_mid=low_mid, in_type=in_type)
    run = harness.DeviceRun(desc, net, causal=causal)
    fwd = run.kernels[AttentionKernelType.forward]
    assert fwd.variant.startswith("attn_fwd16w_") and ("_d320_" if D <= 320 else "_d384_") in fwd.variant, fwd.variant
    assert fwd.blockDimensions == (128, 32, 320 if D <= 320 else 384) and fwd.threadgroupMemoryAllocation == 3 * (32 * ((320 if D <= 320 else 384) * 2 + 16) + 32 * (320 if D <= 320 else 384) * 2)
    hb = 320 if D <= 320 else 384
    dq, dkv = run.kernels[AttentionKernelType.backwardQuery], run.kernels[AttentionKernelType.backwardKeyValue]
    assert dq.variant.startswith("attn_dq16w_") and "_d%d_" % hb in dq.variant and dq.blockDimensions == (128, 32, hb), dq.variant
    assert dkv.variant.startswith("attn_dkv16w_") and "_d%d_" % hb in dkv.variant and dkv.blockDimensions == (64, 32, hb), dkv.variant
    assert dkv.threadgroupMemoryAllocation == 2 * (2 * 32 * hb * 2 + 256) + 2 * 4096
    assert dq.launchForm(run.buffers, row=R, column=C, causal=causal).startswith("attn_dq16w_")
    got = run.execute()
    round_inputs(net, desc)
    ref = net.run(causal=causal)
    tol = dict(O=1.5e-2, L=7e-3 if low_mid else 1e-3, D=1e-1 if low_mid else 5e-2, dV=2e-2, dK=2e-2, dQ=2e-2)
    failures, report = harness.compare(ref, got, tol)
    assert not failures, (failures, report)
    assert all(run.tails_ok.values())


@pytest.mark.parametrize("tr", [(True, True, True, True), (False, True, False, False), (True, False, True, False)])
@pytest.mark.parametrize("shape,causal", [((200, 300, 384), False), ((260, 260, 320), True)])
def test_large_head_dimension_16bit_transposed_operands(shape, causal, tr):
    """256 < D <= 384 with transposeState: these head blocks have no in-place code objects -- given a workspace every transposed
    operand is re-laid out row-major (inputs before, outputs after the launch) and the 16-bit matrix-core code object runs, the
    forward kernel included (the other buckets' forward kernels read transposed operands in place); without one the general kernel."""
    R, C, D = shape
    net = Network(NetworkDescriptor(R, C, D), seed=R + C + D)
    desc = make_desc(R, C, D, low_in=True, in_type=P.BF16, tr=tr)
    run = harness.DeviceRun(desc, net, causal=causal)
    for t, k in run.kernels.items():
        assert k.variant.startswith(("attn_fwd16w_", "attn_dq16w_", "attn_dkv16w_")) and k.needsWorkspaceForFastPath, (t, k.variant)
        assert k.workspaceSize(row=R, column=C) > 0
        assert "general kernel" in k.launchForm(run.buffers, row=R, column=C, causal=causal)     # (no workspace in this query)
    got = run.execute(with_workspace=True)
    round_inputs(net, desc)
    ref = net.run(causal=causal)
    failures, report = harness.compare(ref, got, TOL_MIXED)
    assert not failures, (failures, [k.variant for k in run.kernels.values()])
    assert all(run.tails_ok.values()), run.tails_ok
    slow = harness.DeviceRun(desc, net, causal=causal).execute()      # no workspace: the general kernels in place
    failures, report = harness.compare(ref, slow, TOL_MIXED)
    assert not failures, failures


def test_large_head_dimension_16bit_batched_lengths_and_fused_output_cast():
    """attn_fwd16_wide: heads and batches with strides, per-batch lengths (an entry shorter than one workgroup, one with fewer keys
    than a step), causal with C > R, and O stored in the inputs' 16-bit type by the kernel itself."""
    import torch
    B, H, Rmax, Cmax, D = 3, 2, 200, 333, 384
    rlen, clen = [200, 77, 130], [333, 20, 300]
    for causal, low_out in ((False, False), (True, True)):
        desc = make_desc(Rmax, Cmax, D, low_in=True, in_type=P.BF16)
        desc.lowPrecisionOutputs = low_out
        kernel = AttentionKernel(desc.kernelDescriptor(AttentionKernelType.forward))
        assert kernel.variant == "attn_fwd16w_bf16_d384_w4x32_thr8", kernel.variant
        rng = np.random.default_rng(31)
        host = {n: round_trip(rng.standard_normal((B, H, Rmax if n == "Q" else Cmax, D)).astype(np.float32), int(P.BF16)) for n in ("Q", "K", "V")}
        pack = lambda x: torch.from_numpy((np.ascontiguousarray(x).view(np.uint32) >> 16).astype(np.uint16).view(np.int16)).cuda()
        bufs = {Op.Q: pack(host["Q"]), Op.K: pack(host["K"]), Op.V: pack(host["V"]),
                Op.O: torch.full((B, H, Rmax, D), float("nan"), device="cuda", dtype=torch.bfloat16 if low_out else torch.float32),
                Op.L: torch.full((B, H, Rmax), float("nan"), device="cuda")}
        hs = {Op.Q: Rmax * D, Op.K: Cmax * D, Op.V: Cmax * D, Op.O: Rmax * D, Op.L: Rmax}
        bs = {op: v * H for op, v in hs.items()}
        rl, cl = (torch.tensor(x, dtype=torch.int32, device="cuda") for x in (rlen, clen))
        kernel.dispatch(bufs, row=Rmax, column=Cmax, heads=H, batches=B, headStrides=hs, batchStrides=bs, rowLengths=rl, columnLengths=cl,
                        causal=causal, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        o, l = bufs[Op.O].float().cpu().numpy(), bufs[Op.L].cpu().numpy() / np.float32(harness.LOG2E)
        for b in range(B):
            R, C = rlen[b], clen[b]
            for h in range(H):
                net = Network(NetworkDescriptor(R, C, D), seed=0)
                net.Q, net.K, net.V = (np.ascontiguousarray(host[n][b, h, :(R if n == "Q" else C)]) for n in ("Q", "K", "V"))
                net.invalidate()
                ref = net.run(backward=False, causal=causal)
                assert np.abs(o[b, h, :R] - ref["O"]).max() < (3e-2 if low_out else 1.5e-2), (b, h, causal)
                assert np.abs(l[b, h, :R] - ref["L"]).max() < 1e-3, (b, h, causal)
                assert np.isnan(o[b, h, R:]).all() and np.isnan(l[b, h, R:]).all()


@pytest.mark.parametrize("tr", [(False,) * 4, (True, False, True, False), (True, True, True, True)])
@pytest.mark.parametrize("shape", [(100, 130, 392), (64, 200, 512), (33, 77, 777), (257, 65, 1000)])
def test_any_head_dimension_fp32(shape, tr):
    """D > 384: the reference falls through to its tables' last row and pages the accumulators through the output buffers
    (+Parameters.swift:60-65, +Accumulate.swift:403-469).  So do attn_paged_*: every head dimension, fp32 arithmetic, the
    reference's fp32 tolerance scaled by sqrt(D / 384) for the longer dot products."""
    R, C, D = shape
    tol = 5e-5 * max(1.0, (D / 384.0) ** 0.5)
    report, run = run_case(R, C, D, seed=sum(shape), tolerances={k: tol for k in TOL_FP32}, tr=tr)
    assert all(k.variant.startswith("attn_paged_") and k.variant.endswith("_any_d") for k in run.kernels.values()), [k.variant for k in run.kernels.values()]
    for t, k in run.kernels.items():     # nothing cached, accumulators paged: what the effective descriptor reports
        assert k.blockDimensions == (32, 32, 64)


@pytest.mark.parametrize("in_type", [P.BF16, P.FP16])
@pytest.mark.parametrize("causal", [False, True])
def test_any_head_dimension_16bit_inputs_and_causal(in_type, causal):
    R, C, D = 150, 170, 448
    net = Network(NetworkDescriptor(R, C, D), seed=9 + D)
    desc = make_desc(R, C, D, low_in=True, in_type=in_type)
    run = harness.DeviceRun(desc, net, causal=causal)
    assert all(k.variant.startswith("attn_paged_") for k in run.kernels.values())
    got = run.execute()
    round_inputs(net, desc)
    ref = net.run(causal=causal)
    failures, report = harness.compare(ref, got, dict(O=1e-4, L=1e-4, D=2e-2, dV=2e-2, dK=2e-2, dQ=2e-2))
    assert not failures, failures
    assert all(run.tails_ok.values())


def test_any_head_dimension_needs_fp32_outputs():
    """the accumulators are paged through the OUTPUT buffers, which are always FP32 in the reference (+Precisions.swift:140-143):
    the fused 16-bit output cast (this library's extension) is not available beyond D = 384, and says so"""
    from metal_flash_attention_amd import MFAError
    desc = make_desc(64, 64, 392, low_in=True, in_type=P.BF16)
    desc.lowPrecisionOutputs = True
    with pytest.raises(MFAError) as e:
        AttentionKernel(desc.kernelDescriptor(AttentionKernelType.forward))
    assert e.value.status == 3 and "must be FP32" in str(e.value)
