"""GPU tests of the code-object families no other test selects (found by the variant-coverage map, tests/test_variant_coverage.py):
FP16 gradients next to FP16 Q / K / V (a caller's override of the kernel descriptor's memory precisions -- the reference's own
descriptors store dO in BF16, +Precisions.swift:13-17), every (K, V) transposition pattern x head-dimension bucket x 16-bit type
of the forward kernel, the eight-wave / four-wave 32-row forward objects at every bucket, and the column-parallel / block-sparse
siblings the hand-placed streams keep.  All against the oracle at the reference's mixed tolerances."""
import numpy as np
import pytest

import harness
from metal_flash_attention_amd import (AttentionDescriptor, AttentionKernel, AttentionKernelType, AttentionOperand as Op,
                                       GEMMOperandPrecision as P)
from oracle import Network, NetworkDescriptor, round_trip
from test_attention_gpu import DKV_RS, DKV_W4, DQ_W4, FWD_8x32, TOL_MIXED, make_desc, parameter_rows

pytestmark = pytest.mark.gpu

FWD_X32 = (AttentionKernelType.forward, FWD_8x32[1], FWD_8x32[2] + "| 160 | 128 | 32 | 160 | Q, O |\n| 192 | 128 | 32 | 192 | Q, O |\n| 256 | 128 | 32 | 256 | Q, O |\n")


def _rounded(net, precisions):
    net.Q, net.K, net.V, net.dO = (round_trip(a, int(precisions[o])) for a, o in ((net.Q, Op.Q), (net.K, Op.K), (net.V, Op.V), (net.dO, Op.dO)))
    net.invalidate()


@pytest.mark.parametrize("low_mid", [False, True])
@pytest.mark.parametrize("rows", ["default", "rs", "w4"])
@pytest.mark.parametrize("D", [64, 96, 128, 160, 192, 256, 320, 384])
@pytest.mark.parametrize("grad", [P.FP16, P.BF16])
def test_fp16_inputs_with_either_gradient_type(D, rows, low_mid, grad):
    """FP16 Q / K / V with dO stored in FP16 (descriptor override) or BF16 (the reference's mix), every backward code-object family"""
    if rows == "w4" and (D not in (64, 128) or low_mid):
        pytest.skip("the one-wave-per-key-block kernel exists for the 64 and 128 buckets, FP32 intermediates")
    if D > 256 and rows != "default":
        pytest.skip("one backward code-object family above 256 (attn_bwd16_wide.hip)")
    R, C = 200, 328
    net = Network(NetworkDescriptor(R, C, D), seed=3 * R + C + D)
    desc = make_desc(R, C, D, low_in=True, low_mid=low_mid, in_type=P.FP16)
    with parameter_rows(*({"default": [], "rs": [DKV_RS, DQ_W4], "w4": [DKV_W4]}[rows])):
        run = harness.DeviceRun(desc, net, memory_overrides={Op.dO: grad})
    names = {t.name: k.variant for t, k in run.kernels.items()}
    tag = "_f16_d" if grad == P.FP16 else "_f16_dObf16_d"
    assert tag in names["backwardQuery"] and tag in names["backwardKeyValue"], names
    if rows == "rs":
        assert "attn_dkv16rs" in names["backwardKeyValue"], names
    if rows == "w4":
        assert names["backwardKeyValue"].startswith("attn_dkv16_f16"), names
    got = run.execute()
    _rounded(net, run.precisions)
    failures, report = harness.compare(net.run(), got, TOL_MIXED)
    assert not failures, (failures, names)
    assert all(run.tails_ok.values()), run.tails_ok


@pytest.mark.parametrize("in_type", [P.BF16, P.FP16])
@pytest.mark.parametrize("pattern", [(True, False, False, True), (False, True, False, False), (True, False, True, False), (False, True, True, True)])
@pytest.mark.parametrize("D", [32, 64, 128, 160, 192, 256])
def test_forward_transposition_patterns_at_every_bucket(D, pattern, in_type):
    """transposeState (Q, K, V, O): the code object of each (K, V) pattern -- _tr (Q / O only), _tr_k, _tr_v, _tr_kv -- at every
    head-dimension bucket and both 16-bit types, read in place; a key count that is not whole steps of aligned rows keeps the 8 x 32 /
    4 x 32 object itself (the hand-placed streams take the aligned launches and are tested elsewhere)"""
    R, C = 136, 203
    net = Network(NetworkDescriptor(R, C, D), seed=R + C + D)
    desc = make_desc(R, C, D, low_in=True, in_type=in_type, tr=pattern)
    run = harness.DeviceRun(desc, net, run_backward=False)
    k = run.kernels[AttentionKernelType.forward]
    suffix = {(False, False): "_tr", (True, False): "_tr_k", (False, True): "_tr_v", (True, True): "_tr_kv"}[(pattern[1], pattern[2])]
    assert k.variant.endswith(suffix) and ("_bf16_" if in_type == P.BF16 else "_f16_") in k.variant and "_d%d_" % D in k.variant, k.variant
    assert k.launchForm(run.buffers, row=R, column=C).startswith("attn_fwd16v3"), k.launchForm(run.buffers, row=R, column=C)
    got = run.execute()
    _rounded(net, run.precisions)
    failures, report = harness.compare(net.run(backward=False), got, dict(O=1.5e-2, L=2e-3))
    assert not failures, (failures, k.variant)
    assert all(run.tails_ok.values()), run.tails_ok


@pytest.mark.parametrize("in_type", [P.BF16, P.FP16])
@pytest.mark.parametrize("D", [128, 160, 192, 256])
def test_forward_32_row_wave_objects_at_every_bucket(D, in_type):
    """| D | 128 or 256 | 32 | D | rows select the compiler-scheduled 32-row-wave objects (attn_fwd16_v3.h) as the variant itself"""
    R, C = 300, 449
    net = Network(NetworkDescriptor(R, C, D), seed=R + C + D + 1)
    desc = make_desc(R, C, D, low_in=True, in_type=in_type)
    with parameter_rows(FWD_X32):
        run = harness.DeviceRun(desc, net, run_backward=False)
    k = run.kernels[AttentionKernelType.forward]
    assert k.variant.startswith("attn_fwd16v3_") and "_d%d_" % D in k.variant, k.variant
    got = run.execute()
    _rounded(net, run.precisions)
    failures, report = harness.compare(net.run(backward=False), got, dict(O=1.5e-2, L=2e-3))
    assert not failures, (failures, k.variant)


@pytest.mark.parametrize("in_type,grad", [(P.BF16, None), (P.FP16, None), (P.FP16, P.FP16)])
@pytest.mark.parametrize("kind", ["split", "sparse"])
@pytest.mark.parametrize("D", [64, 128, 160, 256])
def test_launch_form_names_the_sibling_code_object(D, kind, in_type, grad):
    """column-parallel and block-sparse launches of a hand-placed variant run the sibling kernel's code objects: the launch form says
    which (mfa_attention_kernel_launch_form), and the results are the oracle's"""
    import torch
    from metal_flash_attention_amd.torch_binding import pack_block_mask
    R = C = 1024
    net = Network(NetworkDescriptor(R, C, D), seed=D + 11)
    desc = make_desc(R, C, D, low_in=True, in_type=in_type)
    run = harness.DeviceRun(desc, net, memory_overrides=({Op.dO: grad} if grad is not None else None))
    _rounded(net, run.precisions)
    stream = torch.cuda.current_stream().cuda_stream
    forms = {}
    for t, k in run.kernels.items():
        kw = dict(row=R, column=C)
        if kind == "split":
            need = k.workspaceSize(row=R, column=C)
            kw["workspace"] = torch.empty(max(need, 16), dtype=torch.uint8, device="cuda")
        else:
            bits = torch.ones((R // 256, C // 128), dtype=torch.bool, device="cuda")
            mask = pack_block_mask(bits)
            kw.update(blockMask=mask, blockMaskWords=int(mask.shape[-1]))
        forms[t.name] = k.launchForm(run.buffers, **kw)
        k.dispatch(run.buffers, stream=stream, **kw)
    torch.cuda.synchronize()
    failures, report = harness.compare(net.run(), run.results(), TOL_MIXED)
    assert not failures, (failures, forms)
    if D <= 128 and kind == "sparse":
        assert any("sibling attn_" in f for f in forms.values()), forms
