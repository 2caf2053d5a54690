"""Parity anchors that do not depend on this repository's restatement of the reference: hand-derivable closed forms of
the reference's CPU network (tests/golden/closed_form.py cites Network.swift per formula).

  * CPU (`-m "not gpu"`): the C oracle (oracle/network.c) must reproduce every closed form -- this is what pins the
    oracle; plus the key-permutation property and the committed C1 fixture (BASELINE config 1: N=128, D=64, fp32).
  * GPU (`-m gpu`): the HIP kernels, through the C ABI, must reproduce the same closed forms at BASELINE's C1, C2 and
    C3 shapes, fp32 and bf16.
"""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
import closed_form  # noqa: E402
from oracle import Network, NetworkDescriptor, round_trip  # noqa: E402

FP32_TOL = 2e-5   # SquareAttentionTest.swift:539-554


def _oracle(inputs, threads=0, causal=False):
    R, D = inputs["Q"].shape
    C = inputs["K"].shape[0]
    net = Network(NetworkDescriptor(R, C, D), seed=0, threads=threads)
    net.Q, net.K, net.V, net.dO = (np.ascontiguousarray(inputs[n], np.float32) for n in ("Q", "K", "V", "dO"))
    return net.run(backward=True, causal=causal)


def _assert_close(got, exp, tol, what, inputs=None):
    for name, e in exp.items():
        scale = max(1.0, float(np.abs(e).max()))
        if inputs is not None and name in ("dQ", "dK"):
            # dQ = dS K / sqrt(D) (dK = dS^T Q / sqrt(D)): the rounding of dS = P (dP - D) is multiplied by the largest key
            # (query) entry -- the dominant-key case has a key of length 60 sqrt(D) / 4 by construction
            other = inputs["K" if name == "dQ" else "Q"]
            scale *= max(1.0, float(np.abs(other).max()) / np.sqrt(other.shape[1]))
        if name in ("dK", "dQ", "dV"):
            # sums of R (or C) fp32 terms that cancel (dS = P (dP - D) with |dP|, |D| ~ sqrt(D)): the rounding floor grows
            # with the square root of the number of terms -- the reference's 2e-5 was set on N <= 777 random problems
            scale *= max(1.0, np.sqrt(max(np.asarray(got["dQ"]).shape[0], np.asarray(got["dK"]).shape[0])) / 10.0)
        err = float(np.abs(np.asarray(got[name], np.float64) - e).max())
        assert err <= tol * scale, f"{what}: {name} differs from the closed form by {err:.3e} (scale {scale:.3g}, tol {tol:g})"


@pytest.mark.parametrize("shape", [(128, 128, 64), (37, 53, 20), (256, 300, 128), (1, 9, 3)])
@pytest.mark.parametrize("case", closed_form.CASES)
def test_oracle_reproduces_closed_forms(case, shape):
    inputs, exp = closed_form.build(case, *shape, seed=11)
    _assert_close(_oracle(inputs), exp, FP32_TOL, f"oracle {case} {shape}")


def test_oracle_closed_forms_at_c2_shape():
    """BASELINE config 2's shape (N=4096, D=64): the oracle against the closed forms, all host threads"""
    for case in ("q_zero", "dominant_key"):
        inputs, exp = closed_form.build(case, 4096, 4096, 64, seed=12)
        _assert_close(_oracle(inputs), exp, 5e-5, f"oracle {case} C2")


def test_oracle_key_permutation_property():
    rng = np.random.default_rng(5)
    R, C, D = 40, 77, 24
    inputs = {n: rng.standard_normal(s).astype(np.float32) for n, s in (("Q", (R, D)), ("K", (C, D)), ("V", (C, D)), ("dO", (R, D)))}
    perm = rng.permutation(C)
    a = _oracle(inputs)
    b = _oracle(dict(inputs, K=inputs["K"][perm], V=inputs["V"][perm]))
    for name in ("O", "L", "D", "dQ"):
        assert np.abs(a[name] - b[name]).max() < 1e-5, name
    for name in ("dK", "dV"):
        assert np.abs(a[name][perm] - b[name]).max() < 1e-5, name


def _random_inputs(R, C, D, seed):
    rng = np.random.default_rng(seed)
    return {n: rng.standard_normal(sh).astype(np.float32) for n, sh in (("Q", (R, D)), ("K", (C, D)), ("V", (C, D)), ("dO", (R, D)))}


def _assert_relation(base_out, new_out, expect, tol, what, scale_by=None, relative=False):
    for name in ("O", "L", "D", "dV", "dK", "dQ"):
        want = expect(name, np.asarray(base_out[name], np.float64))
        err = float(np.abs(np.asarray(new_out[name], np.float64) - want).max())
        bound = tol[name] if isinstance(tol, dict) else tol
        if scale_by is not None:
            bound *= scale_by.get(name, 1.0)
        if relative:      # 16-bit storage of an output (FP16 L, BF16 D): the rounding step grows with the magnitude
            bound *= max(1.0, float(np.abs(want).max()))
        assert err <= bound, f"{what}: {name} violates the relation by {err:.3e} (bound {bound:.3g})"


@pytest.mark.parametrize("shape", [(128, 128, 64), (37, 53, 20), (256, 300, 128), (4096, 4096, 64)])
@pytest.mark.parametrize("relation", closed_form.RELATIONS)
def test_oracle_metamorphic_relations_on_random_inputs(relation, shape):
    """Non-degenerate anchors: RANDOM Q, K, V, dO (P neither uniform nor one-hot); what Network.swift:134-402 implies for the
    transformed problem must hold between two runs of the C oracle.  (4096, 4096, 64) is BASELINE config 2's shape."""
    R, C, D = shape
    base, new, expect = closed_form.transform(relation, _random_inputs(R, C, D, seed=31), seed=R + D)
    a, b = _oracle(base), _oracle(new)
    if relation == "qk_rescale":     # exact powers of two: bit-identical
        for name in ("O", "L", "D", "dV", "dK", "dQ"):
            assert np.array_equal(np.asarray(b[name]), expect(name, np.asarray(a[name])).astype(np.float32)), name
        return
    # fp32 rounding floor of sums of N cancelling terms (as in _assert_close)
    grow = max(1.0, np.sqrt(max(R, C)) / 10.0)
    _assert_relation(a, b, expect, FP32_TOL, f"oracle {relation} {shape}",
                     scale_by=dict(dQ=4 * grow, dK=4 * grow, dV=grow, D=4.0, O=2.0, L=2.0))


def test_c1_fixture_single_thread_and_all_cores():
    """BASELINE config 1 (forward, one head, N=128, D=64, fp32, CPU only): the committed fixture is reproduced bit for
    bit by one thread (the unparallelised reference's analogue) and by all host threads."""
    path = os.path.join(os.path.dirname(__file__), "golden", "network_golden.npz")
    g = np.load(path)
    cases = [tuple(int(x) for x in c) for c in g["cases"]]
    assert (10, 128, 128, 64) in cases, "C1 (seed 10, N=128, D=64) missing from tests/golden/network_golden.npz"
    for threads in (1, 0):
        net = Network(NetworkDescriptor(128, 128, 64), seed=10, threads=threads)
        res = net.run(backward=True)
        for name in ("O", "L", "D", "dV", "dK", "dQ"):
            assert np.array_equal(res[name], g[f"s10_{name}"]), (threads, name)


# ------------------------------------------------------------------------------------------------ GPU
def _device(inputs, R, C, D, low_in=False, in_type=None):
    import harness
    from test_attention_gpu import make_desc
    from metal_flash_attention_amd import GEMMOperandPrecision as P
    desc = make_desc(R, C, D, low_in=low_in, in_type=in_type if in_type is not None else P.FP16)
    run = harness.DeviceRun(desc, closed_form.FixedNetwork(inputs), seed=77)
    got = run.execute()
    assert all(run.tails_ok.values()), run.tails_ok
    return got, [k.variant for k in run.kernels.values()]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(128, 128, 64), (300, 520, 128), (777, 640, 96)])
@pytest.mark.parametrize("case", closed_form.CASES)
def test_hip_fp32_reproduces_closed_forms(case, shape):
    inputs, exp = closed_form.build(case, *shape, seed=21)
    R, D = inputs["Q"].shape
    got, variants = _device(inputs, R, inputs["K"].shape[0], D)
    _assert_close(got, exp, FP32_TOL, f"HIP fp32 {case} {shape} {variants}", inputs)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["q_zero", "dominant_key", "keys_equal"])
def test_hip_fp32_closed_forms_at_c3_shape(case):
    """BASELINE config 3 (forward + backward, N=4096, D=128, fp32)"""
    inputs, exp = closed_form.build(case, 4096, 4096, 128, seed=22)
    got, variants = _device(inputs, 4096, 4096, 128)
    _assert_close(got, exp, 1e-4, f"HIP fp32 {case} C3 {variants}", inputs)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(4096, 4096, 64), (1024, 4096, 128), (256, 256, 128)])
@pytest.mark.parametrize("case", closed_form.CASES)
def test_hip_bf16_reproduces_closed_forms(case, shape):
    """BASELINE config 2's shape (N=4096, D=64, bf16) and the headline head dimension on the 16-bit matrix cores; the
    closed form is evaluated on the bf16-rounded inputs, the tolerances are the reference's mixed-precision ones
    (SquareAttentionTest.swift:539-554) scaled by the magnitude of each expected tensor."""
    from harness import TOL_MIXED
    from metal_flash_attention_amd import GEMMOperandPrecision as P
    inputs, exp = closed_form.build(case, *shape, seed=23, quantize=lambda x: round_trip(x, int(P.BF16)))
    R, D = inputs["Q"].shape
    got, variants = _device(inputs, R, inputs["K"].shape[0], D, low_in=True, in_type=P.BF16)
    for name, e in exp.items():
        scale = max(1.0, float(np.abs(e).max()))
        err = float(np.abs(np.asarray(got[name], np.float64) - e).max())
        assert err <= TOL_MIXED[name] * scale, f"HIP bf16 {case} {shape} {variants}: {name} err {err:.3e} (scale {scale:.3g})"


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(128, 128, 64), (300, 520, 128), (4096, 4096, 128)])
@pytest.mark.parametrize("relation", closed_form.RELATIONS)
def test_hip_fp32_metamorphic_relations_on_random_inputs(relation, shape):
    """The same relations between two launches of the HIP kernels on RANDOM fp32 inputs ((4096, 4096, 128) = BASELINE config 3)."""
    R, C, D = shape
    base, new, expect = closed_form.transform(relation, _random_inputs(R, C, D, seed=41), seed=R + D)
    a, va = _device(base, R, C, D)
    b, vb = _device(new, R, C, D)
    grow = max(1.0, np.sqrt(max(R, C)) / 10.0)
    tol = 0.0 if relation == "qk_rescale" else FP32_TOL
    _assert_relation(a, b, expect, tol, f"HIP fp32 {relation} {shape} {va}",
                     scale_by=dict(dQ=4 * grow, dK=4 * grow, dV=grow, D=4.0, O=2.0, L=2.0))


@pytest.mark.gpu
@pytest.mark.parametrize("low_mid", [False, True])
@pytest.mark.parametrize("shape", [(4096, 4096, 64), (4096, 4096, 128), (1000, 777, 256)])
@pytest.mark.parametrize("relation", closed_form.RELATIONS)
def test_hip_bf16_metamorphic_relations_on_random_inputs(relation, shape, low_mid):
    """The 16-bit matrix-core kernels (BASELINE config 2's shape, the headline head dimension, the D = 256 kernels; both precision
    modes): inputs and shift vectors on a 1/8 grid so that every transformed operand is exact in bf16.  qk_rescale must be
    bit-identical (every product is scaled by an exact power of two, the folded scale included); the shifts move O / L / D by
    the predicted amounts within the 16-bit kernels' own error (P rounded to bf16, tile order)."""
    import harness
    from test_attention_gpu import make_desc
    from metal_flash_attention_amd import GEMMOperandPrecision as P
    R, C, D = shape
    base, new, expect = closed_form.transform(relation, _random_inputs(R, C, D, seed=51), seed=R + D, grid=0.125)
    for x in list(base.values()) + list(new.values()):
        assert np.array_equal(round_trip(x, int(P.BF16)), x), "operands must be exact in bf16"

    def run(inputs):
        desc = make_desc(R, C, D, low_in=True, low_mid=low_mid, in_type=P.BF16)
        r = harness.DeviceRun(desc, closed_form.FixedNetwork(inputs), seed=78)
        out = r.execute()
        assert all(r.tails_ok.values())
        return out, [k.variant for k in r.kernels.values()]

    a, va = run(base)
    b, vb = run(new)
    if relation == "qk_rescale":
        _assert_relation(a, b, expect, 0.0, f"HIP bf16 {relation} {shape} {va}")
    else:
        _assert_relation(a, b, expect, dict(O=1.5e-2, L=2e-3 if low_mid else 1e-3, D=2e-2, dV=2e-2, dK=2e-2, dQ=2e-2),
                         f"HIP bf16 {relation} {shape} {va}", relative=True)
