"""CPU tests of the oracle (test infrastructure): golden fixtures, fp64 twin, finite differences,
torch autograd, dtype round trips.  No GPU."""
import os

import numpy as np
import pytest

from oracle import Network, NetworkDescriptor, round_trip
from oracle.network_np import attention_f64

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "network_golden.npz")
NAMES = ("O", "L", "D", "dV", "dK", "dQ")


def test_golden_fixtures_bit_exact():
    """The oracle reproduces its committed seeded fixtures bit-for-bit, for 1 and many threads
    (thread count must not change any fp32 sum order)."""
    g = np.load(GOLDEN)
    for seed, R, C, D in g["cases"]:
        for threads in (1, 4):
            net = Network(NetworkDescriptor(int(R), int(C), int(D)), seed=int(seed), threads=threads)
            for name, arr in (("Q", net.Q), ("K", net.K), ("V", net.V), ("dO", net.dO)):
                assert np.array_equal(arr, g[f"s{seed}_{name}"]), (seed, name)
            res = net.run()
            for name in NAMES:
                assert np.array_equal(res[name], g[f"s{seed}_{name}"]), (seed, name, threads)


@pytest.mark.parametrize("shape", [(10, 10, 3), (23, 23, 2), (64, 64, 40), (7, 33, 5), (40, 9, 17), (192, 192, 77)])
def test_matches_independent_fp64_restatement(shape):
    R, C, D = shape
    net = Network(NetworkDescriptor(R, C, D), seed=11)
    res = net.run()
    ref = attention_f64(net.Q, net.K, net.V, net.dO)
    twin = net.run_f64()
    for name in NAMES:
        assert np.abs(twin[name] - ref[name]).max() < 1e-11, name          # C fp64 == numpy fp64
        assert np.abs(res[name] - ref[name]).max() < 2e-5, name             # fp32 within the FP32 tolerance


def test_inputs_are_standard_normal():
    net = Network(NetworkDescriptor(256, 256, 64), seed=5)
    for arr in (net.Q, net.K, net.V, net.dO):
        assert abs(float(arr.mean())) < 0.05 and abs(float(arr.std()) - 1.0) < 0.05
    # Q/dO and K/V come from the same Box-Muller pair (Network.swift:100-112) yet are uncorrelated
    assert abs(float(np.corrcoef(net.Q.ravel(), net.dO.ravel())[0, 1])) < 0.05


def test_finite_differences():
    """Analytic gradients vs central differences of the loss, in fp64
    (Documentation/Archive/FiniteDifferencingTest.swift:86-131)."""
    net = Network(NetworkDescriptor(6, 9, 4), seed=2)
    Q, K, V, dO = (a.astype(np.float64) for a in (net.Q, net.K, net.V, net.dO))
    ref = attention_f64(Q, K, V, dO)

    def loss(q, k, v):
        return float((attention_f64(q, k, v)["O"] * dO).sum())

    h = 1e-5
    for name, X in (("dQ", Q), ("dK", K), ("dV", V)):
        num = np.zeros_like(X)
        for idx in np.ndindex(*X.shape):
            old = X[idx]
            X[idx] = old + h
            lp = loss(Q, K, V)
            X[idx] = old - h
            lm = loss(Q, K, V)
            X[idx] = old
            num[idx] = (lp - lm) / (2 * h)
        assert np.abs(num - ref[name]).max() < 1e-7, name
    # and the fp32 oracle's loss agrees with the definition (Network.swift:314-326)
    assert abs(net.loss() - float((net.run()["O"] * net.dO).sum())) < 1e-3


def test_against_torch_autograd():
    torch = pytest.importorskip("torch")
    net = Network(NetworkDescriptor(33, 47, 24), seed=3)
    q, k, v = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (net.Q, net.K, net.V))
    g = torch.tensor(net.dO, dtype=torch.float64)
    s = (q @ k.T) / np.sqrt(24.0)
    o = torch.softmax(s, dim=-1) @ v
    (o * g).sum().backward()
    res = net.run()
    assert np.abs(res["O"] - o.detach().numpy()).max() < 2e-5
    assert np.abs(res["L"] - torch.logsumexp(s, -1).detach().numpy()).max() < 2e-5
    assert np.abs(res["dQ"] - q.grad.numpy()).max() < 2e-5
    assert np.abs(res["dK"] - k.grad.numpy()).max() < 2e-5
    assert np.abs(res["dV"] - v.grad.numpy()).max() < 2e-5


def test_round_trips_match_reference_packing():
    """MTLContext+Buffers.swift:29-42: FP16 = Float16(x) (RNE), BF16 = upper half (truncation)."""
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.normal(0, 1, 4096), rng.normal(0, 1e-6, 256), rng.uniform(-70000, 70000, 256),
                        [0.0, -0.0, 65504.0, 65519.9, 65520.0, 5.96e-8, 2.98e-8, 2.99e-8]]).astype(np.float32)
    with np.errstate(over="ignore"):
        want16 = x.astype(np.float16).astype(np.float32)
    assert np.array_equal(round_trip(x, 1), want16)
    wantbf = ((x.view(np.uint32) >> 16) << 16).view(np.float32)
    assert np.array_equal(round_trip(x, 2), wantbf)
    assert np.array_equal(round_trip(x, 0), x)


@pytest.mark.parametrize("shape", [(17, 17, 8), (9, 33, 5), (64, 64, 16), (40, 100, 24)])
def test_causal_extension_matches_fp64_restatement(shape):
    """Causal masking (this project's extension of the unmasked reference): the C oracle with columns
    c > r + (C - R) left out of every sum equals the matrix-form fp64 restatement with S = -inf there;
    causal=False still reproduces the committed goldens (test above)."""
    R, C, D = shape
    net = Network(NetworkDescriptor(R, C, D), seed=17)
    got = net.run(causal=True)
    ref = attention_f64(net.Q, net.K, net.V, net.dO, causal=True)
    for name in NAMES:
        assert np.abs(got[name] - ref[name]).max() < 2e-5, name
    # the strictly-upper-triangular part of the problem must not influence anything
    net2 = Network(NetworkDescriptor(R, C, D), seed=17)
    net2.V[C - 1] += 100.0   # only the last row may see the last column
    got2 = net2.run(causal=True)
    assert np.array_equal(got2["O"][:-1], got["O"][:-1])


def test_swift_pin_kit_round_trip(tmp_path):
    """tests/golden/swift_pin.py: the exported input file carries the seeded inputs of the golden npz,
    and an output file in the format the Swift test writes -- here filled from the oracle itself -- passes the checker; a
    perturbed one does not."""
    import importlib.util
    import os
    import numpy as np
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("swift_pin", os.path.join(here, "swift_pin.py"))
    kit = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(kit)
    kit.export(str(tmp_path))
    g = np.load(os.path.join(here, "network_golden.npz"))
    raw = open(tmp_path / "network_golden_inputs.bin", "rb").read()
    assert int(np.frombuffer(raw[:4], np.int32)[0]) == len(g["cases"])
    seed, R, C, D = (int(x) for x in np.frombuffer(raw[4:20], np.int32))
    assert np.array_equal(np.frombuffer(raw[20:20 + 4 * R * D], np.float32), np.asarray(g[f"s{seed}_Q"], np.float32).reshape(-1))

    def write(perturb):
        with open(tmp_path / "network_golden_swift.bin", "wb") as f:
            f.write(np.int32(len(g["cases"])).tobytes())
            for seed, R, C, D in g["cases"]:
                f.write(np.array([seed, R, C, D], np.int32).tobytes())
                for name in ("O", "L", "D", "dV", "dK", "dQ"):
                    a = np.ascontiguousarray(g[f"s{seed}_{name}"], np.float32).reshape(-1).copy()
                    if perturb and name == "dK":
                        a[0] += 1e-2
                    f.write(a.tobytes())
    write(False)
    kit.check(str(tmp_path))
    write(True)
    with pytest.raises(AssertionError):
        kit.check(str(tmp_path))
