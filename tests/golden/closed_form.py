"""Closed-form answers that pin the attention oracle (and the HIP kernels) to something outside this repository's
own code: each case picks inputs for which the formulas of the reference's CPU network
(/root/reference/Tests/FlashAttentionTests/Utilities/Network.swift:134-402) collapse to expressions that can be
written down by hand and evaluated WITHOUT any softmax code:

    S = Q K^T / sqrt(D)                      createMatrixSRow      :134-149
    P = softmax rows of S                    createMatrixPRow      :151-179
    L_r = ln sum_c exp S_rc                  createLTerm           :181-203
    dP = dO V^T, dS = P o (dP - D)           createDerivativePRow / SRow :205-257
    D_r = sum_d dO_rd O_rd                   createDTerm           :259-284
    O = P V                                  inferenceAttention    :286-312
    dV = P^T dO                              derivativeV           :329-350
    dK = dS^T Q / sqrt(D), dQ = dS K / sqrt(D)   derivativeK / derivativeQ :352-402

Cases (expected values are float64 numpy einsums over the INPUTS only):
  q_zero        Q = 0             -> P = 1/C: O = mean V, L = ln C, dK = 0, dV_c = mean-weighted sum of dO,
                                     dQ_r = sum_c (dO_r . (V_c - Vbar)) K_c / (C sqrt D)
  keys_equal    all K rows equal  -> P = 1/C again, L_r = q_r . k / sqrt D + ln C, dQ = 0 (row-sum identity sum_c dS = 0)
  one_key       C = 1             -> P = 1: O = V_0, L = S, dS = 0: dQ = dK = 0, dV_0 = sum_r dO_r
  dominant_key  one key 60 nats above the rest -> P = one-hot to 1e-26: O = V_c*, L = S_rc*, dQ = dK = 0,
                                     dV = one-hot row sum
  permutation   property, not a value: permuting the keys permutes dK, dV and leaves O, L, D, dQ unchanged

`build(name, R, C, D, seed)` returns (inputs, expected) with inputs = dict(Q, K, V, dO) float32 and expected = the six
outputs in the reference's units (L natural log, D unscaled), float64.
"""
import numpy as np

CASES = ("q_zero", "keys_equal", "one_key", "dominant_key")


def _rand(rng, shape):
    return rng.standard_normal(shape).astype(np.float32)


def build(name, R, C, D, seed=0, quantize=None):
    """quantize: optional function float32 array -> float32 array (storage round trip of a 16-bit type); applied to the
    inputs BEFORE the closed form is evaluated, so the expectation is exact for what the kernel reads."""
    rng = np.random.default_rng(seed)
    if name == "one_key":
        C = 1
    Q, K, V, dO = _rand(rng, (R, D)), _rand(rng, (C, D)), _rand(rng, (C, D)), _rand(rng, (R, D))
    if name == "q_zero":
        Q[:] = 0
    elif name == "keys_equal":
        K[:] = K[0]
    elif name == "dominant_key":
        # every query points along e_0 with length a; key c* = b e_0 with a b / sqrt(D) = 60; all other keys are
        # orthogonal to e_0, so their scores are exactly 0
        cstar = C // 3
        a = 4.0
        b = 60.0 * np.sqrt(D) / a
        Q[:] = 0
        Q[:, 0] = a
        K[:, 0] = 0
        K[cstar] = 0
        K[cstar, 0] = b
    if quantize is not None:
        Q, K, V, dO = (quantize(x) for x in (Q, K, V, dO))
    q, k, v, g = (x.astype(np.float64) for x in (Q, K, V, dO))
    rs = 1.0 / np.sqrt(D)
    exp = {}
    if name in ("q_zero", "keys_equal"):
        vbar = v.mean(axis=0)
        exp["O"] = np.broadcast_to(vbar, (R, D)).copy()
        exp["L"] = (q @ k[0]) * rs + np.log(C)
        exp["D"] = g @ vbar
        exp["dV"] = np.broadcast_to(g.sum(axis=0) / C, (C, D)).copy()
        ds = (g @ (v - vbar).T) / C                    # dS_rc = P (dP - D) with P = 1/C
        exp["dK"] = ds.T @ q * rs
        exp["dQ"] = ds @ k * rs                        # keys_equal: = (sum_c dS_rc) k / sqrt(D) = 0 up to rounding
    elif name == "one_key":
        exp["O"] = np.broadcast_to(v[0], (R, D)).copy()
        exp["L"] = (q @ k[0]) * rs
        exp["D"] = g @ v[0]
        exp["dV"] = g.sum(axis=0, keepdims=True)
        exp["dK"] = np.zeros((1, D))
        exp["dQ"] = np.zeros((R, D))
    elif name == "dominant_key":
        cstar = C // 3
        exp["O"] = np.broadcast_to(v[cstar], (R, D)).copy()
        exp["L"] = (q @ k[cstar]) * rs                 # + ln(1 + (C - 1) e^-60) = + 1e-23
        exp["D"] = g @ v[cstar]
        dv = np.zeros((C, D))
        dv[cstar] = g.sum(axis=0)
        exp["dV"] = dv
        exp["dK"] = np.zeros((C, D))
        exp["dQ"] = np.zeros((R, D))
    else:
        raise ValueError(name)
    return dict(Q=Q, K=K, V=V, dO=dO), exp


class FixedNetwork:
    """Duck type of oracle.Network for tests.harness.DeviceRun: just the four input matrices."""

    def __init__(self, inputs):
        self.Q, self.K, self.V, self.dO = (np.ascontiguousarray(inputs[n], np.float32) for n in ("Q", "K", "V", "dO"))
        self.rowDimension, self.headDimension = self.Q.shape
        self.columnDimension = self.K.shape[0]
