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
  two_keys      C = 2, RANDOM Q, K, V, dO -> P_r0 = 1 / (1 + exp(s_r1 - s_r0)) (the logistic function: a non-degenerate,
                                     row-dependent P written without softmax code), L = s_0 + ln(1 + exp(s_1 - s_0)),
                                     dS_r0 = P_0 P_1 (dP_0 - dP_1) = -dS_r1, dQ_r = dS_r0 (K_0 - K_1) / sqrt D
  permutation   property, not a value: permuting the keys permutes dK, dV and leaves O, L, D, dQ unchanged
Metamorphic relations on RANDOM inputs (`RELATIONS`, `transform`): what Network.swift:134-402 implies for a transformed
problem, stated as (which outputs are unchanged, which shift / scale and by how much):
  key_shift     K -> K + 1 u^T: S_rc gains q_r.u / sqrt D for every c -> P unchanged: O, D, dV, dK unchanged, dQ unchanged
                                     (dQ gains (sum_c dS_rc) u / sqrt D = 0), L_r gains q_r.u / sqrt D
  value_shift   V -> V + 1 w^T: O gains w (sum_c P = 1), D_r gains dO_r.w, dP - D unchanged -> dS, dQ, dK unchanged; dV, L unchanged
  qk_rescale    (Q, K) -> (a Q, K / a): S unchanged -> everything unchanged except dQ -> dQ / a, dK -> a dK; with a = 2 every
                                     product is scaled by an exact power of two: the transformed run is BIT-IDENTICAL

`build(name, R, C, D, seed)` returns (inputs, expected) with inputs = dict(Q, K, V, dO) float32 and expected = the six
outputs in the reference's units (L natural log, D unscaled), float64.
"""
import numpy as np

CASES = ("q_zero", "keys_equal", "one_key", "dominant_key", "two_keys")
RELATIONS = ("key_shift", "value_shift", "qk_rescale")


def transform(relation, inputs, seed=0, grid=None):
    """-> (transformed inputs, expect) where expect(name, base_output) is the output the relation predicts (float64) from the
    base problem's output.  grid: optional step (e.g. 1/8) -- inputs and shift vectors are snapped to multiples of it so that
    the transformed operands are exactly representable in a 16-bit storage type."""
    rng = np.random.default_rng(1000 + seed)
    Q, K, V, dO = (np.array(inputs[n], np.float32) for n in ("Q", "K", "V", "dO"))
    D = Q.shape[1]
    snap = (lambda x: (np.round(np.asarray(x, np.float64) / grid) * grid).astype(np.float32)) if grid else (lambda x: np.asarray(x, np.float32))
    Q, K, V, dO = snap(Q), snap(K), snap(V), snap(dO)
    base = dict(Q=Q, K=K, V=V, dO=dO)
    rs = 1.0 / np.sqrt(D)
    if relation == "key_shift":
        u = snap(rng.standard_normal(D) * 0.5)
        new = dict(base, K=(K + u[None, :]).astype(np.float32))
        shift = (Q.astype(np.float64) @ u.astype(np.float64)) * rs
        expect = lambda name, out: out + shift if name == "L" else out
    elif relation == "value_shift":
        w = snap(rng.standard_normal(D))
        new = dict(base, V=(V + w[None, :]).astype(np.float32))
        dshift = dO.astype(np.float64) @ w.astype(np.float64)
        expect = lambda name, out: out + w[None, :].astype(np.float64) if name == "O" else (out + dshift if name == "D" else out)
    elif relation == "qk_rescale":
        a = 2.0
        new = dict(base, Q=(Q * np.float32(a)), K=(K / np.float32(a)))
        expect = lambda name, out: out / a if name == "dQ" else (out * a if name == "dK" else out)
    else:
        raise ValueError(relation)
    return base, new, expect


def _rand(rng, shape):
    return rng.standard_normal(shape).astype(np.float32)


def build(name, R, C, D, seed=0, quantize=None):
    """quantize: optional function float32 array -> float32 array (storage round trip of a 16-bit type); applied to the
    inputs BEFORE the closed form is evaluated, so the expectation is exact for what the kernel reads."""
    rng = np.random.default_rng(seed)
    if name == "one_key":
        C = 1
    if name == "two_keys":
        C = 2
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
    elif name == "two_keys":
        s = (q @ k.T) * rs                              # [R, 2]
        p0 = 1.0 / (1.0 + np.exp(s[:, 1] - s[:, 0]))    # logistic; no softmax code
        p1 = 1.0 - p0
        exp["O"] = p0[:, None] * v[0] + p1[:, None] * v[1]
        exp["L"] = s[:, 0] + np.log1p(np.exp(s[:, 1] - s[:, 0]))
        dp = g @ v.T                                    # [R, 2]
        exp["D"] = p0 * dp[:, 0] + p1 * dp[:, 1]
        ds0 = p0 * p1 * (dp[:, 0] - dp[:, 1])           # = -dS_r1
        exp["dV"] = np.stack([p0 @ g, p1 @ g])
        exp["dK"] = np.stack([ds0 @ q, -ds0 @ q]) * rs
        exp["dQ"] = ds0[:, None] * (k[0] - k[1])[None, :] * rs
    else:
        raise ValueError(name)
    return dict(Q=Q, K=K, V=V, dO=dO), exp


class FixedNetwork:
    """Duck type of oracle.Network for tests.harness.DeviceRun: just the four input matrices."""

    def __init__(self, inputs):
        self.Q, self.K, self.V, self.dO = (np.ascontiguousarray(inputs[n], np.float32) for n in ("Q", "K", "V", "dO"))
        self.rowDimension, self.headDimension = self.Q.shape
        self.columnDimension = self.K.shape[0]
