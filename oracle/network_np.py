"""Independent numpy fp64 restatement of the reference Network formulas
(Network.swift:134-402) -- TEST INFRASTRUCTURE ONLY.  Written in matrix form on
purpose (different code shape from network.c) so that it can catch a mistake in
the C restatement."""
import numpy as np


def block_mask_to_dense(bits, R, C, block_rows=256, block_columns=128):
    """bits: bool [ceil(R / block_rows)][ceil(C / block_columns)] -> bool [R][C] (extension: block-sparse mask)."""
    return np.repeat(np.repeat(np.asarray(bits, bool), block_rows, axis=0), block_columns, axis=1)[:R, :C]


def attention_f64(Q, K, V, dO=None, causal=False, mask=None):
    """mask (extension): bool [R][C], True = attended.  Rows that see no column at all get O = 0, L = -inf and
    contribute nothing to any gradient."""
    Q, K, V = (np.asarray(a, np.float64) for a in (Q, K, V))
    D = Q.shape[-1]
    scale = 1.0 / np.sqrt(np.float64(D))
    S = (Q @ K.T) * scale                                   # Network.swift:134-149, :153
    if causal:  # extension: row r sees column c iff c <= r + max(C - R, 0)
        R_, C_ = S.shape
        S = np.where(np.arange(C_)[None, :] <= np.arange(R_)[:, None] + max(C_ - R_, 0), S, -np.inf)
    if mask is not None:
        S = np.where(mask, S, -np.inf)
    empty = ~np.isfinite(S).any(axis=1, keepdims=True)
    S = np.where(empty, 0.0, S)                             # placeholder rows, zeroed below
    m = S.max(axis=1, keepdims=True)                        # :156-160
    lse = m + np.log(np.exp(S - m).sum(axis=1, keepdims=True))   # :163-171
    P = np.where(empty, 0.0, np.exp(S - lse))               # :172-176
    lse = np.where(empty, -np.inf, lse)
    out = {"O": P @ V, "L": lse[:, 0]}                      # :286-311, :181-203
    if dO is None:
        return out
    dO = np.asarray(dO, np.float64)
    Dt = (out["O"] * dO).sum(axis=1)                        # :259-281
    dP = dO @ V.T                                           # :205-218
    dS = P * (dP - Dt[:, None]) * scale                     # :245-255
    out.update(D=Dt, dV=P.T @ dO, dK=dS.T @ Q, dQ=dS @ K)   # :329-402
    return out
