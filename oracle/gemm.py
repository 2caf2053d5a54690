"""ctypes front end of oracle/gemm.c (TEST INFRASTRUCTURE ONLY).

naive(...)               -- the CPU loop of Tests/FlashAttentionTests/GEMM/AdversarialShapeTest.swift:205-243
laplacian_matrix(n)      -- Tests/FlashAttentionTests/GEMM/LaplacianTest.swift:137-149
laplacian_expected(...)  -- the closed-form expected values of LaplacianTest.swift:286-332 (a known answer the
                            reference pins itself to), as a full matrix laid out like the test's C buffer
tolerance(...)           -- AdversarialShapeTest.swift:283-337 ; laplacian_threshold -- LaplacianTest.swift:264-283
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle_gemm.so")
_lib = None


def _load():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "gemm.c")
        if not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle_gemm.so"], stdout=subprocess.DEVNULL)
        lib = ctypes.CDLL(_LIB_PATH)
        fp, dp, u32 = ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_double), ctypes.c_uint32
        lib.oracle_gemm_naive.argtypes = [u32, u32, u32, fp, fp, fp, fp, u32, u32, u32, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        lib.oracle_gemm_f64.argtypes = [u32, u32, u32, fp, fp, fp, dp, u32, u32, u32, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        lib.oracle_laplacian_matrix.argtypes = [u32, fp]
        lib.oracle_laplacian_expected.argtypes = [u32, fp, fp, u32, u32, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                  ctypes.POINTER(ctypes.c_size_t)]
        lib.oracle_laplacian_expected.restype = ctypes.c_float
        _lib = lib
    return _lib


def _fp(a):
    assert a.dtype == np.float32 and a.flags.c_contiguous
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def naive(M, N, K, A, B, previousC, ld, transA, transB, loadPreviousC, f64=False):
    """A, B, previousC: flat float32 arrays sized trailing x leading as in the reference's test."""
    ldA, ldB, ldC = ld
    out = np.zeros(M * ldC, np.float64 if f64 else np.float32)
    prev = previousC if previousC is not None else np.zeros(M * ldC, np.float32)
    if f64:
        _load().oracle_gemm_f64(M, N, K, _fp(A), _fp(B), _fp(prev), out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                ldA, ldB, ldC, int(transA), int(transB), int(loadPreviousC))
    else:
        _load().oracle_gemm_naive(M, N, K, _fp(A), _fp(B), _fp(prev), _fp(out), ldA, ldB, ldC, int(transA), int(transB),
                                  int(loadPreviousC))
    return out


def laplacian_matrix(n: int) -> np.ndarray:
    a = np.zeros((n, n), np.float32)
    _load().oracle_laplacian_matrix(n, _fp(a.reshape(-1)))
    return a


def laplacian_expected(n, source, previousC, transA, transB, loadPreviousC):
    """Expected contents of the n x n C buffer (row-major, as the reference's test indexes it)."""
    out = np.zeros(n * n, np.float32)
    prev = previousC if previousC is not None else np.zeros(n * n, np.float32)
    idx = ctypes.c_size_t()
    lib = _load()
    src, prv = _fp(np.ascontiguousarray(source.reshape(-1))), _fp(np.ascontiguousarray(prev.reshape(-1)))
    for m in range(n):
        for col in range(n):
            e = lib.oracle_laplacian_expected(n, src, prv, m, col, int(transA), int(transB), int(loadPreviousC), ctypes.byref(idx))
            out[idx.value] = e
    return out.reshape(n, n)


def laplacian_threshold(precisions) -> float:
    """createErrorThreshold, LaplacianTest.swift:264-283."""
    table = {0: 1e-5, 1: 5e-3, 2: 5e-2}
    return max(table[int(p)] for p in precisions)


def tolerance(precisions, K: int) -> float:
    """createTolerance, AdversarialShapeTest.swift:283-337 (FP32 = 0, FP16 = 1, BF16 = 2)."""
    a, b, c = (int(p) for p in precisions)
    noise = float(np.sqrt(np.float32(K)))
    tol = 3e-7
    if a == 1 or b == 1:
        tol = max(tol, 1e-5, 1e-3 / noise)
    if c == 1:
        tol = max(tol, 3e-4)
    if a == 1 and b == 1 and c == 1:
        tol = max(tol, 3e-3, 1e-5 * K)
    if 2 in (a, b, c):
        tol = max(tol, 2e-2 if K < 1000 else 5e-3)
    tol += 2.0 ** -8 if c == 2 else (2.0 ** -10 if c == 1 else 2.0 ** -22)
    return tol
