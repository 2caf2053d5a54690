"""ctypes front end of oracle/network.c (TEST INFRASTRUCTURE ONLY).

Mirrors the reference's test utility API
(/root/reference/Tests/FlashAttentionTests/Utilities/Network.swift:64-130):
`NetworkDescriptor{rowDimension,columnDimension,headDimension}` -> `Network`
with fields Q, K, V, dO and methods inferenceAttention(), createLTerm-as-vector,
createDTerm-as-vector, derivativeV/K/Q(), loss().
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from dataclasses import dataclass
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle_network.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile network.c with gcc (Makefile in this directory)."""
    src = os.path.join(_HERE, "network.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle_network.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB_PATH


def _load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    lib = ctypes.CDLL(_LIB_PATH)
    fp = ctypes.POINTER(ctypes.c_float)
    dp = ctypes.POINTER(ctypes.c_double)
    lib.oracle_network_init.argtypes = [ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int, fp, fp, fp, fp]
    lib.oracle_network_init.restype = None
    lib.oracle_network_run.argtypes = [ctypes.c_int] * 3 + [fp] * 10 + [ctypes.c_int]
    lib.oracle_network_run.restype = ctypes.c_int
    lib.oracle_network_run_masked.argtypes = [ctypes.c_int] * 3 + [fp] * 10 + [ctypes.c_int, ctypes.c_int]
    lib.oracle_network_run_masked.restype = ctypes.c_int
    lib.oracle_network_run_f64.argtypes = [ctypes.c_int] * 3 + [dp] * 10
    lib.oracle_network_run_f64.restype = ctypes.c_int
    lib.oracle_network_loss.argtypes = [ctypes.c_int] * 3 + [fp] * 4
    lib.oracle_network_loss.restype = ctypes.c_float
    lib.oracle_round_trip.argtypes = [fp, ctypes.c_size_t, ctypes.c_int]
    lib.oracle_round_trip.restype = None
    lib.oracle_max_threads.restype = ctypes.c_int
    _lib = lib
    return lib


def max_threads() -> int:
    return int(_load().oracle_max_threads())


def _fp(a: Optional[np.ndarray]):
    if a is None:
        return None
    assert a.dtype == np.float32 and a.flags.c_contiguous
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _dp(a: Optional[np.ndarray]):
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def round_trip(x: np.ndarray, precision: int) -> np.ndarray:
    """Float -> {FP32=0, FP16=1 (RNE), BF16=2 (truncation)} -> Float, as the reference's
    createBuffer/copy pair does (MTLContext+Buffers.swift:29-42, :60-75)."""
    out = np.ascontiguousarray(x, dtype=np.float32).copy()
    _load().oracle_round_trip(_fp(out.reshape(-1)), out.size, int(precision))
    return out


@dataclass
class NetworkDescriptor:
    rowDimension: Optional[int] = None
    columnDimension: Optional[int] = None
    headDimension: Optional[int] = None


class Network:
    """Network.swift:70-130.  `seed` replaces the reference's unseeded system RNG."""

    def __init__(self, descriptor: NetworkDescriptor, seed: int = 0, threads: int = 0):
        if descriptor.rowDimension is None or descriptor.columnDimension is None \
                or descriptor.headDimension is None:
            raise ValueError("Descriptor was incomplete.")  # Network.swift:84
        self.rowDimension = int(descriptor.rowDimension)
        self.columnDimension = int(descriptor.columnDimension)
        self.headDimension = int(descriptor.headDimension)
        self.threads = threads
        R, C, D = self.rowDimension, self.columnDimension, self.headDimension
        self.Q = np.zeros((R, D), np.float32)
        self.K = np.zeros((C, D), np.float32)
        self.V = np.zeros((C, D), np.float32)
        self.dO = np.zeros((R, D), np.float32)
        _load().oracle_network_init(seed, R, C, D, _fp(self.Q), _fp(self.K), _fp(self.V), _fp(self.dO))
        self._cache = None

    # -- everything in one pass (shares the row intermediates; same arithmetic) --
    def run(self, backward: bool = True, causal: bool = False):
        R, C, D = self.rowDimension, self.columnDimension, self.headDimension
        out = {"O": np.empty((R, D), np.float32), "L": np.empty(R, np.float32)}
        if backward:
            out.update(D=np.empty(R, np.float32), dV=np.empty((C, D), np.float32),
                       dK=np.empty((C, D), np.float32), dQ=np.empty((R, D), np.float32))
        for a in (self.Q, self.K, self.V, self.dO):
            assert a.dtype == np.float32 and a.flags.c_contiguous
        rc = _load().oracle_network_run_masked(
            R, C, D, _fp(self.Q), _fp(self.K), _fp(self.V), _fp(self.dO),
            _fp(out["O"]), _fp(out["L"]), _fp(out.get("D")), _fp(out.get("dV")),
            _fp(out.get("dK")), _fp(out.get("dQ")), self.threads, int(bool(causal)))
        if rc != 0:
            raise RuntimeError(f"oracle_network_run failed: {rc}")
        return out

    def _all(self):
        if self._cache is None:
            self._cache = self.run(backward=True)
        return self._cache

    def invalidate(self):
        self._cache = None

    # -- the reference's method names --
    def inferenceAttention(self):
        return self._all()["O"]

    def createLTerms(self):
        return self._all()["L"]

    def createDTerms(self):
        return self._all()["D"]

    def derivativeV(self):
        return self._all()["dV"]

    def derivativeK(self):
        return self._all()["dK"]

    def derivativeQ(self):
        return self._all()["dQ"]

    def loss(self) -> float:
        R, C, D = self.rowDimension, self.columnDimension, self.headDimension
        return float(_load().oracle_network_loss(R, C, D, _fp(self.Q), _fp(self.K), _fp(self.V), _fp(self.dO)))

    # -- fp64 twin (C) --
    def run_f64(self):
        R, C, D = self.rowDimension, self.columnDimension, self.headDimension
        q, k, v, g = (np.ascontiguousarray(a, np.float64) for a in (self.Q, self.K, self.V, self.dO))
        out = {"O": np.empty((R, D)), "L": np.empty(R), "D": np.empty(R),
               "dV": np.empty((C, D)), "dK": np.empty((C, D)), "dQ": np.empty((R, D))}
        _load().oracle_network_run_f64(R, C, D, _dp(q), _dp(k), _dp(v), _dp(g),
                                       _dp(out["O"]), _dp(out["L"]), _dp(out["D"]),
                                       _dp(out["dV"]), _dp(out["dK"]), _dp(out["dQ"]))
        return out
