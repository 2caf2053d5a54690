"""Test harness: the flow of the reference's validateProblemSize / runCorrectnessTest
(Tests/FlashAttentionTests/Attention/SquareAttentionTest.swift:214-555,
 RectangularAttentionTest.swift:39-473) against the HIP kernels, through the C ABI.

  Network (oracle) -> pack buffers in the descriptor's memory precisions, transposed where asked,
  each followed by an equal-size U(-20,20) canary tail (MTLContext+Buffers.swift:13-18), O poisoned
  with NaN (SquareAttentionTest.swift:286) -> forward, backwardQuery, backwardKeyValue dispatches
  (:355-368) -> copy back, undo the storage scaling of L and D (:408-413) -> compare.

torch is used only to own device memory.
"""
from __future__ import annotations

import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from metal_flash_attention_amd import (  # noqa: E402
    AttentionDescriptor,
    AttentionKernel,
    AttentionKernelType,
    AttentionOperand,
    GEMMOperandPrecision,
)

LOG2E = 1.44269504089  # SquareAttentionTest.swift:410

# the 20 shapes of SquareAttentionTest.testCorrectness (SquareAttentionTest.swift:6-25)
SQUARE_SHAPES = [
    (10, 3), (10, 80), (8, 2), (9, 2), (23, 2), (24, 2), (25, 2), (192, 77), (192, 80), (93, 32),
    (99, 35), (64, 32), (64, 34), (64, 36), (64, 40), (32, 64), (4, 1), (4, 2), (384, 95), (777, 199),
]

# tolerances: SquareAttentionTest.swift:539-554, RectangularAttentionTest.swift:451-472
TOL_FP32 = dict(O=2e-5, L=2e-5, D=2e-5, dV=2e-5, dK=2e-5, dQ=2e-5)
TOL_MIXED = dict(O=5e-2, L=7e-3, D=1e-1, dV=5e-2, dK=5e-2, dQ=5e-2)
TOL_MIXED_SHORT = dict(O=5e-2, L=1e-2, D=3e-1)  # column <= 20: gradients unchecked (:451-458)


def rectangular_cases(count: int = 15, seed: int = 0):
    """Seeded re-creation of the generator in RectangularAttentionTest.swift:8-33."""
    rng = np.random.default_rng(seed)
    cases = []
    for _ in range(count):
        vec = rng.random(2, dtype=np.float32)
        vec = vec * vec * vec
        ints = (vec * 128).astype(np.int64)
        ints[ints == 0] = 1
        row, head = int(ints[0]), int(ints[1])
        column = int(rng.integers(1, 11)) if rng.random() < 0.5 else int(rng.integers(10, 129))
        cases.append(dict(
            row=row, column=column, head=head,
            lowPrecisionInputs=bool(rng.integers(0, 2)),
            lowPrecisionIntermediates=bool(rng.integers(0, 2)),
            transposeState=tuple(bool(x) for x in rng.integers(0, 2, 4)),
        ))
    return cases


# ---- buffer packing: MTLContext+Buffers.swift:5-45 / :47-78 ------------------------------------
def pack(array: np.ndarray, precision: GEMMOperandPrecision, rng: np.random.Generator, tail: bool = True):
    """float32 array -> raw bytes in `precision`, followed by an equal-length random tail."""
    flat = np.ascontiguousarray(array, np.float32).reshape(-1)
    if tail:
        flat = np.concatenate([flat, rng.uniform(-20, 20, flat.size).astype(np.float32)])
    if precision == GEMMOperandPrecision.FP32:
        raw = flat.view(np.uint8)
    elif precision == GEMMOperandPrecision.FP16:
        raw = flat.astype(np.float16).view(np.uint8)           # Float16(x): RNE
    else:
        raw = (flat.view(np.uint32) >> 16).astype(np.uint16).view(np.uint8)  # BF16: truncation (:36-42)
    return raw.copy()


def unpack(raw: np.ndarray, precision: GEMMOperandPrecision, count: int, offset: int = 0) -> np.ndarray:
    """MTLContext.copy(_:into:precision:), :47-78.  `offset`/`count` in elements."""
    if precision == GEMMOperandPrecision.FP32:
        return raw.view(np.float32)[offset:offset + count].copy()
    if precision == GEMMOperandPrecision.FP16:
        return raw.view(np.float16)[offset:offset + count].astype(np.float32)
    return (raw.view(np.uint16)[offset:offset + count].astype(np.uint32) << 16).view(np.float32)


def transpose_in(x: np.ndarray) -> np.ndarray:
    """[seq][D] -> [D][seq] (RectangularAttentionTest.swift:88-103)."""
    return np.ascontiguousarray(x.T)


class DeviceRun:
    """One forward+backward pass of the three kernels for a single (R, C, D) problem."""

    def __init__(self, desc: AttentionDescriptor, network, seed: int = 1234, heads: int = 1,
                 run_backward: bool = True, causal: bool = False, memory_overrides=None):
        """memory_overrides: {operand: precision} written into every kernel descriptor after the table lookup (the Swift struct
        lets a caller do the same, AttentionKernelDescriptor.swift:8-49) -- e.g. dO stored in FP16 next to FP16 Q / K / V instead
        of the BF16 the reference's descriptor picks (+Precisions.swift:13-17)"""
        import torch

        self.torch = torch
        self.causal = causal
        self.desc = desc
        self.network = network
        R, C, D = desc.matrixDimensions
        self.R, self.C, self.D = R, C, D
        self.heads = heads
        self.precisions = dict(desc.memoryPrecisions)
        self.precisions.update(memory_overrides or {})
        tQ, tK, tV, tO = desc.transposeState
        self.transposed = {
            AttentionOperand.Q: tQ, AttentionOperand.K: tK, AttentionOperand.V: tV, AttentionOperand.O: tO,
            AttentionOperand.dO: tO, AttentionOperand.dV: tV, AttentionOperand.dK: tK, AttentionOperand.dQ: tQ,
        }
        rng = np.random.default_rng(seed)
        self.shapes = {
            AttentionOperand.Q: (R, D), AttentionOperand.K: (C, D), AttentionOperand.V: (C, D),
            AttentionOperand.O: (R, D), AttentionOperand.L: (R,), AttentionOperand.D: (R,),
            AttentionOperand.dO: (R, D), AttentionOperand.dV: (C, D), AttentionOperand.dK: (C, D),
            AttentionOperand.dQ: (R, D),
        }
        inputs = {AttentionOperand.Q: network.Q, AttentionOperand.K: network.K,
                  AttentionOperand.V: network.V, AttentionOperand.dO: network.dO}
        self.host_raw = {}
        self.buffers = {}
        for op, shape in self.shapes.items():
            if op in inputs:
                data = inputs[op]
                if self.transposed.get(op):
                    data = transpose_in(data)
            else:
                data = np.zeros(shape, np.float32)
                if op == AttentionOperand.O:
                    data.reshape(-1)[0] = np.nan  # resultO[0] = .nan (SquareAttentionTest.swift:286)
            raw = pack(data, self.precisions[op], rng)
            self.host_raw[op] = raw
            self.buffers[op] = torch.from_numpy(raw).cuda()

        self.kernels = {}
        types = [AttentionKernelType.forward]
        if run_backward:
            types += [AttentionKernelType.backwardQuery, AttentionKernelType.backwardKeyValue]
        for t in types:
            kdesc = desc.kernelDescriptor(t)
            for op, prec in (memory_overrides or {}).items():
                if op in kdesc.memoryPrecisions:
                    kdesc.memoryPrecisions[op] = prec
            self.kernels[t] = AttentionKernel(kdesc)

    def execute(self, with_workspace: bool = False):
        """with_workspace: give every launch the scratch it asks for (workspaceSize): split launches, and the row-major
        copies that let the matrix-core kernels serve transposed operands"""
        torch = self.torch
        stream = torch.cuda.current_stream().cuda_stream
        self.workspace_bytes = {}
        for t, kernel in self.kernels.items():  # forward -> backwardQuery -> backwardKeyValue
            ws = None
            if with_workspace:
                need = kernel.workspaceSize(row=self.R, column=self.C)
                self.workspace_bytes[t] = need
                ws = torch.empty(need + 256, dtype=torch.uint8, device="cuda") if need else None
            kernel.dispatch(self.buffers, row=self.R, column=self.C, stream=stream, causal=self.causal, workspace=ws)
        torch.cuda.synchronize()
        return self.results()

    def results(self):
        out = {}
        self.tails_ok = {}
        for op in (AttentionOperand.O, AttentionOperand.L, AttentionOperand.D, AttentionOperand.dV,
                   AttentionOperand.dK, AttentionOperand.dQ):
            raw = self.buffers[op].cpu().numpy()
            shape = self.shapes[op]
            n = int(np.prod(shape))
            vals = unpack(raw, self.precisions[op], n)
            if len(shape) == 2 and self.transposed.get(op):
                vals = vals.reshape(shape[1], shape[0]).T  # transposeOut (:105-121)
            out[op.description] = np.ascontiguousarray(vals.reshape(shape))
            # canary tail must be untouched
            size = GEMMOperandPrecision(self.precisions[op]).size
            self.tails_ok[op.description] = bool(
                np.array_equal(raw[n * size:], self.host_raw[op][n * size:]))
        # undo the kernels' storage scaling (SquareAttentionTest.swift:408-413)
        out["L"] = out["L"] / np.float32(LOG2E)
        out["D"] = out["D"] / (np.float32(1) / np.sqrt(np.float32(self.D)))
        return out


def check(expected: np.ndarray, actual: np.ndarray, tolerance: float):
    """check(expected:actual:tolerance:) of SquareAttentionTest.swift:512-536, but it RETURNS the
    failures instead of printing ten of them.  Entries that are NaN/Inf on both sides are ignored."""
    expected = np.asarray(expected, np.float32).reshape(-1)
    actual = np.asarray(actual, np.float32).reshape(-1)
    assert expected.shape == actual.shape, "Arrays had different length."
    with np.errstate(invalid="ignore"):
        error = np.abs(expected - actual)
    bad = (error > tolerance) | np.isnan(error)
    both_nonfinite = ~np.isfinite(expected) & ~np.isfinite(actual)
    bad &= ~both_nonfinite
    max_err = float(np.nanmax(np.where(both_nonfinite, 0, error))) if error.size else 0.0
    return int(bad.sum()), max_err


def compare(reference: dict, results: dict, tolerances: dict):
    report = {}
    failures = []
    for name, tol in tolerances.items():
        nbad, max_err = check(reference[name], results[name], tol)
        report[name] = max_err
        if nbad:
            failures.append(f"{name}: {nbad} elements over {tol:g} (max err {max_err:.3e})")
    return failures, report
