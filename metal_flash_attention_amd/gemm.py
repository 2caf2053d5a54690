"""Host-side mirror of the reference's GEMM API over the C ABI (include/mfa_gemm.h).

    gemmDesc = GEMMDescriptor()
    gemmDesc.loadPreviousC = False
    gemmDesc.matrixDimensions = (M, N, K)
    gemmDesc.memoryPrecisions = (A, B, C)
    gemmDesc.transposeState = (False, False)
    kernel = GEMMKernel(GEMMKernelDescriptor(descriptor=gemmDesc))
    kernel.blockDimensions, kernel.threadgroupSize, kernel.threadgroupMemoryAllocation
    kernel.dispatch(bufferA, bufferB, bufferC, descriptor=gemmDesc)   # function constants + encoder + dispatch

Reference types mirrored (the call sequence is that of Tests/FlashAttentionTests/GEMM/LaplacianTest.swift:25-41
and :177-215):
  GEMMDescriptor          Sources/FlashAttention/GEMM/GEMMDescriptor/GEMMDescriptor.swift:11-47
  GEMMKernelDescriptor    Sources/FlashAttention/GEMM/GEMMKernelDescriptor.swift  (+ init(descriptor:), GEMMDescriptor.swift:98-322)
  GEMMKernel              Sources/FlashAttention/GEMM/GEMMKernel/GEMMKernel.swift
All logic lives in the C++ library; this file only marshals.
"""
from __future__ import annotations

import ctypes
from typing import Optional, Tuple

from . import _abi
from ._abi import check, lib
from .attention import GEMMOperandPrecision, _pointer as _device_pointer


class GEMMDescriptor:
    def __init__(self):
        self.batchDimension: int = 1
        self.leadingDimensions: Optional[Tuple[int, int, int]] = None
        self.loadPreviousC: bool = False
        self.matrixDimensions: Optional[Tuple[int, int, int]] = None      # (M, N, K)
        self.memoryPrecisions: Optional[Tuple[GEMMOperandPrecision, GEMMOperandPrecision, GEMMOperandPrecision]] = None
        self.transposeState: Optional[Tuple[bool, bool]] = None

    def _c(self) -> _abi.mfa_gemm_descriptor:
        c = _abi.mfa_gemm_descriptor()
        lib().mfa_gemm_descriptor_init(ctypes.byref(c))
        c.batchDimension = int(self.batchDimension)
        c.loadPreviousC = 1 if self.loadPreviousC else 0
        if self.leadingDimensions is not None:
            c.hasLeadingDimensions = 1
            c.leadingDimensionA, c.leadingDimensionB, c.leadingDimensionC = (int(x) for x in self.leadingDimensions)
        if self.matrixDimensions is not None:
            c.hasMatrixDimensions = 1
            c.M, c.N, c.K = (int(x) for x in self.matrixDimensions)
        if self.memoryPrecisions is not None:
            c.hasMemoryPrecisions = 1
            c.precisionA, c.precisionB, c.precisionC = (int(x) for x in self.memoryPrecisions)
        if self.transposeState is not None:
            c.hasTransposeState = 1
            c.transposeA, c.transposeB = (1 if x else 0 for x in self.transposeState)
        return c


class GEMMKernelDescriptor:
    """GEMMKernelDescriptor(descriptor:) -- GEMMDescriptor.swift:98-322."""

    def __init__(self, descriptor: Optional[GEMMDescriptor] = None):
        self._c = _abi.mfa_gemm_kernel_descriptor()
        if descriptor is not None:
            d = descriptor._c()
            check(lib().mfa_gemm_descriptor_kernel_descriptor(ctypes.byref(d), ctypes.byref(self._c)))

    @property
    def blockDimensions(self):
        return (self._c.blockM, self._c.blockN, self._c.blockK) if self._c.blockM else None

    @blockDimensions.setter
    def blockDimensions(self, v):
        self._c.blockM, self._c.blockN, self._c.blockK = (int(x) for x in v)

    @property
    def memoryPrecisions(self):
        return tuple(GEMMOperandPrecision(x) for x in (self._c.memoryPrecisionA, self._c.memoryPrecisionB, self._c.memoryPrecisionC))

    @property
    def registerPrecisions(self):
        return tuple(GEMMOperandPrecision(x) for x in (self._c.registerPrecisionA, self._c.registerPrecisionB, self._c.registerPrecisionC))

    @property
    def splits(self):
        return (self._c.splitsM, self._c.splitsN)

    @property
    def transposeState(self):
        return (bool(self._c.transposeA), bool(self._c.transposeB))

    @property
    def preferAsyncLoad(self):
        return bool(self._c.preferAsyncLoad)

    @property
    def leadingBlockDimensions(self):
        return (self._c.leadingBlockA, self._c.leadingBlockB, self._c.leadingBlockC)


class GEMMKernel:
    def __init__(self, descriptor: GEMMKernelDescriptor):
        handle = ctypes.c_void_p()
        check(lib().mfa_gemm_kernel_create(ctypes.byref(descriptor._c), ctypes.byref(handle)))
        self._handle = handle
        m, n, k = ctypes.c_uint16(), ctypes.c_uint16(), ctypes.c_uint16()
        check(lib().mfa_gemm_kernel_block_dimensions(handle, ctypes.byref(m), ctypes.byref(n), ctypes.byref(k)))
        self.blockDimensions = (m.value, n.value, k.value)
        self.threadgroupSize = int(lib().mfa_gemm_kernel_threadgroup_size(handle))
        self.threadgroupMemoryAllocation = int(lib().mfa_gemm_kernel_threadgroup_memory_allocation(handle))
        self.variant = lib().mfa_gemm_kernel_variant(handle).decode()

    def __del__(self):
        h, self._handle = getattr(self, "_handle", None), None
        if h and _abi._lib is not None:
            _abi._lib.mfa_gemm_kernel_destroy(h)

    @staticmethod
    def _params(descriptor: GEMMDescriptor, batchStrides=None) -> _abi.mfa_gemm_launch_params:
        if descriptor.matrixDimensions is None:
            raise _abi.MFAError(1, "Descriptor was incomplete.")
        p = _abi.mfa_gemm_launch_params()
        lib().mfa_gemm_launch_params_init(ctypes.byref(p))
        p.M, p.N, p.K = (int(x) for x in descriptor.matrixDimensions)
        if descriptor.leadingDimensions is not None:
            p.leadingDimensionA, p.leadingDimensionB, p.leadingDimensionC = (int(x) for x in descriptor.leadingDimensions)
        p.loadPreviousC = 1 if descriptor.loadPreviousC else 0
        p.batchDimension = int(descriptor.batchDimension)
        if batchStrides is not None:
            p.batchStrideA, p.batchStrideB, p.batchStrideC = (int(x) for x in batchStrides)
        return p

    def dispatch(self, A, B, C, *, descriptor: GEMMDescriptor, batchStrides=None, stream: Optional[int] = None) -> None:
        p = self._params(descriptor, batchStrides)
        check(lib().mfa_gemm_kernel_launch(self._handle, _device_pointer(A), _device_pointer(B), _device_pointer(C),
                                           ctypes.byref(p), ctypes.c_void_p(stream or 0)))

    def time(self, A, B, C, *, descriptor: GEMMDescriptor, batchStrides=None, stream: Optional[int] = None,
             warmup: int = 1, iterations: int = 20) -> float:
        p = self._params(descriptor, batchStrides)
        ms = ctypes.c_float()
        check(lib().mfa_gemm_kernel_time(self._handle, _device_pointer(A), _device_pointer(B), _device_pointer(C),
                                         ctypes.byref(p), ctypes.c_void_p(stream or 0), int(warmup), int(iterations),
                                         ctypes.byref(ms)))
        return float(ms.value)
