"""ctypes declarations for include/mfa.h (the C-ABI drop-in boundary).

The product path FAILS LOUDLY when the HIP library is missing: there is no CPU or PyTorch
fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MFA_LIBRARY: developer tools (tools/ab_*.py) point this at libmfa_hip_dev.so, the -DMFA_DEV_VARIANTS build that carries
# the A/B knobs; it only chooses WHICH build of the same C ABI is loaded -- there is no fallback either way.
LIB_PATH = os.environ.get("MFA_LIBRARY") or os.path.join(_HERE, "libmfa_hip.so")

MFA_OPERAND_COUNT = 14
MFA_BUFFER_SLOTS = 10

MFA_OK = 0
STATUS_NAMES = {
    0: "MFA_OK",
    1: "MFA_ERR_INCOMPLETE_DESCRIPTOR",
    2: "MFA_ERR_INVALID_ARGUMENT",
    3: "MFA_ERR_UNSUPPORTED",
    4: "MFA_ERR_HIP",
    5: "MFA_ERR_PARSE",
}


class mfa_attention_descriptor(ctypes.Structure):
    _fields_ = [
        ("lowPrecisionInputs", ctypes.c_uint8),
        ("lowPrecisionIntermediates", ctypes.c_uint8),
        ("hasMatrixDimensions", ctypes.c_uint8),
        ("hasTransposeState", ctypes.c_uint8),
        ("row", ctypes.c_uint32),
        ("column", ctypes.c_uint32),
        ("head", ctypes.c_uint16),
        ("transposeQ", ctypes.c_uint8),
        ("transposeK", ctypes.c_uint8),
        ("transposeV", ctypes.c_uint8),
        ("transposeO", ctypes.c_uint8),
        ("lowPrecisionInputType", ctypes.c_uint8),
        ("lowPrecisionOutputs", ctypes.c_uint8),
        ("reserved", ctypes.c_uint8 * 2),
    ]


class mfa_attention_kernel_descriptor(ctypes.Structure):
    _fields_ = [
        ("hasBlockDimensions", ctypes.c_uint8),
        ("hasHeadDimension", ctypes.c_uint8),
        ("parallelization", ctypes.c_uint16),
        ("traversal", ctypes.c_uint16),
        ("headBlock", ctypes.c_uint16),
        ("headDimension", ctypes.c_uint16),
        ("cacheState", ctypes.c_int8 * MFA_OPERAND_COUNT),
        ("memoryPrecisions", ctypes.c_int8 * MFA_OPERAND_COUNT),
        ("registerPrecisions", ctypes.c_int8 * MFA_OPERAND_COUNT),
        ("transposeState", ctypes.c_int8 * MFA_OPERAND_COUNT),
        ("preferAsyncCache", ctypes.c_int8),
        ("preferAsyncLoad", ctypes.c_int8),
        ("type", ctypes.c_int8),
        ("strictBlockDimensions", ctypes.c_int8),
    ]


class mfa_parameter_row(ctypes.Structure):
    _fields_ = [
        ("maximumHeadDimension", ctypes.c_uint16),
        ("parallelization", ctypes.c_uint16),
        ("traversal", ctypes.c_uint16),
        ("head", ctypes.c_uint16),
        ("cached", ctypes.c_int8 * MFA_OPERAND_COUNT),
    ]


class mfa_launch_params(ctypes.Structure):
    _fields_ = [
        ("row", ctypes.c_uint32),
        ("column", ctypes.c_uint32),
        ("heads", ctypes.c_uint32),
        ("batches", ctypes.c_uint32),
        ("leadingDimension", ctypes.c_int64 * MFA_BUFFER_SLOTS),
        ("headStride", ctypes.c_int64 * MFA_BUFFER_SLOTS),
        ("batchStride", ctypes.c_int64 * MFA_BUFFER_SLOTS),
        ("workspace", ctypes.c_void_p),
        ("workspaceBytes", ctypes.c_uint64),
        ("causal", ctypes.c_uint32),
        ("reserved", ctypes.c_uint32),
        ("rowLengths", ctypes.c_void_p),
        ("columnLengths", ctypes.c_void_p),
        ("blockMask", ctypes.c_void_p),
        ("blockMaskWords", ctypes.c_uint32),
        ("reserved2", ctypes.c_uint32),
        ("blockMaskHeadStride", ctypes.c_int64),
        ("blockMaskBatchStride", ctypes.c_int64),
    ]

MFA_MASK_BLOCK_ROWS, MFA_MASK_BLOCK_COLUMNS = 256, 128


# every symbol include/mfa.h declares: (name, restype, argtypes)
_P = ctypes.POINTER
_KERNEL = ctypes.c_void_p
_BUFS = ctypes.c_void_p * MFA_BUFFER_SLOTS
class mfa_gemm_descriptor(ctypes.Structure):   # include/mfa_gemm.h
    _fields_ = [
        ("batchDimension", ctypes.c_uint32),
        ("hasLeadingDimensions", ctypes.c_uint8), ("loadPreviousC", ctypes.c_uint8),
        ("hasMatrixDimensions", ctypes.c_uint8), ("hasMemoryPrecisions", ctypes.c_uint8),
        ("hasTransposeState", ctypes.c_uint8), ("transposeA", ctypes.c_uint8), ("transposeB", ctypes.c_uint8),
        ("reserved", ctypes.c_uint8),
        ("leadingDimensionA", ctypes.c_uint32), ("leadingDimensionB", ctypes.c_uint32), ("leadingDimensionC", ctypes.c_uint32),
        ("M", ctypes.c_uint32), ("N", ctypes.c_uint32), ("K", ctypes.c_uint32),
        ("precisionA", ctypes.c_int32), ("precisionB", ctypes.c_int32), ("precisionC", ctypes.c_int32),
    ]


class mfa_gemm_kernel_descriptor(ctypes.Structure):
    _fields_ = [
        ("blockM", ctypes.c_uint16), ("blockN", ctypes.c_uint16), ("blockK", ctypes.c_uint16),
        ("leadingBlockA", ctypes.c_uint16), ("leadingBlockB", ctypes.c_uint16), ("leadingBlockC", ctypes.c_uint16),
        ("memoryPrecisionA", ctypes.c_int32), ("memoryPrecisionB", ctypes.c_int32), ("memoryPrecisionC", ctypes.c_int32),
        ("registerPrecisionA", ctypes.c_int32), ("registerPrecisionB", ctypes.c_int32), ("registerPrecisionC", ctypes.c_int32),
        ("splitsM", ctypes.c_uint16), ("splitsN", ctypes.c_uint16),
        ("preferAsyncLoad", ctypes.c_uint8), ("preferAsyncStore", ctypes.c_uint8),
        ("transposeA", ctypes.c_uint8), ("transposeB", ctypes.c_uint8),
        ("complete", ctypes.c_uint8), ("reserved", ctypes.c_uint8 * 3),
    ]


class mfa_gemm_launch_params(ctypes.Structure):
    _fields_ = [
        ("M", ctypes.c_uint32), ("N", ctypes.c_uint32), ("K", ctypes.c_uint32),
        ("leadingDimensionA", ctypes.c_uint32), ("leadingDimensionB", ctypes.c_uint32), ("leadingDimensionC", ctypes.c_uint32),
        ("loadPreviousC", ctypes.c_uint32), ("batchDimension", ctypes.c_uint32),
        ("batchStrideA", ctypes.c_uint64), ("batchStrideB", ctypes.c_uint64), ("batchStrideC", ctypes.c_uint64),
    ]


_GEMM = ctypes.c_void_p

GEMM_SYMBOLS = [
    ("mfa_gemm_descriptor_init", None, [ctypes.POINTER(mfa_gemm_descriptor)]),
    ("mfa_gemm_launch_params_init", None, [ctypes.POINTER(mfa_gemm_launch_params)]),
    ("mfa_gemm_descriptor_kernel_descriptor", ctypes.c_int,
     [ctypes.POINTER(mfa_gemm_descriptor), ctypes.POINTER(mfa_gemm_kernel_descriptor)]),
    ("mfa_gemm_kernel_create", ctypes.c_int, [ctypes.POINTER(mfa_gemm_kernel_descriptor), ctypes.POINTER(_GEMM)]),
    ("mfa_gemm_kernel_destroy", None, [_GEMM]),
    ("mfa_gemm_kernel_block_dimensions", ctypes.c_int,
     [_GEMM, ctypes.POINTER(ctypes.c_uint16), ctypes.POINTER(ctypes.c_uint16), ctypes.POINTER(ctypes.c_uint16)]),
    ("mfa_gemm_kernel_threadgroup_size", ctypes.c_uint32, [_GEMM]),
    ("mfa_gemm_kernel_threadgroup_memory_allocation", ctypes.c_uint32, [_GEMM]),
    ("mfa_gemm_kernel_variant", ctypes.c_char_p, [_GEMM]),
    ("mfa_gemm_kernel_launch", ctypes.c_int,
     [_GEMM, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(mfa_gemm_launch_params), ctypes.c_void_p]),
    ("mfa_gemm_kernel_time", ctypes.c_int,
     [_GEMM, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(mfa_gemm_launch_params), ctypes.c_void_p,
      ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]),
]

SYMBOLS = [
    ("mfa_precision_name", ctypes.c_char_p, [ctypes.c_int]),
    ("mfa_precision_size", ctypes.c_int, [ctypes.c_int]),
    ("mfa_operand_name", ctypes.c_char_p, [ctypes.c_int]),
    ("mfa_operand_buffer_binding", ctypes.c_int, [ctypes.c_int]),
    ("mfa_attention_descriptor_init", None, [_P(mfa_attention_descriptor)]),
    ("mfa_attention_kernel_descriptor_init", None, [_P(mfa_attention_kernel_descriptor)]),
    ("mfa_attention_descriptor_memory_precisions", ctypes.c_int,
     [_P(mfa_attention_descriptor), _P(ctypes.c_int8)]),
    ("mfa_attention_descriptor_register_precisions", ctypes.c_int,
     [_P(mfa_attention_descriptor), _P(ctypes.c_int8)]),
    ("mfa_attention_descriptor_kernel_descriptor", ctypes.c_int,
     [_P(mfa_attention_descriptor), ctypes.c_int, _P(mfa_attention_kernel_descriptor)]),
    ("mfa_parameter_table_get", ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]),
    ("mfa_parameter_table_set", ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_char_p]),
    ("mfa_parameter_table_reset", ctypes.c_int, []),
    ("mfa_parameter_table_select", ctypes.c_int, [ctypes.c_char_p, ctypes.c_uint16, _P(mfa_parameter_row)]),
    ("mfa_attention_kernel_create", ctypes.c_int, [_P(mfa_attention_kernel_descriptor), _P(_KERNEL)]),
    ("mfa_attention_kernel_destroy", None, [_KERNEL]),
    ("mfa_attention_kernel_block_dimensions", ctypes.c_int,
     [_KERNEL, _P(ctypes.c_uint16), _P(ctypes.c_uint16), _P(ctypes.c_uint16)]),
    ("mfa_attention_kernel_threadgroup_size", ctypes.c_uint32, [_KERNEL]),
    ("mfa_attention_kernel_threadgroup_memory_allocation", ctypes.c_uint32, [_KERNEL]),
    ("mfa_attention_kernel_variant", ctypes.c_char_p, [_KERNEL]),
    ("mfa_attention_kernel_fallback_variant", ctypes.c_char_p, [_KERNEL]),
    ("mfa_attention_kernel_needs_workspace_for_fast_path", ctypes.c_int, [_KERNEL]),
    ("mfa_attention_kernel_launch_form", ctypes.c_int, [_KERNEL, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]),
    ("mfa_attention_kernel_effective_descriptor", ctypes.c_int, [_KERNEL, _P(mfa_attention_kernel_descriptor)]),
    ("mfa_launch_params_init", None, [_P(mfa_launch_params)]),
    ("mfa_attention_kernel_launch", ctypes.c_int, [_KERNEL, _P(_BUFS), _P(mfa_launch_params), ctypes.c_void_p]),
    ("mfa_attention_kernel_workspace_size", ctypes.c_int, [_KERNEL, _P(mfa_launch_params), _P(ctypes.c_uint64)]),
    ("mfa_attention_kernel_time", ctypes.c_int,
     [_KERNEL, _P(_BUFS), _P(mfa_launch_params), ctypes.c_void_p, ctypes.c_int, ctypes.c_int, _P(ctypes.c_float)]),
    ("mfa_device_count", ctypes.c_int, [_P(ctypes.c_int)]),
    ("mfa_device_name", ctypes.c_int, [ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]),
    ("mfa_last_error_string", ctypes.c_char_p, []),
    ("mfa_abi_version", ctypes.c_int, []),
]

_lib = None
EXPECTED_ABI = 6   # MFA_ABI_VERSION of include/mfa.h this file mirrors


class MFAError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


def lib() -> ctypes.CDLL:
    """Load libmfa_hip.so.  Raises (never falls back) if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C metal_flash_attention_amd/csrc`. There is no fallback path.")
    # PyTorch-ROCm bundles its own libamdhip64.so.7; when a process uses both, torch's copy must be
    # the one that is resident (loading /opt/rocm's first leaves torch with a runtime that reports
    # hipErrorNoDevice).  torch is only plumbing here (device memory, streams): import it first if
    # it is installed, so the C-ABI library binds to the same HIP runtime instance.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    handle = ctypes.CDLL(LIB_PATH)
    # version first: a stale library (the .so is git-ignored and travels separately from the sources) must fail with this
    # message, not with a bare AttributeError on the first symbol it lacks
    handle.mfa_abi_version.restype = ctypes.c_int
    handle.mfa_abi_version.argtypes = []
    got = int(handle.mfa_abi_version())
    if got != EXPECTED_ABI:   # the struct mirrors above describe exactly one layout of mfa_launch_params & co.
        raise ImportError(f"{LIB_PATH} reports ABI version {got}, these bindings were written for {EXPECTED_ABI}: "
                          f"rebuild the library (make -C metal_flash_attention_amd/csrc)")
    for name, restype, argtypes in SYMBOLS + GEMM_SYMBOLS:
        fn = getattr(handle, name)  # AttributeError if the ABI is incomplete
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = handle
    return handle


def library_path() -> str:
    """Path of the C-ABI library these bindings load (bench.py hashes it to tie a PMC profile to a build)."""
    return LIB_PATH


def check(status: int) -> None:
    if status != MFA_OK:
        raise MFAError(status, lib().mfa_last_error_string().decode())
