"""Host-side mirror of the reference's attention API, over the C ABI (include/mfa.h).

Same names, argument meaning and error behaviour as the Swift types, so tests read like the
reference's own (Tests/FlashAttentionTests/Attention/SquareAttentionTest.swift:214-263):

    attentionDesc = AttentionDescriptor()
    attentionDesc.lowPrecisionInputs = False
    attentionDesc.lowPrecisionIntermediates = False
    attentionDesc.matrixDimensions = (row, column, head)
    attentionDesc.transposeState = (False, False, False, False)
    kernelDesc = attentionDesc.kernelDescriptor(type=AttentionKernelType.forward)
    kernel = AttentionKernel(descriptor=kernelDesc)
    kernel.blockDimensions, kernel.threadgroupSize, kernel.threadgroupMemoryAllocation
    kernel.dispatch(buffers, row=R, column=C)    # replaces createSource + pipeline + encoder

Reference types mirrored:
  AttentionDescriptor        Sources/FlashAttention/Attention/AttentionDescriptor/AttentionDescriptor.swift:10-148
  AttentionKernelDescriptor  Sources/FlashAttention/Attention/AttentionKernelDescriptor.swift:8-49
  AttentionKernelType        Sources/FlashAttention/Attention/AttentionKernelType.swift:10-23
  AttentionOperand           Sources/FlashAttention/Attention/AttentionOperand.swift:9-71
  AttentionKernel            Sources/FlashAttention/Attention/AttentionKernel/AttentionKernel.swift:10-51
  GEMMOperandPrecision       Sources/FlashAttention/GEMM/GEMMOperandPrecision.swift:33-60
Where the reference calls fatalError this raises MFAError (status codes of include/mfa.h).
All logic lives in the C++ library; this file only marshals.
"""
from __future__ import annotations

import ctypes
import enum
from typing import Dict, Iterable, Mapping, Optional, Sequence, Tuple, Union

from . import _abi
from ._abi import MFAError, check, lib


class GEMMOperandPrecision(enum.IntEnum):
    FP32 = 0
    FP16 = 1
    BF16 = 2

    @property
    def name_in_shader(self) -> str:  # `.name` in Swift (GEMMOperandPrecision.swift:39-48)
        return lib().mfa_precision_name(int(self)).decode()

    @property
    def size(self) -> int:  # GEMMOperandPrecision.swift:51-59
        return int(lib().mfa_precision_size(int(self)))


class AttentionKernelType(enum.IntEnum):
    forward = 0
    backwardQuery = 1
    backwardKeyValue = 2


class AttentionOperand(enum.IntEnum):
    Q = 0
    K = 1
    S = 2
    P = 3
    V = 4
    O = 5  # noqa: E741
    L = 6
    D = 7
    dO = 8
    dV = 9
    dP = 10
    dS = 11
    dK = 12
    dQ = 13

    @property
    def description(self) -> str:
        return lib().mfa_operand_name(int(self)).decode()

    @property
    def bufferBinding(self) -> Optional[int]:
        b = int(lib().mfa_operand_buffer_binding(int(self)))
        return None if b < 0 else b


def _dict_from(array, cast=int) -> Dict[AttentionOperand, object]:
    out = {}
    for op in AttentionOperand:
        v = int(array[int(op)])
        if v >= 0:
            out[op] = cast(v)
    return out


def _fill(array, mapping: Mapping[AttentionOperand, object]) -> None:
    for i in range(_abi.MFA_OPERAND_COUNT):
        array[i] = -1
    for op, v in mapping.items():
        array[int(AttentionOperand(op))] = int(v)


class AttentionKernelDescriptor:
    """AttentionKernelDescriptor.swift:8-49 (every field optional / a dictionary)."""

    def __init__(self):
        self.blockDimensions: Optional[Tuple[int, int, int]] = None  # (parallelization, traversal, head)
        self.cacheState: Dict[AttentionOperand, bool] = {}
        self.headDimension: Optional[int] = None
        self.memoryPrecisions: Dict[AttentionOperand, GEMMOperandPrecision] = {}
        self.preferAsyncCache: Optional[bool] = None
        self.preferAsyncLoad: Optional[bool] = None
        self.registerPrecisions: Dict[AttentionOperand, GEMMOperandPrecision] = {}
        self.transposeState: Dict[AttentionOperand, bool] = {}
        self.type: Optional[AttentionKernelType] = None
        # extension: True = block dimensions / cache state no compiled variant implements are an error
        self.strictBlockDimensions: bool = False

    def _to_c(self) -> _abi.mfa_attention_kernel_descriptor:
        c = _abi.mfa_attention_kernel_descriptor()
        lib().mfa_attention_kernel_descriptor_init(ctypes.byref(c))
        if self.blockDimensions is not None:
            c.hasBlockDimensions = 1
            c.parallelization, c.traversal, c.headBlock = (int(x) for x in self.blockDimensions)
        if self.headDimension is not None:
            c.hasHeadDimension = 1
            c.headDimension = int(self.headDimension)
        _fill(c.cacheState, {k: int(bool(v)) for k, v in self.cacheState.items()})
        _fill(c.memoryPrecisions, self.memoryPrecisions)
        _fill(c.registerPrecisions, self.registerPrecisions)
        _fill(c.transposeState, {k: int(bool(v)) for k, v in self.transposeState.items()})
        c.preferAsyncCache = -1 if self.preferAsyncCache is None else int(bool(self.preferAsyncCache))
        c.preferAsyncLoad = -1 if self.preferAsyncLoad is None else int(bool(self.preferAsyncLoad))
        c.type = -1 if self.type is None else int(self.type)
        c.strictBlockDimensions = int(bool(self.strictBlockDimensions))
        return c

    @classmethod
    def _from_c(cls, c: _abi.mfa_attention_kernel_descriptor) -> "AttentionKernelDescriptor":
        d = cls()
        if c.hasBlockDimensions:
            d.blockDimensions = (int(c.parallelization), int(c.traversal), int(c.headBlock))
        if c.hasHeadDimension:
            d.headDimension = int(c.headDimension)
        d.cacheState = _dict_from(c.cacheState, bool)
        d.memoryPrecisions = _dict_from(c.memoryPrecisions, GEMMOperandPrecision)
        d.registerPrecisions = _dict_from(c.registerPrecisions, GEMMOperandPrecision)
        d.transposeState = _dict_from(c.transposeState, bool)
        d.preferAsyncCache = None if c.preferAsyncCache < 0 else bool(c.preferAsyncCache)
        d.preferAsyncLoad = None if c.preferAsyncLoad < 0 else bool(c.preferAsyncLoad)
        d.type = None if c.type < 0 else AttentionKernelType(int(c.type))
        return d


class AttentionDescriptor:
    """AttentionDescriptor.swift:10-27."""

    def __init__(self):
        self.lowPrecisionInputs: bool = False          # Q, K, V, dO
        self.lowPrecisionIntermediates: bool = False   # S, P, L, D, dP, dS
        self.matrixDimensions: Optional[Tuple[int, int, int]] = None  # (row, column, head)
        self.transposeState: Optional[Tuple[bool, bool, bool, bool]] = None  # (Q, K, V, O)
        # extension: storage type of low-precision inputs (FP16 = reference behaviour)
        self.lowPrecisionInputType: GEMMOperandPrecision = GEMMOperandPrecision.FP16
        self.lowPrecisionOutputs: bool = False   # extension: O, dQ, dK, dV stored in lowPrecisionInputType

    def _to_c(self) -> _abi.mfa_attention_descriptor:
        c = _abi.mfa_attention_descriptor()
        lib().mfa_attention_descriptor_init(ctypes.byref(c))
        c.lowPrecisionInputs = int(bool(self.lowPrecisionInputs))
        c.lowPrecisionIntermediates = int(bool(self.lowPrecisionIntermediates))
        c.lowPrecisionInputType = int(self.lowPrecisionInputType)
        c.lowPrecisionOutputs = 1 if self.lowPrecisionOutputs else 0
        if self.matrixDimensions is not None:
            c.hasMatrixDimensions = 1
            c.row, c.column, c.head = (int(x) for x in self.matrixDimensions)
        if self.transposeState is not None:
            c.hasTransposeState = 1
            c.transposeQ, c.transposeK, c.transposeV, c.transposeO = (int(bool(x)) for x in self.transposeState)
        return c

    @property
    def memoryPrecisions(self) -> Dict[AttentionOperand, GEMMOperandPrecision]:
        """AttentionDescriptor+Precisions.swift:10-146."""
        out = (ctypes.c_int8 * _abi.MFA_OPERAND_COUNT)()
        c = self._to_c()
        check(lib().mfa_attention_descriptor_memory_precisions(ctypes.byref(c), out))
        return _dict_from(out, GEMMOperandPrecision)

    @property
    def registerPrecisions(self) -> Dict[AttentionOperand, GEMMOperandPrecision]:
        """AttentionDescriptor+Precisions.swift:149-215."""
        out = (ctypes.c_int8 * _abi.MFA_OPERAND_COUNT)()
        c = self._to_c()
        check(lib().mfa_attention_descriptor_register_precisions(ctypes.byref(c), out))
        return _dict_from(out, GEMMOperandPrecision)

    def kernelDescriptor(self, type: AttentionKernelType) -> AttentionKernelDescriptor:  # noqa: A002
        """AttentionDescriptor.swift:33-130."""
        c = self._to_c()
        out = _abi.mfa_attention_kernel_descriptor()
        check(lib().mfa_attention_descriptor_kernel_descriptor(ctypes.byref(c), int(type), ctypes.byref(out)))
        return AttentionKernelDescriptor._from_c(out)


def parameterFile(type: AttentionKernelType, mixed: bool) -> str:  # noqa: A002
    """AttentionDescriptor.parameterFile(type:) (+Parameters.swift:13-39) for gfx950."""
    buf = ctypes.create_string_buffer(8192)
    check(lib().mfa_parameter_table_get(int(type), int(bool(mixed)), buf, len(buf)))
    return buf.value.decode()


def setParameterFile(type: AttentionKernelType, mixed: bool, text: str) -> None:  # noqa: A002
    """Install a parameter table (text format of AttentionParameterRow.parseTable).  `mixed=True` is the table consulted for
    every descriptor with lowPrecisionInputs (16-bit Q, K, V -> the 16-bit matrix-core code objects), with or without
    lowPrecisionIntermediates; `mixed=False` the table of FP32 inputs.  The reference consults its mixed tables only when
    BOTH flags are set (+Parameters.swift:16) -- include/mfa.h and DESIGN.md 5 give the reason for the departure."""
    check(lib().mfa_parameter_table_set(int(type), int(bool(mixed)), text.encode()))


def resetParameterFiles() -> None:
    check(lib().mfa_parameter_table_reset())


def selectParameterRow(text: str, headDimension: int) -> dict:
    """AttentionParameterRow.parseTable + AttentionDescriptor.row(table:)
    (AttentionParameterRow.swift:22-74, +Parameters.swift:41-66)."""
    row = _abi.mfa_parameter_row()
    check(lib().mfa_parameter_table_select(text.encode(), int(headDimension), ctypes.byref(row)))
    return {
        "maximumHeadDimension": int(row.maximumHeadDimension),
        "parallelization": int(row.parallelization),
        "traversal": int(row.traversal),
        "head": int(row.head),
        "cachedOperands": [op for op in AttentionOperand if row.cached[int(op)] == 1],
    }


BufferLike = Union[int, None, object]


def _pointer(b: BufferLike) -> Optional[int]:
    if b is None:
        return None
    if isinstance(b, int):
        return b
    if hasattr(b, "data_ptr"):  # torch.Tensor (device memory plumbing only)
        return int(b.data_ptr())
    raise TypeError(f"cannot take a device pointer from {type(b)!r}")


class AttentionKernel:
    """AttentionKernel.swift:10-51.  `dispatch` stands in for what the reference's callers do with
    createSource(): makeLibrary + makeComputePipelineState + setBuffer(0...9) +
    setThreadgroupMemoryLength + dispatchThreadgroups (SquareAttentionTest.swift:244-260, :319-368)."""

    def __init__(self, descriptor: AttentionKernelDescriptor):
        self._handle = ctypes.c_void_p()
        c = descriptor._to_c()
        check(lib().mfa_attention_kernel_create(ctypes.byref(c), ctypes.byref(self._handle)))
        self.type = descriptor.type

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h is not None and h.value and _abi._lib is not None:   # module globals may be gone at exit
            _abi._lib.mfa_attention_kernel_destroy(h)
            self._handle = ctypes.c_void_p()

    @property
    def blockDimensions(self) -> Tuple[int, int, int]:
        p, t, h = ctypes.c_uint16(), ctypes.c_uint16(), ctypes.c_uint16()
        check(lib().mfa_attention_kernel_block_dimensions(self._handle, ctypes.byref(p), ctypes.byref(t), ctypes.byref(h)))
        return (p.value, t.value, h.value)

    @property
    def threadgroupSize(self) -> int:
        return int(lib().mfa_attention_kernel_threadgroup_size(self._handle))

    @property
    def threadgroupMemoryAllocation(self) -> int:
        return int(lib().mfa_attention_kernel_threadgroup_memory_allocation(self._handle))

    @property
    def variant(self) -> str:
        return lib().mfa_attention_kernel_variant(self._handle).decode()

    @property
    def fallbackVariant(self) -> str:
        """the general code object behind launches the selected variant cannot take ("" if it is the general one)"""
        return lib().mfa_attention_kernel_fallback_variant(self._handle).decode()

    @property
    def needsWorkspaceForFastPath(self) -> bool:
        """transposed operands: the matrix-core kernel runs on row-major copies in the caller's workspace (workspaceSize);
        without a workspace the launch runs `fallbackVariant`"""
        return bool(lib().mfa_attention_kernel_needs_workspace_for_fast_path(self._handle))

    @property
    def effectiveDescriptor(self) -> AttentionKernelDescriptor:
        out = _abi.mfa_attention_kernel_descriptor()
        check(lib().mfa_attention_kernel_effective_descriptor(self._handle, ctypes.byref(out)))
        return AttentionKernelDescriptor._from_c(out)

    # -- launch ------------------------------------------------------------------------------
    @staticmethod
    def _marshal(buffers, row, column, heads, batches, leadingDimensions, headStrides, batchStrides,
                 workspace=None, causal=False, rowLengths=None, columnLengths=None, blockMask=None,
                 blockMaskWords=0, blockMaskStrides=(0, 0)):
        """`buffers`: dict {AttentionOperand: tensor | int} or a 10-sequence indexed by bufferBinding."""
        slots = [None] * _abi.MFA_BUFFER_SLOTS
        if isinstance(buffers, Mapping):
            for op, b in buffers.items():
                binding = AttentionOperand(op).bufferBinding
                if binding is None:
                    raise ValueError(f"operand {AttentionOperand(op).description} has no buffer binding")
                slots[binding] = b
        else:
            for i, b in enumerate(buffers):
                slots[i] = b
        arr = (ctypes.c_void_p * _abi.MFA_BUFFER_SLOTS)(*[_pointer(b) for b in slots])
        params = _abi.mfa_launch_params()
        lib().mfa_launch_params_init(ctypes.byref(params))
        params.row, params.column, params.heads, params.batches = int(row), int(column), int(heads), int(batches)
        for name, src in (("leadingDimension", leadingDimensions), ("headStride", headStrides),
                          ("batchStride", batchStrides)):
            if src:
                dst = getattr(params, name)
                for op, v in src.items():
                    dst[AttentionOperand(op).bufferBinding] = int(v)
        params.causal = int(bool(causal))
        # variable sequence lengths (extension): device arrays of `batches` uint32 / int32 entries
        params.rowLengths = _pointer(rowLengths)
        params.columnLengths = _pointer(columnLengths)
        # block-sparse mask (extension): device bitmap, one bit per 256 x 128 block, blockMaskWords words per row block
        params.blockMask = _pointer(blockMask)
        params.blockMaskWords = int(blockMaskWords)
        params.blockMaskHeadStride, params.blockMaskBatchStride = (int(x) for x in blockMaskStrides)
        if workspace is not None:   # caller-owned scratch for column-parallel forward launches
            params.workspace = _pointer(workspace)
            params.workspaceBytes = int(workspace.numel() * workspace.element_size()) \
                if hasattr(workspace, "numel") else int(getattr(workspace, "nbytes"))
        return arr, params, slots

    def workspaceSize(self, *, row: int, column: int, heads: int = 1, batches: int = 1) -> int:
        """Bytes of scratch a forward launch of this shape would use if given a workspace (0 = the
        launch fills the GPU without splitting the key range)."""
        params = _abi.mfa_launch_params()
        lib().mfa_launch_params_init(ctypes.byref(params))
        params.row, params.column, params.heads, params.batches = int(row), int(column), int(heads), int(batches)
        out = ctypes.c_uint64()
        check(lib().mfa_attention_kernel_workspace_size(self._handle, ctypes.byref(params), ctypes.byref(out)))
        return int(out.value)

    def dispatch(self, buffers, *, row: int, column: int, heads: int = 1, batches: int = 1,
                 leadingDimensions: Optional[Mapping] = None, headStrides: Optional[Mapping] = None,
                 batchStrides: Optional[Mapping] = None, stream: Optional[int] = None,
                 workspace=None, causal: bool = False, rowLengths=None, columnLengths=None, blockMask=None,
                 blockMaskWords: int = 0, blockMaskStrides=(0, 0)) -> None:
        arr, params, _keep = self._marshal(buffers, row, column, heads, batches, leadingDimensions,
                                           headStrides, batchStrides, workspace, causal, rowLengths, columnLengths,
                                           blockMask, blockMaskWords, blockMaskStrides)
        check(lib().mfa_attention_kernel_launch(self._handle, ctypes.byref(arr), ctypes.byref(params),
                                                ctypes.c_void_p(stream or 0)))

    def launchForm(self, buffers, *, row: int, column: int, heads: int = 1, batches: int = 1,
                   leadingDimensions: Optional[Mapping] = None, headStrides: Optional[Mapping] = None,
                   batchStrides: Optional[Mapping] = None, workspace=None, causal: bool = False, rowLengths=None,
                   columnLengths=None, blockMask=None, blockMaskWords: int = 0, blockMaskStrides=(0, 0)) -> str:
        """What `dispatch` with the same arguments would run (nothing is launched): the variant's own kernel, the general
        kernel, column-parallel pieces + combine, a re-layout pass in front, or the persistent form `attn_fwd16_p4p` --
        the name rocprofv3 shows for dense / causal forward launches at D <= 128 (mfa_attention_kernel_launch_form)."""
        arr, params, _keep = self._marshal(buffers, row, column, heads, batches, leadingDimensions, headStrides, batchStrides,
                                           workspace, causal, rowLengths, columnLengths, blockMask, blockMaskWords, blockMaskStrides)
        out = ctypes.create_string_buffer(512)
        check(lib().mfa_attention_kernel_launch_form(self._handle, ctypes.byref(arr), ctypes.byref(params), out, len(out)))
        return out.value.decode()

    def time(self, buffers, *, row: int, column: int, heads: int = 1, batches: int = 1,
             leadingDimensions: Optional[Mapping] = None, headStrides: Optional[Mapping] = None,
             batchStrides: Optional[Mapping] = None, stream: Optional[int] = None,
             warmup: int = 1, iterations: int = 5, workspace=None, causal: bool = False) -> float:
        """Milliseconds for `iterations` back-to-back launches (HIP events on `stream`)."""
        arr, params, _keep = self._marshal(buffers, row, column, heads, batches, leadingDimensions,
                                           headStrides, batchStrides, workspace, causal)
        ms = ctypes.c_float()
        check(lib().mfa_attention_kernel_time(self._handle, ctypes.byref(arr), ctypes.byref(params),
                                              ctypes.c_void_p(stream or 0), int(warmup), int(iterations),
                                              ctypes.byref(ms)))
        return float(ms.value)


def deviceCount() -> int:
    n = ctypes.c_int()
    check(lib().mfa_device_count(ctypes.byref(n)))
    return n.value


def deviceName(device: int = 0) -> str:
    buf = ctypes.create_string_buffer(256)
    check(lib().mfa_device_name(device, buf, len(buf)))
    return buf.value.decode()
