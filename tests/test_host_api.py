"""CPU tests: the C-ABI library loads, exports every symbol include/mfa.h declares, and the host
logic (descriptors, precision policy, parameter tables, error codes) behaves like the reference's
Swift types.  No compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

import metal_flash_attention_amd as mfa
from metal_flash_attention_amd import (
    AttentionDescriptor, AttentionKernel, AttentionKernelDescriptor, AttentionKernelType,
    AttentionOperand, GEMMOperandPrecision, MFAError, _abi,
)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
T = AttentionKernelType
Op = AttentionOperand
P = GEMMOperandPrecision


@pytest.fixture(scope="module", autouse=True)
def _built(built_library):
    yield


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "mfa.h")).read()
    declared = set(re.findall(r"\b(mfa_[a-z0-9_]+)\s*\(", header))
    handle = ctypes.CDLL(_abi.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(handle, name), f"{name} declared in include/mfa.h but not exported"
    assert declared == {s[0] for s in _abi.SYMBOLS}, "ctypes table out of sync with the header"
    assert _abi.lib().mfa_abi_version() == 4


def test_struct_layouts_match_header():
    # sizes are part of the ABI: a cgo/JNI/Swift binding relies on them
    assert ctypes.sizeof(_abi.mfa_attention_descriptor) == 24
    assert ctypes.sizeof(_abi.mfa_attention_kernel_descriptor) == 70
    assert ctypes.sizeof(_abi.mfa_launch_params) == 16 + 3 * 80 + 16 + 8 + 16 + 32
    assert ctypes.sizeof(_abi.mfa_parameter_row) == 22


def test_operand_bindings_and_precisions():
    # AttentionOperand.swift:52-71 ; GEMMOperandPrecision.swift:33-60
    want = {"Q": 0, "K": 1, "V": 2, "O": 3, "L": 4, "D": 5, "dO": 6, "dV": 7, "dK": 8, "dQ": 9}
    for op in Op:
        assert op.description == op.name
        assert op.bufferBinding == want.get(op.name)
    assert [(p.value, p.name_in_shader, p.size) for p in P] == [(0, "float", 4), (1, "half", 2), (2, "bfloat", 2)]


def test_incomplete_descriptor_is_an_error_not_an_abort():
    d = AttentionDescriptor()
    with pytest.raises(MFAError) as e:
        d.kernelDescriptor(T.forward)
    assert e.value.status == 1 and "Descriptor was incomplete." in str(e.value)
    d.matrixDimensions = (8, 8, 8)
    with pytest.raises(MFAError):
        d.kernelDescriptor(T.forward)  # transposeState still missing
    with pytest.raises(MFAError) as e:
        AttentionKernel(AttentionKernelDescriptor())
    assert e.value.status == 1


def _desc(dims=(64, 64, 64), low_in=False, low_mid=False, tr=(False,) * 4, in_type=P.FP16):
    d = AttentionDescriptor()
    d.lowPrecisionInputs, d.lowPrecisionIntermediates = low_in, low_mid
    d.matrixDimensions, d.transposeState, d.lowPrecisionInputType = dims, tr, in_type
    return d


def test_memory_precisions_policy():
    # AttentionDescriptor+Precisions.swift:10-146
    m = _desc().memoryPrecisions
    assert all(m[o] == P.FP32 for o in (Op.Q, Op.K, Op.V, Op.dO, Op.L, Op.D, Op.O, Op.dV, Op.dK, Op.dQ))
    m = _desc(low_in=True, low_mid=True).memoryPrecisions
    assert (m[Op.Q], m[Op.K], m[Op.V], m[Op.dO]) == (P.FP16, P.FP16, P.FP16, P.BF16)
    assert (m[Op.L], m[Op.D]) == (P.FP16, P.BF16)
    assert all(m[o] == P.FP32 for o in (Op.O, Op.dV, Op.dK, Op.dQ))  # always FP32 (:140-143)
    m = _desc(low_in=True, in_type=P.BF16).memoryPrecisions
    assert (m[Op.Q], m[Op.K], m[Op.V], m[Op.dO], m[Op.L]) == (P.BF16, P.BF16, P.BF16, P.BF16, P.FP32)
    r = _desc(low_in=True, low_mid=True).registerPrecisions
    assert (r[Op.S], r[Op.P], r[Op.dP], r[Op.dS]) == (P.FP16, P.FP16, P.FP32, P.BF16)
    assert _desc(low_mid=True).registerPrecisions[Op.S] == P.FP32  # (:197) S is FP16 only with both flags


def test_kernel_descriptor_follows_reference_rules():
    d = _desc(dims=(100, 50, 3), tr=(True, False, True, False))
    for t, expected in ((T.forward, {Op.Q, Op.O}), (T.backwardQuery, {Op.Q, Op.dO, Op.dQ}),
                        (T.backwardKeyValue, {Op.K, Op.V, Op.dV, Op.dK})):
        k = d.kernelDescriptor(t)
        assert set(k.cacheState) == expected                     # AttentionDescriptor.swift:56-86
        assert k.blockDimensions[2] == 8                          # head block <= pad8(D) (:48-53)
        assert k.headDimension == 3 and k.type == t
        ts = k.transposeState                                     # gradients inherit (:95-111)
        assert (ts[Op.Q], ts[Op.K], ts[Op.V], ts[Op.O]) == (True, False, True, False)
        assert (ts[Op.dQ], ts[Op.dK], ts[Op.dV], ts[Op.dO]) == (True, False, True, False)
        assert k.preferAsyncLoad is True and k.preferAsyncCache is False


def test_parameter_table_text_format_and_row_selection():
    # the reference's own M1 forward-mixed table (AttentionDescriptor+Parameters.swift:118-122)
    table = "| 96  | 32 | 128 | 32 | Q, O |\n| 128 | 32 | 128 | 32 | Q    |\n| 384 | 32 | 128 | 32 |      |\n\n"
    assert mfa.selectParameterRow(table, 64) == dict(maximumHeadDimension=96, parallelization=32, traversal=128,
                                                     head=32, cachedOperands=[Op.Q, Op.O])
    assert mfa.selectParameterRow(table, 97)["cachedOperands"] == [Op.Q]
    assert mfa.selectParameterRow(table, 128)["maximumHeadDimension"] == 128
    assert mfa.selectParameterRow(table, 1000)["maximumHeadDimension"] == 384   # else last (+Parameters.swift:60-65)
    for bad in ("| 96 | 32 | 128 | 32 |\n", "| x | 32 | 128 | 32 | Q |\n", "| 96 | 32 | 128 | 32 | Z |\n", ""):
        with pytest.raises(MFAError) as e:
            mfa.selectParameterRow(bad, 64)
        assert e.value.status == 5


def test_parameter_tables_can_be_replaced_and_validated():
    try:
        for t in T:
            for mixed in (False, True):
                text = mfa.parameterFile(t, mixed)
                assert text.count("|") % 6 == 0 and mfa.selectParameterRow(text, 128)["parallelization"] % 32 == 0
        mfa.setParameterFile(T.forward, False, "| 384 | 64 | 32 | 16 | O |\n")
        k = _desc(dims=(10, 10, 100)).kernelDescriptor(T.forward)
        assert k.blockDimensions == (64, 32, 16) and k.cacheState == {Op.Q: False, Op.O: True}
        # an operand the kernel type does not own is rejected (AttentionDescriptor.swift:68-73)
        mfa.setParameterFile(T.forward, False, "| 384 | 64 | 32 | 16 | K |\n")
        with pytest.raises(MFAError) as e:
            _desc().kernelDescriptor(T.forward)
        assert "Unexpected operand: K" in str(e.value)
        with pytest.raises(MFAError):
            mfa.setParameterFile(T.forward, False, "| 384 | 64 | 32 |\n")
    finally:
        mfa.resetParameterFiles()
    assert _desc().kernelDescriptor(T.forward).blockDimensions[0] % 32 == 0


def test_kernel_object_reports_geometry_without_a_gpu():
    for dims in ((10, 10, 3), (4096, 4096, 128), (777, 777, 199)):
        for t in T:
            k = AttentionKernel(_desc(dims=dims).kernelDescriptor(t))
            par, trav, head = k.blockDimensions
            assert par % 32 == 0 and trav % 32 == 0 and head >= dims[2]
            assert k.threadgroupSize == 64 * (par // 32)                  # one wave per 32 rows/columns
            assert 0 < k.threadgroupMemoryAllocation <= 160 * 1024          # LDS per CU
            assert k.variant
            eff = k.effectiveDescriptor
            assert eff.blockDimensions == (par, trav, head)
    with pytest.raises(MFAError) as e:
        AttentionKernel(_desc(dims=(16, 16, 1000)).kernelDescriptor(T.forward))
    assert e.value.status == 3


def test_launch_argument_validation_happens_before_any_gpu_call():
    k = AttentionKernel(_desc().kernelDescriptor(T.forward))
    with pytest.raises(MFAError) as e:
        k.dispatch({Op.Q: 4096, Op.K: 4096, Op.V: 4096, Op.O: 4096}, row=64, column=64)  # L missing
    assert e.value.status == 2 and "operand L" in str(e.value)
    with pytest.raises(MFAError):
        k.dispatch({Op.Q: 4096, Op.K: 4096, Op.V: 4096, Op.O: 4096, Op.L: 4096}, row=0, column=64)
    with pytest.raises(MFAError):
        k.dispatch({Op.Q: 4096, Op.K: 4096, Op.V: 4096, Op.O: 4096, Op.L: 4096}, row=64, column=64,
                   leadingDimensions={Op.Q: 8})


def test_workspace_size_query_follows_the_split_heuristic():
    """Column-parallel forward (extension): only 16-bit row-major forward kernels split, only when the
    row-parallel grid cannot fill 256 CUs and the key range is long; the size is
    splits x heads x batches x R x (D + 2) floats."""
    low = _desc(dims=(4096, 4096, 64), low_in=True, in_type=P.BF16)
    k = AttentionKernel(low.kernelDescriptor(T.forward))
    one = k.workspaceSize(row=4096, column=4096)
    assert one > 0 and one % (4096 * 66 * 4) == 0
    splits = one // (4096 * 66 * 4)
    assert 2 <= splits <= 16
    assert k.workspaceSize(row=4096, column=4096, heads=32, batches=8) == 0     # already 4096 workgroups
    assert k.workspaceSize(row=4096, column=256) == 0                           # traversal too short
    assert AttentionKernel(_desc(dims=(4096, 4096, 64)).kernelDescriptor(T.forward)).workspaceSize(row=4096, column=4096) == 0
    # the 16-bit backward kernels split their traversal too: dQ slabs (splits x R x D floats), dV + dK slabs
    dq = AttentionKernel(low.kernelDescriptor(T.backwardQuery)).workspaceSize(row=4096, column=4096)
    dkv = AttentionKernel(low.kernelDescriptor(T.backwardKeyValue)).workspaceSize(row=4096, column=4096)
    assert dq > 0 and dq % (4096 * 64 * 4) == 0 and dkv > 0 and dkv % (2 * 4096 * 64 * 4) == 0
    assert AttentionKernel(_desc(dims=(4096, 4096, 64)).kernelDescriptor(T.backwardQuery)).workspaceSize(row=4096, column=4096) == 0


def test_oversized_slices_route_to_general_kernels_without_a_gpu():
    """The 16-bit kernels use 32-bit byte offsets per (head, batch) slice; a launch whose slice would
    exceed them must be planned on the general kernel.  The plan is visible through the workspace query
    (only the 16-bit forward kernel ever asks for one) -- no GPU call is made."""
    d = _desc(dims=(4096, 4096, 64), low_in=True, in_type=P.BF16)
    k = AttentionKernel(d.kernelDescriptor(T.forward))
    assert k.variant.startswith("attn_fwd16")
    assert k.workspaceSize(row=4096, column=4096) > 0
