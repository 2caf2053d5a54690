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
    assert _abi.lib().mfa_abi_version() == _abi.EXPECTED_ABI == 6


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
    # any head dimension: beyond 384 the D-blocked kernels that page the accumulators through the FP32 output buffers
    # (+Accumulate.swift:403-469); nothing cached, and no fused 16-bit output cast there
    k = AttentionKernel(_desc(dims=(16, 16, 1000)).kernelDescriptor(T.forward))
    assert k.variant == "attn_paged_fwd_f32_any_d" and k.blockDimensions == (32, 32, 64)
    assert not any(k.effectiveDescriptor.cacheState.values())
    low = _desc(dims=(16, 16, 1000), low_in=True, in_type=P.BF16)
    low.lowPrecisionOutputs = True
    with pytest.raises(MFAError) as e:
        AttentionKernel(low.kernelDescriptor(T.forward))
    assert e.value.status == 3 and "must be FP32" in str(e.value)


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


def test_piece_count_of_one_head_launches():
    """round 6 (`choose_splits`, profiles/r06_sweep_splits.txt): the piece count is rounded DOWN to one round of workgroups (24 row
    blocks x 11 pieces = 264 workgroups ran a second round of eight), capped near sqrt(117 traversal / parallel) (the combine pass reads
    one slab per piece) and prefers equal pieces of whole 256-key blocks (what the persistent split streams serve)"""
    for N, D, want in ((6144, 128, 8), (4096, 64, 8), (4096, 128, 8), (3072, 128, 6), (2048, 64, 8), (8192, 128, 8), (16384, 128, 4)):
        low = _desc(dims=(N, N, D), low_in=True, in_type=P.BF16)
        k = AttentionKernel(low.kernelDescriptor(T.forward))
        assert k.workspaceSize(row=N, column=N) == want * N * (D + 2) * 4, (N, D)
        dq = AttentionKernel(low.kernelDescriptor(T.backwardQuery)).workspaceSize(row=N, column=N)
        assert dq % (N * D * 4) == 0 and 2 <= dq // (N * D * 4) <= 11 and (N // 256) * (dq // (N * D * 4)) <= 256
    # D > 128 (round 6): the role-split backward kernels cut the launch themselves -- 128-row / 128-key workgroups, at most 256 of them
    low = _desc(dims=(8192, 8192, 256), low_in=True, in_type=P.BF16)
    assert AttentionKernel(low.kernelDescriptor(T.backwardQuery)).workspaceSize(row=8192, column=8192) == 4 * 8192 * 256 * 4
    assert AttentionKernel(low.kernelDescriptor(T.backwardKeyValue)).workspaceSize(row=8192, column=8192) == 2 * 4 * 8192 * 256 * 4


def test_oversized_slices_route_to_general_kernels_without_a_gpu():
    """The 16-bit kernels use 32-bit byte offsets per (head, batch) slice; a launch whose slice would
    exceed them must be planned on the general kernel.  The plan is visible through the workspace query
    (only the 16-bit forward kernel ever asks for one) -- no GPU call is made."""
    d = _desc(dims=(4096, 4096, 64), low_in=True, in_type=P.BF16)
    k = AttentionKernel(d.kernelDescriptor(T.forward))
    assert k.variant.startswith("attn_fwd16")
    assert k.workspaceSize(row=4096, column=4096) > 0


# ---- the parameter table drives kernel selection (AttentionDescriptor.swift:37-54 -> AttentionKernel.swift:27-50) ----
def _low(dims, in_type=P.BF16, low_mid=False):
    d = _desc(dims=dims, low_in=True, low_mid=low_mid)
    d.lowPrecisionInputType = in_type
    return d


def test_default_table_rows_are_compiled_variants():
    """every row of every built-in table is (parallelization, traversal, head, cached operands) of a code object that
    exists: creating the kernel with strictBlockDimensions succeeds and the effective descriptor equals the request"""
    for t in T:
        for mixed in (False, True):
            rows = [r for r in mfa.parameterFile(t, mixed).split("\n") if r.strip()]
            for r in rows:
                max_d = int(r.split("|")[1])
                for D in {max_d, max(8, max_d - 8)}:
                    d = _low((512, 512, D)) if mixed else _desc(dims=(512, 512, D))
                    kd = d.kernelDescriptor(t)
                    kd.strictBlockDimensions = True
                    k = AttentionKernel(kd)     # MFAError (status 3) if no variant implements the row
                    eff = k.effectiveDescriptor
                    assert eff.blockDimensions[:2] == kd.blockDimensions[:2], (t, mixed, D, k.variant)
                    assert eff.blockDimensions[2] >= kd.blockDimensions[2]
                    assert eff.cacheState == kd.cacheState, (t, mixed, D, k.variant)


def test_parameter_table_row_selects_the_code_object():
    """forward, 16-bit inputs, D = 128: the default row (256, 64, 128) is the four-wave hand-placed kernel; a row with 32-key
    steps selects the eight-wave kernel; a tuple nothing implements falls back (reported) or fails (strict)"""
    try:
        d = _low((4096, 4096, 128))
        assert AttentionKernel(d.kernelDescriptor(T.forward)).variant.startswith("attn_fwd16p4_bf16_d128_w4x64")
        mfa.setParameterFile(T.forward, True, "| 64 | 256 | 32 | 64 | Q, O |\n| 128 | 256 | 32 | 128 | Q, O |\n| 256 | 128 | 32 | 256 | Q, O |\n")
        k = AttentionKernel(d.kernelDescriptor(T.forward))
        assert k.variant.startswith("attn_fwd16v3_bf16_d128_w8x32") and k.blockDimensions == (256, 32, 128)
        # D <= 64: the default row (256, 64, 64) is the persistent four-wave kernel (round 5); 32-key steps select the eight-wave one;
        # D <= 32 keeps its own 4 x 32 object by default and takes the four-wave kernel (zero-padded chunks) with a (256, 64, 64) row
        d64, d32 = _low((4096, 4096, 64), low_mid=True), _low((4096, 4096, 32), low_mid=True)
        assert AttentionKernel(d64.kernelDescriptor(T.forward)).variant.startswith("attn_fwd16v3_bf16_d64_w8x32")
        mfa.resetParameterFiles()
        assert AttentionKernel(d64.kernelDescriptor(T.forward)).variant == "attn_fwd16p6_bf16_d64_w4x64_thr8_fold"
        assert AttentionKernel(d32.kernelDescriptor(T.forward)).variant.startswith("attn_fwd16v3_bf16_d32_w4x32")
        mfa.setParameterFile(T.forward, True, "| 32 | 256 | 64 | 64 | Q, O |\n| 64 | 256 | 64 | 64 | Q, O |\n| 128 | 256 | 64 | 128 | Q, O |\n")
        for dd in (d64, d32):
            k = AttentionKernel(dd.kernelDescriptor(T.forward))
            assert k.variant == "attn_fwd16p6_bf16_d64_w4x64_thr8_fold" and k.blockDimensions == (256, 64, 64), k.variant
        mfa.setParameterFile(T.backwardKeyValue, True, "| 128 | 128 | 64 | 128 | K, V, dV, dK |\n")
        assert AttentionKernel(d.kernelDescriptor(T.backwardKeyValue)).variant.startswith("attn_dkv16_bf16_d128_w4x32")
        mfa.setParameterFile(T.backwardKeyValue, True, "| 128 | 128 | 32 | 128 | K, V, dV, dK |\n")
        assert AttentionKernel(d.kernelDescriptor(T.backwardKeyValue)).variant.startswith("attn_dkv16rs_bf16_d128")
        mfa.resetParameterFiles()   # default row (256, 32, 128): four waves x 64 keys; the stream follows the types of L and D
        assert AttentionKernel(d.kernelDescriptor(T.backwardKeyValue)).variant == "attn_dkv16p4_bf16_d128_w4x64_exact"
        dm = _low((4096, 4096, 128), low_mid=True)
        assert AttentionKernel(dm.kernelDescriptor(T.backwardKeyValue)).variant == "attn_dkv16p4_bf16_d128_w4x64"
        # nothing implements 64 rows x 128 keys with Q streamed: nearest variant + report, or an error when strict
        mfa.setParameterFile(T.forward, True, "| 384 | 64 | 128 | 32 | O |\n")
        kd = d.kernelDescriptor(T.forward)
        k = AttentionKernel(kd)
        assert k.effectiveDescriptor.blockDimensions != kd.blockDimensions and k.effectiveDescriptor.cacheState[Op.Q] is True
        kd.strictBlockDimensions = True
        with pytest.raises(MFAError) as e:
            AttentionKernel(kd)
        assert e.value.status == 3 and "compiled (parallelization, traversal, head)" in str(e.value)
    finally:
        mfa.resetParameterFiles()


def test_transposed_descriptors_select_the_in_place_forward_code_objects():
    """transposeState with 16-bit inputs (no GPU needed to plan): the forward kernel has a code object per pattern of (K, V) and
    head-dimension bucket that reads / writes the operands where they lie -- no workspace; the backward kernels keep the
    re-layout path (workspace) with the general kernel behind it; FP32 inputs run the general kernel, which reads any layout."""
    suffix = {(False, False): "_tr", (True, False): "_tr_k", (False, True): "_tr_v", (True, True): "_tr_kv"}
    for D, bucket in ((24, 32), (64, 64), (72, 128), (128, 128), (136, 160), (192, 192), (200, 256), (256, 256)):
        for in_type, tname in ((P.BF16, "bf16"), (P.FP16, "f16")):
            for (tk, tv), suf in suffix.items():
                for tq, to in ((True, False), (False, True)):
                    if not (tq or tk or tv or to):
                        continue
                    d = _desc(dims=(512, 512, D), low_in=True, in_type=in_type, tr=(tq, tk, tv, to))
                    k = AttentionKernel(d.kernelDescriptor(T.forward))
                    assert k.variant.startswith("attn_fwd16v3_%s_d%d_" % (tname, bucket)) and k.variant.endswith(suf), (D, k.variant)
                    assert not k.needsWorkspaceForFastPath and k.workspaceSize(row=512, column=512) == 0
                    assert k.fallbackVariant.startswith("attn_generic")
    d = _desc(dims=(512, 512, 128), low_in=True, in_type=P.BF16, tr=(False, True, True, False))
    for t in (T.backwardQuery, T.backwardKeyValue):
        k = AttentionKernel(d.kernelDescriptor(t))
        assert k.needsWorkspaceForFastPath and k.workspaceSize(row=512, column=512) > 0 and not k.variant.startswith("attn_generic")
    k = AttentionKernel(_desc(dims=(512, 512, 128), tr=(True, True, True, True)).kernelDescriptor(T.forward))   # FP32 inputs
    assert k.variant.startswith("attn_generic") and not k.needsWorkspaceForFastPath
    k = AttentionKernel(_desc(dims=(512, 512, 260), low_in=True, in_type=P.BF16, tr=(True,) * 4).kernelDescriptor(T.forward))   # D > 256
    assert k.variant.startswith("attn_generic")


def test_transposed_launches_are_routed_to_the_in_place_streams():
    """K^T + V^T at the buckets 160 / 192 / 256 keep their variant, and a launch of whole 32-key steps of aligned rows is handed
    to attn_fwd16_p5_tr (transposed operands read where they lie, AttentionKernel.swift:189-204) -- planned without a GPU (host
    pointers only decide the alignment)."""
    torch = pytest.importorskip("torch")
    N = 512
    for D, bucket in ((136, 160), (192, 192), (200, 256), (256, 256)):
        for low_mid in (False, True):
            d = _desc(dims=(N, N, D), low_in=True, low_mid=low_mid, in_type=P.BF16, tr=(False, True, True, False))
            k = AttentionKernel(d.kernelDescriptor(T.forward))
            assert k.variant.startswith("attn_fwd16v3_bf16_d%d_" % bucket) and k.variant.endswith("_tr_kv")
            for C, stream in ((512, True), (520, False), (32, True)):
                b = {Op.Q: torch.zeros((N, D), dtype=torch.bfloat16), Op.K: torch.zeros((D, C), dtype=torch.bfloat16),
                     Op.V: torch.zeros((D, C), dtype=torch.bfloat16), Op.O: torch.zeros((N, D)),
                     Op.L: torch.zeros(N, dtype=torch.float16 if low_mid else torch.float32)}
                form = k.launchForm(b, row=N, column=C)
                assert form.startswith("attn_fwd16_p5_tr" if stream else "attn_fwd16v3"), (D, C, form)
                assert not stream or ("folded" in form) == low_mid
    # backward kernels, every operand transposed, no workspace: in place when whole tiles / steps of aligned rows (D in (64, 128])
    for in_type in (P.BF16, P.FP16):
        d = _desc(dims=(N, N, 128), low_in=True, low_mid=True, in_type=in_type, tr=(True,) * 4)
        mem = d.memoryPrecisions
        tt = {P.FP32: torch.float32, P.FP16: torch.float16, P.BF16: torch.bfloat16}
        for C, R, (fq, fkv) in ((512, 512, (True, True)), (520, 512, (False, True)), (512, 520, (True, False))):
            b = {op: torch.zeros((128, R if op in (Op.Q, Op.O, Op.dO, Op.dQ) else C), dtype=tt[mem[op]])
                 for op in (Op.Q, Op.K, Op.V, Op.O, Op.dO, Op.dQ, Op.dK, Op.dV)}
            b[Op.L], b[Op.D] = torch.zeros(R, dtype=tt[mem[Op.L]]), torch.zeros(R, dtype=tt[mem[Op.D]])
            for t, name, want in ((T.backwardQuery, "attn_dq16_p4_tr", fq), (T.backwardKeyValue, "attn_dkv16_p4_tr", fkv)):
                form = AttentionKernel(d.kernelDescriptor(t)).launchForm(b, row=R, column=C)
                assert form.startswith(name) == want, (in_type, R, C, t, form)
    # one operand transposed: the stream family of that pattern (generated at build time, model-verified)
    for (tk, tv), which in (((True, False), "transposed K"), ((False, True), "transposed V")):
        d = _desc(dims=(N, N, 256), low_in=True, in_type=P.BF16, tr=(False, tk, tv, False))
        k = AttentionKernel(d.kernelDescriptor(T.forward))
        b = {Op.Q: torch.zeros((N, 256), dtype=torch.bfloat16), Op.K: torch.zeros((256, N) if tk else (N, 256), dtype=torch.bfloat16),
             Op.V: torch.zeros((256, N) if tv else (N, 256), dtype=torch.bfloat16), Op.O: torch.zeros((N, 256)), Op.L: torch.zeros(N)}
        form = k.launchForm(b, row=N, column=N)
        assert form.startswith("attn_fwd16_p5_tr") and which in form and "K / V" not in form, form
        assert k.launchForm(b, row=N, column=N, causal=True).startswith("attn_fwd16_p5_tr")


def test_strict_descriptors_refuse_the_silent_general_kernel_on_transposed_backward_launches():
    """A transposed backward launch without workspace that no in-place kernel takes (every bucket but 128) would run the general
    fp32 kernel, 20-50 x slower than the selected code object: with strictBlockDimensions the launch fails with MFA_ERR_UNSUPPORTED
    and the message names the workspace to pass; without the flag it keeps running (launch form says 'general kernel'); in the
    128 bucket the in-place kernels take it either way.  Planned without a GPU."""
    torch = pytest.importorskip("torch")
    N = 512
    tt = {P.FP32: torch.float32, P.FP16: torch.float16, P.BF16: torch.bfloat16}
    for D, in_place in ((64, False), (128, True), (192, False)):
        d = _desc(dims=(N, N, D), low_in=True, low_mid=True, in_type=P.BF16, tr=(True,) * 4)
        mem = d.memoryPrecisions
        b = {op: torch.zeros((D, N), dtype=tt[mem[op]]) for op in (Op.Q, Op.K, Op.V, Op.O, Op.dO, Op.dQ, Op.dK, Op.dV)}
        b[Op.L], b[Op.D] = torch.zeros(N, dtype=tt[mem[Op.L]]), torch.zeros(N, dtype=tt[mem[Op.D]])
        for t in (T.backwardQuery, T.backwardKeyValue):
            kd = d.kernelDescriptor(t)
            loose = AttentionKernel(kd).launchForm(b, row=N, column=N)
            assert ("general kernel" in loose) == (not in_place), (D, t, loose)
            kd.strictBlockDimensions = True
            k = AttentionKernel(kd)
            if in_place:
                assert k.launchForm(b, row=N, column=N).startswith(("attn_dq16_p4_tr", "attn_dkv16_p4_tr"))
                continue
            with pytest.raises(MFAError) as e:
                k.launchForm(b, row=N, column=N)
            need = k.workspaceSize(row=N, column=N)
            assert e.value.status == 3 and "workspace of %d bytes" % need in str(e.value), str(e.value)
            ws = torch.zeros(need + 256, dtype=torch.uint8)     # with the workspace the re-layout path of the selected code object runs
            off = (-ws.data_ptr()) % 256
            form = k.launchForm(b, row=N, column=N, workspace=ws[off:off + need])
            assert form.startswith("attn_relayout") and "general kernel" not in form, form


def test_fp32_descriptors_are_the_fp32_production_kernels():
    """FP32 descriptors of the 64 / 128 head blocks with row-major operands and D % 4 == 0 ARE the FP32 production kernels
    (csrc/attn_f32.h; round 5: own variant name and own LDS bytes -- what rocprofv3 shows as kernel name and lds_bytes); the general
    kernel is their sibling: it keeps block-sparse launches and launches whose rows are not 16-byte aligned, and the launch form says
    so -- planned without a GPU: host pointers only decide the alignment"""
    torch = pytest.importorskip("torch")
    R, C = 300, 520
    names = {T.forward: "attn_f32_fwd", T.backwardQuery: "attn_f32_dq", T.backwardKeyValue: "attn_f32_dkv"}
    lds = {(T.forward, 128): 65536, (T.backwardQuery, 128): 98304, (T.backwardKeyValue, 128): 99840,
           (T.forward, 64): 32768, (T.backwardQuery, 64): 49152, (T.backwardKeyValue, 64): 50688}

    def buffers(D, ld=None):
        ld = ld or D
        b = {op: torch.zeros((R if op in (Op.Q, Op.O, Op.dO, Op.dQ) else C, ld)) for op in (Op.Q, Op.K, Op.V, Op.O, Op.dO, Op.dQ, Op.dK, Op.dV)}
        b[Op.L], b[Op.D] = torch.zeros(R), torch.zeros(R)
        return b

    for D, block in ((128, 128), (100, 128), (64, 64), (36, 64)):
        d = _desc(dims=(R, C, D))
        for t, name in names.items():
            k = AttentionKernel(d.kernelDescriptor(t))
            assert k.variant == "%s_d%d_w4x32" % (name, block)
            assert k.blockDimensions == (128, 32, block) and k.threadgroupMemoryAllocation == lds[(t, block)]
            b = buffers(D)
            assert k.launchForm(b, row=R, column=C) == k.variant
            assert k.launchForm(b, row=R, column=C, causal=True) == k.variant
            # rows of D + 1 floats are not 16-byte aligned: the general kernel serves the launch; rows of D + 4 are
            form = k.launchForm(buffers(D, D + 1), row=R, column=C, leadingDimensions={op: D + 1 for op in b if op not in (Op.L, Op.D)})
            assert form.startswith("attn_generic_") and "_d%d_" % block in form and "general kernel" in form
            assert k.launchForm(buffers(D, D + 4), row=R, column=C, leadingDimensions={op: D + 4 for op in b if op not in (Op.L, Op.D)}) == k.variant
            # a block mask keeps the general kernel's own sparse code object (named as the sibling)
            mask = torch.full((2, 1), -1, dtype=torch.int32)
            form = k.launchForm(b, row=R, column=C, blockMask=mask, blockMaskWords=1)
            assert "attn_generic_" in form and "block-sparse" in form
    for D in (30, 200):   # D % 4 != 0; the 256 head block: the general kernels
        d = _desc(dims=(R, C, D))
        for t in names:
            k = AttentionKernel(d.kernelDescriptor(t))
            assert k.variant.startswith("attn_generic_") and k.launchForm(buffers(D), row=R, column=C) == k.variant
    k = AttentionKernel(_desc(dims=(R, C, 128), tr=(False, True, False, False)).kernelDescriptor(T.forward))   # a transposed operand
    assert k.variant.startswith("attn_generic_")


def test_low_precision_intermediates_select_the_folded_scale_stream():
    """S and P in FP32 registers (lowPrecisionIntermediates = false): the scale is applied in fp32 per score; with
    lowPrecisionIntermediates the reference itself keeps S / P in 16 bits (+Precisions.swift:149-215) and the kernel may
    pre-multiply Q by the scale in the 16-bit type"""
    exact = AttentionKernel(_low((4096, 4096, 128)).kernelDescriptor(T.forward)).variant
    folded = AttentionKernel(_low((4096, 4096, 128), low_mid=True).kernelDescriptor(T.forward)).variant
    assert "fold" not in exact and folded.endswith("_fold"), (exact, folded)


def test_effective_descriptor_reports_the_register_precisions_really_used():
    """matrix-core kernels round P (and dS) to the inputs' 16-bit type also when the descriptor asks for FP32 intermediates
    (+Precisions.swift:201-205): the effective descriptor says so instead of echoing the request"""
    d = _low((512, 512, 128))
    for t in T:
        kd = d.kernelDescriptor(t)
        assert kd.registerPrecisions[Op.P] == P.FP32
        eff = AttentionKernel(kd).effectiveDescriptor
        assert eff.registerPrecisions[Op.P] == P.BF16 and eff.registerPrecisions[Op.S] == P.FP32, t
        if t != T.forward:
            assert eff.registerPrecisions[Op.dS] == P.BF16
    eff = AttentionKernel(_desc(dims=(512, 512, 128)).kernelDescriptor(T.forward)).effectiveDescriptor   # fp32 kernels: as requested
    assert eff.registerPrecisions[Op.P] == P.FP32


def test_product_library_carries_no_developer_knobs():
    """no environment knob, no superseded kernel, no timing-only ablation in libmfa_hip.so (they live in the -DMFA_DEV_VARIANTS
    build libmfa_hip_dev.so used by tools/ab_*.py)"""
    import os
    import re
    from metal_flash_attention_amd import _abi
    path = os.path.join(os.path.dirname(_abi.__file__), "libmfa_hip.so")
    blob = open(path, "rb").read()
    for needle in (b"WRONG_RESULTS", b"MFA_FWD16_IMPL", b"MFA_DKV16_IMPL", b"MFA_BWD16_DISABLE", b"MFA_GEMM_IMPL", b"PROF_CLOBBERS_O",
                   b"attn_fwd16_v2", b"attn_fwd16_v4"):
        assert needle not in blob, needle
    assert re.search(rb"getenv", blob) is None
