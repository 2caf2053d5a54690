"""What hipcc made of the C++ around the hand-placed statements (attn_fwd16_p4 / p5, attn_dq16_p4, attn_dkv16_p4), read back
from the gfx950 code objects of the last build.  The statements own all but ~28 registers per lane, so any per-lane value
hipcc keeps live across them goes to scratch; in the causal forward kernels that once turned every prologue LDS-DMA piece
into `scratch_load; s_waitcnt vmcnt(0); buffer_load ... lds` -- one piece in flight at a time
(profiles/r02_fwd16_causal_pass_loop_scratch.txt).  No GPU needed: llvm-objdump on csrc/build/*.o."""
import os
import re
import subprocess
import tempfile

import pytest

CSRC = os.path.join(os.path.dirname(__file__), "..", "metal_flash_attention_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
OBJECTS = ["attn_fwd16_p4", "attn_fwd16_p4p", "attn_fwd16_p6", "attn_fwd16_p5", "attn_dq16_p4", "attn_dkv16_p4", "attn_dq16_p5", "attn_dkv16_p5", "attn_f32",
           "attn_fwd16_p4_tr", "attn_fwd16_p5_tr", "attn_bwd16_p4_tr"]
# scratch instructions hipcc may leave AROUND a hand-placed statement, per translation unit (the worst kernel of the unit today;
# pinned so that they cannot grow unnoticed -- the statements themselves never touch scratch)
SCRATCH_BUDGET = {"attn_fwd16_p4": 40, "attn_fwd16_p4p": 0, "attn_fwd16_p6": 0, "attn_fwd16_p5": 40, "attn_dq16_p4": 40, "attn_dkv16_p4": 40, "attn_dq16_p5": 0,
                  "attn_dkv16_p5": 8, "attn_f32": 8, "attn_fwd16_p4_tr": 60, "attn_fwd16_p5_tr": 100, "attn_bwd16_p4_tr": 540}   # (attn_dq16_p4_tr gathers Q^T / dO^T / O^T in its C++ prologue: 526 today)
# (kernels of the unit that spill at all, most spilled vector registers in one kernel): the state of the round-5 build
SPILL_BUDGET = {"attn_fwd16_p4p": (0, 0), "attn_fwd16_p6": (0, 0), "attn_dq16_p5": (0, 0), "attn_dkv16_p5": (28, 2), "attn_dq16_p4": (18, 2), "attn_dkv16_p4": (21, 6),
                "attn_f32": (1, 2), "attn_fwd16_p4": (4, 15), "attn_fwd16_p5": (12, 2), "attn_fwd16_p4_tr": (24, 19),
                "attn_fwd16_p5_tr": (56, 34), "attn_bwd16_p4_tr": (24, 73),
                # round 6, the 384 head blocks (compiler-scheduled; the 320 ones and attn_dkv16_wide do not spill)
                "attn_fwd16_wide": (4, 31), "attn_bwd16_wide": (6, 16)}


def _kernels(obj):
    path = os.path.join(CSRC, "build", obj + ".o")
    tools = [os.path.join(LLVM, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump")]
    if not os.path.exists(path) or not all(os.path.exists(t) for t in tools):
        pytest.skip("needs the object files of the last build and the ROCm llvm tools")
    with tempfile.TemporaryDirectory() as tmp:
        fat, dev = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "dev.o")
        subprocess.check_call([tools[0], "-O", "binary", "--only-section=.hip_fatbin", path, fat])
        subprocess.check_call([tools[1], "--unbundle", "--type=o", "--input=" + fat,
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + dev])
        text = subprocess.check_output([tools[2], "-d", "--no-show-raw-insn", dev], text=True)
    kernels, name = {}, None
    for line in text.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(_ZN3mfa[^>]+)>:", line)
        if m:
            name = m.group(1)
            kernels[name] = []
        elif name and line.startswith("\t"):
            kernels[name].append(line.split("//")[0].strip())
    return kernels


@pytest.mark.parametrize("obj", OBJECTS)
def test_lds_dma_pieces_are_issued_back_to_back(obj):
    kernels = _kernels(obj)
    assert kernels
    for name, ins in kernels.items():
        for i, t in enumerate(ins):
            if t.startswith("buffer_load") and t.endswith("lds"):
                window = ins[max(0, i - 4):i]
                # (a wait FOLLOWED BY THE TILE'S BARRIER is the seam of the hand-placed streams, whose first LDS-DMA piece leads the
                # phase behind it by design -- round 6: the branch-free loop has no block-switch test between the two any more)
                for k, w in enumerate(window):
                    if w.startswith("s_waitcnt vmcnt(0)"):
                        assert any(x.startswith("s_barrier") for x in window[k + 1:]), (name, i, window)
                assert not any(w.startswith("scratch_load") for w in window), (name, i, window)


@pytest.mark.parametrize("obj", OBJECTS)
def test_scratch_stays_out_of_the_way(obj):
    """a handful of spills around the statement are fine (values the epilogue needs); dozens mean loop invariants again"""
    for name, ins in _kernels(obj).items():
        n = sum(1 for t in ins if t.startswith("scratch_"))
        assert n <= SCRATCH_BUDGET[obj], (name, n)


# prologue loads of the hand-placed units: a global load whose s_waitcnt vmcnt(0) comes before the next load is a round trip to
# memory of its own.  Round 5 found the backward prologues written one (row block, k-step) at a time around asm statements --
# hipcc keeps loads on their side of a volatile statement or a branch -- 133 of attn_dq16_p5's 166 loads were serial
# (profiles/r05_prologue_loads.txt); the budgets are today's numbers (single loads of L / per-batch lengths and the last of a batch)
SERIAL_LOAD_BUDGET = {"attn_dq16_p5": 8, "attn_dkv16_p5": 3, "attn_dq16_p4": 8, "attn_dkv16_p4": 4, "attn_fwd16_p5": 6}


@pytest.mark.parametrize("obj", sorted(SERIAL_LOAD_BUDGET))
def test_prologue_loads_are_in_flight_together(obj):
    def is_load(t):
        return (t.startswith("buffer_load") or t.startswith("global_load")) and not t.endswith("lds")
    for name, ins in _kernels(obj).items():
        if "combine" in name:
            continue
        serial = 0
        for i, t in enumerate(ins):
            if not is_load(t):
                continue
            for w in ins[i + 1:i + 40]:
                if is_load(w):
                    break
                if w.startswith("s_waitcnt") and "vmcnt(0)" in w:
                    serial += 1
                    break
        assert serial <= SERIAL_LOAD_BUDGET[obj], (name, serial)


# translation units whose kernels are allowed to CALL a device function hipcc did not inline (by-reference captures in scratch): none.
# (Round 3 found the transposed 8 x 32 code objects at D > 128 calling the tile-load lambda of attn_fwd16_v3.h -- ~0.1 PFLOP/s,
# profiles/r03_dev_transposed_streams.txt; MFA_V3_INLINE_LOADS forces it inline in every build since round 4.)
KNOWN_CALLERS = set()


def _audit(build):
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
    import audit_code_objects
    path = os.path.join(CSRC, build)
    if not os.path.isdir(path) or not os.path.exists(os.path.join(LLVM, "llvm-readelf")):
        pytest.skip("needs the object files of the last build and the ROCm llvm tools")
    report = audit_code_objects.audit(path)
    if not report:
        pytest.skip("no object files in " + build)
    return report


def test_no_kernel_calls_a_function():
    """Every lambda and helper of a kernel is inlined: a device function that is not a kernel means s_swappc_b64 in a loop and
    its captures in scratch memory.  tools/audit_code_objects.py reads the symbol tables (FUNC symbols without a kernel descriptor)."""
    report = _audit("build")
    callers = {tu for tu, r in report.items() if r["functions"]}
    assert callers == (KNOWN_CALLERS & set(report)), {tu: report[tu]["functions"] for tu in callers ^ (KNOWN_CALLERS & set(report))}
    for tu in callers:   # only the one lambda: void() of attn_fwd16_v3
        assert all(f.startswith("_ZZN3mfa13attn_fwd16_v3I") and f.endswith("ENKUlvE_clEv") for f in report[tu]["functions"]), report[tu]["functions"]


@pytest.mark.parametrize("build", ["build", "build_dev"])
def test_transposed_code_objects_have_the_tile_loads_inline(build):
    """MFA_V3_INLINE_LOADS: no translation unit calls a function, and the transposed code objects at 160 / 192 have no stack"""
    report = _audit(build)
    assert not {tu for tu, r in report.items() if r["functions"]}
    for tu in ("attn_fwd16_v3_tr_d160", "attn_fwd16_v3_tr_d192"):
        if tu in report:
            assert not report[tu]["stack"], report[tu]["stack"]


def test_spills_of_the_hand_placed_units_stay_within_their_budgets():
    """llvm-readelf notes of the product build: `.vgpr_spill_count` per kernel.  The hand-placed statements own ~480 of the 512
    registers, so hipcc's share around them spills a few values in some wrappers (masked / transposed variants); the budgets are
    today's numbers -- a change that makes a wrapper spill more fails here instead of showing up as per-block cost"""
    report = _audit("build")
    for tu, (kernels, most) in SPILL_BUDGET.items():
        if tu not in report:
            continue
        spills = report[tu]["spills"]
        assert len(spills) <= kernels, (tu, len(spills), kernels)
        assert max([n for _, n in spills] or [0]) <= most, (tu, sorted(spills, key=lambda x: -x[1])[:3])
