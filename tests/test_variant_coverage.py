"""Every code-object variant the product library can select is launched by at least one `-m gpu` test.

The GPU session records variant name -> test ids (tests/conftest.py, written to gpurun_out/variant_coverage.json); a copy of a
full run on an MI355X is committed as tests/golden/variant_coverage.json.  Here, on the CPU: the variant names compiled into
libmfa_hip.so (its string table) are held against that map.  A new variant without a GPU test fails this test until the map
is re-recorded (python -m pytest tests -m gpu on the GPU box, then copy the file) or the name is listed in NOT_LAUNCHED with
the reason."""
import json
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "variant_coverage.json")

# variant names no GPU test launches, with the reason (kept empty unless there is one)
NOT_LAUNCHED = {
}


def library_variant_names():
    from metal_flash_attention_amd import _abi
    out = subprocess.run(["strings", "-n", "12", _abi.library_path()], check=True, capture_output=True, text=True).stdout
    # variant names: kernel family, input type, head-dimension bucket, geometry (launch-form words such as attn_bwd16_p4_tr are not)
    # (the general kernels compose their names at run time: attn_generic_* are not in the string table and are launched by every fp32 test)
    pat = re.compile(r"^attn_[a-z0-9]+_(bf16|f16)(_dObf16)?_d\d+_[a-z0-9_]+$")
    return sorted({s for s in out.splitlines() if pat.match(s)})


def test_every_variant_is_launched_by_a_gpu_test(built_library):
    if not os.path.exists(GOLDEN):
        pytest.skip("no recorded map yet: run the GPU suite and copy gpurun_out/variant_coverage.json to tests/golden/")
    names = library_variant_names()
    assert len(names) > 100, names
    recorded = json.load(open(GOLDEN))
    launched = set(recorded["variants"]) | set(recorded["forms"])   # a kernel's own variant, or the sibling its launch form names
    missing = [n for n in names if n not in launched and n not in NOT_LAUNCHED]
    assert not missing, "variants without a GPU test (%d of %d): %s" % (len(missing), len(names), missing)
    stale = sorted(n for n in NOT_LAUNCHED if n in launched)
    assert not stale, "listed as not launched but recorded: %s" % stale


def test_recorded_map_names_real_tests():
    if not os.path.exists(GOLDEN):
        pytest.skip("no recorded map yet")
    recorded = json.load(open(GOLDEN))
    files = {t.split("::")[0] for tests in recorded["variants"].values() for t in tests if t != "?"}
    for f in files:
        assert os.path.exists(os.path.join(ROOT, f)), f
