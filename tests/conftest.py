import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built_library():
    """Make sure libmfa_hip.so and the oracle exist (cross-compiles without a GPU)."""
    import __graft_entry__ as entry

    entry.build()
    return True


# ---- which code objects do the GPU tests launch?  (-m gpu sessions on a GPU box only) -----------------------------------------
# OPT-IN (MFA_VARIANT_COVERAGE=1; the default session leaves AttentionKernel untouched -- no extra prepare_launch per dispatch in
# the timing-sensitive tests): every AttentionKernel.dispatch / .time of the session is recorded as variant name -> the test ids
# that launched it, written to gpurun_out/variant_coverage.json at session end; a copy of a full `-m gpu` run is committed as
# tests/golden/variant_coverage.json and tests/test_variant_coverage.py (CPU) holds the library's variant names against it.
_COVERAGE = {"variants": {}, "forms": {}}
_CURRENT = {"id": None}


def _gpu_session():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:  # noqa: BLE001
        return False


@pytest.fixture(autouse=True)
def _current_test_id(request):
    _CURRENT["id"] = request.node.nodeid
    yield
    _CURRENT["id"] = None


@pytest.fixture(autouse=True, scope="session")
def _record_launched_variants():
    if os.environ.get("MFA_VARIANT_COVERAGE", "0") != "1" or not _gpu_session():
        yield
        return
    import json
    from metal_flash_attention_amd import AttentionKernel
    real = {name: getattr(AttentionKernel, name) for name in ("dispatch", "time")}

    def wrap(name):
        def method(self, buffers, *args, **kw):
            tests = _COVERAGE["variants"].setdefault(self.variant, [])
            tid = _CURRENT["id"] or "?"
            if tid not in tests and len(tests) < 4:
                tests.append(tid)
            try:   # the launch form names the kernel family that really runs (persistent form, split pieces, re-layout, fallback)
                form_kw = {k: v for k, v in kw.items() if k not in ("stream", "warmup", "iterations")}
                import re
                for form in set(re.findall(r"attn_[A-Za-z0-9_]+", self.launchForm(buffers, **form_kw))):   # (incl. a named sibling)
                    _COVERAGE["forms"][form] = _COVERAGE["forms"].get(form, 0) + 1
            except Exception as exc:  # noqa: BLE001 -- a launch that is about to fail validation: the real call below reports it
                _COVERAGE.setdefault("form_errors", []).append("%s: %s" % (tid, exc))
            return real[name](self, buffers, *args, **kw)
        return method

    for name in real:
        setattr(AttentionKernel, name, wrap(name))
    yield
    for name, fn in real.items():
        setattr(AttentionKernel, name, fn)
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    from metal_flash_attention_amd import _abi
    worker = os.environ.get("PYTEST_XDIST_WORKER")   # pytest -n N: every worker records its own part, the controller merges them below
    with open(os.path.join(out, "variant_coverage.%s.json" % worker if worker else "variant_coverage.json"), "w") as f:
        json.dump({"library": os.path.basename(_abi.library_path()), "variants": dict(sorted(_COVERAGE["variants"].items())),
                   "forms": dict(sorted(_COVERAGE["forms"].items())), "form_errors": _COVERAGE.get("form_errors", [])[:50]}, f, indent=1)


def pytest_sessionfinish(session, exitstatus):
    """pytest -n N: merge the workers' parts of the variant coverage map (the controller runs this after the last worker has finished)"""
    if os.environ.get("PYTEST_XDIST_WORKER"):
        return
    import glob
    import json
    parts = sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "variant_coverage.gw*.json")))
    if not parts:
        return
    merged = {"library": None, "variants": {}, "forms": {}}
    for part in parts:
        data = json.load(open(part))
        merged["library"] = data["library"]
        for name, tests in data["variants"].items():
            have = merged["variants"].setdefault(name, [])
            have.extend(t for t in tests if t not in have and len(have) < 4)
        for name, count in data["forms"].items():
            merged["forms"][name] = merged["forms"].get(name, 0) + count
        os.remove(part)
    merged["variants"] = dict(sorted(merged["variants"].items()))
    merged["forms"] = dict(sorted(merged["forms"].items()))
    with open(os.path.join(ROOT, "gpurun_out", "variant_coverage.json"), "w") as f:
        json.dump(merged, f, indent=1)
