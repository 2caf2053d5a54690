"""The drop-in boundary exercised from plain C (no Python, no torch in the callee): compile
tests/c_abi/c_abi_check.c against include/mfa.h + libmfa_hip.so (+ the oracle as checker) with plain gcc and run it."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_abi", "c_abi_check.c")
EXE = os.path.join(ROOT, "tests", "c_abi", "c_abi_check.out")


def _build():
    import __graft_entry__ as entry
    entry.build()
    pkg = os.path.join(ROOT, "metal_flash_attention_amd")
    cmd = ["gcc", "-std=c11", "-O2", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", SRC, "-o", EXE,
           "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
           "-L", pkg, "-lmfa_hip", "-L", os.path.join(ROOT, "oracle"), "-loracle_network",
           "-L", "/opt/rocm/lib", "-lamdhip64", "-lm",
           f"-Wl,-rpath,{pkg}", f"-Wl,-rpath,{os.path.join(ROOT, 'oracle')}", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)


CPP_SRC = os.path.join(ROOT, "tests", "c_abi", "cpp_mirror_check.cpp")
CPP_EXE = os.path.join(ROOT, "tests", "c_abi", "cpp_mirror_check.out")


def _build_cpp():
    import __graft_entry__ as entry
    entry.build()
    pkg = os.path.join(ROOT, "metal_flash_attention_amd")
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", CPP_SRC, "-o", CPP_EXE,
           "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
           "-L", pkg, "-lmfa_hip", "-L", os.path.join(ROOT, "oracle"), "-loracle_network", "-loracle_gemm",
           "-L", "/opt/rocm/lib", "-lamdhip64", "-lm",
           f"-Wl,-rpath,{pkg}", f"-Wl,-rpath,{os.path.join(ROOT, 'oracle')}", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)


def test_cpp_mirror_compiles_against_the_headers():
    """include/mfa.hpp (the C++ mirror of the Swift types) is valid C++17 and links."""
    _build_cpp()
    assert os.path.exists(CPP_EXE)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(93, 130, 32), (192, 192, 80)])
def test_cpp_mirror_matches_oracle_on_gpu(shape):
    if not os.path.exists(CPP_EXE):
        _build_cpp()
    out = subprocess.run([CPP_EXE] + [str(x) for x in shape], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "C++ MIRROR OK" in out.stdout


def test_c_program_compiles_and_links_against_the_abi():
    """CPU: the header is valid C11 and the library satisfies every reference the program makes."""
    _build()
    assert os.path.exists(EXE)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(200, 333, 80), (64, 64, 64), (10, 10, 3)])
def test_c_program_matches_oracle_on_gpu(shape):
    if not os.path.exists(EXE):
        _build()
    out = subprocess.run([EXE] + [str(x) for x in shape], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "C ABI OK" in out.stdout
