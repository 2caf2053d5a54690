#!/usr/bin/env python3
"""Static audit of the gfx950 code objects in the product library's object files (metal_flash_attention_amd/csrc/build/*.o):
per translation unit the number of kernels, DEVICE FUNCTIONS THAT ARE NOT KERNELS (a lambda or helper hipcc did not inline: the
kernel calls it with s_swappc_b64 and its by-reference captures live in scratch memory), kernels with a stack
(.private_segment_fixed_size > 0) and kernels with spilled vector registers.  No GPU needed; tests/test_build_artifacts.py runs
`audit()` and pins the list of known offenders.  Usage: python tools/audit_code_objects.py [build-dir]"""
import glob
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def code_object(obj, tmp):
    """the gfx950 code object embedded in a hipcc object file (section .hip_fatbin), or None"""
    fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, "x.co")
    for f in (fat, co):
        if os.path.exists(f):
            os.remove(f)
    subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True, capture_output=True)
    if not os.path.exists(fat) or os.path.getsize(fat) == 0:
        return None
    r = subprocess.run([LLVM + "/clang-offload-bundler", "--type=o", "--targets=" + TARGET, "--input=" + fat, "--output=" + co,
                        "--unbundle"], capture_output=True)
    return co if r.returncode == 0 and os.path.exists(co) and os.path.getsize(co) else None


def audit(build_dir):
    """{translation unit: dict(kernels, functions=[non-kernel device functions], stack=[kernels with a stack], spills=[...])}"""
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for obj in sorted(glob.glob(os.path.join(build_dir, "*.o"))):
            co = code_object(obj, tmp)
            if co is None:
                continue
            syms = subprocess.run([LLVM + "/llvm-readelf", "-s", "-W", co], check=True, capture_output=True, text=True).stdout
            funcs, descriptors = set(), set()
            for line in syms.splitlines():
                f = line.split()
                if len(f) >= 8 and f[3] == "FUNC" and f[6] != "UND":
                    funcs.add(f[7])
                elif len(f) >= 8 and f[3] == "OBJECT" and f[7].endswith(".kd"):
                    descriptors.add(f[7][:-3])
            notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
            stack, spills, name = [], [], None
            # (the metadata lists a kernel's keys alphabetically: .name comes before .private_segment_fixed_size and .vgpr_spill_count)
            for line in notes.splitlines():
                m = re.match(r"\s*-?\s*\.name:\s+(\S+)", line)
                if m:
                    name = m.group(1)
                m = re.match(r"\s*\.private_segment_fixed_size:\s+(\d+)", line)
                if m and int(m.group(1)) > 0:
                    stack.append((name, int(m.group(1))))
                m = re.match(r"\s*\.vgpr_spill_count:\s+(\d+)", line)
                if m and int(m.group(1)) > 0:
                    spills.append((name, int(m.group(1))))
            out[os.path.basename(obj)[:-2]] = dict(kernels=len(descriptors), functions=sorted(funcs - descriptors), stack=stack, spills=spills)
    return out


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    build = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "metal_flash_attention_amd", "csrc", "build")
    for tu, r in audit(build).items():
        print("%-26s kernels %3d   un-inlined device functions %2d   kernels with a stack %2d   with spilled VGPRs %2d"
              % (tu, r["kernels"], len(r["functions"]), len(r["stack"]), len(r["spills"])))
        for f in r["functions"]:
            print("      function", f[:150])
