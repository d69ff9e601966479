#!/usr/bin/env python3
"""Register / scratch / LDS use of every gfx950 kernel in the built objects (normalizing-flows_amd/lib/obj/*.o), read from
the AMDGPU metadata notes of the embedded code objects (llvm-objcopy --dump-section .hip_fatbin -> clang-offload-bundler
--unbundle -> llvm-readelf --notes).  Used by tests/test_host.py to assert that the fused kernels have no spills.

    python tools/kernel_resources.py [object-or-source ...]      # a .hip source is compiled device-only first
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
          "group_segment_fixed_size", "max_flat_workgroup_size")


def demangle(names):
    try:
        out = subprocess.run([os.path.join(LLVM, "llvm-cxxfilt")] + names, capture_output=True, text=True, check=True).stdout
        return out.strip().split("\n")
    except Exception:
        return names


def code_object(path, tmp):
    """Path of the gfx950 code object inside `path` (a host object / shared object with a .hip_fatbin section, or already a
    device ELF)."""
    fat = os.path.join(tmp, os.path.basename(path) + ".fat")
    r = subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, path, os.devnull],
                       capture_output=True, text=True)
    if r.returncode != 0 or not os.path.exists(fat):
        return path
    co = os.path.join(tmp, os.path.basename(path) + ".co")
    subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    "--input=" + fat, "--output=" + co, "--unbundle"], check=True, capture_output=True)
    return co


def compile_device_only(src, tmp, extra=()):
    out = os.path.join(tmp, os.path.basename(src) + ".o")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed", "-c", src,
                    "-o", out] + list(extra), check=True)
    return out


def resources(path, extra=()):
    """{demangled kernel name: {field: int}} for every kernel of `path`."""
    with tempfile.TemporaryDirectory() as tmp:
        if path.endswith(".hip"):
            path = compile_device_only(path, tmp, extra)
        co = code_object(path, tmp)
        txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
    res = {}
    for blk in re.split(r"\n\s*- \.agpr_count:", "\n" + txt):
        m = re.search(r"\.name:\s+(\S+)", blk)
        if not m or ".vgpr_count" not in blk:
            continue
        d = {}
        blk = ".agpr_count:" + blk
        for f in FIELDS:
            mm = re.search(r"\.%s:\s+(\d+)" % f, blk)
            if mm:
                d[f] = int(mm.group(1))
        res[m.group(1)] = d
    names = list(res)
    return dict(zip(demangle(names), (res[n] for n in names))) if names else {}


def main():
    paths = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "normalizing-flows_amd", "lib", "obj", "*.o")))
    print("%-92s %5s %5s %6s %6s %8s %7s" % ("kernel", "vgpr", "agpr", "vspill", "sspill", "scratchB", "ldsB"))
    for p in paths:
        for k, d in sorted(resources(p).items()):
            print("%-92s %5d %5d %6d %6d %8d %7d" % (k[:92], d.get("vgpr_count", -1), d.get("agpr_count", -1),
                                                     d.get("vgpr_spill_count", -1), d.get("sgpr_spill_count", -1),
                                                     d.get("private_segment_fixed_size", -1), d.get("group_segment_fixed_size", -1)))


if __name__ == "__main__":
    main()
