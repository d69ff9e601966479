#!/usr/bin/env python3
"""Ablation builds of libnf_mi355x.so: the named sources recompiled with extra -D flags, every other object reused from
normalizing-flows_amd/lib/obj, linked into normalizing-flows_amd/lib/variants/<name>.so (git-ignored, travels with gpurun).
Select at run time with NF_MI355X_LIB=<path>.

    python tools/build_variant.py gt4 "-DNF_GT_NW=4" glow_conv.hip
"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "normalizing-flows_amd")


def main():
    name, flags, srcs = sys.argv[1], sys.argv[2].split(), sys.argv[3:]
    sys.path.insert(0, ROOT)
    from normflows_amd import _lib
    _lib.build()
    out_dir = os.path.join(PKG, "lib", "variants")
    os.makedirs(out_dir, exist_ok=True)
    objs = []
    for o in sorted(glob.glob(os.path.join(PKG, "lib", "obj", "*.o"))):
        base = os.path.basename(o)[:-2] + ".hip"
        if base in srcs:
            vo = os.path.join(out_dir, "%s_%s.o" % (name, base[:-4]))
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed"]
                                  + flags + ["-c", os.path.join(PKG, "csrc", base), "-o", vo])
            objs.append(vo)
        else:
            objs.append(o)
    so = os.path.join(out_dir, name + ".so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs)
    print(so)


if __name__ == "__main__":
    main()
