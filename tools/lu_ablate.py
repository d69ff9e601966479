#!/usr/bin/env python3
"""Within-process A/B of lu_linear.hip build variants (-D flags), launches interleaved round-robin.
Usage: python tools/lu_ablate.py "" "-DNF_LU_ABL_NOCOMPUTE" ...   (timing only; ablations are NOT correct)."""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import normflows_amd as nfa  # noqa: E402
from normflows_amd._lib import f64, i32, i64, ptr  # noqa: E402

CSRC = os.path.join(ROOT, "normalizing-flows_amd", "csrc")
OUT = os.path.join(ROOT, "gpurun_out", "ablate")
os.makedirs(OUT, exist_ok=True)


def build(flags, idx):
    so = os.path.join(OUT, "lu%d.so" % idx)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           "-I" + os.path.join(ROOT, "include"), "-o", so, os.path.join(CSRC, "lu_linear.hip")] + flags.split())
    return C.CDLL(so)


def main():
    variants = sys.argv[1:] or [""]
    libs = [build(f, i) for i, f in enumerate(variants)]
    dev = torch.device("cuda:0")
    B, D = int(os.environ.get("NF_B", 65536)), int(os.environ.get("NF_D", 64))
    direction = int(os.environ.get("NF_DIR", "0"))
    x = torch.randn(B, D, device=dev)
    y = torch.empty_like(x)
    ld = torch.zeros(B, device=dev)
    perm = torch.randperm(D, device=dev)
    lo = torch.randn(D * (D - 1) // 2, device=dev) * 0.1
    up = torch.randn(D * (D - 1) // 2, device=dev) * 0.1
    ud = torch.randn(D, device=dev)
    bias = torch.randn(D, device=dev)
    st = nfa._lib.stream()

    def launch(lib):
        rc = lib.nf_lu_linear_permute(ptr(x), ptr(y), ptr(ld), ptr(perm), ptr(lo), ptr(up), ptr(ud), ptr(bias), i64(B), i32(D),
                                      f64(1e-3), i32(direction), i32(1), i32(0), st)
        assert rc == 0, rc

    for lib in libs:
        for _ in range(3):
            launch(lib)
    torch.cuda.synchronize()
    res = [[] for _ in libs]
    for _ in range(10):
        for i, lib in enumerate(libs):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(8):
                launch(lib)
            e.record()
            torch.cuda.synchronize()
            res[i].append(s.elapsed_time(e) / 8 * 1e3)
    for f, r in zip(variants, res):
        r = sorted(r)
        print("%-50s median %.1f us  min %.1f us" % (f or "(baseline)", r[len(r) // 2], r[0]))


if __name__ == "__main__":
    main()
