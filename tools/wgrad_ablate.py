#!/usr/bin/env python3
"""Within-process A/B of wgrad.hip build variants (-D flags).  Usage: python tools/wgrad_ablate.py "" "-DNF_WG_ABL_NOLOAD" ...
Env: NF_M, NF_N, NF_B.  (timing only; ablations are NOT correct)."""
import ctypes as C
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import normflows_amd as nfa  # noqa: E402
from normflows_amd._lib import i32, i64, ptr  # noqa: E402

CSRC = os.path.join(ROOT, "normalizing-flows_amd", "csrc")
OUT = os.path.join(ROOT, "gpurun_out", "ablate")
os.makedirs(OUT, exist_ok=True)


def build(flags, idx):
    so = os.path.join(OUT, "wg%d.so" % idx)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           "-o", so, os.path.join(CSRC, "wgrad.hip")] + flags.split())
    lib = C.CDLL(so)
    lib.nf_linear_wgrad_scratch_floats.restype = C.c_int64
    return lib


def main():
    variants = sys.argv[1:] or [""]
    libs = [build(f, i) for i, f in enumerate(variants)]
    dev = torch.device("cuda:0")
    B, M, N = int(os.environ.get("NF_B", 65536)), int(os.environ.get("NF_M", 736)), int(os.environ.get("NF_N", 128))
    dy = torch.randn(B, M, device=dev)
    x = torch.randn(B, N, device=dev)
    dW = torch.empty(M, N, device=dev)
    db = torch.empty(M, device=dev)
    n = max(lib.nf_linear_wgrad_scratch_floats(i64(B), i32(M), i32(N)) for lib in libs)
    scratch = torch.empty(n, device=dev)
    st = nfa._lib.stream()

    def launch(lib):
        rc = lib.nf_linear_wgrad(ptr(dy), ptr(x), ptr(dW), ptr(db), ptr(scratch), i64(B), i32(M), i32(N), i32(0), st)
        assert rc == 0, rc

    for lib in libs:
        for _ in range(3):
            launch(lib)
    torch.cuda.synchronize()
    res = [[] for _ in libs]
    for _ in range(8):
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
        print("%-40s median %.1f us  min %.1f us" % (f or "(baseline)", r[len(r) // 2], r[0]))


if __name__ == "__main__":
    main()
