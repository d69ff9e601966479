#!/usr/bin/env python3
"""Within-process A/B of rqs_bwd.hip build variants (-D flags) on the benchmark layer shape (timing only)."""
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
    so = os.path.join(OUT, "bwd%d.so" % idx)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-pass-failed",
                           "-o", so, os.path.join(CSRC, "rqs_bwd.hip")] + flags.split())
    return C.CDLL(so)


def main():
    variants = sys.argv[1:] or [""]
    libs = [build(f, i) for i, f in enumerate(variants)]
    dev = torch.device("cuda:0")
    B, D, K = 65536, 64, 8
    nT = nI = 32
    M = 3 * K - 1
    x = torch.randn(B, D, device=dev)
    gy = torch.randn(B, D, device=dev)
    gld = torch.randn(B, device=dev)
    cond = torch.randn(B, nT * M, device=dev)
    uw, uh, ud = torch.randn(nI, K, device=dev), torch.randn(nI, K, device=dev), torch.randn(nI, K - 1, device=dev)
    ii = torch.arange(0, D, 2, device=dev)
    ti = torch.arange(1, D, 2, device=dev)
    gx = torch.zeros_like(x)
    gcond = torch.empty_like(cond)
    guw, guh, gud = torch.zeros_like(uw), torch.zeros_like(uh), torch.zeros_like(ud)
    st = nfa._lib.stream()

    def launch(lib):
        rc = lib.nf_rqs_coupling_bwd(ptr(x), ptr(gy), ptr(gld), ptr(cond), ptr(uw), ptr(uh), ptr(ud), ptr(ii), i32(nI), ptr(ti),
                                     i32(nT), i64(B), i32(D), i32(K), i32(1), f64(3.0), f64(1e-3), f64(1e-3), f64(1e-3),
                                     f64(128 ** 0.5), i32(0), ptr(gx), ptr(gcond), ptr(guw), ptr(guh), ptr(gud), i32(0), st)
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
            for _ in range(4):
                launch(lib)
            e.record()
            torch.cuda.synchronize()
            res[i].append(s.elapsed_time(e) / 4 * 1e3)
    for f, r in zip(variants, res):
        r = sorted(r)
        print("%-40s median %.1f us  min %.1f us" % (f or "(baseline)", r[len(r) // 2], r[0]))


if __name__ == "__main__":
    main()
