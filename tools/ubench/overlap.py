#!/usr/bin/env python3
"""Runs tools/ubench/overlap.hip on the GPU box: cycles per MFMA with and without VALU fillers, 1 and 2 waves/SIMD."""
import ctypes as C, os, subprocess, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "..", "..", "gpurun_out", "overlap.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", so,
                       os.path.join(here, "overlap.hip")])
lib = C.CDLL(so)
out = torch.empty(256 * 8 * 512, device="cuda")
iters = 2000
def t(mode, nv, blocks_per_cu, block):
    grid = 256 * blocks_per_cu
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(2):
        lib.run(mode, nv, C.c_void_p(out.data_ptr()), iters, grid, block, st)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); lib.run(mode, nv, C.c_void_p(out.data_ptr()), iters, grid, block, st); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3  # us
for block, bpc, label in ((256, 1, "1 wave/SIMD"), (256, 2, "2 waves/SIMD")):
    print("==", label)
    for mode, nv, name in ((0, 0, "MFMA only"), (1, 2, "VALU only 4 fma/slot"), (1, 4, "VALU only 8 fma/slot"), (1, 8, "VALU only 16 fma/slot"),
                           (2, 2, "MFMA + 4 fma"), (2, 4, "MFMA + 8 fma"), (2, 8, "MFMA + 16 fma"),
                           (3, 1, "TRANS only 2 exp/slot"), (3, 2, "TRANS only 4 exp/slot"), (4, 1, "MFMA + 2 exp"), (4, 2, "MFMA + 4 exp")):
        us = t(mode, nv, bpc, block)
        slots = iters * 16
        print("  %-26s %8.1f us  -> %.1f ns per slot (per wave)" % (name, us, us * 1e3 / slots))
