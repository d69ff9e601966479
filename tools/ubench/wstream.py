#!/usr/bin/env python3
"""Runs tools/ubench/wstream.hip: can 4 waves per CU (one per SIMD) stream the 0.69 MB/layer weight stream straight from
L2 into VGPRs (8 loads in flight per wave, 8 MFMAs per 1 KB load) at the MFMA-bound rate?"""
import ctypes as C, os, subprocess, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "..", "..", "gpurun_out", "wstream.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", so,
                       os.path.join(here, "wstream.hip")])
lib = C.CDLL(so)
layers, nloads = 32, 672                       # 672 KB of A operands per layer and wave
w = torch.randn(layers * nloads * 256, device="cuda") * 1e-3
out = torch.empty(256 * 2 * 512, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
def t(mode, grid, block):
    for _ in range(2):
        lib.run(mode, C.c_void_p(w.data_ptr()), C.c_void_p(out.data_ptr()), nloads, layers, grid, block, st)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); lib.run(mode, C.c_void_p(w.data_ptr()), C.c_void_p(out.data_ptr()), nloads, layers, grid, block, st); e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / layers
for grid, block, label in ((256, 256, "256 WG x 4 waves (1 wave/SIMD)"), (512, 256, "512 WG x 4 waves (2 waves/SIMD)")):
    a, b = t(0, grid, block), t(1, grid, block)
    print("%-34s MFMA only %.1f us/layer   L2->VGPR stream + MFMA %.1f us/layer  (+%.1f%%)" % (label, a, b, 100 * (b / a - 1)))
