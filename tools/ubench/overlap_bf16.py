#!/usr/bin/env python3
import ctypes as C, os, subprocess, torch
here = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(here, "..", "..", "gpurun_out", "overlap_bf16.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", so,
                       os.path.join(here, "overlap_bf16.hip")])
lib = C.CDLL(so)
out = torch.empty(256 * 8 * 512, device="cuda")
iters = 2000
def t(mode, nv, bpc):
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(2):
        lib.run(mode, nv, C.c_void_p(out.data_ptr()), iters, 256 * bpc, 256, st)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); lib.run(mode, nv, C.c_void_p(out.data_ptr()), iters, 256 * bpc, 256, st); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3
for bpc, label in ((1, "1 wave/SIMD"), (2, "2 waves/SIMD")):
    print("==", label, "(slot = 2 bf16 MFMAs 32x32x16 on independent accumulators)")
    for mode, nv, name in ((0, 0, "MFMA only"), (1, 2, "VALU only 4 fma"), (1, 4, "VALU only 8 fma"), (1, 8, "VALU only 16 fma"),
                           (2, 2, "MFMA + 4 fma"), (2, 4, "MFMA + 8 fma"), (2, 8, "MFMA + 16 fma")):
        us = t(mode, nv, bpc)
        print("  %-22s %8.1f us -> %.1f ns per slot" % (name, us, us * 1e3 / (iters * 16)))
