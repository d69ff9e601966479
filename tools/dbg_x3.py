import sys, os, subprocess, ctypes, numpy as np, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import normflows_amd as nfa
from normflows_amd import _lib
flags = sys.argv[1:]
if flags:
    so = os.path.join(ROOT, "gpurun_out", "dbg_x3.so")
    srcs = [os.path.join(_lib.CSRC, f) for f in os.listdir(_lib.CSRC) if f.endswith(".hip")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", so] + srcs + flags)
    _lib._lib = ctypes.CDLL(so)
    _lib._lib.nf_version.restype = ctypes.c_char_p; _lib._lib.nf_strerror.restype = ctypes.c_char_p
    _lib._lib.nf_rqs_fused_pack_size.restype = ctypes.c_int64
torch.manual_seed(17)
dev = "cuda:0"
layer = nfa.flows.CoupledRationalQuadraticSpline(64, 2, 128, num_bins=8, init_identity=False).to(dev)
x = torch.randn(33, 64, device=dev)
with torch.no_grad():
    z32, ld32 = layer.inverse(x)
    nfa.config.set_fused_gemm("bf16x3")
    z3, ld3 = layer.inverse(x)
d = (z3 - z32).abs()
print(flags, "max |dz|", float(d.max()), "ld diff", float((ld3 - ld32).abs().max()))
