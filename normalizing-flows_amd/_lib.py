"""ctypes binding of libnf_mi355x.so (the C ABI declared in include/nf_mi355x.h).

The library is built in-tree by `build()` (hipcc --offload-arch=gfx950) and loaded from
normalizing-flows_amd/lib/.  There is NO fallback: if the shared object is missing or a tensor does not
live on a HIP device, the calling layer raises.  torch is used here only to obtain device pointers and the
current HIP stream.
"""
import ctypes as C
import glob
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIBDIR = os.path.join(_HERE, "lib")
LIBPATH = os.path.join(LIBDIR, "libnf_mi355x.so")
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")

NF_F32, NF_F64 = 0, 1
LD_WRITE, LD_ADD, LD_SUB = 0, 1, -1
TAILS = {None: 0, "linear": 1, "circular": 2}
SCALE = {"exp": 0, "sigmoid": 1, "sigmoid_inv": 2, None: 3}
RQS_DENSITY, RQS_SAMPLE_IDENTITY, RQS_SAMPLE_TRANSFORM = 0, 1, 2

_lib = None


class NativeLibraryError(RuntimeError):
    pass


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _hip_includes(src):
    """Other .hip sources a .hip source #includes (rqs_fused_nw4.hip is a second build of rqs_fused.hip)."""
    import re
    return [os.path.join(CSRC, m) for m in re.findall(r'^#include "([^"]+\.hip)"', open(src).read(), flags=re.M)]


def build(force=False, verbose=False):
    """Compile every HIP source into lib/libnf_mi355x.so for gfx950 (cross-compiles without a GPU)."""
    srcs = sources()
    deps = srcs + glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(INCLUDE, "*.h"))
    if not force and os.path.exists(LIBPATH) and all(os.path.getmtime(d) <= os.path.getmtime(LIBPATH) for d in deps):
        return LIBPATH
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if not force and os.path.exists(o) and all(
                os.path.getmtime(d) <= os.path.getmtime(o)
                for d in [s] + _hip_includes(s) + glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(INCLUDE, "*.h"))):
            continue
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed"] + os.environ.get("NF_HIPCC_FLAGS", "").split() + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise NativeLibraryError("hipcc failed: %s\n%s" % (" ".join(cmd), out.decode(errors="replace")))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIBPATH] + objs
    subprocess.check_call(cmd)
    return LIBPATH


SAFE_WAITS_LIB = os.path.join(LIBDIR, "variants", "safe_waits.so")


def build_safe_waits(force=False):
    """The differential build of the counted-wait lint (tests/test_gpu_hygiene.py): every source that uses NF_WAIT_VMCNT recompiled
    with -DNF_SAFE_WAITS (each hand-counted `s_waitcnt vmcnt(N)` becomes a full drain), the other objects reused; selected at run
    time with NF_MI355X_LIB.  Built by __graft_entry__.build() so that it travels to the GPU box with the tree."""
    build()
    srcs = [s for s in sources() if "NF_WAIT_VMCNT" in open(s).read() or any("NF_WAIT_VMCNT" in open(i).read() for i in _hip_includes(s))]
    deps = srcs + glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(INCLUDE, "*.h")) + [LIBPATH]
    if not force and os.path.exists(SAFE_WAITS_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(SAFE_WAITS_LIB) for d in deps):
        return SAFE_WAITS_LIB
    vdir = os.path.dirname(SAFE_WAITS_LIB)
    os.makedirs(vdir, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    procs, objs = [], []
    for s in sources():
        base = os.path.basename(s)[:-4]
        if s in srcs:
            o = os.path.join(vdir, "safe_waits_%s.o" % base)
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed", "-DNF_SAFE_WAITS", "-c", s, "-o", o]
            procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        else:
            o = os.path.join(LIBDIR, "obj", base + ".o")
        objs.append(o)
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise NativeLibraryError("hipcc failed: %s\n%s" % (" ".join(cmd), out.decode(errors="replace")))
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SAFE_WAITS_LIB] + objs)
    return SAFE_WAITS_LIB


def exported_symbols_declared():
    """Names of every function declared in include/nf_mi355x.h (used by the symbol-export test)."""
    import re
    txt = open(os.path.join(INCLUDE, "nf_mi355x.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(nf_[a-z0-9_]+)\s*\(", txt)))


def int64_functions_declared():
    """Names of the functions include/nf_mi355x.h declares as returning int64_t (scratch / pack sizes): ctypes assumes a 32-bit int
    unless told otherwise, and a size above 2^31 elements then arrives truncated.  (Round 6, last session: nf_maf_solve_t_scratch_floats
    had its return type set only on the ablation-build path -- the density-direction backward of config 5's layer failed above
    ~720 000 rows per call with a scratch buffer a fraction of the size the kernel addressed.)"""
    import re
    txt = open(os.path.join(INCLUDE, "nf_mi355x.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"^\s*int64_t\s+(nf_[a-z0-9_]+)\s*\(", txt, flags=re.M)))


def _declare_return_types(handle):
    handle.nf_version.restype = C.c_char_p
    handle.nf_strerror.restype = C.c_char_p
    for fn in int64_functions_declared():
        if hasattr(handle, fn):
            getattr(handle, fn).restype = C.c_int64


def lib():
    global _lib
    if _lib is None:
        override = os.environ.get("NF_MI355X_LIB")   # ablation builds (tools/*_ablate.py): another build of the SAME sources
        if override:
            _lib = C.CDLL(override)
            _declare_return_types(_lib)
            return _lib
        if not os.path.exists(LIBPATH):
            raise NativeLibraryError(
                "libnf_mi355x.so is missing (%s). Build it with `python -c 'import __graft_entry__ as g; g.build()'`."
                " There is no CPU or eager fallback." % LIBPATH)
        _lib = C.CDLL(LIBPATH)
        _declare_return_types(_lib)       # (every int64_t function of the header, not a hand-kept list)
    return _lib


def check(rc, what):
    if rc == 0:
        return
    msg = lib().nf_strerror(rc).decode()
    if rc == -22:
        raise ValueError("%s: %s" % (what, msg))
    if rc in (-95, -34):
        raise NotImplementedError("%s: %s" % (what, msg))
    raise RuntimeError("%s: %s (code %d)" % (what, msg, rc))


def dtype_code(t):
    if t.dtype == torch.float32:
        return NF_F32
    if t.dtype == torch.float64:
        return NF_F64
    raise TypeError("nf_mi355x kernels support float32/float64 tensors, got %s" % t.dtype)


def require_device(*tensors):
    cur = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "normflows_amd layers run only on an MI355X (HIP) device; got a %s tensor. There is no CPU path."
                % t.device)
        if cur is None:
            cur = torch.cuda.current_device()
        if t.device.index != cur:   # kernels are enqueued on the CURRENT device's stream (one process per GPU)
            raise RuntimeError("tensor on %s but the current device is cuda:%d: call torch.cuda.set_device first"
                               % (t.device, cur))


def ptr(t):
    if t is None:
        return C.c_void_p(0)
    if not t.is_contiguous():
        raise ValueError("tensor handed to the C ABI must be contiguous")
    return C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


i64 = C.c_int64
i32 = C.c_int
f64 = C.c_double
