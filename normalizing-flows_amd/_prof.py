"""roctx ranges around the layers of the hot path (profiling hooks).

Off by default.  NF_ROCTX=1 (or `normflows_amd.config.set_roctx(True)`) wraps every layer call issued by the containers
(flows/base.run_flow, the fused NSF chains, the Glow level chains) in a roctxRangePush / roctxRangePop pair named after the
layer type, so that `rocprofv3 --marker-trace --kernel-trace` groups kernels by layer.  libroctx64 is loaded lazily with
ctypes; when it is missing the ranges are silently no-ops (profiling aid, never part of the product path's results)."""
import contextlib
import ctypes
import os

enabled = os.environ.get("NF_ROCTX") == "1"
_lib = None
_tried = False


def _load():
    global _lib, _tried
    if not _tried:
        _tried = True
        for name in ("libroctx64.so", "/opt/rocm/lib/libroctx64.so", "librocprofiler-sdk-roctx.so",
                     "/opt/rocm/lib/librocprofiler-sdk-roctx.so"):
            try:
                _lib = ctypes.CDLL(name)
                _lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                break
            except (OSError, AttributeError):
                _lib = None
    return _lib


def push(name):
    lib = _load()
    if lib is not None:
        lib.roctxRangePushA(name.encode())


def pop():
    lib = _load()
    if lib is not None:
        lib.roctxRangePop()


@contextlib.contextmanager
def range_(name):
    if not enabled:
        yield
        return
    push(name)
    try:
        yield
    finally:
        pop()
