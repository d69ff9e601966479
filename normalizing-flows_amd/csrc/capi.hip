// capi.hip -- library identification and error strings of libnf_mi355x.so.
#include "common.hpp"

extern "C" const char *nf_version(void) { return "nf_mi355x 0.1.0 (gfx950)"; }

extern "C" int nf_max_bins(void) { return NF_MAX_BINS; }

extern "C" const char *nf_strerror(int code) {
    switch (code) {
        case NF_OK: return "ok";
        case NF_EIO: return "HIP launch failed (hipGetLastError != hipSuccess)";
        case NF_EFAULT: return "null pointer for a required buffer";
        case NF_EINVAL: return "invalid argument";
        case NF_ERANGE: return "argument outside the range compiled into the kernels";
        case NF_ENOTSUP: return "shape / dtype not supported by this build";
        default: return "unknown error code";
    }
}
