"""normflows_amd -- MI355X (gfx950) native implementation of normflows' coupling-layer hot path.

Same operator API as VincentStimper/normalizing-flows 1.7.3 (`Flow.forward / inverse -> (z, log_det)`,
state_dict-compatible layer classes, NormalizingFlow / MultiscaleFlow containers); the arithmetic of every layer
is a hand-written HIP kernel behind the C ABI of include/nf_mi355x.h (libnf_mi355x.so).  There is no CPU or
eager fallback: layers raise if the library is missing or a tensor is not on a HIP device.

The on-disk package directory is `normalizing-flows_amd/`; import it as `normflows_amd` (see normflows_amd.py
at the repository root).
"""
from . import _lib, config, ops, nets, flows, distributions, transforms, utils, dp
from .core import NormalizingFlow, ConditionalNormalizingFlow, MultiscaleFlow, invalidate_caches
from .distributions import DiagGaussian, ConditionalDiagGaussian, ClassCondDiagGaussian, GlowBase

__version__ = "0.1.0"

from . import _refpaths as _refpaths   # noqa: E402
import sys as _sys                      # noqa: E402
_refpaths.install(_sys.modules[__name__])


def native_library_path():
    return _lib.LIBPATH


def native_version():
    return _lib.lib().nf_version().decode()
