import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def golden_state(g, prefix="sd__"):
    """State dict stored by make_golden.sd(): 'sd__a__b' -> 'a.b'."""
    return {k[len(prefix):].replace("__", "."): v for k, v in g.items() if k.startswith(prefix)}


def assert_close(a, b, rtol, atol, what=""):
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    both_nan = np.isnan(a) & np.isnan(b)
    same_inf = np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b))
    ok = both_nan | same_inf | (np.abs(a - b) <= atol + rtol * np.abs(b))
    if not ok.all():
        idx = np.argwhere(~ok)[0]
        raise AssertionError("%s: %d/%d mismatches, first at %s: got %r want %r (max abs err %.3e)" % (
            what, (~ok).sum(), ok.size, tuple(idx), a[tuple(idx)], b[tuple(idx)],
            np.nanmax(np.where(np.isfinite(a - b), np.abs(a - b), 0))))


@pytest.fixture(scope="session")
def oracle():
    import nf_oracle
    nf_oracle.build()
    return nf_oracle


TOL = {np.dtype("float32"): dict(rtol=2e-5, atol=2e-5), np.dtype("float64"): dict(rtol=1e-10, atol=1e-10)}


def ld_tol(dtype, root_finding=False):
    """Tolerance of a PER-LAYER log|det| against the reference's output in the same precision (round 6, VERDICT r05: was 1e-3).
    Closed-form direction (the spline itself, the density direction of a coupling layer): 1e-4 relative + absolute, the north-star
    bar.  Root-finding direction (the spline's inverse, a coupling layer's sampling direction): the quadratic's discriminant cancels
    in float32 and the REFERENCE's own float32 evaluation sits 9.8e-5 (absolute) away from its float64 evaluation on these fixtures
    (measured here by running the reference on spline_K10_f32 in both precisions: DESIGN.md section 5) -- two float32 orderings cannot agree better than that: 5e-4 absolute there."""
    t = TOL[np.dtype(dtype)]
    if root_finding:
        return dict(rtol=t["rtol"] * 5, atol=t["atol"] * 25)
    return dict(rtol=t["rtol"] * 5, atol=t["atol"] * 5)
