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
