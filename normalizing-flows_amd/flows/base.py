"""Operator API of the path: `Flow.forward(z) -> (z', log_det)`, `Flow.inverse(z) -> (z', log_det)`.

Mirrors normflows/flows/base.py:5-82 (Flow, Reverse, Composite, zero_log_det_like_z) so that reference
containers (nf.NormalizingFlow, nf.MultiscaleFlow) can drive these layers unchanged.

Extension used by our own containers (core.py): every layer implements
    _run(z, inverse, ld, acc) -> z'
which enqueues the layer's kernel(s) and folds the log-det into the caller's (B,) accumulator `ld`
(`acc` = +1 for `log_q += log_det`, -1 for `log_q -= log_det`) inside the kernel, instead of returning a
fresh tensor that the container then adds with another launch.
"""
import torch
from torch import nn

from .. import _prof




class Flow(nn.Module):
    """Generic flow layer (flows/base.py:5-24)."""

    def __init__(self):
        super().__init__()

    def forward(self, z):
        raise NotImplementedError("Forward pass has not been implemented.")

    def inverse(self, z):
        raise NotImplementedError("This flow has no algebraic inverse.")

    # -- accumulate protocol ------------------------------------------------------------------------
    def _run(self, z, inverse, ld, acc, **kw):
        """Default: call the public API and fold the result in with one extra elementwise launch."""
        z, log_det = (self.inverse(z, **kw) if inverse else self.forward(z, **kw))
        if acc > 0:
            ld += log_det
        else:
            ld -= log_det
        return z


def run_flow(flow, z, inverse, ld, acc, **kw):
    """Drive any flow (ours or a reference/duck-typed one) through the accumulate protocol."""
    if _prof.enabled:     # roctx range per layer type (NF_ROCTX=1)
        with _prof.range_("%s.%s" % (type(flow).__name__, "inverse" if inverse else "forward")):
            return _run_flow(flow, z, inverse, ld, acc, **kw)
    return _run_flow(flow, z, inverse, ld, acc, **kw)


def _run_flow(flow, z, inverse, ld, acc, **kw):
    if hasattr(flow, "_run"):
        return flow._run(z, inverse, ld, acc, **kw)
    z, log_det = (flow.inverse(z, **kw) if inverse else flow(z, **kw))
    if acc > 0:
        ld += log_det
    else:
        ld -= log_det
    return z


class Reverse(Flow):
    """Swaps forward and inverse of a flow (flows/base.py:27-45)."""

    def __init__(self, flow):
        super().__init__()
        self.flow = flow

    def forward(self, z):
        return self.flow.inverse(z)

    def inverse(self, z):
        return self.flow.forward(z)

    def _run(self, z, inverse, ld, acc, **kw):
        return run_flow(self.flow, z, not inverse, ld, acc, **kw)


class Composite(Flow):
    """Chain of flows applied in order (flows/base.py:48-78)."""

    def __init__(self, flows):
        super().__init__()
        self._flows = nn.ModuleList(flows)

    def _cascade(self, z, inverse):
        ld = torch.zeros(z.shape[0], dtype=z.dtype, device=z.device)
        return self._run(z, inverse, ld, +1), ld

    def forward(self, inputs):
        return self._cascade(inputs, False)

    def inverse(self, inputs):
        return self._cascade(inputs, True)

    def _run(self, z, inverse, ld, acc, **kw):
        seq = reversed(self._flows) if inverse else self._flows
        for f in seq:
            z = run_flow(f, z, inverse, ld, acc)
        return z


def zero_log_det_like_z(z):
    return torch.zeros(z.shape[0], dtype=z.dtype, device=z.device)


def new_ld(z):
    return torch.empty(z.shape[0], dtype=z.dtype, device=z.device)
