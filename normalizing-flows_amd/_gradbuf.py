"""Where a parameter's gradient is WRITTEN by the backward kernels.

The one-call layer backward (ops.coupling_train_bwd / nf_coupling_train_bwd) takes a destination address per gradient.  By
default that is a fresh tensor per parameter; dp.FlatParameters registers, for each of its parameters, a view of ONE flat gradient
buffer here, so the kernels write the whole model's gradient contiguously: the optimizer steps one flat tensor in one launch
(torch.optim.Adam(fused=True) on 608 parameter tensors: 17 launches, 0.72 ms per step of the benchmark model; on one flat
tensor: one launch, ~0.05 ms) and data-parallel all-reduces run on slices of that buffer in place (no torch.cat, no copy back).

Keyed by id() with a weak reference to the parameter (tensors cannot be dictionary keys by value); an entry dies with its
parameter or when its owner calls release().
"""
import weakref

import torch

_targets = {}      # id(param) -> (weakref to param, flat view)


def register(param, view):
    pid = id(param)

    def _gone(_, pid=pid):
        _targets.pop(pid, None)
    _targets[pid] = (weakref.ref(param, _gone), view)


def release(param):
    _targets.pop(id(param), None)


def target(param):
    """The registered destination view of `param` (the registered object itself), or None."""
    ent = _targets.get(id(param))
    if ent is None or ent[0]() is not param:
        return None
    return ent[1]


def out(param):
    """A tensor the backward kernels write `param`'s gradient into: a FRESH view of the registered destination (autograd's
    AccumulateGrad adopts a returned gradient without copying only when nobody else holds that tensor object), or a new tensor."""
    v = target(param)
    if v is not None and v.dtype == param.dtype and v.device == param.device:
        return v.view(v.shape)
    return torch.empty_like(param, memory_format=torch.contiguous_format)
