"""Per-step weight packing of a whole model in two launches (training).

A training step re-packs every benchmark-shaped coupling layer's weights (nf_rqs_fused_pack_all) and re-assembles every
LULinearPermute's factors (nf_lu_factors) once: 2 x 32 launches of 5-12 us in front of kernels that fill the chip.  run_chain
(core.py) calls begin() before it walks the layers: the eligible layers of the chain are packed by ONE launch per kind
(nf_rqs_fused_pack_all_multi / nf_lu_factors_multi, blockIdx.y = layer) from a cached device table of pointers, and each layer
is handed a token; a layer whose token is the current one skips its own pack launch.  The token dies with the call (end()), so a
layer used outside run_chain, or after an exception, packs itself as before.
"""
import weakref

import torch

from . import config as _config
from . import ops

_current = None
_tables = {}


def current():
    return _current


def _table(key_name, rows, device):
    """Device tensor of the rows' pointers, rebuilt only when a pointer changed (parameters are updated in place)."""
    flat = tuple(p for r in rows for p in r)
    hit = _tables.get(key_name)
    if hit is None or hit[0] != flat or hit[1].device != device:
        hit = _tables[key_name] = (flat, torch.tensor(flat, dtype=torch.int64, device=device))
    return hit[1]


_plans = {}


def _nsf_row(c, z):
    net, u = c.transform_net, c.unconditional_transform
    _, wpad, _, wfull_t = c._train_buffers(z)
    lin = [l for blk in net.blocks for l in blk.linear_layers]
    return [c._train_blob_for(z), net.initial_layer.weight, net.initial_layer.bias, net.final_layer.weight, net.final_layer.bias,
            u.unnormalized_widths, u.unnormalized_heights, u.unnormalized_derivatives, wfull_t, wpad, c.identity_features] \
        + [l.weight for l in lin] + [l.bias for l in lin]


def _lu_row(f, z):
    lin = f.linear
    return [f.permutation._permutation, lin.lower_entries, lin.upper_entries, lin.unconstrained_upper_diag,
            f._factors_buffer(z.device)]


def _sentinels(kind, layer):
    """Two of the layer's parameters, read from the module: the cached rows are valid while these are the cached objects and
    still require gradients (.to() / load_state_dict keep the Parameter objects; only re-assigned attributes replace them)."""
    if kind == "nsf":
        return layer.transform_net.initial_layer.weight, layer.transform_net.final_layer.bias
    return layer.linear.lower_entries, layer.linear.unconstrained_upper_diag


def _plan(flows, z):
    """Which layers of `flows` are packed together and the tensors of their table rows: decided once per (model, batch shape,
    training flags, configuration) -- reading ~1300 module attributes every step costs the host several ms.  Per step only the
    sentinels are re-read and the cached tensors asked for their (possibly new) data pointers."""
    from .flows.mixing import LULinearPermute
    from .flows.neural_spline import CoupledRationalQuadraticSpline
    groups = {}
    for f in flows:
        if isinstance(f, CoupledRationalQuadraticSpline):
            c = f.prqct
            if c._train_full_ok(z, None, False) and all(p.requires_grad for p in c.parameters()):
                key = ("nsf", len(c.transform_net.blocks), float(c.tail_bound), c.min_bin_width, c.min_bin_height, c.min_derivative)
                groups.setdefault(key, []).append((c, _nsf_row(c, z), _sentinels("nsf", c)))
        elif isinstance(f, LULinearPermute) and f._train_factors_ok(z) and all(p.requires_grad for p in f.parameters()):
            groups.setdefault(("lu", f.linear.features, float(f.linear.eps)), []).append((f, _lu_row(f, z), _sentinels("lu", f)))
    return list(groups.items())


def _plan_valid(plan):
    for key, entries in plan:
        for layer, _, sen in entries:
            a, b = _sentinels(key[0], layer)
            if a is not sen[0] or b is not sen[1] or not a.requires_grad:
                return False
    return True


def begin(flows, z, inverse):
    """Pack the eligible layers of `flows` for a differentiable density pass over z; returns the token (None: nothing done)."""
    global _current
    if not (_config.train_prepack and inverse and torch.is_grad_enabled() and torch.is_tensor(z) and z.is_cuda and z.dim() == 2
            and z.dtype == torch.float32 and z.shape[0] >= 1024 and _current is None):
        return None
    # the plan depends on what decides the layers' training path: shapes, module flags, configuration
    sig = (tuple(z.shape), z.device, _config.train_full, len(flows), tuple(f.training for f in flows))
    hit = _plans.get(id(flows))
    if hit is None or hit[2]() is not flows or hit[0] != sig or not _plan_valid(hit[1]):
        if len(_plans) > 16:        # models come and go (tests, sweeps): do not keep their layers alive through old plans
            _plans.clear()
            _tables.clear()
        hit = _plans[id(flows)] = (sig, _plan(flows, z), weakref.ref(flows))
    plan = hit[1]
    if sum(len(es) for _, es in plan) < 2:
        return None
    token = object()
    for key, entries in plan:
        rows = [[t.data_ptr() for t in row] for _, row, _ in entries]
        table = _table((key, id(flows)), rows, z.device)
        if key[0] == "nsf":
            ops.rqs_fused_pack_all_multi(table, len(entries), key[1], tail_bound=key[2], min_bin_width=key[3], min_bin_height=key[4],
                                         min_derivative=key[5])
        else:
            ops.lu_factors_multi(table, len(entries), key[2], key[1])
        for layer, _, _ in entries:
            layer.__dict__["_prepacked"] = token
    _current = token
    return token


def end(token):
    global _current
    if token is not None and _current is token:
        _current = None


def take(layer):
    """True once per prepack: the layer's blob / factors were written by the current call's multi-launch."""
    tok = layer.__dict__.get("_prepacked")
    if tok is not None and tok is _current:
        layer.__dict__["_prepacked"] = None
        return True
    return False
