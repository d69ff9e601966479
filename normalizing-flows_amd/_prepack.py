"""Per-step weight packing of a whole model in two launches (training).

A training step re-packs every benchmark-shaped coupling layer's weights (nf_rqs_fused_pack_all) and re-assembles every
LULinearPermute's factors (nf_lu_factors) once: 2 x 32 launches of 5-12 us in front of kernels that fill the chip.  run_chain
(core.py) calls begin() before it walks the layers: the eligible layers of the chain are packed by ONE launch per kind
(nf_rqs_fused_pack_all_multi / nf_lu_factors_multi, blockIdx.y = layer) from a cached device table of pointers, and each layer
is handed a token; a layer whose token is the current one skips its own pack launch.  The token dies with the call (end()), so a
layer used outside run_chain, or after an exception, packs itself as before.
"""
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


def begin(flows, z, inverse):
    """Pack the eligible layers of `flows` for a differentiable density pass over z; returns the token (None: nothing done)."""
    global _current
    if not (_config.train_prepack and inverse and torch.is_grad_enabled() and torch.is_tensor(z) and z.is_cuda and z.dim() == 2
            and z.dtype == torch.float32 and z.shape[0] >= 1024 and _current is None):
        return None
    from .flows.mixing import LULinearPermute
    from .flows.neural_spline import CoupledRationalQuadraticSpline
    nsf, lus = {}, []
    for f in flows:
        if isinstance(f, CoupledRationalQuadraticSpline):
            c = f.prqct
            if c._train_full_ok(z, None, False) and any(p.requires_grad for p in c.parameters()):
                nsf.setdefault(len(c.transform_net.blocks), []).append(c)
        elif isinstance(f, LULinearPermute) and f._train_factors_ok(z) and any(p.requires_grad for p in f.parameters()):
            lus.append(f)
    if sum(len(v) for v in nsf.values()) + len(lus) < 2:
        return None
    token = object()
    dev = z.device
    for nb, layers in nsf.items():
        rows, first = [], layers[0]
        for c in layers:
            net, u = c.transform_net, c.unconditional_transform
            _, wpad, _, wfull_t = c._train_buffers(z)
            blob = c._train_blob_for(z)
            lin = [l for b in net.blocks for l in b.linear_layers]
            rows.append([blob.data_ptr(), net.initial_layer.weight.data_ptr(), net.initial_layer.bias.data_ptr(),
                         net.final_layer.weight.data_ptr(), net.final_layer.bias.data_ptr(), u.unnormalized_widths.data_ptr(),
                         u.unnormalized_heights.data_ptr(), u.unnormalized_derivatives.data_ptr(), wfull_t.data_ptr(),
                         wpad.data_ptr(), c.identity_features.data_ptr()] + [l.weight.data_ptr() for l in lin]
                        + [l.bias.data_ptr() for l in lin])
        same = all((float(c.tail_bound), c.min_bin_width, c.min_bin_height, c.min_derivative)
                   == (float(first.tail_bound), first.min_bin_width, first.min_bin_height, first.min_derivative) for c in layers)
        if not same:
            continue
        ops.rqs_fused_pack_all_multi(_table(("nsf", nb, id(flows)), rows, dev), len(layers), nb, tail_bound=float(first.tail_bound),
                                     min_bin_width=first.min_bin_width, min_bin_height=first.min_bin_height,
                                     min_derivative=first.min_derivative)
        for c in layers:
            c.__dict__["_prepacked"] = token
    by_shape = {}
    for f in lus:
        by_shape.setdefault((f.linear.features, float(f.linear.eps)), []).append(f)
    for (D, eps), layers in by_shape.items():
        rows = []
        for f in layers:
            lin = f.linear
            rows.append([f.permutation._permutation.data_ptr(), lin.lower_entries.data_ptr(), lin.upper_entries.data_ptr(),
                         lin.unconstrained_upper_diag.data_ptr(), f._factors_buffer(dev).data_ptr()])
        ops.lu_factors_multi(_table(("lu", D, eps, id(flows)), rows, dev), len(layers), eps, D)
        for f in layers:
            f.__dict__["_prepacked"] = token
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
