"""Per-step weight packing of a whole model in two launches (training).

A training step re-packs every benchmark-shaped coupling layer's weights (nf_rqs_fused_pack_all) and re-assembles every
LULinearPermute's factors (nf_lu_factors) once: 2 x 32 launches of 5-12 us in front of kernels that fill the chip.  run_chain
(core.py) calls begin() before it walks the layers: the eligible layers of the chain are packed by ONE launch per kind
(nf_rqs_fused_pack_all_multi / nf_lu_factors_multi, blockIdx.y = layer) from a cached device table of pointers, and each layer
is handed a token; a layer whose token is the current one skips its own pack launch.  The token dies with the call (end()), so a
layer used outside run_chain, or after an exception, packs itself as before.

State: the plan of a model (which layers, which tensors, the device pointer tables) lives in a WeakKeyDictionary keyed by the
model's flow list object itself -- nothing is keyed by id(), nothing keeps a dead model's tensors alive; the token of the pass
in flight is thread-local.  A plan is trusted for a step only if EVERY tensor of every row is still the object registered in
its owning module (one dict lookup per entry, ~0.1 ms for the 32-pair model; re-walking the modules costs 3-8 ms) and the
layers' path switches are unchanged: re-assigning any parameter / buffer, .requires_grad_(False), use_fused* or a config
switch rebuilds it.
"""
import threading
import weakref

import torch

from . import config as _config
from . import ops

_tls = threading.local()
_plans = weakref.WeakKeyDictionary()      # flow list (nn.ModuleList) -> _Plan


def current():
    return getattr(_tls, "token", None)


class _Plan:
    __slots__ = ("sig", "groups", "tables")

    def __init__(self, sig, groups):
        self.sig, self.groups, self.tables = sig, groups, {}


def _owned(module, name):
    """(tensor, owner dict, key): the registered parameter / buffer `name` of `module` and where to look it up again."""
    d = module._parameters if name in module._parameters else module._buffers
    return d[name], d, name


def _nsf_row(c, z):
    """Table row of one coupling layer (order = nf_rqs_fused_pack_all_multi's, include/nf_mi355x.h) + its ownership records."""
    net, u = c.transform_net, c.unconditional_transform
    _, wpad, _, wfull_t = c._train_buffers(z)
    lin = [l for blk in net.blocks for l in blk.linear_layers]
    own = [_owned(net.initial_layer, "weight"), _owned(net.initial_layer, "bias"), _owned(net.final_layer, "weight"),
           _owned(net.final_layer, "bias"), _owned(u, "unnormalized_widths"), _owned(u, "unnormalized_heights"),
           _owned(u, "unnormalized_derivatives"), _owned(c, "identity_features")] \
        + [_owned(l, "weight") for l in lin] + [_owned(l, "bias") for l in lin]
    t = [o[0] for o in own]
    row = [c._train_blob_for(z)] + t[:7] + [wfull_t, wpad, t[7]] + t[8:]
    # layer-owned images (blob, wfull_t, wpad) live in the layer's __dict__: checked by identity as well
    held = [(row[0], c.__dict__, "_train_blob"), (c.__dict__["_train_wbufs"], c.__dict__, "_train_wbufs")]
    return row, own + held


def _lu_row(f, z):
    lin = f.linear
    own = [_owned(f.permutation, "_permutation"), _owned(lin, "lower_entries"), _owned(lin, "upper_entries"),
           _owned(lin, "unconstrained_upper_diag")]
    fbuf = f._factors_buffer(z.device)
    return [o[0] for o in own] + [fbuf], own + [(fbuf, f.__dict__, "_lu_fbuf")]


def _pair_row(c, f, z):
    """Table row of nf_lu_pack_train_multi for the pair (coupling c, LULinearPermute f): the LU's tensors, the coupling's training
    blob and the LU's (64, 64) composed-matrix buffer (round 6)."""
    lin = f.linear
    own = [_owned(f.permutation, "_permutation"), _owned(lin, "lower_entries"), _owned(lin, "upper_entries"),
           _owned(lin, "unconstrained_upper_diag"), _owned(lin, "bias")]
    blob, wd = c._train_blob_for(z), f._wd_buffer(z.device)
    return [o[0] for o in own] + [blob, wd], own + [(blob, c.__dict__, "_train_blob"), (wd, f.__dict__, "_lu_wd")]


def _switches(kind, layer):
    if kind == "nsf":
        return (layer.use_fused, layer.use_fused_train, layer.training)
    return (layer.training,)


def _trainable(flows):
    """Cheap eligibility fingerprint: per layer, whether its FIRST parameter requires a gradient (one attribute read per layer)."""
    out = []
    for f in flows:
        p = next(f.parameters(), None)
        out.append(None if p is None else p.requires_grad)
    return tuple(out)


def _plan(flows, z, sig):
    """Which layers of `flows` are packed together and the tensors of their table rows: decided once per (model, batch shape,
    training flags, configuration)."""
    from .flows.mixing import LULinearPermute
    from .flows.neural_spline import CoupledRationalQuadraticSpline
    groups, packed = {}, set()
    for f in flows:
        if isinstance(f, CoupledRationalQuadraticSpline):
            c = f.prqct
            if c._train_full_ok(z, None, False) and all(p.requires_grad for p in c.parameters()):
                key = ("nsf", len(c.transform_net.blocks), float(c.tail_bound), c.min_bin_width, c.min_bin_height, c.min_derivative)
                row, own = _nsf_row(c, z)
                groups.setdefault(key, []).append((c, row, own, _switches("nsf", c)))
                packed.add(id(f))
        elif isinstance(f, LULinearPermute) and f._train_factors_ok(z) and all(p.requires_grad for p in f.parameters()):
            row, own = _lu_row(f, z)
            groups.setdefault(("lu", f.linear.features, float(f.linear.eps)), []).append((f, row, own, _switches("lu", f)))
            packed.add(id(f))
    out = list(groups.items())
    # round 6: adjacent [CoupledRQS, LULinearPermute] pairs whose two layers are both packed above run as autograd.PairTrainFn: one
    # more multi-layer launch writes the LU stage of the coupling's blob (AFTER the "nsf" launch: it re-initialises the header)
    if _config.train_pair and _config.train_bwd_onecall and _config.final_bwd_fused and _config.resblock_bwd:
        pairs = {}
        fl = list(flows)
        for a, b in zip(fl[:-1], fl[1:]):
            if (isinstance(a, CoupledRationalQuadraticSpline) and isinstance(b, LULinearPermute) and id(a) in packed and id(b) in packed
                    and b.linear.features == 64 and z.shape[0] % 64 == 0 and 1 <= len(a.prqct.transform_net.blocks) <= 5):
                row, own = _pair_row(a.prqct, b, z)
                key = ("pair", len(a.prqct.transform_net.blocks), float(b.linear.eps))
                pairs.setdefault(key, []).append(((a.prqct, b), row, own, _switches("nsf", a.prqct) + _switches("lu", b)))
        out += list(pairs.items())
    return _Plan(sig, out)


def _plan_valid(plan):
    for key, entries in plan.groups:
        for layer, _, own, sw in entries:
            cur = (_switches("nsf", layer[0]) + _switches("lu", layer[1])) if key[0] == "pair" else _switches(key[0], layer)
            if cur != sw:
                return False
            for t, d, name in own:
                cur = d.get(name)
                if cur is not t or (t.__class__ is torch.nn.Parameter and not t.requires_grad):
                    return False
    return True


def _table(plan, key, rows, device):
    """Device tensor of the rows' pointers, rebuilt only when a pointer changed (parameters are updated in place)."""
    flat = tuple(p for r in rows for p in r)
    hit = plan.tables.get(key)
    if hit is None or hit[0] != flat or hit[1].device != device:
        hit = plan.tables[key] = (flat, torch.tensor(flat, dtype=torch.int64, device=device))
    return hit[1]


def begin(flows, z, inverse):
    """Pack the eligible layers of `flows` for a differentiable density pass over z; returns the token (None: nothing done)."""
    if not (_config.train_prepack and inverse and torch.is_grad_enabled() and torch.is_tensor(z) and z.is_cuda and z.dim() == 2
            and z.dtype == torch.float32 and z.shape[0] >= 1024 and current() is None and isinstance(flows, torch.nn.Module)):
        return None
    # the plan depends on what decides the layers' training path: shapes, configuration, the list itself -- and WHICH layers are
    # trainable right now: a plan built while parameters were frozen (reverse_kld(score_fn=False), a freeze-then-unfreeze
    # fine-tune) holds too few layers and would otherwise stay valid for the same batch shape forever (round-3 ADVICE)
    sig = (tuple(z.shape), z.device, _config.train_full, _config.resblock_bwd, _config.lu_bwd_fused, len(flows), _trainable(flows),
           _config.train_pair, _config.train_bwd_onecall, _config.final_bwd_fused)
    plan = _plans.get(flows)
    if plan is None or plan.sig != sig or not _plan_valid(plan):
        plan = _plan(flows, z, sig)
        if sum(len(es) for _, es in plan.groups) >= 2:
            _plans[flows] = plan            # (a plan with fewer than two eligible layers is not worth a launch and is not kept)
        else:
            _plans.pop(flows, None)
    if sum(len(es) for k_, es in plan.groups if k_[0] != "pair") < 2:
        return None
    token = object()
    for key, entries in plan.groups:
        rows = [[t.data_ptr() for t in row] for _, row, _, _ in entries]
        table = _table(plan, key, rows, z.device)
        if key[0] == "nsf":
            ops.rqs_fused_pack_all_multi(table, len(entries), key[1], tail_bound=key[2], min_bin_width=key[3], min_bin_height=key[4],
                                         min_derivative=key[5])
        elif key[0] == "pair":
            ops.lu_pack_train_multi(table, len(entries), key[1], key[2])
            for (c, f), _, _, _ in entries:
                c.__dict__["_pair_prepacked"] = (token, f)
            continue
        else:
            ops.lu_factors_multi(table, len(entries), key[2], key[1])
        for layer, _, _, _ in entries:
            layer.__dict__["_prepacked"] = token
    _tls.token = token
    return token


def end(token):
    if token is not None and current() is token:
        _tls.token = None


def take_pair(c, f):
    """True once per prepack: coupling c's blob holds LULinearPermute f's stage, both layers' own packs are current, and nobody has
    consumed them yet (autograd.PairTrainFn then runs the pair; both layers' tokens are consumed with it)."""
    ent = c.__dict__.get("_pair_prepacked")
    tok = current()
    if ent is None or tok is None or ent[0] is not tok or ent[1] is not f:
        return False
    if c.__dict__.get("_prepacked") is not tok or f.__dict__.get("_prepacked") is not tok:
        return False
    c.__dict__["_pair_prepacked"] = None
    c.__dict__["_prepacked"] = None
    f.__dict__["_prepacked"] = None
    return True


def take(layer):
    """True once per prepack: the layer's blob / factors were written by the current call's multi-launch."""
    tok = layer.__dict__.get("_prepacked")
    if tok is not None and tok is current():
        layer.__dict__["_prepacked"] = None
        return True
    return False
