"""Neural spline flow coupling layers.

Mirrors normflows/flows/neural_spline/coupling.py:16-362 (Coupling, PiecewiseCoupling,
PiecewiseRationalQuadraticCDF, PiecewiseRationalQuadraticCoupling) and
normflows/flows/neural_spline/wrapper.py:14-85 (CoupledRationalQuadraticSpline): constructor signatures,
buffers `identity_features` / `transform_features`, module tree `prqct.transform_net.*`,
`prqct.unconditional_transform.unnormalized_{widths,heights,derivatives}` (state_dict compatible).

The coupling transform (index split, spline on the transform half with conditioner outputs, batch-shared
spline on the identity half, scatter, per-sample log-det) is one launch of nf_rqs_coupling per direction
phase; the spline arithmetic of normflows/utils/splines.py lives in csrc/common.hpp.
"""
import warnings

import numpy as np
import torch
from .. import _keys
from torch import nn

from .. import _lib as L
from .. import _prepack
from .. import config as _config
from .. import ops
from ..autograd import CouplingDensityFn, CouplingTrainFn, FinalSplineDensityFn, IdentLinearFn, SplineFn, needs_grad
from ..nets import PeriodicFeaturesElementwise, ResidualNet
from ..utils.masks import create_alternating_binary_mask
from .base import Flow

DEFAULT_MIN_BIN_WIDTH = 1e-3
DEFAULT_MIN_BIN_HEIGHT = 1e-3
DEFAULT_MIN_DERIVATIVE = 1e-3


_TAIL_CODE = {"linear": 1, "circular": 2}


def _check_tails(tails, tail_bound):
    if isinstance(tails, (list, tuple)):
        for t in tails:
            if t not in _TAIL_CODE:
                raise RuntimeError("{} tails are not implemented.".format(t))
    elif tails not in (None, "linear", "circular"):
        raise RuntimeError("{} tails are not implemented.".format(tails))


def _tails_kwargs(tails, tail_bound, role, device=None, cache=None):
    """Kernel arguments for scalar or per-feature tails / bounds (utils/splines.py:48-66); role = "t" | "i".
    `cache` (a dict owned by the module) keeps the device copy of the type codes."""
    kw = {}
    if isinstance(tails, (list, tuple)):
        kw["tails"] = "feature"
        key = (role, str(device))
        if cache is not None and key in cache:
            codes = cache[key]
        else:
            codes = torch.tensor([_TAIL_CODE[t] for t in tails], dtype=torch.int32, device=device)
            if cache is not None:
                cache[key] = codes
        kw["tails_" + role] = codes
    else:
        kw["tails"] = tails
    if torch.is_tensor(tail_bound):
        if tails is None:
            raise NotImplementedError("tensor tail_bound without tails")
        kw["bound_" + role] = tail_bound.reshape(-1)
        kw["tail_bound"] = 1.0
    else:
        kw["tail_bound"] = tail_bound
    return kw


FUSED_D, FUSED_H = 64, 128     # the shape of csrc/rqs_fused.hip (narrower layers are zero-padded into it)


class PiecewiseRationalQuadraticCDF(Flow):
    """Batch-shared monotone RQ spline per feature (nsf/coupling.py:170-259).  Inputs (B, *shape)."""

    def __init__(self, shape, num_bins=10, tails=None, tail_bound=1.0, identity_init=True,
                 min_bin_width=DEFAULT_MIN_BIN_WIDTH, min_bin_height=DEFAULT_MIN_BIN_HEIGHT,
                 min_derivative=DEFAULT_MIN_DERIVATIVE):
        super().__init__()
        _check_tails(tails, tail_bound)
        self.min_bin_width = min_bin_width
        self.min_bin_height = min_bin_height
        self.min_derivative = min_derivative
        if torch.is_tensor(tail_bound):
            self.register_buffer("tail_bound", tail_bound)
        else:
            self.tail_bound = tail_bound
        self.tails = tails
        self.num_bins = num_bins
        if self.tails == "linear":
            num_derivatives = num_bins - 1
        elif self.tails == "circular":
            num_derivatives = num_bins
        else:
            num_derivatives = num_bins + 1
        if identity_init:
            self.unnormalized_widths = nn.Parameter(torch.zeros(*shape, num_bins))
            self.unnormalized_heights = nn.Parameter(torch.zeros(*shape, num_bins))
            constant = np.log(np.exp(1 - min_derivative) - 1)
            self.unnormalized_derivatives = nn.Parameter(constant * torch.ones(*shape, num_derivatives))
        else:
            self.unnormalized_widths = nn.Parameter(torch.rand(*shape, num_bins))
            self.unnormalized_heights = nn.Parameter(torch.rand(*shape, num_bins))
            self.unnormalized_derivatives = nn.Parameter(torch.rand(*shape, num_derivatives))

    def _spline(self, inputs, inverse, ld=None, acc=None):
        uw, uh, ud = self.unnormalized_widths, self.unnormalized_heights, self.unnormalized_derivatives
        tails, bound = self.tails, self.tail_bound
        shape = tuple(uw.shape[:-1])
        if tuple(inputs.shape[1:]) != shape:
            raise ValueError("PiecewiseRationalQuadraticCDF: inputs of shape (batch,) + %s expected" % (shape,))
        x2 = inputs
        if len(shape) != 1:   # (B, *shape) inputs (nsf/coupling.py:221-253 expands the parameters over the batch): the
            x2 = inputs.reshape(inputs.shape[0], -1)   # kernel sees prod(shape) independent features
            uw, uh, ud = uw.reshape(-1, uw.shape[-1]), uh.reshape(-1, uh.shape[-1]), ud.reshape(-1, ud.shape[-1])
            outer = int(np.prod(shape[:-1]))
            if isinstance(tails, (list, tuple)):
                tails = list(tails) * outer             # the list runs along the last feature dim (utils/splines.py:51-57)
            if torch.is_tensor(bound):
                bound = torch.broadcast_to(bound, shape).reshape(-1)
        elif torch.is_tensor(bound):
            bound = torch.broadcast_to(bound, shape)
        cache = self.__dict__.setdefault("_tcache", {})
        kw = dict(min_bin_width=self.min_bin_width, min_bin_height=self.min_bin_height,
                  min_derivative=self.min_derivative, **_tails_kwargs(tails, bound, "i", inputs.device, cache))
        if needs_grad(inputs, self):   # standalone use in a trained model: forward kernel + backward kernel
            y, log_det = SplineFn.apply(x2.contiguous(), None, uw, uh, ud, self.num_bins, inverse, kw)
            y = y.reshape(inputs.shape)
            if ld is None:
                return y, log_det
            if acc is None or acc > 0:
                ld += log_det
            else:
                ld -= log_det
            return y, ld
        idx = torch.arange(x2.shape[1], device=inputs.device)
        none = idx[:0]
        y, ld = ops.rqs_coupling(x2, None, uw.detach(), uh.detach(), ud.detach(), idx, none, self.num_bins,
                                 L.RQS_SAMPLE_IDENTITY if inverse else L.RQS_DENSITY, logdet=ld, acc=acc, **kw)
        return y.reshape(inputs.shape), ld

    def forward(self, inputs, context=None):
        return self._spline(inputs, False)

    def inverse(self, inputs, context=None):
        return self._spline(inputs, True)

    def _run(self, z, inverse, ld, acc, **kw):
        y, _ = self._spline(z, inverse, ld=ld, acc=acc)
        return y


class Coupling(Flow):
    """Coupling layer base: 1-D mask -> identity / transform feature index buffers, conditioner network,
    optional unconditional transform of the identity half (nsf/coupling.py:16-140)."""

    def __init__(self, mask, transform_net_create_fn, unconditional_transform=None):
        mask = torch.as_tensor(mask)
        if mask.dim() != 1:
            raise ValueError("Mask must be a 1-dim tensor.")
        if mask.numel() <= 0:
            raise ValueError("Mask can't be empty.")
        super().__init__()
        self.features = len(mask)
        features_vector = torch.arange(self.features)
        self.register_buffer("identity_features", features_vector.masked_select(mask <= 0))
        self.register_buffer("transform_features", features_vector.masked_select(mask > 0))
        assert self.num_identity_features + self.num_transform_features == self.features
        self.transform_net = transform_net_create_fn(
            self.num_identity_features, self.num_transform_features * self._transform_dim_multiplier())
        if unconditional_transform is None:
            self.unconditional_transform = None
        else:
            self.unconditional_transform = unconditional_transform(features=self.num_identity_features)

    @property
    def num_identity_features(self):
        return len(self.identity_features)

    @property
    def num_transform_features(self):
        return len(self.transform_features)

    def _check(self, inputs):
        if inputs.dim() not in [2, 4]:
            raise ValueError("Inputs must be a 2D or a 4D tensor.")
        if inputs.shape[1] != self.features:
            raise ValueError("Expected features = {}, got {}.".format(self.features, inputs.shape[1]))

    def _transform_dim_multiplier(self):
        raise NotImplementedError()


class PiecewiseRationalQuadraticCoupling(Coupling):
    """RQ-spline coupling (nsf/coupling.py:262-362).  forward = density direction of the spline,
    inverse = quadratic-root direction."""

    def __init__(self, mask, transform_net_create_fn, num_bins=10, tails=None, tail_bound=1.0,
                 apply_unconditional_transform=False, img_shape=None, min_bin_width=DEFAULT_MIN_BIN_WIDTH,
                 min_bin_height=DEFAULT_MIN_BIN_HEIGHT, min_derivative=DEFAULT_MIN_DERIVATIVE):
        _check_tails(tails, tail_bound)
        self.num_bins = num_bins
        self.min_bin_width = min_bin_width
        self.min_bin_height = min_bin_height
        self.min_derivative = min_derivative
        # per-feature tails / bounds are split between the two halves (nsf/coupling.py:283-318)
        mask_t = torch.as_tensor(mask)
        features_vector = torch.arange(len(mask_t))
        identity_features = features_vector.masked_select(mask_t <= 0)
        transform_features = features_vector.masked_select(mask_t > 0)
        if isinstance(tails, (list, tuple)):
            self.tails = [tails[i] for i in transform_features]
            tails_ = [tails[i] for i in identity_features]
        else:
            self.tails = tails
            tails_ = tails
        if torch.is_tensor(tail_bound):
            tail_bound_ = tail_bound[identity_features]
        else:
            self.tail_bound = tail_bound
            tail_bound_ = tail_bound
        if apply_unconditional_transform:
            unconditional_transform = lambda features: PiecewiseRationalQuadraticCDF(
                shape=[features] + (img_shape if img_shape else []), num_bins=num_bins, tails=tails_,
                tail_bound=tail_bound_, min_bin_width=min_bin_width, min_bin_height=min_bin_height,
                min_derivative=min_derivative)
        else:
            unconditional_transform = None
        super().__init__(mask, transform_net_create_fn, unconditional_transform=unconditional_transform)
        if torch.is_tensor(tail_bound):
            self.register_buffer("tail_bound", tail_bound[transform_features])
        self._per_feature = isinstance(self.tails, list) or torch.is_tensor(tail_bound)
        self._fused_ok = None      # lazily decided: shape handled by the fused MFMA kernel?
        self._fused_parity = 0
        self._fused_cache = None   # (parameter-version key, packed weight blob)
        self._fused_x3_cache = None  # (same key object, split-bf16 blob)
        self.use_fused = True      # set False to force the unfused (library GEMM + nf_rqs_coupling) path
        self.use_fused_train = True   # training: final Linear + coupling transform as one launch (FinalSplineDensityFn)

    def _transform_dim_multiplier(self):
        if self.tails == "linear":
            return self.num_bins * 3 - 1
        elif self.tails == "circular":
            return self.num_bins * 3
        return self.num_bins * 3 + 1

    def _wh_div(self):
        net = self.transform_net
        if hasattr(net, "hidden_features"):
            return float(np.sqrt(net.hidden_features))
        if hasattr(net, "hidden_channels"):
            return float(np.sqrt(net.hidden_channels))
        warnings.warn("Inputs to the softmax are not scaled down: initialization might be bad.")
        return 1.0

    def _kernel_kwargs(self):
        kw = dict(min_bin_width=self.min_bin_width, min_bin_height=self.min_bin_height,
                  min_derivative=self.min_derivative, wh_div=self._wh_div())
        dev = self.identity_features.device
        cache = self.__dict__.setdefault("_tcache", {})
        kw.update(_tails_kwargs(self.tails, self.tail_bound, "t", dev, cache))
        u = self.unconditional_transform
        if self._per_feature and u is not None:   # the identity half carries its own per-feature description
            ki = _tails_kwargs(u.tails, u.tail_bound, "i", dev, cache)
            for k in ("tails_i", "bound_i"):
                if k in ki:
                    kw[k] = ki[k]
        return kw

    def _uncond(self):
        u = self.unconditional_transform
        if u is None:
            return None, None, None
        return u.unnormalized_widths.detach(), u.unnormalized_heights.detach(), u.unnormalized_derivatives.detach()

    def _conditioner(self, rows, context):
        ident = rows.index_select(1, self.identity_features)
        out = self.transform_net(ident, context)
        return out.contiguous()

    def _density(self, inputs, context=None, ld=None, acc=None):
        """prqct.forward (nsf/coupling.py:71-98): conditioner on the raw identity features."""
        self._check(inputs)
        if inputs.dim() == 4:
            return self._image(inputs, context, False, ld, acc)
        if needs_grad(inputs, context, self):
            return self._autograd(inputs, context, False, ld, acc)
        if self.use_fused and self._fused_eligible(inputs, context):
            return self._fused(inputs, 0, ld, acc)
        wide = self._wide_pack(inputs, context)
        if wide is not None:
            return self._wide(inputs, wide, 0, ld, acc)
        cond = self._conditioner(inputs, context)
        uw, uh, ud = self._uncond()
        return ops.rqs_coupling(inputs, cond, uw, uh, ud, self.identity_features, self.transform_features,
                                self.num_bins, L.RQS_DENSITY, logdet=ld, acc=acc, **self._kernel_kwargs())

    def _sample(self, inputs, context=None, ld=None, acc=None):
        """prqct.inverse (nsf/coupling.py:100-128): CDF^-1 on the identity half first, conditioner on ITS output."""
        self._check(inputs)
        if inputs.dim() == 4:
            return self._image(inputs, context, True, ld, acc)
        if needs_grad(inputs, context, self):
            return self._autograd(inputs, context, True, ld, acc)
        if self.use_fused and self._fused_eligible(inputs, context):
            return self._fused(inputs, 1, ld, acc)
        wide = self._wide_pack(inputs, context)
        if wide is not None:
            return self._wide(inputs, wide, 1, ld, acc)
        uw, uh, ud = self._uncond()
        kw = self._kernel_kwargs()
        y, ld = ops.rqs_coupling(inputs, None, uw, uh, ud, self.identity_features, self.transform_features,
                                 self.num_bins, L.RQS_SAMPLE_IDENTITY, logdet=ld, acc=acc, **kw)
        cond = self._conditioner(y, context)
        acc2 = acc if acc not in (None, L.LD_WRITE) else L.LD_ADD
        return ops.rqs_coupling(inputs, cond, uw, uh, ud, self.identity_features, self.transform_features,
                                self.num_bins, L.RQS_SAMPLE_TRANSFORM, y=y, logdet=ld, acc=acc2, **kw)

    def forward(self, inputs, context=None):
        return self._density(inputs, context)

    def inverse(self, inputs, context=None):
        return self._sample(inputs, context)

    # -- shapes beyond the benchmark kernel's (D <= 128, hidden <= 512): the whole layer as one launch (csrc/nsf_wide.hip) ---
    def _wide_pack(self, inputs, context, lu=None, direction=0):
        """Device copies of flows/nsf_wide_pack.py's streams + the batch-shared spline's knot tables, rebuilt when a parameter
        changes; None when the layer is outside nf_nsf_wide's structure (then: library GEMMs + nf_rqs_coupling).  With `lu` (the
        adjacent LULinearPermute, D <= 128) the pack carries its dense matrix for `direction` and the pair runs as one launch."""
        if not (self.use_fused and _config.nsf_wide and context is None and inputs.dim() == 2 and inputs.dtype == torch.float32
                and inputs.is_cuda):
            return None
        net, u = self.transform_net, self.unconditional_transform
        if u is None or not isinstance(net, ResidualNet):
            return None
        tensors = list(net.parameters()) + [u.unnormalized_widths, u.unnormalized_heights, u.unnormalized_derivatives]
        if lu is not None:
            lin = lu.linear
            tensors = tensors + [lin.lower_entries, lin.upper_entries, lin.unconstrained_upper_diag, lin.bias]
        key = _keys.pkey(tensors) + (str(inputs.device),)
        caches = self.__dict__.setdefault("_wide_cache", {})
        slot = (id(lu), direction) if lu is not None else None
        cache = caches.get(slot)
        if cache is None or cache[0] != key:
            from . import nsf_wide_pack
            lu_np = lad = None
            if lu is not None:
                Wd, Ws, bd, bs, lad = lu._dense_matrices()
                lu_np = (Wd.cpu().numpy(), bd.cpu().numpy()) if direction == 0 else (Ws.cpu().numpy(), bs.cpu().numpy())
            packed = nsf_wide_pack.pack_nsf_wide(self, lu=lu_np, direction=direction)
            if packed is not None:
                blob, table = packed
                tabs = ops.nsf_wide_tables(u.unnormalized_widths.detach(), u.unnormalized_heights.detach(),
                                           u.unnormalized_derivatives.detach(), self.num_bins, self.tail_bound, self.min_bin_width,
                                           self.min_bin_height, self.min_derivative)
                packed = (torch.from_numpy(blob).to(inputs.device), torch.from_numpy(table).to(inputs.device), tabs, int(table[3]), lad)
            cache = caches[slot] = (key, packed)
        return cache[1]

    def _wide(self, inputs, packed, direction, ld, acc):
        blob, table, tabs, hp, lad = packed
        return ops.nsf_wide(inputs, blob, table, tabs, hp, direction, self.tail_bound, self.min_bin_width, self.min_bin_height,
                            self.min_derivative, logdet=ld, acc=acc, lu_logdet=lad, K=self.num_bins)

    # -- images (nsf/coupling.py:150-160): every pixel is a row of C channel features for the 2-D coupling kernel ----
    def _image(self, inputs, context, sample, ld, acc):
        """4-D inputs: mask over channels, conv conditioner output (B, nT*M, H, W) -> per pixel (nT, M) parameter rows
        (:152-156); the unconditional transform carries per-pixel parameters (`img_shape`), so it runs through the same
        kernel as a second 'conditioner' whose rows are shared across the batch."""
        if needs_grad(inputs, context, self):
            return self._image_autograd(inputs, context, sample, ld, acc)
        B, C, H, W = inputs.shape
        HW = H * W
        I, Tf = self.identity_features, self.transform_features
        rows = inputs.permute(0, 2, 3, 1).reshape(B * HW, C).contiguous()
        kw = self._kernel_kwargs()
        kw.pop("tails_i", None)
        kw.pop("bound_i", None)
        u = self.unconditional_transform
        ld_rows = torch.zeros(B * HW, dtype=inputs.dtype, device=inputs.device)

        def to_rows(p, n):   # (B, n*M, H, W) -> (B*HW, n*M) with the M numbers of a feature contiguous
            return p.reshape(B, n, -1, H, W).permute(0, 3, 4, 1, 2).reshape(B * HW, -1).contiguous()

        def uncond_call(x_rows, mode, y=None):
            prm = torch.cat([u.unnormalized_widths.detach(), u.unnormalized_heights.detach(),
                             u.unnormalized_derivatives.detach()], -1)          # (nI, H, W, M_i)
            if prm.dim() != 4:
                raise NotImplementedError("image coupling needs an unconditional transform built with img_shape")
            prm = prm.permute(1, 2, 0, 3).reshape(1, HW, -1).expand(B, HW, -1).reshape(B * HW, -1).contiguous()
            ukw = dict(min_bin_width=u.min_bin_width, min_bin_height=u.min_bin_height,
                       min_derivative=u.min_derivative, wh_div=1.0)
            ukw.update(_tails_kwargs(u.tails, u.tail_bound, "t", inputs.device, self.__dict__.setdefault("_tcache_u", {})))
            return ops.rqs_coupling(x_rows, prm, None, None, None, Tf, I, self.num_bins, mode, y=y, logdet=ld_rows,
                                    acc=L.LD_ADD, **ukw)   # roles swapped: the identity half is the 'transform' set here

        if not sample:
            cond = to_rows(self.transform_net(inputs[:, I, ...], context), len(Tf))
            y_rows, _ = ops.rqs_coupling(rows, cond, None, None, None, I, Tf, self.num_bins, L.RQS_DENSITY,
                                         logdet=ld_rows, acc=L.LD_ADD, **kw)
            if u is not None:
                yi, _ = uncond_call(rows, L.RQS_DENSITY)
                y_rows.index_copy_(1, I, yi.index_select(1, I))
        else:
            y_rows = rows.clone()
            if u is not None:
                uncond_call(rows, L.RQS_SAMPLE_TRANSFORM, y=y_rows)
            img = y_rows.view(B, H, W, C).permute(0, 3, 1, 2)
            cond = to_rows(self.transform_net(img[:, I, ...].contiguous(), context), len(Tf))
            ops.rqs_coupling(rows, cond, None, None, None, I, Tf, self.num_bins, L.RQS_SAMPLE_TRANSFORM, y=y_rows,
                             logdet=ld_rows, acc=L.LD_ADD, **kw)
        out = y_rows.view(B, H, W, C).permute(0, 3, 1, 2).contiguous()
        log_det = ld_rows.view(B, HW).sum(1)
        if ld is not None:
            if acc is None or acc > 0:
                ld += log_det
            else:
                ld -= log_det
            return out, ld
        return out, log_det

    def _image_autograd(self, inputs, context, sample, ld, acc):
        """Training path of _image: the same pixel-row view, each half through SplineFn (forward + backward kernel); the
        per-pixel unconditional parameters are expanded over the batch by torch, whose autograd sums their gradient."""
        B, C, H, W = inputs.shape
        HW = H * W
        I, Tf = self.identity_features, self.transform_features
        K = self.num_bins
        rows = inputs.permute(0, 2, 3, 1).reshape(B * HW, C)
        ident, trans = rows.index_select(1, I), rows.index_select(1, Tf)
        kw = self._kernel_kwargs()
        kw.pop("tails_i", None)
        kw.pop("bound_i", None)
        u = self.unconditional_transform

        def to_rows(p, n):
            return p.reshape(B, n, -1, H, W).permute(0, 3, 4, 1, 2).reshape(B * HW, -1).contiguous()

        def uncond(x_rows, inverse):
            prm = torch.cat([u.unnormalized_widths, u.unnormalized_heights, u.unnormalized_derivatives], -1)
            if prm.dim() != 4:
                raise NotImplementedError("image coupling needs an unconditional transform built with img_shape")
            prm = prm.permute(1, 2, 0, 3).reshape(1, HW, -1).expand(B, HW, -1).reshape(B * HW, -1).contiguous()
            ukw = dict(min_bin_width=u.min_bin_width, min_bin_height=u.min_bin_height,
                       min_derivative=u.min_derivative, wh_div=1.0)
            ukw.update(_tails_kwargs(u.tails, u.tail_bound, "t", inputs.device, self.__dict__.setdefault("_tcache_u", {})))
            return SplineFn.apply(x_rows.contiguous(), prm, None, None, None, K, inverse, ukw)

        ld_i = None
        if not sample:
            cond = to_rows(self.transform_net(inputs[:, I, ...], context), len(Tf))
            trans, ld_t = SplineFn.apply(trans.contiguous(), cond, None, None, None, K, False, kw)
            if u is not None:
                ident, ld_i = uncond(ident, False)
        else:
            if u is not None:
                ident, ld_i = uncond(ident, True)
            img_i = ident.reshape(B, H, W, len(I)).permute(0, 3, 1, 2).contiguous()
            cond = to_rows(self.transform_net(img_i, context), len(Tf))
            trans, ld_t = SplineFn.apply(trans.contiguous(), cond, None, None, None, K, True, kw)
        ld_rows = ld_t if ld_i is None else ld_t + ld_i
        y_rows = torch.empty(B * HW, C, dtype=inputs.dtype, device=inputs.device)
        y_rows = y_rows.index_copy(1, I, ident).index_copy(1, Tf, trans)
        out = y_rows.view(B, H, W, C).permute(0, 3, 1, 2).contiguous()
        log_det = ld_rows.view(B, HW).sum(1)
        if ld is not None:
            if acc is None or acc > 0:
                ld += log_det
            else:
                ld -= log_det
            return out, ld
        return out, log_det

    def _train_buffers(self, inputs):
        """Zero-padded images of the initial (H, D) and final (nT, 24, H) weights used by the fused training path: owned by
        this layer (its identity columns / real rows are the only entries ever written, the rest stays zero)."""
        net = self.transform_net
        w0, wf = net.initial_layer.weight, net.final_layer.weight
        buf = self.__dict__.get("_train_wbufs")
        if buf is None or buf[0].device != inputs.device or buf[0].dtype != w0.dtype:
            nT = self.transform_features.numel()
            col_map = torch.full((inputs.shape[1],), -1, dtype=torch.int32, device=inputs.device)
            col_map[self.identity_features] = torch.arange(self.identity_features.numel(), dtype=torch.int32, device=inputs.device)
            buf = (torch.zeros(w0.shape[0], inputs.shape[1], dtype=w0.dtype, device=inputs.device),
                   torch.zeros(nT, 24, wf.shape[1], dtype=wf.dtype, device=inputs.device), col_map,
                   torch.zeros(inputs.shape[1], w0.shape[0], dtype=w0.dtype, device=inputs.device))      # (D, H): w0^T on full rows
            self.__dict__["_train_wbufs"] = buf
        return buf

    def _train_fused_ok(self, inputs, context, sample):
        """The benchmark shape under autograd: trunk / whole layer on the fused training kernels."""
        return (not sample and self.use_fused and self.use_fused_train and inputs.is_cuda and inputs.shape[0] >= 1024
                and self._fused_eligible(inputs, context) and not self._fused_padded()
                and self.unconditional_transform is not None and self.num_bins == 8)

    def _train_full_ok(self, inputs, context, sample):
        """... and the whole layer's forward as one launch (CouplingTrainFn)."""
        return (self._train_fused_ok(inputs, context, sample) and _config.train_full and inputs.shape[1] == 64
                and self.transform_net.initial_layer.weight.shape[0] == 128)

    def _train_blob_for(self, inputs):
        blob = self.__dict__.get("_train_blob")
        if blob is None or blob.device != inputs.device:
            blob = self._train_blob = ops.rqs_fused_train_blob(len(self.transform_net.blocks), inputs.device)
        return blob

    # -- training path: same kernels through torch.autograd.Function (autograd.py), split/merge by torch indexing ---
    def _autograd(self, inputs, context, sample, ld, acc):
        kw = self._kernel_kwargs()
        u = self.unconditional_transform
        if self._train_fused_ok(inputs, context, sample):
            # the benchmark shape: trunk (initial layer + residual blocks, autograd-tracked), then the final Linear + the
            # coupling transform as ONE launch (FinalSplineDensityFn)
            net = self.transform_net
            inputs = inputs.contiguous()
            wfull, wpad, col_map, wfull_t = self._train_buffers(inputs)
            fkw = dict(tail_bound=float(self.tail_bound), min_bin_width=self.min_bin_width, min_bin_height=self.min_bin_height,
                       min_derivative=self.min_derivative, wh_div=self._wh_div(), col_map=col_map)
            if self._train_full_ok(inputs, context, sample):
                blob = self._train_blob_for(inputs)
                fkw["prepacked"] = _prepack.take(self)      # packed by run_chain's one launch for the whole model
                fkw["holder"] = self.__dict__.setdefault("_img_holder", {})    # whose weights the layer-owned images hold (autograd._stamp)
                blk = [p for b in net.blocks for l in b.linear_layers for p in (l.weight, l.bias)]
                return CouplingTrainFn.apply(inputs, net.initial_layer.weight, net.initial_layer.bias, net.final_layer.weight,
                                             net.final_layer.bias, u.unnormalized_widths, u.unnormalized_heights,
                                             u.unnormalized_derivatives, self.identity_features, self.transform_features, blob,
                                             self._fused_parity, fkw, wfull_t, wpad, ld, 1 if (acc is None or acc > 0) else -1,
                                             *blk)
            h2 = IdentLinearFn.apply(inputs, net.initial_layer.weight, net.initial_layer.bias, self.identity_features, wfull)
            for block in net.blocks:
                h2 = block(h2)
            blob = self.__dict__.get("_train_blob")
            if blob is None or blob.device != inputs.device:
                blob = self._train_blob = ops.rqs_fused_train_blob(len(net.blocks), inputs.device)
            fkw = dict(tail_bound=float(self.tail_bound), min_bin_width=self.min_bin_width, min_bin_height=self.min_bin_height,
                       min_derivative=self.min_derivative, wh_div=self._wh_div(),
                       holder=self.__dict__.setdefault("_img_holder", {}))
            outputs, log_det = FinalSplineDensityFn.apply(inputs.contiguous(), h2, net.final_layer.weight, net.final_layer.bias,
                                                          u.unnormalized_widths, u.unnormalized_heights,
                                                          u.unnormalized_derivatives, self.identity_features,
                                                          self.transform_features, blob, self._fused_parity, len(net.blocks), fkw,
                                                          wpad, ld, 1 if (acc is None or acc > 0) else -1)
            return outputs, log_det      # log_det IS ld (updated inside the launch) when the caller passed its accumulator
        ident = inputs.index_select(1, self.identity_features)
        if not sample:   # nsf/coupling.py:71-98 as one forward + one backward kernel on full rows
            cond = self.transform_net(ident, context)
            uw, uh, ud = (u.unnormalized_widths, u.unnormalized_heights, u.unnormalized_derivatives) if u is not None \
                else (None, None, None)
            outputs, log_det = CouplingDensityFn.apply(inputs.contiguous(), cond, uw, uh, ud, self.identity_features,
                                                       self.transform_features, self.num_bins, kw)
            if ld is not None:
                if acc is None or acc > 0:
                    ld += log_det
                else:
                    ld -= log_det
                return outputs, ld
            return outputs, log_det
        trans = inputs.index_select(1, self.transform_features)
        ld_i = None
        if not sample:
            cond = self.transform_net(ident, context)
            trans, ld_t = SplineFn.apply(trans, cond, None, None, None, self.num_bins, False, kw)
            if u is not None:
                ident, ld_i = SplineFn.apply(ident, None, u.unnormalized_widths, u.unnormalized_heights,
                                             u.unnormalized_derivatives, self.num_bins, False, kw)
        else:            # nsf/coupling.py:100-128
            if u is not None:
                ident, ld_i = SplineFn.apply(ident, None, u.unnormalized_widths, u.unnormalized_heights,
                                             u.unnormalized_derivatives, self.num_bins, True, kw)
            cond = self.transform_net(ident, context)
            trans, ld_t = SplineFn.apply(trans, cond, None, None, None, self.num_bins, True, kw)
        log_det = ld_t if ld_i is None else ld_t + ld_i
        outputs = torch.empty_like(inputs)
        outputs = outputs.index_copy(1, self.identity_features, ident).index_copy(1, self.transform_features, trans)
        if ld is not None:
            if acc is None or acc > 0:
                ld += log_det
            else:
                ld -= log_det
            return outputs, ld
        return outputs, log_det

    # -- fused path: conditioner on MFMA + spline epilogue in one kernel (csrc/rqs_fused.hip) -----------------
    def _fused_eligible(self, inputs, context):
        net = self.transform_net
        if not (isinstance(net, ResidualNet) and net.is_plain_relu()):
            return False
        if context is not None or inputs.dim() != 2 or inputs.dtype != torch.float32 or not inputs.is_cuda:
            return False
        if self.tails != "linear" or self._per_feature or self.unconditional_transform is None:
            return False
        if self._fused_ok is None:
            ii = self.identity_features.cpu()
            ti = self.transform_features.cpu()
            n = self.features
            alt0 = torch.equal(ii, torch.arange(0, n, 2)) and torch.equal(ti, torch.arange(1, n, 2))
            alt1 = torch.equal(ii, torch.arange(1, n, 2)) and torch.equal(ti, torch.arange(0, n, 2))
            # the kernel's shape is (64 features, 128 hidden units; 4, 8 or 16 bins); narrower layers run on it zero-padded
            # (_fused_blob / _pad_rows below): 2 <= features <= 64, hidden <= 128, any number of blocks
            ok = ((alt0 or alt1) and 2 <= n <= FUSED_D and net.hidden_features <= FUSED_H and self.num_bins in (4, 8, 16)
                  and self.min_bin_width * self.num_bins <= 1.0 and self.min_bin_height * self.num_bins <= 1.0
                  and ops.rqs_fused_supported(FUSED_D // 2, FUSED_D // 2, FUSED_H, len(net.blocks), self.num_bins))
            self._fused_ok = bool(ok)
            self._fused_parity = 0 if alt0 else 1
        return self._fused_ok

    def _fused_hidden(self):
        """`hidden` argument of the fused entry points: 64 / 32 tell the kernel that the units beyond of the (always 128-unit)
        blob are zero padding, so it skips their row-blocks and k-groups (the HB = 2 / 1 instantiations)."""
        h = self.transform_net.hidden_features
        return FUSED_H // 4 if h <= FUSED_H // 4 else (FUSED_H // 2 if h <= FUSED_H // 2 else FUSED_H)

    def _fused_live_d(self):
        """Columns in use, rounded up to the kernel's 16-column chunks: the final-layer groups of all-padding chunks are skipped."""
        return min(FUSED_D, 16 * ((self.features + 15) // 16))

    def _fused_padded(self):
        return self.features != FUSED_D or self.transform_net.hidden_features != FUSED_H

    def _pad_rows(self, x):
        """(B, D) rows -> (B, 64): the padding columns hold tail_bound + 1, outside every spline's interval, so they pass
        through the layer (and the padded LU's identity block) unchanged with log-det 0 (utils/splines.py:28, :40-41)."""
        if x.shape[1] == FUSED_D:
            return x
        xp = torch.full((x.shape[0], FUSED_D), float(self.tail_bound) + 1.0, dtype=x.dtype, device=x.device)
        xp[:, :x.shape[1]] = x
        return xp

    def _padded_tensors(self, d, lu_t, eps):
        """The layer's weights embedded in the kernel's shape: zero rows / columns for the missing hidden units and features
        (a zero hidden unit stays zero through bias-free ReLU blocks; a padding transform feature's parameters never matter,
        its x sits in the tail), the width / height rows of the final layer rescaled for the kernel's fixed 1 / sqrt(128)
        (nsf/coupling.py:334-339 divides by sqrt(hidden)), LU factors extended by an identity block."""
        net = self.transform_net
        h, nb = net.hidden_features, len(net.blocks)
        nI, nT = len(self.identity_features), len(self.transform_features)
        dev = d[0].device
        M = 3 * self.num_bins - 1
        z = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=dev)   # noqa: E731
        w0, b0 = z(FUSED_H, FUSED_D // 2), z(FUSED_H)
        w0[:h, :nI], b0[:h] = d[0], d[1]
        out = [w0, b0]
        for i in range(2 * nb):
            w, b = z(FUSED_H, FUSED_H), z(FUSED_H)
            w[:h, :h], b[:h] = d[2 + 2 * i], d[3 + 2 * i]
            out += [w, b]
        o = 2 + 4 * nb
        wf, bf = z(FUSED_D // 2 * M, FUSED_H), z(FUSED_D // 2 * M)
        sc = torch.ones(M, dtype=torch.float32, device=dev)
        sc[:2 * self.num_bins] = float(np.sqrt(FUSED_H / h))
        wf[:nT * M, :h] = (d[o].view(nT, M, h) * sc.view(1, M, 1)).reshape(nT * M, h)
        bf[:nT * M] = (d[o + 1].view(nT, M) * sc.view(1, M)).reshape(-1)
        out += [wf, bf]
        for t, width in ((d[o + 2], self.num_bins), (d[o + 3], self.num_bins), (d[o + 4], self.num_bins - 1)):
            u = z(FUSED_D // 2, width)
            u[:nI] = t
            out.append(u)
        lu_out = None
        if lu_t is not None:
            perm, low, up, diag, bias = lu_t
            D = self.features
            li, ui = torch.tril_indices(D, D, -1, device=dev), torch.triu_indices(D, D, 1, device=dev)
            Lm, Um = z(FUSED_D, FUSED_D), z(FUSED_D, FUSED_D)
            Lm[li[0], li[1]] = low
            Um[ui[0], ui[1]] = up
            LI, UI = torch.tril_indices(FUSED_D, FUSED_D, -1, device=dev), torch.triu_indices(FUSED_D, FUSED_D, 1, device=dev)
            dg = torch.full((FUSED_D,), float(np.log(np.exp(1.0 - eps) - 1.0)), dtype=torch.float32, device=dev)  # softplus + eps = 1
            dg[:D] = diag
            bp = z(FUSED_D)
            bp[:D] = bias
            pp = torch.arange(FUSED_D, device=dev, dtype=perm.dtype)
            pp[:D] = perm
            lu_out = (pp, Lm[LI[0], LI[1]].contiguous(), Um[UI[0], UI[1]].contiguous(), dg, bp)
        return out, lu_out

    def _fused_blob(self, lu=None):
        net, u = self.transform_net, self.unconditional_transform
        tensors = [net.initial_layer.weight, net.initial_layer.bias]
        for b in net.blocks:
            for lin in b.linear_layers:
                tensors += [lin.weight, lin.bias]
        tensors += [net.final_layer.weight, net.final_layer.bias, u.unnormalized_widths, u.unnormalized_heights,
                    u.unnormalized_derivatives]
        key = _keys.pkey(tensors)
        if lu is not None:
            lin = lu.linear
            lu_t = [lu.permutation._permutation, lin.lower_entries, lin.upper_entries, lin.unconstrained_upper_diag,
                    lin.bias]
            key = key + _keys.pkey(lu_t) + (lin.eps,)
        if self._fused_cache is None or self._fused_cache[0] != key:
            d = [t.detach() for t in tensors]
            nb = len(net.blocks)
            if self._fused_padded():
                d, lu_pad = self._padded_tensors(d, None if lu is None else [t.detach() for t in lu_t], lin.eps if lu is not None else 0.0)
                if lu is not None:
                    lu_t = list(lu_pad)
            wb = [d[2 + 2 * i] for i in range(2 * nb)]
            bb = [d[3 + 2 * i] for i in range(2 * nb)]
            o = 2 + 4 * nb
            blob = ops.rqs_fused_pack(d[0], d[1], wb, bb, d[o], d[o + 1], d[o + 2], d[o + 3], d[o + 4], self.num_bins,
                                      self.tail_bound, self.min_bin_width, self.min_bin_height, self.min_derivative)
            if lu is not None:
                ops.rqs_fused_pack_lu(blob, nb, lu_t[0], lu_t[1].detach(), lu_t[2].detach(), lu_t[3].detach(),
                                      lu_t[4].detach(), eps=lin.eps, K=self.num_bins)
            self._fused_cache = (key, blob)
        return self._fused_cache[1]

    def _fused_x3_blob(self, lu):
        """Split-bf16 blob of this layer (+ its LU), derived from the fp32 blob once per parameter version."""
        blob = self._fused_blob(lu)
        key = self._fused_cache[0]
        if self._fused_x3_cache is None or self._fused_x3_cache[0] is not key:
            self._fused_x3_cache = (key, ops.rqs_fused_x3_pack(blob, len(self.transform_net.blocks), lu is not None))
        return self._fused_x3_cache[1]

    def _fused(self, inputs, direction, ld=None, acc=None, lu=None):
        self._check(inputs)
        if inputs.shape[1] != FUSED_D:     # narrower layer: the kernel's 64 columns, padding in the tails
            y, l = self._fused_run(self._pad_rows(inputs), direction, ld, acc, lu)
            return y[:, :inputs.shape[1]].contiguous(), l
        return self._fused_run(inputs, direction, ld, acc, lu)

    def _fused_run(self, inputs, direction, ld=None, acc=None, lu=None):
        net = self.transform_net
        from .. import config
        if config.fused_gemm == "bf16x3" and self.num_bins == 8:      # the split-bf16 kernel is instantiated for 8 bins only
            return ops.rqs_fused_x3(inputs, self._fused_x3_blob(lu), self._fused_parity, FUSED_H,
                                    len(net.blocks), self.num_bins, direction, logdet=ld, acc=acc,
                                    tail_bound=self.tail_bound, min_bin_width=self.min_bin_width,
                                    min_bin_height=self.min_bin_height, min_derivative=self.min_derivative,
                                    fuse_lu=lu is not None)
        return ops.rqs_fused(inputs, self._fused_blob(lu), self._fused_parity, self._fused_hidden(), len(net.blocks),
                             self.num_bins, direction, logdet=ld, acc=acc, tail_bound=self.tail_bound,
                             min_bin_width=self.min_bin_width, min_bin_height=self.min_bin_height,
                             min_derivative=self.min_derivative, fuse_lu=lu is not None, live_d=self._fused_live_d())


class CoupledRationalQuadraticSpline(Flow):
    """Neural spline flow coupling layer (wrapper.py:14-85).  NOTE the direction swap of the reference:
    forward (generative) = prqct.inverse, inverse (normalising) = prqct.forward."""

    def __init__(self, num_input_channels, num_blocks, num_hidden_channels, num_context_channels=None, num_bins=8,
                 tails="linear", tail_bound=3.0, activation=nn.ReLU, dropout_probability=0.0, reverse_mask=False,
                 init_identity=True):
        super().__init__()

        def transform_net_create_fn(in_features, out_features):
            net = ResidualNet(in_features=in_features, out_features=out_features,
                              context_features=num_context_channels, hidden_features=num_hidden_channels,
                              num_blocks=num_blocks, activation=activation(),
                              dropout_probability=dropout_probability, use_batch_norm=False)
            if init_identity:
                torch.nn.init.constant_(net.final_layer.weight, 0.0)
                torch.nn.init.constant_(net.final_layer.bias, np.log(np.exp(1 - DEFAULT_MIN_DERIVATIVE) - 1))
            return net

        self.prqct = PiecewiseRationalQuadraticCoupling(
            mask=create_alternating_binary_mask(num_input_channels, even=reverse_mask),
            transform_net_create_fn=transform_net_create_fn, num_bins=num_bins, tails=tails, tail_bound=tail_bound,
            apply_unconditional_transform=True)

    def forward(self, z, context=None):
        z, log_det = self.prqct._sample(z, context)
        return z, log_det.view(-1)

    def inverse(self, z, context=None):
        z, log_det = self.prqct._density(z, context)
        return z, log_det.view(-1)

    def _run(self, z, inverse, ld, acc, context=None, **kw):
        if inverse:
            y, _ = self.prqct._density(z, context, ld=ld, acc=acc)
        else:
            y, _ = self.prqct._sample(z, context, ld=ld, acc=acc)
        return y

    # -- pair fusion with the LULinearPermute that follows this layer in the flow list ----------------------------
    def _pair_eligible(self, z, lu):
        """True when [this layer, lu] can run as ONE kernel: density = lu.inverse then self.inverse (the order
        NormalizingFlow.log_prob visits them, core.py:193-195), sample = self.forward then lu.forward."""
        p = self.prqct
        if needs_grad(z, self, lu):
            return False
        if not (lu.linear.features == p.features and lu.linear.bias.dtype == torch.float32 and lu.linear.bias.is_cuda):
            return False
        if p.use_fused and p._fused_eligible(z, None):
            return True
        # beyond the benchmark kernel's shapes: nf_nsf_wide with the LU layer's dense matrix in the same launch
        return p.features <= 128 and lu.use_dense and p._wide_pack(z, None) is not None

    def _run_pair_train(self, z, lu, ld, acc):
        """The pair's density direction under autograd as autograd.PairTrainFn (round 6); the caller (core._run_chain_impl) has
        checked _prepack.take_pair: every image the kernels read was written by this step's multi-layer packs."""
        from ..autograd import PairTrainFn
        p = self.prqct
        net, u, lin = p.transform_net, p.unconditional_transform, lu.linear
        z = z.contiguous()
        wfull, wpad, col_map, wfull_t = p._train_buffers(z)
        fkw = dict(tail_bound=float(p.tail_bound), min_bin_width=p.min_bin_width, min_bin_height=p.min_bin_height,
                   min_derivative=p.min_derivative, wh_div=p._wh_div(), col_map=col_map, prepacked=True,
                   holder=p.__dict__.setdefault("_img_holder", {}))
        blk = [q for b in net.blocks for l in b.linear_layers for q in (l.weight, l.bias)]
        y, _ = PairTrainFn.apply(z, lu.permutation._permutation, lin.lower_entries, lin.upper_entries, lin.unconstrained_upper_diag,
                                 lin.bias, lin.eps, lu._factors_buffer(z.device), lu._wd_buffer(z.device), net.initial_layer.weight,
                                 net.initial_layer.bias, net.final_layer.weight, net.final_layer.bias, u.unnormalized_widths,
                                 u.unnormalized_heights, u.unnormalized_derivatives, p.identity_features, p.transform_features,
                                 p._train_blob_for(z), p._fused_parity, fkw, wfull_t, wpad, ld, 1 if (acc is None or acc > 0) else -1,
                                 *blk)
        return y

    def _run_pair(self, z, lu, inverse, ld, acc):
        p = self.prqct
        if p.use_fused and p._fused_eligible(z, None):
            y, _ = p._fused(z, 0 if inverse else 1, ld, acc, lu=lu)
            return y
        direction = 0 if inverse else 1
        y, _ = p._wide(z, p._wide_pack(z, None, lu=lu, direction=direction), direction, ld, acc)
        return y


class CircularCoupledRationalQuadraticSpline(Flow):
    """NSF coupling layer with circular coordinates (wrapper.py:88-185): the features `ind_circ` get circular tails,
    the others linear ones, and the conditioner sees periodic features of the circular identity coordinates."""

    def __init__(self, num_input_channels, num_blocks, num_hidden_channels, ind_circ, num_context_channels=None,
                 num_bins=8, tail_bound=3.0, activation=nn.ReLU, dropout_probability=0.0, reverse_mask=False,
                 mask=None, init_identity=True):
        super().__init__()
        if mask is None:
            mask = create_alternating_binary_mask(num_input_channels, even=reverse_mask)
        features_vector = torch.arange(num_input_channels)
        identity_features = features_vector.masked_select(mask <= 0)
        ind_circ = torch.tensor(ind_circ)
        ind_circ_id = [i for i, idf in enumerate(identity_features) if idf in ind_circ]
        if torch.is_tensor(tail_bound):
            scale_pf = np.pi / tail_bound[ind_circ_id]
        else:
            scale_pf = np.pi / tail_bound

        def transform_net_create_fn(in_features, out_features):
            pf = PeriodicFeaturesElementwise(in_features, ind_circ_id, scale_pf) if len(ind_circ_id) > 0 else None
            net = ResidualNet(in_features=in_features, out_features=out_features,
                              context_features=num_context_channels, hidden_features=num_hidden_channels,
                              num_blocks=num_blocks, activation=activation(), dropout_probability=dropout_probability,
                              use_batch_norm=False, preprocessing=pf)
            if init_identity:
                torch.nn.init.constant_(net.final_layer.weight, 0.0)
                torch.nn.init.constant_(net.final_layer.bias, float(np.log(np.exp(1 - DEFAULT_MIN_DERIVATIVE) - 1)))
            return net

        tails = ["circular" if i in ind_circ else "linear" for i in range(num_input_channels)]
        self.prqct = PiecewiseRationalQuadraticCoupling(mask=mask, transform_net_create_fn=transform_net_create_fn,
                                                        num_bins=num_bins, tails=tails, tail_bound=tail_bound,
                                                        apply_unconditional_transform=True)

    def forward(self, z, context=None):
        z, log_det = self.prqct.inverse(z, context)
        return z, log_det.view(-1)

    def inverse(self, z, context=None):
        z, log_det = self.prqct(z, context)
        return z, log_det.view(-1)
