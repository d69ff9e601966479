"""Autoregressive flows: Autoregressive, MaskedAffineAutoregressive (MAF), MaskedPiecewiseRationalQuadraticAutoregressive
and its wrapper AutoregressiveRationalQuadraticSpline (AR-NSF).

Mirrors normflows/flows/affine/autoregressive.py:10-128 (constructor signature, `autoregressive_net.*` state_dict
keys, direction semantics: `forward` = one MADE pass, `inverse` = D sequential MADE passes).  The element-wise
affine transform and its log-det are one HIP kernel (nf_maf_affine).  Without gradient tracking the single-pass direction is ONE
launch: nf_made_forward_affine (MADE on fp32 MFMA over the masks' non-zero blocks + the affine epilogue, csrc/made_fwd.hip), and
MADE itself is one launch (nf_made_forward) for the other element-wise transforms; under autograd MADE runs on
nf_made_forward_train / nf_made_backward / nf_made_wgrad (autograd.MadeFn, csrc/made_bwd.hip); with a context, in float64 or for
structures outside flows/made_pack.py, MADE's masked linears are library GEMMs on pre-masked weights.

The inverse of MaskedAffineAutoregressive runs as ONE launch of nf_maf_inverse when
the MADE has the supported structure (flows/maf_pack.py): every hidden unit is finalised once, total work = one MADE
pass instead of D; the autoregressive spline layer (AR-NSF sampling) does the same through nf_arnsf_inverse.  Under autograd the
affine layer's inverse is differentiated implicitly (autograd.MafInverseFn: the one-pass kernel forward, chain sweeps + one
weight-gradient launch backward); other structures, float64 and the spline layer's gradient-tracking inverse keep the reference's
D-pass loop.
"""
import numpy as np
import torch
from .. import _keys
from torch.nn import functional as F

from .. import autograd, nets, ops
from . import maf_pack
from .base import Flow


class Autoregressive(Flow):
    """Element-wise invertible transform with parameters from an autoregressive net (autoregressive.py:10-47)."""

    def __init__(self, autoregressive_net):
        super().__init__()
        self.autoregressive_net = autoregressive_net

    def forward(self, inputs, context=None):
        params = self.autoregressive_net(inputs, context)
        return self._elementwise(inputs, params, 0)

    def inverse(self, inputs, context=None):
        if autograd.needs_grad(inputs, *self.parameters()) and autograd.ArInverseImplicitFn.eligible(self, inputs, context):
            # implicit differentiation (one graph-free inverse + <= D backward sweeps) instead of D recorded passes
            return autograd.ArInverseImplicitFn.apply(self, inputs, *self.parameters())
        return self._inverse_loop(inputs, context)

    def _inverse_loop(self, inputs, context=None):
        """autoregressive.py:29-40: D passes of the net, feature i final after pass i (recorded one by one under autograd)."""
        num_inputs = int(np.prod(inputs.shape[1:]))
        outputs = torch.zeros_like(inputs)
        logabsdet = None
        for i in range(num_inputs):
            params = self.autoregressive_net(outputs, context)
            last = i == num_inputs - 1
            outputs, ld = self._elementwise(inputs, params, 1, want_logdet=last)
            if last:
                logabsdet = ld
        return outputs, logabsdet

    def _elementwise(self, inputs, params, direction, want_logdet=True):
        raise NotImplementedError()

    def _output_dim_multiplier(self):
        raise NotImplementedError()


class MaskedAffineAutoregressive(Autoregressive):
    """Masked autoregressive flow, arXiv 1705.07057 (autoregressive.py:50-128)."""

    def __init__(self, features, hidden_features, context_features=None, num_blocks=2, use_residual_blocks=True,
                 random_mask=False, activation=F.relu, dropout_probability=0.0, use_batch_norm=False):
        self.features = features
        made = nets.MADE(features=features, hidden_features=hidden_features, context_features=context_features,
                         num_blocks=num_blocks, output_multiplier=self._output_dim_multiplier(),
                         use_residual_blocks=use_residual_blocks, random_mask=random_mask, activation=activation,
                         dropout_probability=dropout_probability, use_batch_norm=use_batch_norm)
        super().__init__(made)

    def _output_dim_multiplier(self):
        return 2

    @staticmethod
    def _tri():
        """Format 1 (regular tiles on the triangular sequential part, nf_maf_inverse_h_tri) -- the default on the half-sharing mapping."""
        from .. import config
        return bool(config.maf_halves and config.maf_tri)

    def _packed(self, device):
        """Device copies of the incremental-inverse pack, rebuilt when any MADE parameter changes:
        (blob, table, hidden_padded, num_blocks, table_host); table_host = the host copy of a format-1 table, else None."""
        tri = self._tri()
        key = _keys.pkey(self.autoregressive_net.parameters()) + (str(device), tri)
        cache = getattr(self, "_maf_pack_cache", None)
        if cache is None or cache[0] != key:
            if str(device) != "cpu":      # structure (shared by layers with these masks) + one gather on the device
                st = self._inverse_struct(device)
                packed = None
                if st is not None:
                    plist = [t for l in self.autoregressive_net._linears() for t in (l.weight, l.bias)]
                    packed = (ops.pack_gather(plist, st[0]), st[1], st[2], st[3], st[4])
            else:
                packed = maf_pack.pack_made(self.autoregressive_net, blocks=(1, 2, 3), tri=tri)    # 1..3 residual blocks: nf_maf_inverse_h
                if packed is not None:
                    blob, table = packed
                    packed = (torch.from_numpy(blob).to(device), torch.from_numpy(table).to(device), int(table[3]), int(table[6]),
                              table if tri else None)
            self._maf_pack_cache = cache = (key, packed)
        return cache[1]

    def _inverse_struct(self, device):
        """(gather indices, table, hidden_padded, num_blocks, table_host) of the one-pass inverse kernel's pack on the device;
        None = unsupported."""
        from ..flows import made_pack
        net = self.autoregressive_net
        tri = self._tri()
        skey = (str(device), tri) + tuple((l.mask.data_ptr(), l.mask._version) for l in net._linears())
        st = self.__dict__.get("_inv_struct")
        if st is None or st[0] != skey:
            struct = made_pack.maf_inverse_structure(net, tri=tri)
            if struct is not None:
                struct = (torch.from_numpy(struct[0]).to(device), torch.from_numpy(struct[1]).to(device), int(struct[1][3]),
                          int(struct[1][6]), struct[1] if tri else None)
            st = self.__dict__["_inv_struct"] = (skey, struct)
        return st[1]

    def forward(self, inputs, context=None):
        """autoregressive.py:24-27: one MADE pass + the element-wise affine map -- ONE launch (nf_made_forward_affine) for the
        MADE structures flows/made_pack.py takes; layer by layer otherwise (context, float64, gradient tracking, ...)."""
        if (context is None and inputs.dim() == 2 and inputs.dtype == torch.float32 and inputs.is_cuda
                and not autograd.needs_grad(inputs, *self.autoregressive_net.parameters())):
            packed = self.autoregressive_net.packed_forward(inputs.device)
            if packed is not None:
                return ops.made_forward_affine(inputs, packed[0], packed[1], packed[2])
        return super().forward(inputs, context)

    def inverse(self, inputs, context=None):
        if (context is None and inputs.dim() == 2 and inputs.dtype == torch.float32 and inputs.is_cuda
                and not autograd.needs_grad(inputs, *self.autoregressive_net.parameters())):
            packed = self._packed(inputs.device)
            if packed is not None:
                return ops.maf_inverse(inputs, packed[0], packed[1], packed[2], num_blocks=packed[3], table_host=packed[4])
        from .. import config
        if (context is None and inputs.dim() == 2 and inputs.dtype == torch.float32 and inputs.is_cuda and config.maf_implicit
                and config.made_train and config.made_fused):
            packs = self._implicit_packs(inputs.device)
            if packs is not None:          # autograd.MafInverseFn: implicit differentiation instead of autograd through the D passes
                plist = [t for l in self.autoregressive_net._linears() for t in (l.weight, l.bias)]
                return autograd.MafInverseFn.apply(packs[0], packs[1], packs[2], inputs, *plist)
        if not config.maf_implicit:        # (the reference's D recorded passes, asked for explicitly)
            return self._inverse_loop(inputs, context)
        return super().inverse(inputs, context)

    def _implicit_packs(self, device):
        """(inverse pack, forward pack, backward pack) gathered on the device from the CURRENT parameters (value-independent structures
        cached per module: flows/made_pack.maf_inverse_structure / made_train_structure); None outside those structures."""
        from ..flows import made_pack
        net = self.autoregressive_net
        plist = [t for l in net._linears() for t in (l.weight, l.bias)]
        skey = tuple((l.mask.data_ptr(), l.mask._version) for l in net._linears())
        packs = nets._train_packs_from(net, lambda: made_pack.made_train_structure(net, 2), plist, device, skey=skey)
        if packs is None:
            return None
        from .. import config
        if config.maf_onepass and config.maf_halves:
            # round 5: the inverse on its FORMAT-0 pack (it leaves the ReLU masks in format-0 positions) + the transposed pack of the
            # one-pass solve (made_pack.maf_solve_t_structure), both gathered from the current parameters
            tri = self._tri()          # the forward on the format-1 pack (fast kernel) when available; the masks follow its positions
            st = self.__dict__.get("_onepass_struct")
            if st is None or st[0] != (str(device), tri) + skey:
                s0 = made_pack.maf_inverse_structure(net, tri=tri)
                s2 = made_pack.maf_solve_t_structure(net, tri=tri)
                val = None
                if s0 is not None and s2 is not None:
                    dev = lambda a: torch.from_numpy(a).to(device)
                    from . import maf_pack
                    gc = maf_pack.solve_t_gradient_columns(net, tri=tri)
                    fc = maf_pack.solve_t_gradient_columns(net, tri=tri, forward=True)
                    pw = maf_pack.position_wgrad_tables(net, tri=tri)
                    if pw is not None:
                        pw = dict(pw, wtable=dev(pw["wtable"]), stable=dev(pw["stable"]))
                    val = (dev(s0[0]), dev(s0[1]), int(s0[1][3]), int(s0[1][6]), int(s0[1][4]), dev(s2[0]), dev(s2[1]),
                           s0[1] if tri else None, None if gc is None else dev(gc), None if fc is None else dev(fc),
                           dev(maf_pack.final_layer_columns(net)), pw, s2[1] if tri else None)
                st = self.__dict__["_onepass_struct"] = ((str(device), tri) + skey, val)
            if st[1] is not None:
                src0, table0, hp, nb, tiles, src2, table2, th, gcols, fcols, srcf, pw, tth = st[1]
                inv = {"blob": ops.pack_gather(plist, src0), "table": table0, "hp": hp, "nb": nb, "tiles": tiles,
                       "tblob": ops.pack_gather(plist, src2), "ttable": table2, "table_host": th, "gcols": gcols, "fcols": fcols,
                       "wf_src": srcf if fcols is not None else None, "pw": pw, "ttable_host": tth}
                return inv, packs[0], packs[1]
        inv = self._inverse_struct(device)
        if inv is None:
            return None
        src, table, hp, nb, th = inv
        return (ops.pack_gather(plist, src), table, hp, nb, th), packs[0], packs[1]

    def _elementwise(self, inputs, params, direction, want_logdet=True):
        if inputs.dim() != 2:
            raise NotImplementedError("MaskedAffineAutoregressive: (batch, features) inputs only")
        if autograd.needs_grad(inputs, params):
            from .. import config as _cfg
            if _cfg.torch_elementwise:        # config.higher_order_gradients(): affine/autoregressive.py:98-128 in torch ops
                prm = params.view(-1, inputs.shape[1], 2)
                scale = torch.sigmoid(prm[..., 0] + 2.0) + 1e-3
                log_scale = torch.log(scale).sum(1)
                if direction == 0:
                    return scale * inputs + prm[..., 1], log_scale
                return (inputs - prm[..., 1]) / scale, -log_scale
            return autograd.MafAffineFn.apply(inputs.contiguous(), params, direction)
        return ops.maf_affine(inputs, params, direction, want_logdet=want_logdet)


class MaskedPiecewiseRationalQuadraticAutoregressive(Autoregressive):
    """Autoregressive rational-quadratic spline transform (neural_spline/autoregressive.py:17-140).  MADE produces
    3K-1 | 3K | 3K+1 numbers per feature (linear | circular | no tails); the element-wise spline and its row-summed
    log-det are ONE launch of nf_rqs_coupling with every column a transform column (no identity half).  The widths and
    heights are NOT divided by sqrt(hidden): the reference tests `hasattr(net, "hidden_features")` (:107-109) and its
    MADE never sets that attribute."""

    def __init__(self, features, hidden_features, context_features=None, num_bins=10, tails=None, tail_bound=1.0,
                 num_blocks=2, use_residual_blocks=True, random_mask=False, permute_mask=False, activation=F.relu,
                 dropout_probability=0.0, use_batch_norm=False, init_identity=True, min_bin_width=1e-3,
                 min_bin_height=1e-3, min_derivative=1e-3):
        from .neural_spline import _check_tails
        _check_tails(tails, tail_bound)
        self.num_bins = num_bins
        self.min_bin_width = min_bin_width
        self.min_bin_height = min_bin_height
        self.min_derivative = min_derivative
        self.tails = tails
        self.features = features
        if isinstance(self.tails, (list, tuple)):   # periodic features of the circular coordinates (:44-55)
            ind_circ = [i for i in range(features) if self.tails[i] == "circular"]
            scale_pf = np.pi / tail_bound[ind_circ] if torch.is_tensor(tail_bound) else np.pi / tail_bound
            preprocessing = nets.PeriodicFeaturesElementwise(features, ind_circ, scale_pf)
        else:
            preprocessing = None
        made = nets.MADE(features=features, hidden_features=hidden_features, context_features=context_features,
                         num_blocks=num_blocks, output_multiplier=self._output_dim_multiplier(),
                         use_residual_blocks=use_residual_blocks, random_mask=random_mask, permute_mask=permute_mask,
                         activation=activation, dropout_probability=dropout_probability,
                         use_batch_norm=use_batch_norm, preprocessing=preprocessing)
        if init_identity:
            torch.nn.init.constant_(made.final_layer.weight, 0.0)
            torch.nn.init.constant_(made.final_layer.bias, float(np.log(np.exp(1 - min_derivative) - 1)))
        super().__init__(made)
        if torch.is_tensor(tail_bound):
            self.register_buffer("tail_bound", tail_bound)
        else:
            self.tail_bound = tail_bound

    def _output_dim_multiplier(self):
        if self.tails == "linear":
            return self.num_bins * 3 - 1
        if self.tails == "circular":
            return self.num_bins * 3
        return self.num_bins * 3 + 1

    def _packed(self, device):
        """Device copies of the incremental-inverse pack (rows layout), rebuilt when any MADE parameter changes."""
        key = _keys.pkey(self.autoregressive_net.parameters()) + (str(device),)
        cache = getattr(self, "_arnsf_pack_cache", None)
        if cache is None or cache[0] != key:
            packed = maf_pack.pack_made(self.autoregressive_net, mult=self._output_dim_multiplier(), rows=True)
            if packed is not None:
                blob, table = packed
                packed = (torch.from_numpy(blob).to(device), torch.from_numpy(table).to(device), int(table[3]))
            self._arnsf_pack_cache = cache = (key, packed)
        return cache[1]

    def forward(self, inputs, context=None):
        """Density direction (autoregressive.py:24-27 + neural_spline/autoregressive.py:94-134): MADE + the element-wise spline as
        ONE launch (nf_made_forward_spline) for 8 bins, linear tails, a scalar tail bound and the MADE structures
        flows/made_pack.py takes; otherwise one MADE pass (one launch where possible) + nf_rqs_coupling."""
        if (context is None and inputs.dim() == 2 and inputs.dtype == torch.float32 and inputs.is_cuda and self.tails == "linear"
                and self.num_bins == 8 and not torch.is_tensor(self.tail_bound)
                and not hasattr(self.autoregressive_net, "hidden_features")
                and self.min_bin_width * 8 <= 1.0 and self.min_bin_height * 8 <= 1.0
                and not autograd.needs_grad(inputs, *self.autoregressive_net.parameters())):
            packed = self.autoregressive_net.packed_forward(inputs.device, spline=True)
            if packed is not None:
                return ops.made_forward_spline(inputs, packed[0], packed[1], packed[2], float(self.tail_bound), self.min_bin_width,
                                               self.min_bin_height, self.min_derivative)
        return super().forward(inputs, context)

    def inverse(self, inputs, context=None):
        """One launch of nf_arnsf_inverse for the supported MADE structure (scalar tails, float32, no context, no
        sqrt(hidden) scaling, no gradient tracking); the reference's D-pass loop otherwise."""
        if (context is None and inputs.dim() == 2 and inputs.dtype == torch.float32 and inputs.is_cuda
                and (self.tails is None or isinstance(self.tails, str)) and not torch.is_tensor(self.tail_bound)
                and not hasattr(self.autoregressive_net, "hidden_features")
                and not autograd.needs_grad(inputs, *self.autoregressive_net.parameters())):
            packed = self._packed(inputs.device)
            if packed is not None:
                return ops.arnsf_inverse(inputs, packed[0], packed[1], packed[2], self.num_bins, self.tails,
                                         float(self.tail_bound), self.min_bin_width, self.min_bin_height,
                                         self.min_derivative)
        return super().inverse(inputs, context)

    def _elementwise(self, inputs, params, direction, want_logdet=True):
        if inputs.dim() != 2:
            raise NotImplementedError("MaskedPiecewiseRationalQuadraticAutoregressive: (batch, features) inputs only")
        B, D = inputs.shape
        idx = getattr(self, "_all_idx", None)
        if idx is None or idx[0].device != inputs.device or idx[1].numel() != D:
            idx = (torch.empty(0, dtype=torch.long, device=inputs.device),
                   torch.arange(D, dtype=torch.long, device=inputs.device))
            self._all_idx = idx
        wh_div = float(np.sqrt(self.autoregressive_net.hidden_features)) \
            if hasattr(self.autoregressive_net, "hidden_features") else 1.0
        from .neural_spline import _tails_kwargs
        kw = _tails_kwargs(self.tails, self.tail_bound, "t", inputs.device, self.__dict__.setdefault("_tcache", {}))
        if autograd.needs_grad(inputs, params):
            kw = dict(kw, min_bin_width=self.min_bin_width, min_bin_height=self.min_bin_height,
                      min_derivative=self.min_derivative, wh_div=wh_div)
            return autograd.SplineFn.apply(inputs, params, None, None, None, self.num_bins, direction == 1, kw)
        mode = ops.L.RQS_DENSITY if direction == 0 else ops.L.RQS_SAMPLE_TRANSFORM
        return ops.rqs_coupling(inputs, params.contiguous(), None, None, None, idx[0], idx[1], self.num_bins, mode,
                                min_bin_width=self.min_bin_width, min_bin_height=self.min_bin_height,
                                min_derivative=self.min_derivative, wh_div=wh_div, **kw)


class AutoregressiveRationalQuadraticSpline(Flow):
    """NSF autoregressive layer (neural_spline/wrapper.py:188-245): like the coupling wrapper the directions are
    swapped, forward (generative) = D-pass inverse of the transform, inverse (density) = one MADE pass + spline."""

    def __init__(self, num_input_channels, num_blocks, num_hidden_channels, num_context_channels=None, num_bins=8,
                 tail_bound=3, activation=torch.nn.ReLU, dropout_probability=0.0, permute_mask=False,
                 init_identity=True):
        super().__init__()
        act = activation()
        self.mprqat = MaskedPiecewiseRationalQuadraticAutoregressive(
            features=num_input_channels, hidden_features=num_hidden_channels, context_features=num_context_channels,
            num_bins=num_bins, tails="linear", tail_bound=tail_bound, num_blocks=num_blocks, use_residual_blocks=True,
            random_mask=False, permute_mask=permute_mask, activation=F.relu if isinstance(act, torch.nn.ReLU) else act,
            dropout_probability=dropout_probability, use_batch_norm=False, init_identity=init_identity)

    def forward(self, z, context=None):
        z, log_det = self.mprqat.inverse(z, context=context)
        return z, log_det.view(-1)

    def inverse(self, z, context=None):
        z, log_det = self.mprqat(z, context=context)
        return z, log_det.view(-1)


class CircularAutoregressiveRationalQuadraticSpline(Flow):
    """Autoregressive NSF layer with circular coordinates (wrapper.py:247-330)."""

    def __init__(self, num_input_channels, num_blocks, num_hidden_channels, ind_circ, num_context_channels=None,
                 num_bins=8, tail_bound=3, activation=torch.nn.ReLU, dropout_probability=0.0, permute_mask=True,
                 init_identity=True):
        super().__init__()
        tails = ["circular" if i in ind_circ else "linear" for i in range(num_input_channels)]
        act = activation()
        self.mprqat = MaskedPiecewiseRationalQuadraticAutoregressive(
            features=num_input_channels, hidden_features=num_hidden_channels, context_features=num_context_channels,
            num_bins=num_bins, tails=tails, tail_bound=tail_bound, num_blocks=num_blocks, use_residual_blocks=True,
            random_mask=False, permute_mask=permute_mask, activation=F.relu if isinstance(act, torch.nn.ReLU) else act,
            dropout_probability=dropout_probability, use_batch_norm=False, init_identity=init_identity)

    def forward(self, z, context=None):
        z, log_det = self.mprqat.inverse(z, context=context)
        return z, log_det.view(-1)

    def inverse(self, z, context=None):
        z, log_det = self.mprqat(z, context=context)
        return z, log_det.view(-1)
