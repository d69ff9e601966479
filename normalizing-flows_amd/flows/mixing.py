"""Mixing layers: Permute, Invertible1x1Conv, LULinearPermute.

Mirrors normflows/flows/mixing.py:9-54 (Permute), :57-133 (Invertible1x1Conv), :213-254 (_Permutation,
_RandomPermutation), :368-532 (_LULinear), :535-563 (LULinearPermute): constructor signatures, RNG
consumption order at construction (so a seeded build reproduces the reference's permutations / QR
initialisation bit for bit) and state_dict keys.  The arithmetic runs in nf_lu_linear_permute,
nf_inv1x1_assemble and nf_inv1x1_conv.
"""
import numpy as np
import torch
from .. import _keys
from torch import nn
from torch.nn import init

from .. import _prepack
from .. import ops
from ..autograd import LULinearPermuteFn, needs_grad
from .base import Flow


class Permute(Flow):
    """Channel permutation, mode 'shuffle' (random, fixed) or 'swap' (halves) (mixing.py:9-54).
    Pure data movement (torch indexing); log-det is zero."""

    def __init__(self, num_channels, mode="shuffle"):
        super().__init__()
        self.mode = mode
        self.num_channels = num_channels
        if self.mode == "shuffle":
            perm = torch.randperm(self.num_channels)
            inv_perm = torch.empty_like(perm).scatter_(dim=0, index=perm, src=torch.arange(self.num_channels))
            self.register_buffer("perm", perm)
            self.register_buffer("inv_perm", inv_perm)

    def _move(self, z, inverse):
        if self.mode == "shuffle":
            return z[:, self.inv_perm if inverse else self.perm, ...]
        if self.mode == "swap":
            cut = (self.num_channels + 1) // 2 if inverse else self.num_channels // 2
            return torch.cat([z[:, cut:, ...], z[:, :cut, ...]], dim=1)
        raise NotImplementedError("The mode " + self.mode + " is not implemented.")

    def forward(self, z, context=None):
        return self._move(z, False), torch.zeros(len(z), device=z.device)

    def inverse(self, z, context=None):
        return self._move(z, True), torch.zeros(len(z), device=z.device)

    def _run(self, z, inverse, ld, acc, **kw):
        return self._move(z, inverse)


def prefetch_weights(convs):
    """Round 6 (late): the density-direction matrices of several LU-parametrised Invertible1x1Convs under autograd in ONE launch per
    size (autograd.Inv1x1WeightsFn: nf_inv1x1_assemble_multi forward, nf_inv1x1_lu_grads_multi backward) -- they depend on parameters
    only, so a Glow level assembles its K matrices before its first block runs.  Each layer's (W, log|det| per pixel) waits in
    `_w_prefetch` for that layer's next `_weight_train(True)`; the caller drops what was not consumed (clear_prefetched)."""
    from .. import config
    if not config.glow_weights_batched or not torch.is_grad_enabled():
        return
    groups = {}
    for c in convs:
        if (c.use_lu and c.L.is_cuda and c.L.dtype == torch.float32 and c.num_channels <= 64
                and (c.L.requires_grad or c.U.requires_grad or c.log_S.requires_grad)):
            groups.setdefault((c.num_channels, c.L.device), []).append(c)
    from ..autograd import Inv1x1WeightsFn
    for group in groups.values():
        if len(group) < 2:
            continue
        flat = [t for c in group for t in (c.P, c.L, c.U, c.sign_S, c.log_S)]
        out = Inv1x1WeightsFn.apply(len(group), *flat)
        for i, c in enumerate(group):
            c.__dict__["_w_prefetch"] = (out[2 * i], out[2 * i + 1])


def clear_prefetched(convs):
    for c in convs:
        c.__dict__.pop("_w_prefetch", None)


class Invertible1x1Conv(Flow):
    """Glow's invertible 1x1 convolution on NCHW tensors (mixing.py:57-133).

    use_lu=True: W = P L U with fixed P, learnable L, U, log_S; assembled on the device by
    nf_inv1x1_assemble (triangular inverses in fp64 for the sampling direction) and applied by
    nf_inv1x1_conv.  use_lu=False keeps the raw W parameter; its inverse / slogdet are taken with
    torch.linalg (library call, non-default parametrisation) and only the conv runs in our kernel.
    """

    def __init__(self, num_channels, use_lu=False):
        super().__init__()
        self.num_channels = num_channels
        self.use_lu = use_lu
        Q, _ = torch.linalg.qr(torch.randn(self.num_channels, self.num_channels))
        if use_lu:
            P, L, U = torch.lu_unpack(*Q.lu())
            self.register_buffer("P", P)
            self.L = nn.Parameter(L)
            S = U.diag()
            self.register_buffer("sign_S", torch.sign(S))
            self.log_S = nn.Parameter(torch.log(torch.abs(S)))
            self.U = nn.Parameter(torch.triu(U, diagonal=1))
            self.register_buffer("eye", torch.diag(torch.ones(self.num_channels)))
        else:
            self.W = nn.Parameter(Q)

    def _weight(self, inverse_dir):
        """W and the per-pixel log|det| (0-dim) for flow.inverse (inverse_dir=True) or flow.forward.  The assembled
        matrix is kept until a parameter changes: the reference re-assembles (and re-inverts) it on every call."""
        if self.use_lu:
            key = (inverse_dir,) + _keys.pkey((self.L, self.U, self.log_S, self.P))
            cache = getattr(self, "_w_cache", None)
            if cache is None or cache[0] != key:
                cache = (key, ops.inv1x1_assemble(self.P, self.L.detach(), self.U.detach(), self.sign_S,
                                                  self.log_S.detach(), inverse=not inverse_dir))
                self._w_cache = cache
            return cache[1]
        W = self.W.detach()
        sld = torch.slogdet(W)[1]
        if inverse_dir:
            return W.contiguous(), sld
        Winv = torch.inverse(W) if W.dtype == torch.float64 else torch.inverse(W.double()).type(W.dtype)
        return Winv.contiguous(), -sld

    def _weight_train(self, inverse_dir):
        """Differentiable assembly of W (mixing.py:88-104) with torch ops on the C x C matrices."""
        if not self.use_lu:
            W = self.W
            sld = torch.slogdet(W)[1]
            if inverse_dir:
                return W, sld
            Winv = torch.inverse(W) if W.dtype == torch.float64 else torch.inverse(W.double()).type(W.dtype)
            return Winv, -sld
        if inverse_dir and self.L.is_cuda and self.num_channels <= 64:
            pre = self.__dict__.pop("_w_prefetch", None)
            if pre is not None:                          # (assembled with the level's other layers: prefetch_weights below)
                return pre
            from ..autograd import Inv1x1WeightFn      # density direction: assembly and its VJP as one launch each
            return Inv1x1WeightFn.apply(self.P, self.L, self.U, self.sign_S, self.log_S)
        Lm = torch.tril(self.L, diagonal=-1) + self.eye
        Um = torch.triu(self.U, diagonal=1) + torch.diag(self.sign_S * torch.exp(self.log_S))
        if inverse_dir:
            return self.P @ Lm @ Um, torch.sum(self.log_S)
        if self.log_S.dtype == torch.float64:
            Li, Ui = torch.inverse(Lm), torch.inverse(Um)
        else:
            Li, Ui = torch.inverse(Lm.double()).type(self.log_S.dtype), torch.inverse(Um.double()).type(self.log_S.dtype)
        return Ui @ Li @ self.P.t(), -torch.sum(self.log_S)

    def _conv(self, z, inverse_dir, ld=None, acc=None, want_scalar=True):
        from ..autograd import Inv1x1Fn, needs_grad
        if needs_grad(z, self):
            W, ldu = self._weight_train(inverse_dir)
            y, log_det = Inv1x1Fn.apply(z.contiguous(), W, ldu)
            if ld is None:
                return y, log_det
            from .affine import _fold_ld
            _fold_ld(ld, acc, log_det)
            return y, ld
        W, ldu = self._weight(inverse_dir)
        return ops.inv1x1_conv(z, W, ldu, logdet=ld, acc=acc, want_scalar=want_scalar)

    def forward(self, z):
        return self._conv(z, False)

    def inverse(self, z):
        return self._conv(z, True)

    def _run(self, z, inverse, ld, acc, **kw):
        y, _ = self._conv(z, inverse, ld=ld, acc=acc, want_scalar=False)
        return y


class InvertibleAffine(Flow):
    """One-dimensional version of the invertible 1x1 convolution, z_ = z @ W (mixing.py:136-207).  Same parameters as
    Invertible1x1Conv; `z @ W` is the per-sample product with W^T, so the HIP mat-vec kernel runs on (B, C, 1, 1)
    with the transposed assembled matrix (kept until a parameter changes)."""

    def __init__(self, num_channels, use_lu=True):
        super().__init__()
        self.num_channels = num_channels
        self.use_lu = use_lu
        Q, _ = torch.linalg.qr(torch.randn(self.num_channels, self.num_channels))
        if use_lu:
            P, L, U = torch.lu_unpack(*Q.lu())
            self.register_buffer("P", P)
            self.L = nn.Parameter(L)
            S = U.diag()
            self.register_buffer("sign_S", torch.sign(S))
            self.log_S = nn.Parameter(torch.log(torch.abs(S)))
            self.U = nn.Parameter(torch.triu(U, diagonal=1))
            self.register_buffer("eye", torch.diag(torch.ones(self.num_channels)))
        else:
            self.W = nn.Parameter(Q)

    def _weight_t(self, inverse_dir):
        params = (self.L, self.U, self.log_S, self.P) if self.use_lu else (self.W,)
        key = (inverse_dir,) + _keys.pkey(params)
        cache = getattr(self, "_w_cache", None)
        if cache is None or cache[0] != key:
            if self.use_lu:
                W, ldu = ops.inv1x1_assemble(self.P, self.L.detach(), self.U.detach(), self.sign_S, self.log_S.detach(),
                                             inverse=not inverse_dir)
            else:
                W = self.W.detach()
                ldu = torch.slogdet(W)[1]
                if not inverse_dir:
                    W = torch.inverse(W) if W.dtype == torch.float64 else torch.inverse(W.double()).type(W.dtype)
                    ldu = -ldu
            cache = (key, (W.t().contiguous(), ldu))
            self._w_cache = cache
        return cache[1]

    def _torch(self, z, inverse_dir):
        """Differentiable path (mixing.py:165-207 as torch ops), taken only when a gradient is asked for: the matrix is assembled
        from the parameters themselves, so L, U, log_S (or W) receive gradients.  Not a HIP path: the layer is not on the
        benchmark's hot path; the inference kernels stay the default."""
        if self.use_lu:
            lower = torch.tril(self.L, -1) + self.eye
            upper = torch.triu(self.U, 1) + torch.diag(self.sign_S * torch.exp(self.log_S))
            logabs = self.log_S.sum()
            if inverse_dir:                       # Flow.inverse: z @ (P L U)
                return z @ (self.P @ lower @ upper), logabs
            # Flow.forward: z @ (P L U)^-1 = ((z @ U^-1) @ L^-1) @ P^T, by two triangular solves from the right
            t = torch.linalg.solve_triangular(upper, z, upper=True, left=False)
            t = torch.linalg.solve_triangular(lower, t, upper=False, left=False, unitriangular=True)
            return t @ self.P.t(), -logabs
        logabs = torch.linalg.slogdet(self.W)[1]
        if inverse_dir:
            return z @ self.W, logabs
        return torch.linalg.solve(self.W, z, left=False), -logabs

    def _mul(self, z, inverse_dir, ld=None, acc=None, want_scalar=True):
        if z.dim() != 2:
            raise NotImplementedError("InvertibleAffine: (batch, channels) inputs")
        if needs_grad(z, self):
            y, l = self._torch(z, inverse_dir)
            if ld is not None:
                ld.add_(l, alpha=float(acc))
            return y, l
        Wt, ldu = self._weight_t(inverse_dir)
        y, lds = ops.inv1x1_conv(z.reshape(z.shape[0], -1, 1, 1), Wt, ldu, logdet=ld, acc=acc, want_scalar=want_scalar)
        return y.view(z.shape), lds

    def forward(self, z, context=None):
        return self._mul(z, False)

    def inverse(self, z, context=None):
        return self._mul(z, True)

    def _run(self, z, inverse, ld, acc, **kw):
        y, _ = self._mul(z, inverse, ld=ld, acc=acc, want_scalar=False)
        return y


# ---- LU linear + permutation used by neural spline flows ---------------------------------------------------
class _Permutation(Flow):
    """Holds a fixed permutation of dimension `dim` (mixing.py:213-247)."""

    def __init__(self, permutation, dim=1):
        if permutation.ndimension() != 1:
            raise ValueError("Permutation must be a 1D tensor.")
        super().__init__()
        self._dim = dim
        self.register_buffer("_permutation", permutation)

    @property
    def _inverse_permutation(self):
        return torch.argsort(self._permutation)

    @staticmethod
    def _permute(inputs, permutation, dim):
        if dim >= inputs.ndimension():
            raise ValueError("No dimension {} in inputs.".format(dim))
        if inputs.shape[dim] != len(permutation):
            raise ValueError("Dimension {} in inputs must be of size {}.".format(dim, len(permutation)))
        outputs = torch.index_select(inputs, dim, permutation)
        return outputs, torch.zeros(inputs.shape[0], dtype=inputs.dtype, device=inputs.device)

    def forward(self, inputs, context=None):
        return self._permute(inputs, self._permutation, self._dim)

    def inverse(self, inputs, context=None):
        return self._permute(inputs, self._inverse_permutation, self._dim)


class _RandomPermutation(_Permutation):
    def __init__(self, features, dim=1):
        super().__init__(torch.randperm(features), dim)


class _LULinear(Flow):
    """Parameter holder of the LU-parametrised linear map (mixing.py:274-365, :368-532): packed strictly-lower and
    strictly-upper entries, softplus-constrained diagonal, bias."""

    def __init__(self, features, using_cache=False, identity_init=True, eps=1e-3):
        super().__init__()
        self.features = features
        self.bias = nn.Parameter(torch.zeros(features))
        self.using_cache = using_cache
        self.eps = eps
        n_triangular_entries = ((features - 1) * features) // 2
        self.lower_entries = nn.Parameter(torch.zeros(n_triangular_entries))
        self.upper_entries = nn.Parameter(torch.zeros(n_triangular_entries))
        self.unconstrained_upper_diag = nn.Parameter(torch.zeros(features))
        self._initialize(identity_init)

    def _initialize(self, identity_init):
        init.zeros_(self.bias)
        if identity_init:
            init.zeros_(self.lower_entries)
            init.zeros_(self.upper_entries)
            init.constant_(self.unconstrained_upper_diag, np.log(np.exp(1 - self.eps) - 1))
        else:
            stdv = 1.0 / np.sqrt(self.features)
            init.uniform_(self.lower_entries, -stdv, stdv)
            init.uniform_(self.upper_entries, -stdv, stdv)
            init.uniform_(self.unconstrained_upper_diag, -stdv, stdv)

    def use_cache(self, mode=True):
        # The kernel re-assembles L and U in LDS on every launch (8 KB of parameters); nothing to cache.
        self.using_cache = mode


class LULinearPermute(Flow):
    """Fixed random permutation followed by an LU-parametrised linear map (mixing.py:535-563)."""

    def __init__(self, num_channels, identity_init=True):
        super().__init__()
        self.permutation = _RandomPermutation(num_channels)
        self.linear = _LULinear(num_channels, identity_init=identity_init)
        self.use_dense = True    # False: the LDS-tile kernel nf_lu_linear_permute also for D <= 64 float32 (tests / ablation)

    def _apply_kernel(self, z, inverse, ld=None, acc=None):
        if z.dim() != 2:
            raise ValueError("LULinearPermute expects (batch, features) inputs.")
        if z.shape[1] != self.linear.features:
            raise ValueError("Dimension 1 in inputs must be of size {}.".format(self.linear.features))
        lin = self.linear
        if needs_grad(z, self):   # training path (autograd.py): same forward kernel, GEMM-based backward
            fout = self._factors_buffer(z.device) if (inverse and _prepack.take(self)) else None
            y, log_det = LULinearPermuteFn.apply(z, self.permutation._permutation, lin.lower_entries, lin.upper_entries,
                                                 lin.unconstrained_upper_diag, lin.bias, lin.eps, 0 if inverse else 1,
                                                 ld, 1 if (acc is None or acc > 0) else -1, fout,
                                                 self.__dict__.setdefault("_img_holder", {}) if fout is not None else None)
            return y, log_det            # log_det IS ld (updated in place) when the caller passed its accumulator
        if z.dtype == torch.float32 and z.is_cuda and 64 < lin.features <= 128 and self.use_dense:
            # wider layers (round 3): the same ONE dense product on fp32 MFMA (nf_rows_matvec_affine takes D <= 128); the
            # D x D matrices are composed in float64 by a handful of torch launches once per parameter version (the one-workgroup
            # composer keeps four D x D fp64 matrices in LDS: 64 x 64 at most)
            params = (lin.lower_entries, lin.upper_entries, lin.unconstrained_upper_diag, lin.bias)
            key = _keys.pkey(params)
            cache = self.__dict__.get("_dense_cache")
            if cache is None or cache[0] != key:
                cache = self._dense_cache = (key, self._compose_wide())
            Wd, Ws, bd, bs, lad = cache[1]
            if inverse:
                return ops.rows_matvec_affine(z, Wd, bd, lad, +1.0, logdet=ld, acc=acc)
            return ops.rows_matvec_affine(z, Ws, bs, lad, -1.0, logdet=ld, acc=acc)
        if z.dtype == torch.float32 and z.is_cuda and lin.features <= 64 and self.use_dense:
            # the layer as ONE dense D x D product on fp32 MFMA (nf_lu_compose once per parameter version + nf_rows_matvec_affine):
            # HBM-bound, 5x the LDS-tile kernel below, which stays for D > 64 and float64
            params = (lin.lower_entries, lin.upper_entries, lin.unconstrained_upper_diag, lin.bias)
            key = _keys.pkey(params)
            cache = self.__dict__.get("_dense_cache")
            if cache is None or cache[0] != key:
                cache = self._dense_cache = (key, ops.lu_compose(self.permutation._permutation, *[p_.detach() for p_ in params],
                                                                 eps=lin.eps))
            Wd, Ws, bd, bs, lad = cache[1]
            if inverse:
                return ops.rows_matvec_affine(z, Wd, bd, lad, +1.0, logdet=ld, acc=acc)
            return ops.rows_matvec_affine(z, Ws, bs, lad, -1.0, logdet=ld, acc=acc)
        # flow.inverse (density) = kernel direction 0, flow.forward (sample) = kernel direction 1
        return ops.lu_linear_permute(z, self.permutation._permutation, lin.lower_entries.detach(),
                                     lin.upper_entries.detach(), lin.unconstrained_upper_diag.detach(),
                                     lin.bias.detach(), 0 if inverse else 1, eps=lin.eps, logdet=ld, acc=acc)

    def _dense_matrices(self):
        """(Wd, Ws, bias_d, bias_s, log|det|) on the device, cached per parameter version: density y = Wd x + bias_d, sampling
        y = Ws x + bias_s (the dense form of mixing.py:402-473, :535-563 used by nf_rows_matvec_affine and by the fused pair kernels)."""
        lin = self.linear
        params = (lin.lower_entries, lin.upper_entries, lin.unconstrained_upper_diag, lin.bias)
        key = _keys.pkey(params)
        cache = self.__dict__.get("_dense_cache")
        if cache is None or cache[0] != key:
            if lin.features <= 64:
                val = ops.lu_compose(self.permutation._permutation, *[p_.detach() for p_ in params], eps=lin.eps)
            else:
                val = self._compose_wide()
            cache = self._dense_cache = (key, val)
        return cache[1]

    def _compose_wide(self):
        """(Wd, Ws, bias_d, bias_s, log|det|) of mixing.py:402-473, :535-563 as dense matrices, float64 arithmetic on the device:
        density y = L (U x[perm]) + b -> Wd[:, perm] = L U; sample y[perm] = U^-1 L^-1 (x - b) -> Ws[perm] = U^-1 L^-1, bias_s = -Ws b."""
        lin, perm = self.linear, self.permutation._permutation
        D = lin.features
        dev = lin.bias.device
        with torch.no_grad():
            Lm = torch.eye(D, dtype=torch.float64, device=dev)
            Um = torch.zeros(D, D, dtype=torch.float64, device=dev)
            li, ui = torch.tril_indices(D, D, -1, device=dev), torch.triu_indices(D, D, 1, device=dev)
            Lm[li[0], li[1]] = lin.lower_entries.detach().double()
            Um[ui[0], ui[1]] = lin.upper_entries.detach().double()
            diag = (torch.nn.functional.softplus(lin.unconstrained_upper_diag.detach()) + lin.eps).double()
            Um[torch.arange(D, device=dev), torch.arange(D, device=dev)] = diag
            Wd = torch.zeros(D, D, dtype=torch.float64, device=dev)
            Wd[:, perm] = Lm @ Um
            eye = torch.eye(D, dtype=torch.float64, device=dev)
            inv = torch.linalg.solve_triangular(Um, torch.linalg.solve_triangular(Lm, eye, upper=False, unitriangular=True), upper=True)
            Ws = torch.zeros(D, D, dtype=torch.float64, device=dev)
            Ws[perm] = inv
            b = lin.bias.detach().double()
            lad = torch.log(torch.nn.functional.softplus(lin.unconstrained_upper_diag.detach()) + lin.eps).sum().reshape(1)
            return (Wd.float().contiguous(), Ws.float().contiguous(), b.float().contiguous(), (-(Ws @ b)).float().contiguous(),
                    lad.float().contiguous())

    def _train_factors_ok(self, z):
        """The density direction under autograd assembles its factors with nf_lu_factors (LULinearPermuteFn.forward)."""
        return z.is_cuda and z.dtype == torch.float32 and z.dim() == 2 and 2 <= self.linear.features <= 64 \
            and z.shape[1] == self.linear.features

    def _factors_buffer(self, device):
        """(5 D^2 + D + 1) floats owned by the layer: the factor images a multi-layer launch (_prepack.py) writes."""
        buf = self.__dict__.get("_lu_fbuf")
        D = self.linear.features
        if buf is None or buf.device != device:
            buf = self.__dict__["_lu_fbuf"] = torch.empty(5 * D * D + D + 1, dtype=torch.float32, device=device)
        return buf

    def _wd_buffer(self, device):
        """(D, D) floats owned by the layer: the composed density-direction matrix W_d the per-step pair pack (_prepack.py,
        nf_lu_pack_train_multi) leaves for the backward (nf_lu_bwd_composed)."""
        buf = self.__dict__.get("_lu_wd")
        D = self.linear.features
        if buf is None or buf.device != device:
            buf = self.__dict__["_lu_wd"] = torch.empty(D, D, dtype=torch.float32, device=device)
        return buf

    def forward(self, z, context=None):
        return self._apply_kernel(z, False)

    def inverse(self, z, context=None):
        return self._apply_kernel(z, True)

    def _run(self, z, inverse, ld, acc, **kw):
        y, _ = self._apply_kernel(z, inverse, ld=ld, acc=acc)
        return y
