"""Affine family: AffineConstFlow, AffineCoupling, MaskedAffineFlow, AffineCouplingBlock.

Mirrors normflows/flows/affine/coupling.py:9-54, :99-171, :174-229, :232-267 (constructor signatures,
state_dict keys, forward/inverse semantics).  The arithmetic of every layer is one HIP kernel
(nf_actnorm, nf_affine_coupling, nf_masked_affine); the parameter maps (`param_map`, `s`, `t`) are arbitrary
nn.Modules evaluated by the caller-supplied network, exactly as in the reference.
"""
import numpy as np
import torch
from torch import nn

from .. import ops
from ..autograd import ActNormFn, AffineCouplingFn, MaskedAffineFn, needs_grad
from .base import Flow, run_flow
from .reshape import Merge, Split


_LAZY = []       # innermost last: [accumulator tensor, [(negate, log_det), ...]] while a lazy_ld block is open


class lazy_ld:
    """with lazy_ld(ld): the `ld += log_det` / `ld -= log_det` statements the layers inside issue on THIS accumulator under autograd
    (_fold_ld) are collected and applied when the block closes -- or at flush() -- as ONE launch in the same order (autograd.LdFoldFn,
    nf_ld_fold_multi: same bits).  Nothing inside may read `ld` before that: MultiscaleFlow._level_pass (a level's GlowBlocks only
    add to it; round 6, late: 96 launches of three microseconds per level).  A no-op for anything but a contiguous float32 CUDA
    vector, and when config.lazy_logdet is off."""

    def __init__(self, ld):
        from .. import config
        ok = (config.lazy_logdet and ld is not None and ld.is_cuda and ld.dtype == torch.float32 and ld.dim() == 1
              and ld.is_contiguous() and torch.is_grad_enabled())
        self.ent = [ld, []] if ok else None

    def __enter__(self):
        if self.ent is not None:
            _LAZY.append(self.ent)
        return self

    def flush(self):
        if self.ent is not None and self.ent[1]:
            from ..autograd import LdFoldFn
            ld, terms = self.ent
            self.ent[1] = []
            LdFoldFn.apply(ld, tuple(n for n, _ in terms), *[t for _, t in terms])

    def __exit__(self, et, ev, tb):
        if self.ent is not None:
            _LAZY.remove(self.ent)
            if et is None:
                self.flush()
        return False


def _fold_ld(ld, acc, log_det):
    """Accumulate protocol under autograd: the kernels' in-place accumulation is not differentiable."""
    if ld is None:
        return log_det
    if _LAZY and _LAZY[-1][0] is ld and log_det.shape == ld.shape and log_det.dtype == ld.dtype:
        _LAZY[-1][1].append((not (acc is None or acc > 0), log_det))
        return ld
    if acc is None or acc > 0:
        ld += log_det
    else:
        ld -= log_det
    return ld


class AffineConstFlow(Flow):
    """Per-dimension learned scale and shift (coupling.py:9-54); base class of ActNorm."""

    def __init__(self, shape, scale=True, shift=True):
        super().__init__()
        if scale:
            self.s = nn.Parameter(torch.zeros(shape)[None])
        else:
            self.register_buffer("s", torch.zeros(shape)[None])
        if shift:
            self.t = nn.Parameter(torch.zeros(shape)[None])
        else:
            self.register_buffer("t", torch.zeros(shape)[None])
        self.n_dim = self.s.dim()
        self.batch_dims = torch.nonzero(torch.tensor(self.s.shape) == 1, as_tuple=False)[:, 0].tolist()
        shp = tuple(self.s.shape[1:])
        # kernels index parameters as (i / HW) % C: parameter shapes (C,), (C,1), (C,1,1), ... are supported
        self._per_channel = len(shp) >= 1 and all(d == 1 for d in shp[1:])
        self._elementwise = len(self.batch_dims) == 1  # no broadcast dims besides the batch

    def _geometry(self, z):
        if self._per_channel:
            return z
        # every element has its own (s, t) -- or gets one by broadcasting, _operands(): C = prod(sample shape), HW = 1
        return z.reshape(z.shape[0], -1)

    def _operands(self, z, detach):
        """(z as the kernel sees it, s, t as flat vectors).  Parameter shapes that broadcast over inner dimensions other than
        trailing ones -- (1, H, W), (C, 1, W), ... (coupling.py:30-35 takes any broadcastable shape) -- are expanded to one
        parameter per element of a sample; under autograd the expansion is a view, so the gradient is summed back by torch."""
        s, t = (self.s.detach(), self.t.detach()) if detach else (self.s, self.t)
        if not (self._per_channel or self._elementwise):
            full = (1,) + tuple(z.shape[1:])
            s, t = s.expand(full), t.expand(full)
        return self._geometry(z), s.reshape(-1), t.reshape(-1)

    def _apply_kernel(self, z, inverse, ld=None, acc=None, want_scalar=True):
        if needs_grad(z, self.s, self.t):   # training: HIP forward through ActNormFn, per-sample log-det
            zz, sv, tv = self._operands(z, False)
            y, log_det = ActNormFn.apply(zz.contiguous(), sv, tv, 1 if inverse else 0)
            return y.view(z.shape), _fold_ld(ld, acc, log_det)
        zz, sv, tv = self._operands(z, True)
        y, lds = ops.actnorm(zz, sv, tv, 1 if inverse else 0, logdet=ld, acc=acc, want_scalar=want_scalar)
        return y.view(z.shape), lds

    def forward(self, z):
        return self._apply_kernel(z, False)  # 0-dim log_det like the reference (coupling.py:44)

    def inverse(self, z):
        return self._apply_kernel(z, True)

    def _run(self, z, inverse, ld, acc, **kw):
        y, _ = self._apply_kernel(z, inverse, ld=ld, acc=acc, want_scalar=False)
        return y


class CCAffineConst(Flow):
    """Affine constant flow with class-conditional parameters (coupling.py:57-97): s = s0 + y @ s_cc, t = t0 + y @ t_cc
    per sample (y is the (B, num_classes) one-hot / weight matrix the reference multiplies with); the transform and
    its log-det are one launch of nf_masked_affine with an all-zero mask (every element transformed)."""

    def __init__(self, shape, num_classes):
        super().__init__()
        if isinstance(shape, int):
            shape = (shape,)
        self.shape = shape
        self.s = nn.Parameter(torch.zeros(shape)[None])
        self.t = nn.Parameter(torch.zeros(shape)[None])
        self.s_cc = nn.Parameter(torch.zeros(num_classes, int(np.prod(shape))))
        self.t_cc = nn.Parameter(torch.zeros(num_classes, int(np.prod(shape))))
        self.n_dim = self.s.dim()
        self.batch_dims = torch.nonzero(torch.tensor(self.s.shape) == 1, as_tuple=False)[:, 0].tolist()

    def _st(self, z, y):
        y = y.to(self.s.dtype)
        s = self.s.detach() + (y @ self.s_cc.detach()).view(-1, *self.shape)
        t = self.t.detach() + (y @ self.t_cc.detach()).view(-1, *self.shape)
        return s.expand_as(z).contiguous(), t.expand_as(z).contiguous()

    def _torch(self, z, y, direction):
        """Differentiable path (affine/coupling.py:57-96 as torch ops), taken only when a gradient is asked for; the per-sample
        log-det counts every element the (broadcast) scale acts on."""
        yv = y.to(self.s.dtype)
        s = self.s + (yv @ self.s_cc).view(-1, *self.shape)
        t = self.t + (yv @ self.t_cc).view(-1, *self.shape)
        repeats = z[0].numel() // s[0].numel()
        total = s.reshape(s.shape[0], -1).sum(1) * repeats
        if direction == 0:
            return z * torch.exp(s) + t, total
        return (z - t) * torch.exp(-s), -total

    def _transform(self, z, y, direction, ld=None, acc=None):
        if needs_grad(z, self):
            out, l = self._torch(z, y, direction)
            if ld is not None:
                ld.add_(l, alpha=float(acc))
            return out, l
        s, t = self._st(z, y)
        zero = torch.zeros(z.shape[1:], dtype=z.dtype, device=z.device)
        return ops.masked_affine(z, zero, s, t, direction, logdet=ld, acc=acc)

    def forward(self, z, y):
        return self._transform(z, y, 0)

    def inverse(self, z, y):
        return self._transform(z, y, 1)


class AffineCoupling(Flow):
    """Affine coupling on a list [z1, z2] (coupling.py:99-171)."""

    def __init__(self, param_map, scale=True, scale_map="exp"):
        super().__init__()
        self.add_module("param_map", param_map)
        self.scale = scale
        self.scale_map = scale_map
        if scale and scale_map not in ("exp", "sigmoid", "sigmoid_inv"):
            raise NotImplementedError("This scale map is not implemented.")

    def _smap(self):
        return self.scale_map if self.scale else None

    def _transform(self, z, inverse, ld=None, acc=None):
        z1, z2 = z
        param = self.param_map(z1)
        if needs_grad(z1, z2, param):
            y2, log_det = AffineCouplingFn.apply(z2.contiguous(), param, 0, False, self._smap(), 1 if inverse else 0)
            return [z1, y2], _fold_ld(ld, acc, log_det)
        y2, ld = ops.affine_coupling(z2, param, 0, False, self._smap(), 1 if inverse else 0, logdet=ld, acc=acc)
        return [z1, y2], ld

    def forward(self, z):
        return self._transform(z, False)

    def inverse(self, z):
        return self._transform(z, True)

    def _run(self, z, inverse, ld, acc, **kw):
        out, _ = self._transform(z, inverse, ld=ld, acc=acc)
        return out


class MaskedAffineFlow(Flow):
    """RealNVP masked affine flow f(z) = b z + (1-b)(z exp(s(b z)) + t(b z)) (coupling.py:174-229)."""

    def __init__(self, b, t=None, s=None):
        super().__init__()
        self.b_cpu = b.view(1, *b.size())
        self.register_buffer("b", self.b_cpu)
        if s is None:
            self.s = None
        else:
            self.add_module("s", s)
        if t is None:
            self.t = None
        else:
            self.add_module("t", t)

    def _transform(self, z, inverse, ld=None, acc=None):
        need_net = self.s is not None or self.t is not None
        z_masked = self.b * z if need_net else None
        scale = self.s(z_masked) if self.s is not None else None
        trans = self.t(z_masked) if self.t is not None else None
        if needs_grad(z, scale, trans):
            y, log_det = MaskedAffineFn.apply(z.contiguous(), self.b, scale, trans, 1 if inverse else 0)
            return y, _fold_ld(ld, acc, log_det)
        return ops.masked_affine(z, self.b, scale, trans, 1 if inverse else 0, logdet=ld, acc=acc)

    def forward(self, z):
        return self._transform(z, False)

    def inverse(self, z):
        return self._transform(z, True)

    def _run(self, z, inverse, ld, acc, **kw):
        y, _ = self._transform(z, inverse, ld=ld, acc=acc)
        return y


class AffineCouplingBlock(Flow):
    """Split -> AffineCoupling -> Merge (coupling.py:232-267).

    `flows` keeps the reference's three sub-modules (state_dict keys `flows.1.param_map.*`).  For the
    channel / channel_inv split modes the three steps run as ONE kernel that reads z, copies the identity
    half and writes the transformed half into the merged output; checkerboard modes compose the three
    sub-flows like the reference.
    """

    def __init__(self, param_map, scale=True, scale_map="exp", split_mode="channel"):
        super().__init__()
        self.flows = nn.ModuleList([])
        self.flows += [Split(split_mode)]
        self.flows += [AffineCoupling(param_map, scale, scale_map)]
        self.flows += [Merge(split_mode)]
        self.split_mode = split_mode

    def _fused(self, z, inverse, ld, acc):
        C = z.shape[1]
        coupling = self.flows[1]
        if self.split_mode == "channel":
            c1, flip = (C + 1) // 2, False  # chunk(2): first ceil(C/2) channels are z1 (reshape.py:31)
            z1 = z[:, :c1]
        else:
            c1, flip = C // 2, True          # channel_inv: z2 = first ceil(C/2), z1 = the rest (reshape.py:33)
            z1 = z[:, C - c1:]
        split = coupling.param_map.forward_split(z1) if hasattr(coupling.param_map, "forward_split") else None
        if split is not None:   # inference: the last convolution's bias is added inside the coupling kernel
            return ops.affine_coupling(z, split[0], c1, flip, coupling._smap(), 1 if inverse else 0, logdet=ld, acc=acc,
                                       param_bias=split[1])
        param = coupling.param_map(z1)
        if needs_grad(z, param):
            y, log_det = AffineCouplingFn.apply(z.contiguous(), param, c1, flip, coupling._smap(), 1 if inverse else 0)
            return y, _fold_ld(ld, acc, log_det)
        return ops.affine_coupling(z, param, c1, flip, coupling._smap(), 1 if inverse else 0, logdet=ld, acc=acc)

    def _compose(self, z, inverse, ld, acc):
        seq = reversed(self.flows) if inverse else self.flows
        for f in seq:
            z = run_flow(f, z, inverse, ld, acc)
        return z

    def _go(self, z, inverse, ld, acc):
        if self.split_mode in ("channel", "channel_inv") and z.shape[1] >= 2:
            y, _ = self._fused(z, inverse, ld, acc)
            return y
        return self._compose(z, inverse, ld, acc)

    def forward(self, z):
        ld = torch.zeros(z.shape[0], dtype=z.dtype, device=z.device)
        return self._go(z, False, ld, +1), ld

    def inverse(self, z):
        ld = torch.zeros(z.shape[0], dtype=z.dtype, device=z.device)
        return self._go(z, True, ld, +1), ld

    def _run(self, z, inverse, ld, acc, **kw):
        return self._go(z, inverse, ld, acc)
