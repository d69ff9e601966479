"""Base distributions at the end of the path: DiagGaussian (normflows/distributions/base.py:8-103) and
ClassCondDiagGaussian (:273-345, the base of the reference's Glow example).

log_prob is one launch of nf_diag_gaussian_log_prob (optionally accumulating into the caller's log_q);
sampling draws torch.randn on the device and evaluates the same closed form.
"""
import numpy as np
import torch
from . import _keys
from torch import nn

from . import ops
from .autograd import DiagGaussianLogProbFn, GaussianRowsLogProbFn, needs_grad


class BaseDistribution(nn.Module):
    def __init__(self):
        super().__init__()

    def forward(self, num_samples=1):
        raise NotImplementedError

    def log_prob(self, z):
        raise NotImplementedError

    def sample(self, num_samples=1, **kwargs):
        z, _ = self.forward(num_samples, **kwargs)
        return z


class DiagGaussian(BaseDistribution):
    """Multivariate Gaussian with diagonal covariance (base.py:52-103)."""

    def __init__(self, shape, trainable=True):
        super().__init__()
        if isinstance(shape, int):
            shape = (shape,)
        if isinstance(shape, list):
            shape = tuple(shape)
        self.shape = shape
        self.n_dim = len(shape)
        self.d = np.prod(shape)
        if trainable:
            self.loc = nn.Parameter(torch.zeros(1, *self.shape))
            self.log_scale = nn.Parameter(torch.zeros(1, *self.shape))
        else:
            self.register_buffer("loc", torch.zeros(1, *self.shape))
            self.register_buffer("log_scale", torch.zeros(1, *self.shape))
        self.temperature = None

    def _shift(self):
        return 0.0 if self.temperature is None else float(np.log(self.temperature))

    def forward(self, num_samples=1, context=None):
        eps = torch.randn((num_samples,) + self.shape, dtype=self.loc.dtype, device=self.loc.device)
        return self.from_noise(eps)

    def from_noise(self, eps):
        """z = loc + exp(log_scale) eps and log p(z) (base.py:80-92) for given standard-normal noise."""
        if needs_grad(eps, self.loc, self.log_scale):   # reparametrised sample: gradients reach loc / log_scale
            log_scale = self.log_scale + self._shift()
            z = self.loc + torch.exp(log_scale) * eps
            log_p = -0.5 * self.d * np.log(2 * np.pi) - torch.sum(log_scale + 0.5 * torch.pow(eps, 2),
                                                                  list(range(1, self.n_dim + 1)))
            return z, log_p
        log_scale = self.log_scale.detach() + self._shift()
        z = self.loc.detach() + torch.exp(log_scale) * eps
        # log_p = -d/2 log(2 pi) - sum(log_scale + eps^2/2): the same closed form as log_prob at (z - loc)/scale = eps
        zeros = torch.zeros_like(self.loc.detach())
        log_p = ops.diag_gaussian_log_prob(eps, zeros, zeros, 0.0)
        log_p = log_p - log_scale.sum()
        return z, log_p

    def log_prob(self, z, context=None):
        if needs_grad(z, self.loc, self.log_scale):
            return DiagGaussianLogProbFn.apply(z, self.loc, self.log_scale, self._shift())
        return ops.diag_gaussian_log_prob(z, self.loc.detach(), self.log_scale.detach(), self._shift())

    def _log_prob_acc(self, z, log_q, acc=+1):
        """log_q (+|-)= log_prob(z), fused into the kernel's store."""
        if needs_grad(z, self.loc, self.log_scale):
            lp = self.log_prob(z)
            if acc > 0:
                log_q += lp
            else:
                log_q -= lp
            return log_q
        ops.diag_gaussian_log_prob(z, self.loc.detach(), self.log_scale.detach(), self._shift(), out=log_q, acc=acc)
        return log_q


class ClassCondDiagGaussian(BaseDistribution):
    """Class-conditional diagonal Gaussian (base.py:273-345).  Parameters keep the reference layout (*shape,
    num_classes) so state_dicts interchange; the kernel reads the transposed (num_classes, d) rows, refreshed when a
    parameter changes.  y is a vector of labels (row gather inside the kernel) or a (B, num_classes) weight matrix
    (rows blended by one library GEMM, as the reference's `loc @ y`)."""

    def __init__(self, shape, num_classes):
        super().__init__()
        if isinstance(shape, int):
            shape = (shape,)
        if isinstance(shape, list):
            shape = tuple(shape)
        self.shape = shape
        self.n_dim = len(shape)
        self.perm = [self.n_dim] + list(range(self.n_dim))
        self.d = np.prod(shape)
        self.num_classes = num_classes
        self.loc = nn.Parameter(torch.zeros(*self.shape, num_classes))
        self.log_scale = nn.Parameter(torch.zeros(*self.shape, num_classes))
        self.temperature = None
        self._rows_cache = None

    def _shift(self):
        return 0.0 if self.temperature is None else float(np.log(self.temperature))

    def _rows(self):
        """(num_classes, d) copies of loc / log_scale."""
        key = _keys.pkey((self.loc, self.log_scale))
        if self._rows_cache is None or self._rows_cache[0] != key:
            d = int(self.d)
            self._rows_cache = (key, self.loc.detach().reshape(d, self.num_classes).t().contiguous(),
                                self.log_scale.detach().reshape(d, self.num_classes).t().contiguous())
        return self._rows_cache[1], self._rows_cache[2]

    def _select(self, y, num_samples):
        """Per-sample (loc, log_scale) rows and the row index the kernel should use."""
        loc_r, ls_r = self._rows()
        if y.dim() == 1:
            return loc_r, ls_r, y
        w = y.to(loc_r.dtype)
        return w @ loc_r, w @ ls_r, None

    def forward(self, num_samples=1, y=None):
        if y is not None:
            num_samples = len(y)
        else:
            y = torch.randint(self.num_classes, (num_samples,), device=self.loc.device)
        grad = needs_grad(self.loc, self.log_scale)
        if grad:                         # differentiable rows (the cached copies are detached)
            d = int(self.d)
            loc_r, ls_r = self.loc.reshape(d, self.num_classes).t(), self.log_scale.reshape(d, self.num_classes).t()
            idx = y if y.dim() == 1 else None
            if idx is None:
                loc_r, ls_r = y.to(loc_r.dtype) @ loc_r, y.to(loc_r.dtype) @ ls_r
        else:
            loc_r, ls_r, idx = self._select(y, num_samples)
        eps = torch.randn((num_samples,) + self.shape, dtype=self.loc.dtype, device=self.loc.device)
        if idx is not None:
            loc_b, ls_b = loc_r[idx], ls_r[idx]
        else:
            loc_b, ls_b = loc_r, ls_r
        ls_b = (ls_b + self._shift()).view((num_samples,) + self.shape)
        z = loc_b.view((num_samples,) + self.shape) + torch.exp(ls_b) * eps
        if grad:                         # reparametrised sample: closed form under autograd (base.py:296-316)
            log_p = -0.5 * self.d * np.log(2 * np.pi) - (ls_b + 0.5 * eps ** 2).reshape(num_samples, -1).sum(1)
            return z, log_p
        zeros = torch.zeros(1, int(self.d), dtype=z.dtype, device=z.device)
        log_p = ops.diag_gaussian_log_prob(eps, zeros, zeros, 0.0) - ls_b.reshape(num_samples, -1).sum(1)
        return z, log_p

    def log_prob(self, z, y):
        if needs_grad(z, self.loc, self.log_scale):   # training: differentiable row tables, HIP forward
            d = int(self.d)
            loc_r = self.loc.reshape(d, self.num_classes).t()
            ls_r = self.log_scale.reshape(d, self.num_classes).t()
            if y.dim() == 1:
                idx = y
            else:
                w = y.to(loc_r.dtype)
                loc_r, ls_r, idx = w @ loc_r, w @ ls_r, None
            return GaussianRowsLogProbFn.apply(z.contiguous(), loc_r.contiguous(), ls_r.contiguous(), idx, self._shift())
        loc_r, ls_r, idx = self._select(y, len(z))
        return ops.diag_gaussian_log_prob_rows(z, loc_r, ls_r, idx, self._shift())


class GlowBase(BaseDistribution):
    """Base distribution of the Glow model: one mean / log-scale per channel, scaled by exp(logs * logscale_factor),
    optionally class conditional (base.py:348-471).  log_prob expands the per-channel rows over the pixels and runs
    nf_diag_gaussian_log_prob_rows (rows picked by label inside the kernel, or blended per sample for soft labels)."""

    def __init__(self, shape, num_classes=None, logscale_factor=3.0):
        super().__init__()
        if isinstance(shape, int):
            shape = (shape,)
        if isinstance(shape, list):
            shape = tuple(shape)
        self.shape = shape
        self.n_dim = len(shape)
        self.num_pix = int(np.prod(shape[1:]))
        self.d = np.prod(shape)
        self.sum_dim = list(range(1, self.n_dim + 1))
        self.num_classes = num_classes
        self.class_cond = num_classes is not None
        self.logscale_factor = logscale_factor
        pshape = (1, self.shape[0]) + (self.n_dim - 1) * (1,)
        self.loc = nn.Parameter(torch.zeros(*pshape))
        self.loc_logs = nn.Parameter(torch.zeros(*pshape))
        self.log_scale = nn.Parameter(torch.zeros(*pshape))
        self.log_scale_logs = nn.Parameter(torch.zeros(*pshape))
        if self.class_cond:
            self.loc_cc = nn.Parameter(torch.zeros(self.num_classes, self.shape[0]))
            self.log_scale_cc = nn.Parameter(torch.zeros(self.num_classes, self.shape[0]))
        self.temperature = None

    def _channel_params(self, y, num_samples, detach=True):
        """(rows, C) mean and log-scale and the row index per sample (None: row b is sample b)."""
        C = self.shape[0]
        dt = (lambda t: t.detach()) if detach else (lambda t: t)
        loc = (dt(self.loc) * torch.exp(dt(self.loc_logs) * self.logscale_factor)).view(1, C)
        ls = (dt(self.log_scale) * torch.exp(dt(self.log_scale_logs) * self.logscale_factor)).view(1, C)
        idx = None
        if self.class_cond:
            if y.dim() == 1:
                loc, ls, idx = loc + dt(self.loc_cc), ls + dt(self.log_scale_cc), y
            else:
                w = y.to(loc.dtype)
                loc, ls = loc + w @ dt(self.loc_cc), ls + w @ dt(self.log_scale_cc)
        elif num_samples is not None:
            idx = torch.zeros(num_samples, dtype=torch.long, device=loc.device)
        if self.temperature is not None:
            ls = ls + float(np.log(self.temperature))
        return loc, ls, idx

    def _expand(self, rows):
        return rows.unsqueeze(-1).expand(rows.shape[0], rows.shape[1], self.num_pix).reshape(rows.shape[0], -1)

    def forward(self, num_samples=1, y=None):
        if self.class_cond:
            if y is not None:
                num_samples = len(y)
            else:
                y = torch.randint(self.num_classes, (num_samples,), device=self.loc.device)
        grad = needs_grad(*self.parameters())
        loc, ls, idx = self._channel_params(y, num_samples, detach=not grad)
        if idx is not None:
            loc, ls = loc[idx], ls[idx]
        pshape = (num_samples, self.shape[0]) + (self.n_dim - 1) * (1,)
        eps = torch.randn((num_samples,) + self.shape, dtype=self.loc.dtype, device=self.loc.device)
        z = loc.view(pshape) + torch.exp(ls.view(pshape)) * eps
        if grad:                        # reparametrised sample: closed form under autograd (base.py:430-448)
            log_p = (-0.5 * self.d * np.log(2 * np.pi) - self.num_pix * ls.sum(1)
                     - 0.5 * (eps ** 2).reshape(num_samples, -1).sum(1))
            return z, log_p
        zeros = torch.zeros(1, int(self.d), dtype=z.dtype, device=z.device)
        log_p = ops.diag_gaussian_log_prob(eps, zeros, zeros, 0.0) - self.num_pix * ls.sum(1)
        return z, log_p

    def log_prob(self, z, y=None):
        if needs_grad(z, *self.parameters()):
            loc, ls, idx = self._channel_params(y, len(z), detach=False)
            return GaussianRowsLogProbFn.apply(z.contiguous(), self._expand(loc).contiguous(),
                                               self._expand(ls).contiguous(), idx, 0.0)
        loc, ls, idx = self._channel_params(y, len(z))
        return ops.diag_gaussian_log_prob_rows(z, self._expand(loc), self._expand(ls), idx, 0.0)


class ConditionalDiagGaussian(BaseDistribution):
    """Diagonal Gaussian whose mean and log-scale come from a context encoder (base.py:106-155): first half of the
    encoder output = mean, second half = log std; log_prob is nf_diag_gaussian_log_prob_rows with one row per sample."""

    def __init__(self, shape, context_encoder):
        super().__init__()
        if isinstance(shape, int):
            shape = (shape,)
        if isinstance(shape, list):
            shape = tuple(shape)
        self.shape = shape
        self.n_dim = len(shape)
        self.d = np.prod(shape)
        self.context_encoder = context_encoder

    def _params(self, context):
        out = self.context_encoder(context)
        split = out.shape[-1] // 2
        return out[..., :split], out[..., split:]

    def forward(self, num_samples=1, context=None):
        mean, log_scale = self._params(context)
        eps = torch.randn((num_samples,) + self.shape, dtype=mean.dtype, device=mean.device)
        z = mean + torch.exp(log_scale) * eps
        log_p = -0.5 * self.d * np.log(2 * np.pi) - torch.sum(log_scale + 0.5 * torch.pow(eps, 2),
                                                              list(range(1, self.n_dim + 1)))
        return z, log_p

    def log_prob(self, z, context=None):
        mean, log_scale = self._params(context)
        if needs_grad(z, mean, log_scale):
            return -0.5 * self.d * np.log(2 * np.pi) - torch.sum(
                log_scale + 0.5 * torch.pow((z - mean) / torch.exp(log_scale), 2), list(range(1, self.n_dim + 1)))
        B = z.shape[0]
        return ops.diag_gaussian_log_prob_rows(z, mean.expand(B, *mean.shape[1:]).contiguous(),
                                               log_scale.expand(B, *log_scale.shape[1:]).contiguous(), None, 0.0)
