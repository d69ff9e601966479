"""ActNorm (normflows/flows/normalization.py:7-39): AffineConstFlow with data-dependent initialisation.

First call initialises (s, t) from the batch: per-channel mean and unbiased std are reduced by
nf_actnorm_stats and turned into parameters by nf_actnorm_init (forward-first: s = -log(std+1e-6),
t = -mean e^s; inverse-first: s = log(std+1e-6), t = mean).  The `data_dep_init_done` buffer keeps the
reference's name/dtype (state_dict compatible); its value is mirrored in a host flag so that steady-state
calls do not read the device buffer (the reference syncs on it every call, normalization.py:21, :33).
"""
import torch

from .. import ops
from .affine import AffineConstFlow
from .base import Flow


class ActNorm(AffineConstFlow):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.data_dep_init_done_cpu = torch.tensor(0.0)
        self.register_buffer("data_dep_init_done", self.data_dep_init_done_cpu)
        self._init_known = None  # host mirror of data_dep_init_done (None = unknown, read the buffer once)

    def _load_from_state_dict(self, *args, **kwargs):
        self._init_known = None
        return super()._load_from_state_dict(*args, **kwargs)

    def _maybe_init(self, z, inverse):
        if self._init_known is None:
            self._init_known = bool(self.data_dep_init_done.item() > 0.0)
        if self._init_known:
            return
        assert self.s is not None and self.t is not None
        if not (self._per_channel or self._elementwise):
            # parameters that broadcast over inner dimensions (not a shape the reference's models use): the statistics over the
            # broadcast dimensions as plain torch reductions, once (normalization.py:21-39; local batch only)
            with torch.no_grad():
                std = z.std(dim=self.batch_dims, keepdim=True) + 1e-6
                mean = z.mean(dim=self.batch_dims, keepdim=True)
                if inverse:
                    self.s.data.copy_(torch.log(std))
                    self.t.data.copy_(mean)
                else:
                    self.s.data.copy_(-torch.log(std))
                    self.t.data.copy_(-mean / std)
            self.data_dep_init_done.fill_(1.0)
            self._init_known = True
            return
        zz = self._geometry(z)
        mean, std = ops.actnorm_stats(zz)
        from .. import dp
        mean, std = dp.combine_moments(mean, std, zz.numel() // zz.shape[1])   # identity unless data parallel
        ops.actnorm_init(mean, std, self.s.data.view(-1), self.t.data.view(-1), 1 if inverse else 0)
        self.data_dep_init_done.fill_(1.0)
        self._init_known = True

    def forward(self, z):
        self._maybe_init(z, False)
        return super().forward(z)

    def inverse(self, z):
        self._maybe_init(z, True)
        return super().inverse(z)

    def _run(self, z, inverse, ld, acc, **kw):
        self._maybe_init(z, inverse)
        return super()._run(z, inverse, ld, acc)


class BatchNorm(Flow):
    """Batch normalisation flow without gradients through the batch statistics (normalization.py:42-62; forward only,
    like the reference).  Statistics over dim 0 come from nf_actnorm_stats (mean, unbiased std per element), the
    normalisation and its log-det are nf_actnorm with s = -log sqrt(std^2 + eps), t = -mean e^s."""

    def __init__(self, eps=1.0e-10):
        super().__init__()
        self.eps_cpu = torch.tensor(eps)
        self.register_buffer("eps", self.eps_cpu)

    def forward(self, z):
        zz = z.reshape(z.shape[0], -1)
        mean, std = ops.actnorm_stats(zz)
        s = -0.5 * torch.log(std ** 2 + self.eps.to(z.dtype))
        t = -mean * torch.exp(s)
        ld = torch.zeros(z.shape[0], dtype=z.dtype, device=z.device)
        y, _ = ops.actnorm(zz, s, t, 0, logdet=ld, acc=1, want_scalar=False)
        return y.view(z.shape), ld
