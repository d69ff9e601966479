"""Data-side transforms used as `transform=` of MultiscaleFlow / NormalizingFlow (normflows/transforms.py:8-75)."""
import numpy as np
import torch

from . import ops
from .flows.base import Flow


class Logit(Flow):
    """Logit mapping of image tensors, logit(alpha + (1 - 2 alpha) x) (transforms.py:8-47); one HIP launch per call
    (element-wise map + per-sample log-det reduction, nf_logit)."""

    def __init__(self, alpha=0.05):
        super().__init__()
        self.alpha = alpha

    def _torch(self, z, inverse):
        """transforms.py:26-47 as differentiable torch ops (only when the input carries a graph)."""
        beta, n = 1 - 2 * self.alpha, float(np.prod(z.shape[1:]))
        dims = list(range(1, z.dim()))
        if not inverse:
            ld = (-np.log(beta) * n + torch.nn.functional.logsigmoid(z).sum(dims)
                  + torch.nn.functional.logsigmoid(-z).sum(dims))
            return (torch.sigmoid(z) - self.alpha) / beta, ld
        u = self.alpha + beta * z
        logz, log1mz = torch.log(u), torch.log(1 - u)
        return logz - log1mz, np.log(beta) * n - logz.sum(dims) - log1mz.sum(dims)

    def forward(self, z):
        if torch.is_grad_enabled() and z.requires_grad:
            return self._torch(z, False)
        return ops.logit(z, self.alpha, 0)

    def inverse(self, z):
        if torch.is_grad_enabled() and z.requires_grad:
            return self._torch(z, True)
        return ops.logit(z, self.alpha, 1)

    def _run(self, z, inverse, ld, acc):
        if torch.is_grad_enabled() and z.requires_grad:
            y, l = self._torch(z, inverse)
            ld.add_(l, alpha=float(acc))
            return y
        y, _ = ops.logit(z, self.alpha, 1 if inverse else 0, logdet=ld, acc=acc)
        return y


class Shift(Flow):
    """Shift by a constant (transforms.py:50-75).  Like the reference it works IN PLACE on its input."""

    def __init__(self, shift=-0.5):
        super().__init__()
        self.shift = shift

    def forward(self, z):
        z -= self.shift
        return z, torch.zeros(z.shape[0], dtype=z.dtype, device=z.device)

    def inverse(self, z):
        z += self.shift
        return z, torch.zeros(z.shape[0], dtype=z.dtype, device=z.device)
