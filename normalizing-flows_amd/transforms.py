"""Data-side transforms used as `transform=` of MultiscaleFlow / NormalizingFlow (normflows/transforms.py:8-75)."""
import torch

from . import ops
from .flows.base import Flow


class Logit(Flow):
    """Logit mapping of image tensors, logit(alpha + (1 - 2 alpha) x) (transforms.py:8-47); one HIP launch per call
    (element-wise map + per-sample log-det reduction, nf_logit)."""

    def __init__(self, alpha=0.05):
        super().__init__()
        self.alpha = alpha

    def forward(self, z):
        return ops.logit(z, self.alpha, 0)

    def inverse(self, z):
        return ops.logit(z, self.alpha, 1)

    def _run(self, z, inverse, ld, acc):
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
