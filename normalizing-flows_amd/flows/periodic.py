"""Periodic wrap / shift of circular coordinates (normflows/flows/periodic.py:6-73): volume preserving index
arithmetic (log-det 0), kept as tensor ops; they only accompany the circular spline layers."""
import torch

from .base import Flow


class PeriodicWrap(Flow):
    """Map periodic coordinates to [-bound, bound] (periodic.py:6-32)."""

    def __init__(self, ind, bound=1.0):
        super().__init__()
        self.ind = ind
        if torch.is_tensor(bound):
            self.register_buffer("bound", bound)
        else:
            self.bound = bound

    def forward(self, z):
        return z, torch.zeros(len(z), dtype=z.dtype, device=z.device)

    def inverse(self, z):
        z_ = z.clone()
        z_[..., self.ind] = torch.remainder(z_[..., self.ind] + self.bound, 2 * self.bound) - self.bound
        return z_, torch.zeros(len(z), dtype=z.dtype, device=z.device)


class PeriodicShift(Flow):
    """Shift and wrap periodic coordinates (periodic.py:35-73)."""

    def __init__(self, ind, bound=1.0, shift=0.0):
        super().__init__()
        self.ind = ind
        if torch.is_tensor(bound):
            self.register_buffer("bound", bound)
        else:
            self.bound = bound
        if torch.is_tensor(shift):
            self.register_buffer("shift", shift)
        else:
            self.shift = shift

    def forward(self, z):
        z_ = z.clone()
        z_[..., self.ind] = torch.remainder(z_[..., self.ind] + self.shift + self.bound, 2 * self.bound) - self.bound
        return z_, torch.zeros(len(z), dtype=z.dtype, device=z.device)

    def inverse(self, z):
        z_ = z.clone()
        z_[..., self.ind] = torch.remainder(z_[..., self.ind] - self.shift + self.bound, 2 * self.bound) - self.bound
        return z_, torch.zeros(len(z), dtype=z.dtype, device=z.device)
