"""Wrapping and shifting of periodic coordinates (normflows/flows/periodic.py:6-73).  Volume preserving (log-det 0)
index arithmetic that only accompanies the circular spline layers, kept as tensor ops: both flows are the map
x -> ((x + offset + bound) mod 2 bound) - bound on the selected coordinates, with offset 0 (wrap) or +-shift."""
import torch

from .base import Flow


class _PeriodicMap(Flow):
    def __init__(self, ind, bound):
        super().__init__()
        self.ind = ind
        self._keep("bound", bound)

    def _keep(self, name, value):
        """Tensors become buffers (they follow .to()), plain numbers stay attributes -- as in the reference."""
        if torch.is_tensor(value):
            self.register_buffer(name, value)
        else:
            setattr(self, name, value)

    def _fold(self, z, offset):
        out = z.clone()
        sel = out[..., self.ind]
        out[..., self.ind] = torch.remainder(sel + offset + self.bound, 2 * self.bound) - self.bound
        return out, z.new_zeros(len(z))


class PeriodicWrap(_PeriodicMap):
    """Map periodic coordinates into [-bound, bound] (periodic.py:6-32): identity in the generative direction."""

    def __init__(self, ind, bound=1.0):
        super().__init__(ind, bound)

    def forward(self, z):
        return z, z.new_zeros(len(z))

    def inverse(self, z):
        return self._fold(z, 0.0)


class PeriodicShift(_PeriodicMap):
    """Shift periodic coordinates and wrap them back into [-bound, bound] (periodic.py:35-73)."""

    def __init__(self, ind, bound=1.0, shift=0.0):
        super().__init__(ind, bound)
        self._keep("shift", shift)

    def forward(self, z):
        return self._fold(z, self.shift)

    def inverse(self, z):
        return self._fold(z, -self.shift)
