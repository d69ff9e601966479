"""Split / Merge / Squeeze (normflows/flows/reshape.py:9-128).

Split and Merge exist for API parity (lists of tensors in/out); on the hot path the channel split/merge is
folded into the affine-coupling kernel (nf_affine_coupling reads z and writes the merged y directly).
Squeeze is one gather kernel (nf_squeeze).
"""
import torch

from .. import ops
from .base import Flow


def _checkerboard(shape, inverted, device):
    """Boolean checkerboard over the non-batch dims of `shape`: True where the coordinate sum is odd (reshape.py:35-46
    builds the same colouring by nested alternation); `inverted` flips it."""
    parity = torch.zeros(shape[1:], dtype=torch.long, device=device)
    for axis, n in enumerate(shape[1:]):
        view = [1] * (len(shape) - 1)
        view[axis] = n
        parity = parity + torch.arange(n, device=device).view(view)
    odd = (parity % 2 == 1)
    return (~odd if inverted else odd).expand(shape)


_CHANNEL_ORDER = {"channel": (0, 1), "channel_inv": (1, 0)}   # which chunk is z1 / z2


class Split(Flow):
    """Split features into two sets (reshape.py:9-85); modes channel, channel_inv, checkerboard[_inv].  The checkerboard
    halves keep the row-major order of their elements and halve the last dim."""

    def __init__(self, mode="channel"):
        super().__init__()
        self.mode = mode

    def _check_mode(self):
        if self.mode not in _CHANNEL_ORDER and "checkerboard" not in self.mode:
            raise NotImplementedError("Mode " + self.mode + " is not implemented.")

    def forward(self, z):
        self._check_mode()
        if self.mode in _CHANNEL_ORDER:
            halves = z.chunk(2, dim=1)
            i1, i2 = _CHANNEL_ORDER[self.mode]
            return [halves[i1], halves[i2]], 0
        pick = _checkerboard(tuple(z.shape), "inv" in self.mode, z.device)
        half_shape = tuple(z.shape[:-1]) + (-1,)
        return [z[pick].view(half_shape), z[~pick].view(half_shape)], 0

    def inverse(self, z):
        self._check_mode()
        z1, z2 = z
        if self.mode in _CHANNEL_ORDER:
            parts = [None, None]
            i1, i2 = _CHANNEL_ORDER[self.mode]
            parts[i1], parts[i2] = z1, z2
            return torch.cat(parts, 1), 0
        shape = tuple(z1.shape[:-1]) + (2 * z1.shape[-1],)
        pick = _checkerboard(shape, "inv" in self.mode, z1.device)
        out = torch.empty(shape, dtype=z1.dtype, device=z1.device)
        out[pick] = z1.reshape(-1)
        out[~pick] = z2.reshape(-1)
        return out, 0

    def _run(self, z, inverse, ld, acc, **kw):
        z, _ = self.inverse(z) if inverse else self.forward(z)
        return z


class Merge(Split):
    """Split with forward and inverse interchanged (reshape.py:88-100)."""

    def __init__(self, mode="channel"):
        super().__init__(mode)

    def forward(self, z):
        return Split.inverse(self, z)

    def inverse(self, z):
        return Split.forward(self, z)


class Squeeze(Flow):
    """Squeeze of the multi-scale architecture (reshape.py:103-128)."""

    def __init__(self):
        super().__init__()

    @staticmethod
    def _sq(z, direction):
        from ..autograd import SqueezeFn, needs_grad
        if needs_grad(z):
            return SqueezeFn.apply(z.contiguous(), direction)
        return ops.squeeze(z, direction)

    def forward(self, z):
        return self._sq(z, 0), 0

    def inverse(self, z):
        return self._sq(z, 1), 0

    def _run(self, z, inverse, ld, acc, **kw):
        return self._sq(z, 1 if inverse else 0)
