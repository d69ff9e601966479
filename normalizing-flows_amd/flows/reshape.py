"""Split / Merge / Squeeze (normflows/flows/reshape.py:9-128).

Split and Merge exist for API parity (lists of tensors in/out); on the hot path the channel split/merge is
folded into the affine-coupling kernel (nf_affine_coupling reads z and writes the merged y directly).
Squeeze is one gather kernel (nf_squeeze).
"""
import torch

from .. import ops
from .base import Flow


def _checkerboard(z_size, inv, device):
    """Checkerboard colouring of reshape.py:35-46 for a tensor of size z_size (batch dim included)."""
    n_dims = len(z_size)
    cb0, cb1 = 0, 1
    for i in range(1, n_dims):
        cb0_, cb1_ = cb0, cb1
        cb0 = [cb0_ if j % 2 == 0 else cb1_ for j in range(z_size[n_dims - i])]
        cb1 = [cb1_ if j % 2 == 0 else cb0_ for j in range(z_size[n_dims - i])]
    cb = cb1 if inv else cb0
    cb = torch.tensor(cb)[None].repeat(z_size[0], *((n_dims - 1) * [1]))
    return cb.to(device)


class Split(Flow):
    """Split features into two sets (reshape.py:9-85); modes channel, channel_inv, checkerboard[_inv]."""

    def __init__(self, mode="channel"):
        super().__init__()
        self.mode = mode

    def forward(self, z):
        if self.mode == "channel":
            z1, z2 = z.chunk(2, dim=1)
        elif self.mode == "channel_inv":
            z2, z1 = z.chunk(2, dim=1)
        elif "checkerboard" in self.mode:
            cb = _checkerboard(list(z.size()), "inv" in self.mode, z.device)
            z_size = z.size()
            z1 = z.reshape(-1)[torch.nonzero(cb.view(-1), as_tuple=False)].view(*z_size[:-1], -1)
            z2 = z.reshape(-1)[torch.nonzero((1 - cb).view(-1), as_tuple=False)].view(*z_size[:-1], -1)
        else:
            raise NotImplementedError("Mode " + self.mode + " is not implemented.")
        return [z1, z2], 0

    def inverse(self, z):
        z1, z2 = z
        if self.mode == "channel":
            z = torch.cat([z1, z2], 1)
        elif self.mode == "channel_inv":
            z = torch.cat([z2, z1], 1)
        elif "checkerboard" in self.mode:
            n_dims = z1.dim()
            z_size = list(z1.size())
            z_size[-1] *= 2
            cb = _checkerboard(z_size, "inv" in self.mode, z1.device)
            z1 = z1[..., None].repeat(*(n_dims * [1]), 2).view(*z_size[:-1], -1)
            z2 = z2[..., None].repeat(*(n_dims * [1]), 2).view(*z_size[:-1], -1)
            z = cb * z1 + (1 - cb) * z2
        else:
            raise NotImplementedError("Mode " + self.mode + " is not implemented.")
        return z, 0

    def _run(self, z, inverse, ld, acc, **kw):
        z, _ = self.inverse(z) if inverse else self.forward(z)
        return z


class Merge(Split):
    """Split with forward and inverse interchanged (reshape.py:88-100)."""

    def __init__(self, mode="channel"):
        super().__init__(mode)

    def forward(self, z):
        return super().inverse(z)

    def inverse(self, z):
        return super().forward(z)


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
