"""GlowBlock = [AffineCouplingBlock(ConvNet2d), Invertible1x1Conv, ActNorm] (normflows/flows/affine/glow.py:11-84)."""
import torch
from torch import nn

from .. import nets
from .affine import AffineCouplingBlock
from .base import Flow, run_flow
from .mixing import Invertible1x1Conv
from .normalization import ActNorm


class GlowBlock(Flow):
    def __init__(self, channels, hidden_channels, scale=True, scale_map="sigmoid", split_mode="channel", leaky=0.0,
                 init_zeros=True, use_lu=True, net_actnorm=False):
        super().__init__()
        self.flows = nn.ModuleList([])
        kernel_size = (3, 1, 3)
        num_param = 2 if scale else 1
        if "channel" == split_mode:
            channels_ = ((channels + 1) // 2,) + 2 * (hidden_channels,)
            channels_ += (num_param * (channels // 2),)
        elif "channel_inv" == split_mode:
            channels_ = (channels // 2,) + 2 * (hidden_channels,)
            channels_ += (num_param * ((channels + 1) // 2),)
        elif "checkerboard" in split_mode:
            channels_ = (channels,) + 2 * (hidden_channels,)
            channels_ += (num_param * channels,)
        else:
            raise NotImplementedError("Mode " + split_mode + " is not implemented.")
        param_map = nets.ConvNet2d(channels_, kernel_size, leaky, init_zeros, actnorm=net_actnorm)
        self.flows += [AffineCouplingBlock(param_map, scale, scale_map, split_mode)]
        if channels > 1:
            self.flows += [Invertible1x1Conv(channels, use_lu)]
        self.flows += [ActNorm((channels,) + (1, 1))]

    def _run(self, z, inverse, ld, acc, **kw):
        seq = reversed(self.flows) if inverse else self.flows
        for f in seq:
            z = run_flow(f, z, inverse, ld, acc)
        return z

    def forward(self, z):
        ld = torch.zeros(z.shape[0], dtype=z.dtype, device=z.device)
        return self._run(z, False, ld, +1), ld

    def inverse(self, z):
        ld = torch.zeros(z.shape[0], dtype=z.dtype, device=z.device)
        return self._run(z, True, ld, +1), ld
