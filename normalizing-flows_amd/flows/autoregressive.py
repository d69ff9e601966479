"""Masked autoregressive flow (MAF): Autoregressive, MaskedAffineAutoregressive.

Mirrors normflows/flows/affine/autoregressive.py:10-128 (constructor signature, `autoregressive_net.*` state_dict
keys, direction semantics: `forward` = one MADE pass, `inverse` = D sequential MADE passes).  The element-wise
affine transform and its log-det are one HIP kernel (nf_maf_affine); MADE's masked linears are library GEMMs on
pre-masked weights (cached per parameter version).

The inverse of MaskedAffineAutoregressive runs as ONE launch of nf_maf_inverse when
the MADE has the supported structure (flows/maf_pack.py): every hidden unit is finalised once, total work = one MADE
pass instead of D.  Other structures, float64 and gradient-tracking calls keep the reference's D-pass loop.
"""
import numpy as np
import torch
from torch.nn import functional as F

from .. import autograd, nets, ops
from . import maf_pack
from .base import Flow


class Autoregressive(Flow):
    """Element-wise invertible transform with parameters from an autoregressive net (autoregressive.py:10-47)."""

    def __init__(self, autoregressive_net):
        super().__init__()
        self.autoregressive_net = autoregressive_net

    def forward(self, inputs, context=None):
        params = self.autoregressive_net(inputs, context)
        return self._elementwise(inputs, params, 0)

    def inverse(self, inputs, context=None):
        num_inputs = int(np.prod(inputs.shape[1:]))
        outputs = torch.zeros_like(inputs)
        logabsdet = None
        for i in range(num_inputs):
            params = self.autoregressive_net(outputs, context)
            last = i == num_inputs - 1
            outputs, ld = self._elementwise(inputs, params, 1, want_logdet=last)
            if last:
                logabsdet = ld
        return outputs, logabsdet

    def _elementwise(self, inputs, params, direction, want_logdet=True):
        raise NotImplementedError()

    def _output_dim_multiplier(self):
        raise NotImplementedError()


class MaskedAffineAutoregressive(Autoregressive):
    """Masked autoregressive flow, arXiv 1705.07057 (autoregressive.py:50-128)."""

    def __init__(self, features, hidden_features, context_features=None, num_blocks=2, use_residual_blocks=True,
                 random_mask=False, activation=F.relu, dropout_probability=0.0, use_batch_norm=False):
        self.features = features
        made = nets.MADE(features=features, hidden_features=hidden_features, context_features=context_features,
                         num_blocks=num_blocks, output_multiplier=self._output_dim_multiplier(),
                         use_residual_blocks=use_residual_blocks, random_mask=random_mask, activation=activation,
                         dropout_probability=dropout_probability, use_batch_norm=use_batch_norm)
        super().__init__(made)

    def _output_dim_multiplier(self):
        return 2

    def _packed(self, device):
        """Device copies of the incremental-inverse pack, rebuilt when any MADE parameter changes."""
        key = tuple((p.data_ptr(), p._version) for p in self.autoregressive_net.parameters()) + (str(device),)
        cache = getattr(self, "_maf_pack_cache", None)
        if cache is None or cache[0] != key:
            packed = maf_pack.pack_made(self.autoregressive_net)
            if packed is not None:
                blob, table = packed
                packed = (torch.from_numpy(blob).to(device), torch.from_numpy(table).to(device), int(table[3]))
            self._maf_pack_cache = cache = (key, packed)
        return cache[1]

    def inverse(self, inputs, context=None):
        if (context is None and inputs.dim() == 2 and inputs.dtype == torch.float32 and inputs.is_cuda
                and not autograd.needs_grad(inputs, *self.autoregressive_net.parameters())):
            packed = self._packed(inputs.device)
            if packed is not None:
                return ops.maf_inverse(inputs, packed[0], packed[1], packed[2])
        return super().inverse(inputs, context)

    def _elementwise(self, inputs, params, direction, want_logdet=True):
        if inputs.dim() != 2:
            raise NotImplementedError("MaskedAffineAutoregressive: (batch, features) inputs only")
        return ops.maf_affine(inputs, params, direction, want_logdet=want_logdet)
