"""Masked autoregressive flow (MAF): Autoregressive, MaskedAffineAutoregressive.

Mirrors normflows/flows/affine/autoregressive.py:10-128 (constructor signature, `autoregressive_net.*` state_dict
keys, direction semantics: `forward` = one MADE pass, `inverse` = D sequential MADE passes).  The element-wise
affine transform and its log-det are one HIP kernel (nf_maf_affine); MADE's masked linears are library GEMMs on
pre-masked weights (cached per parameter version).

NOTE: the inverse is implemented with the reference's D-pass structure; the
masked-aware incremental evaluation (each hidden unit finalised once, total work = ONE MADE pass) is the planned
persistent-kernel replacement, see DESIGN.md section 7.
"""
import numpy as np
import torch
from torch.nn import functional as F

from .. import nets, ops
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

    def _elementwise(self, inputs, params, direction, want_logdet=True):
        if inputs.dim() != 2:
            raise NotImplementedError("MaskedAffineAutoregressive: (batch, features) inputs only")
        return ops.maf_affine(inputs, params, direction, want_logdet=want_logdet)
