"""Bits per dimension of image models (normflows/utils/eval.py:5-64): host-side bookkeeping around model.log_prob for
data that was logit-transformed with parameter alpha = trans_param[0] before training."""
import math

import torch

_LN2 = math.log(2.0)


def _logit_correction_bits(x):
    """Per-sample sum of log2 sigmoid(x) + log2 sigmoid(-x): the Jacobian of the logit preprocessing, in bits."""
    ls = torch.nn.functional.logsigmoid
    per_elem = (ls(x) + ls(-x)) / _LN2
    return per_elem.flatten(1).sum(1)


def bitsPerDim(model, x, y=None, trans="logit", trans_param=[0.05]):
    """Bits per dimension of a batch under `model` (eval.py:5-34)."""
    if trans != "logit":
        raise NotImplementedError("The transformation " + trans + " is not implemented.")
    dims = x[0].numel()
    log_q = model.log_prob(x, y) if y is not None else model.log_prob(x)
    nats_to_bits = 1.0 / (dims * _LN2)
    return 8.0 - math.log2(1.0 - trans_param[0]) - log_q * nats_to_bits + _logit_correction_bits(x) / dims


def bitsPerDimDataset(model, data_loader, class_cond=True, trans="logit", trans_param=[0.05]):
    """NaN-skipping average of bitsPerDim over a data loader (eval.py:37-64)."""
    total, count = 0.0, 0
    with torch.no_grad():
        for x, y in data_loader:
            b = bitsPerDim(model, x, y.to(x.device) if class_cond else None, trans, trans_param)
            ok = ~torch.isnan(b)
            total += float(b[ok].sum())
            count += int(ok.sum())
    return total / count
