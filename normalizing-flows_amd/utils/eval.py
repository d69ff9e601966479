"""Bits per dimension of image models (normflows/utils/eval.py:5-64): host-side bookkeeping around model.log_prob."""
import numpy as np
import torch


def bitsPerDim(model, x, y=None, trans="logit", trans_param=[0.05]):
    """eval.py:5-34: bits/dim of a batch that was logit-transformed for training."""
    if trans != "logit":
        raise NotImplementedError("The transformation " + trans + " is not implemented.")
    dims = int(np.prod(x.shape[1:]))
    log_q = model.log_prob(x) if y is None else model.log_prob(x, y)
    sum_dims = list(range(1, x.dim()))
    ls = torch.nn.functional.logsigmoid
    sig = (torch.sum(ls(x), sum_dims) + torch.sum(ls(-x), sum_dims)) / np.log(2)
    return -log_q / dims / np.log(2) - np.log2(1 - trans_param[0]) + 8 + sig / dims


def bitsPerDimDataset(model, data_loader, class_cond=True, trans="logit", trans_param=[0.05]):
    """eval.py:37-64: NaN-skipping average over a data loader."""
    n, b_cum = 0, 0.0
    with torch.no_grad():
        for x, y in iter(data_loader):
            b = bitsPerDim(model, x, y.to(x.device) if class_cond else None, trans, trans_param).to("cpu").numpy()
            b_cum += np.nansum(b)
            n += len(x) - np.sum(np.isnan(b))
    return b_cum / n
