"""Tensor helpers of the hot-path files (normflows/utils/nn.py:181-193).  The kernels do these reductions themselves; the
functions exist for callers that import them from the reference's `utils` namespace."""
import torch


def sum_except_batch(x, num_batch_dims=1):
    """Per-sample sum: every dimension after the first `num_batch_dims` is reduced (utils/nn.py:190-193)."""
    if x.dim() <= num_batch_dims:
        return x.sum()      # nothing but batch dimensions: torch.sum over an empty dim list reduces everything, and so does the reference
    return x.flatten(start_dim=num_batch_dims).sum(dim=-1)


def tile(x, n):
    """Every element of x (flattened) repeated n times in place: [a, b] -> [a, a, .., b, b, ..] (utils/nn.py:181-187; MADE's
    output-multiplier masks)."""
    return x.reshape(-1).repeat_interleave(n)
