"""Data-loader transforms of the image examples (normflows/utils/preprocessing.py:4-57).  They run on the loader side on
host tensors (plain tensor arithmetic, not part of the device path): one callable base with an `apply` hook."""
import torch


class _LoaderTransform:
    """Callable wrapper so that instances drop into torchvision.transforms.Compose like the reference's classes."""

    def apply(self, x):
        raise NotImplementedError

    def __call__(self, x):
        return self.apply(x)


class Scale(_LoaderTransform):
    """x * scale (255/256 by default: room for the dequantisation jitter)."""

    def __init__(self, scale=255.0 / 256.0):
        self.scale = scale

    def apply(self, x):
        return torch.mul(x, self.scale)


class Jitter(_LoaderTransform):
    """Uniform dequantisation noise in [0, scale)."""

    def __init__(self, scale=1.0 / 256):
        self.scale = scale

    def apply(self, x):
        return torch.add(x, torch.rand_like(x), alpha=self.scale)


class Logit(_LoaderTransform):
    """logit(alpha + (1 - alpha) x) and its inverse."""

    def __init__(self, alpha=0):
        self.alpha = alpha

    def apply(self, x):
        u = torch.add(x * (1 - self.alpha), self.alpha)
        return torch.log(u / (1 - u))

    def inverse(self, x):
        return (torch.sigmoid(x) - self.alpha) / (1 - self.alpha)
