"""Data-loader transforms (normflows/utils/preprocessing.py:4-57): plain tensor arithmetic on the loader side."""
import torch


class Logit:
    def __init__(self, alpha=0):
        self.alpha = alpha

    def __call__(self, x):
        x_ = self.alpha + (1 - self.alpha) * x
        return torch.log(x_ / (1 - x_))

    def inverse(self, x):
        return (torch.sigmoid(x) - self.alpha) / (1 - self.alpha)


class Jitter:
    def __init__(self, scale=1.0 / 256):
        self.scale = scale

    def __call__(self, x):
        return x + torch.rand_like(x) * self.scale


class Scale:
    def __init__(self, scale=255.0 / 256.0):
        self.scale = scale

    def __call__(self, x):
        return x * self.scale
