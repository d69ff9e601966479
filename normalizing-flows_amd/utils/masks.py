"""Binary feature masks (normflows/utils/masks.py:4-57).  Integer/byte work: bit-exact with the reference."""
import torch


def create_alternating_binary_mask(features, even=True):
    """uint8 mask with ones at even (even=True) or odd (even=False) positions (masks.py:4-17)."""
    mask = torch.zeros(features).byte()
    mask[(0 if even else 1)::2] += 1
    return mask


def create_mid_split_binary_mask(features):
    """uint8 mask with ones on the first ceil(features/2) positions (masks.py:20-32)."""
    mask = torch.zeros(features).byte()
    mask[:(features + 1) // 2] += 1
    return mask


def create_random_binary_mask(features, seed=None):
    """uint8 mask with ceil(features/2) ones at multinomial-sampled positions (masks.py:35-57)."""
    mask = torch.zeros(features).byte()
    weights = torch.ones(features).float()
    num_samples = (features + 1) // 2
    generator = None
    if seed is not None:
        generator = torch.Generator()
        generator.manual_seed(seed)
    indices = torch.multinomial(input=weights, num_samples=num_samples, replacement=False, generator=generator)
    mask[indices] += 1
    return mask
