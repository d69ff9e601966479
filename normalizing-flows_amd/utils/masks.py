"""Binary feature masks (normflows/utils/masks.py:4-57).  Integer/byte work: bit-exact with the reference
(tests/test_host.py::test_masks_bit_exact, ::test_seeded_construction_is_bit_identical_to_reference)."""
import torch


def _ones_at(features, positions):
    """uint8 vector of length `features` with ones at `positions` (an index tensor or a boolean selector)."""
    out = torch.zeros(features, dtype=torch.uint8)
    out[positions] = 1
    return out


def _half(features):
    return (features + 1) // 2   # the reference rounds the half up for odd sizes


def create_alternating_binary_mask(features, even=True):
    """Ones on the even (even=True) or odd (even=False) positions (masks.py:4-17)."""
    parity = 0 if even else 1
    return _ones_at(features, torch.arange(features) % 2 == parity)


def create_mid_split_binary_mask(features):
    """Ones on the first ceil(features / 2) positions (masks.py:20-32)."""
    return _ones_at(features, torch.arange(features) < _half(features))


def create_random_binary_mask(features, seed=None):
    """ceil(features / 2) ones at positions drawn without replacement (masks.py:35-57); the draw consumes the same
    multinomial call as the reference, so seeded masks are identical."""
    gen = None if seed is None else torch.Generator().manual_seed(seed)
    picked = torch.multinomial(torch.ones(features), _half(features), replacement=False, generator=gen)
    return _ones_at(features, picked)
