from .masks import create_alternating_binary_mask, create_mid_split_binary_mask, create_random_binary_mask
from .eval import bitsPerDim, bitsPerDimDataset
from .preprocessing import Logit, Jitter, Scale
from .nn import sum_except_batch, tile
from . import splines
