from .base import Flow, Reverse, Composite, zero_log_det_like_z
from .reshape import Split, Merge, Squeeze
from .affine import AffineConstFlow, CCAffineConst, AffineCoupling, MaskedAffineFlow, AffineCouplingBlock
from .normalization import ActNorm, BatchNorm
from .mixing import Permute, Invertible1x1Conv, InvertibleAffine, LULinearPermute
from .glow import GlowBlock
from .neural_spline import (CoupledRationalQuadraticSpline, CircularCoupledRationalQuadraticSpline,
                            PiecewiseRationalQuadraticCoupling, PiecewiseRationalQuadraticCDF)
from .autoregressive import (Autoregressive, MaskedAffineAutoregressive,
                             MaskedPiecewiseRationalQuadraticAutoregressive, AutoregressiveRationalQuadraticSpline,
                             CircularAutoregressiveRationalQuadraticSpline)
from .periodic import PeriodicWrap, PeriodicShift
