"""The reference's spline entry points under their own names (normflows/utils/splines.py:6-219) on the HIP kernels.

`unconstrained_rational_quadratic_spline` / `rational_quadratic_spline` take the reference's arguments and return the same pair
(outputs, logabsdet), element-wise, on tensors of any leading shape; the arithmetic is nf_rqs_spline (csrc/rqs_spline.hip:
softmax -> knots -> count-based bin search -> rational-quadratic piece, the reference's order of operations).  With gradients
enabled and a tensor that requires them the call goes through the coupling kernels' forward + backward pair (one feature per
row), so inputs and all three parameter tensors receive gradients (`rational_quadratic_spline`: for square boxes, right - left ==
top - bottom, which includes the reference's default [0, 1] x [0, 1]; other boxes only without gradients).  Not mirrored: per-feature tail lists and tensor limits
(`utils/splines.py:48-66, 116-119`) -- the flow classes (`PiecewiseRationalQuadraticCDF`, the coupling layers) take those."""
import torch

from .. import ops
from ..autograd import SplineFn, needs_grad

DEFAULT_MIN_BIN_WIDTH = 1e-3
DEFAULT_MIN_BIN_HEIGHT = 1e-3
DEFAULT_MIN_DERIVATIVE = 1e-3


def searchsorted(bin_locations, inputs, eps=1e-6):
    """Index of the bin each input falls into: the number of knots <= input, minus one, with the last knot nudged up by `eps`
    (utils/splines.py:11-13).  The reference adds eps to the caller's tensor in place; this does not modify its argument."""
    knots = torch.cat([bin_locations[..., :-1], bin_locations[..., -1:] + eps], dim=-1)
    return torch.sum(inputs[..., None] >= knots, dim=-1) - 1


def _scalar(v, what):
    if torch.is_tensor(v):
        if v.numel() != 1:
            raise NotImplementedError("utils.splines: tensor-valued %s; use the flow classes (tensor tail_bound)" % what)
        return float(v)
    return float(v)


def _run(inputs, uw, uh, ud, inverse, tails, bound, box, mins):
    if needs_grad(inputs, uw, uh, ud):
        assert box is None          # (rational_quadratic_spline reduces square boxes to the unit box before it gets here)
        K = uw.shape[-1]
        x = inputs.reshape(-1, 1)
        cond = torch.cat([uw, uh, ud], dim=-1).reshape(x.shape[0], -1)
        kw = dict(tails=tails, tail_bound=bound, min_bin_width=mins[0], min_bin_height=mins[1], min_derivative=mins[2], wh_div=1.0)
        y, lad = SplineFn.apply(x.contiguous(), cond.contiguous(), None, None, None, K, bool(inverse), kw)
        return y.reshape(inputs.shape), lad.reshape(inputs.shape)
    if box is None:
        return ops.rqs_spline(inputs, uw, uh, ud, inverse=bool(inverse), tails=tails, tail_bound=bound, left=-bound, right=bound,
                              bottom=-bound, top=bound, min_bin_width=mins[0], min_bin_height=mins[1], min_derivative=mins[2])
    return ops.rqs_spline(inputs, uw, uh, ud, inverse=bool(inverse), tails=None, left=box[0], right=box[1], bottom=box[2],
                          top=box[3], min_bin_width=mins[0], min_bin_height=mins[1], min_derivative=mins[2])


def unconstrained_rational_quadratic_spline(inputs, unnormalized_widths, unnormalized_heights, unnormalized_derivatives,
                                            inverse=False, tails="linear", tail_bound=1.0,
                                            min_bin_width=DEFAULT_MIN_BIN_WIDTH, min_bin_height=DEFAULT_MIN_BIN_HEIGHT,
                                            min_derivative=DEFAULT_MIN_DERIVATIVE):
    """utils/splines.py:16-97: identity outside [-tail_bound, tail_bound] (logabsdet 0 there), the spline inside; `tails` in
    ("linear", "circular"): K - 1 resp. K derivative logits per element."""
    if tails not in ("linear", "circular"):
        if isinstance(tails, (list, tuple)):
            raise NotImplementedError("utils.splines: per-feature tail lists; use the flow classes")
        raise RuntimeError("{} tails are not implemented.".format(tails))
    return _run(inputs, unnormalized_widths, unnormalized_heights, unnormalized_derivatives, inverse, tails,
                _scalar(tail_bound, "tail_bound"), None, (min_bin_width, min_bin_height, min_derivative))


def rational_quadratic_spline(inputs, unnormalized_widths, unnormalized_heights, unnormalized_derivatives, inverse=False,
                              left=0.0, right=1.0, bottom=0.0, top=1.0, min_bin_width=DEFAULT_MIN_BIN_WIDTH,
                              min_bin_height=DEFAULT_MIN_BIN_HEIGHT, min_derivative=DEFAULT_MIN_DERIVATIVE):
    """utils/splines.py:100-219: the spline on [left, right] -> [bottom, top]; K + 1 derivative logits per element."""
    num_bins = unnormalized_widths.shape[-1]
    if min_bin_width * num_bins > 1.0:
        raise ValueError("Minimal bin width too large for the number of bins")
    if min_bin_height * num_bins > 1.0:
        raise ValueError("Minimal bin height too large for the number of bins")
    l, r, b, t = (_scalar(v, n) for v, n in ((left, "left"), (right, "right"), (bottom, "bottom"), (top, "top")))
    mins = (min_bin_width, min_bin_height, min_derivative)
    if needs_grad(inputs, unnormalized_widths, unnormalized_heights, unnormalized_derivatives):
        # The backward kernel's `tails=None` form is the UNIT box [0, 1] x [0, 1] (the reference's default).  Any SQUARE box of
        # side s is that box scaled by s in both coordinates and translated: bin widths and heights scale alike, so the slopes
        # (the derivative parameters are slopes in actual coordinates, utils/splines.py:150-158), theta and logabsdet do not
        # change: spline_box(x) = bottom + s * spline_unit((x - left) / s) (inverse: (y - bottom) / s -> left + s * .).  A box with
        # right - left != top - bottom rescales the slopes and has no such reduction: not differentiable here.
        if abs((r - l) - (t - b)) <= 1e-12 * max(abs(r - l), abs(t - b), 1.0):
            side = r - l
            o_in, o_out = (b, l) if inverse else (l, b)
            u = inputs if (side == 1.0 and o_in == 0.0) else (inputs - o_in) / side
            y, lad = _run(u, unnormalized_widths, unnormalized_heights, unnormalized_derivatives, inverse, None, 1.0, None, mins)
            return (y if (side == 1.0 and o_out == 0.0) else y * side + o_out), lad
        raise NotImplementedError("utils.splines.rational_quadratic_spline under autograd: right - left must equal top - bottom "
                                  "(any square box, e.g. the default [0, 1] x [0, 1]); other boxes run without gradients only")
    return _run(inputs, unnormalized_widths, unnormalized_heights, unnormalized_derivatives, inverse, None, 0.0, (l, r, b, t), mins)
