"""torch.autograd.Function wrappers: the training path.

The reference trains with `loss = model.forward_kld(x); loss.backward()` (core.py:87-102, examples); every layer is
differentiated by PyTorch autograd.  Here the forward of each Function is the same HIP kernel as in inference; the
backward is
  * a HIP kernel for the spline transform (nf_rqs_coupling_bwd, csrc/rqs_bwd.hip),
  * library GEMMs / triangular solves through torch for the batch reductions of LULinearPermute's parameter
    gradients (dL = tril(gy^T u), dU = triu(gu^T x_p): plain GEMMs over the batch),
  * closed-form elementwise expressions for DiagGaussian.
The conditioner networks are ordinary torch modules and are differentiated by autograd itself.  Layers switch to
these Functions only when gradients are needed (`needs_grad`); under torch.no_grad() the fused inference kernels run.
"""
import torch

from . import _lib as L
from . import ops


def needs_grad(*tensors_or_modules):
    if not torch.is_grad_enabled():
        return False
    for t in tensors_or_modules:
        if t is None:
            continue
        if isinstance(t, torch.nn.Module):
            if any(p.requires_grad for p in t.parameters()):
                return True
        elif torch.is_tensor(t) and t.requires_grad:
            return True
    return False


class SplineFn(torch.autograd.Function):
    """(y, row-summed logabsdet) of the RQ spline on ALL columns of x.

    cond given   -> per-element parameters (transform half of a coupling layer): nsf/coupling.py:329-362
    cond is None -> batch-shared parameters uw/uh/ud (unconditional transform): nsf/coupling.py:221-253
    """

    @staticmethod
    def forward(ctx, x, cond, uw, uh, ud, K, inverse, kw):
        B, D = x.shape
        idx = torch.arange(D, device=x.device)
        none = idx[:0]
        if cond is not None:
            ii, ti, mode = none, idx, (L.RQS_SAMPLE_TRANSFORM if inverse else L.RQS_DENSITY)
            y, ld = ops.rqs_coupling(x, cond.contiguous(), None, None, None, ii, ti, K, mode, **kw)
        else:
            ii, ti, mode = idx, none, (L.RQS_SAMPLE_IDENTITY if inverse else L.RQS_DENSITY)
            kw = dict(kw, wh_div=1.0)
            y, ld = ops.rqs_coupling(x, None, uw, uh, ud, ii, ti, K, mode, **kw)
        ctx.save_for_backward(x, cond, uw, uh, ud, ii, ti)
        ctx.K, ctx.mode, ctx.kw = K, mode, kw
        return y, ld

    @staticmethod
    def backward(ctx, gy, gld):
        x, cond, uw, uh, ud, ii, ti = ctx.saved_tensors
        if gy is None:
            gy = torch.zeros_like(x)
        if gld is None:
            gld = torch.zeros(x.shape[0], dtype=x.dtype, device=x.device)
        gx, gcond, guw, guh, gud = ops.rqs_coupling_bwd(x, gy, gld, cond, uw, uh, ud, ii, ti, ctx.K, ctx.mode, **ctx.kw)
        return gx, gcond, guw, guh, gud, None, None, None


def _assemble_lu(lower_entries, upper_entries, udiag_raw, eps):
    D = udiag_raw.numel()
    dev, dt = udiag_raw.device, udiag_raw.dtype
    li = torch.tril_indices(D, D, -1, device=dev)
    ui = torch.triu_indices(D, D, 1, device=dev)
    Lm = torch.eye(D, device=dev, dtype=dt)
    Lm[li[0], li[1]] = lower_entries
    diag = torch.nn.functional.softplus(udiag_raw) + eps
    Um = torch.diag(diag)
    Um[ui[0], ui[1]] = upper_entries
    return Lm, Um, diag, li, ui


def _batch_outer(a, b, want_colsum=False):
    """a^T b (and the column sums of a) over the batch: the split-K HIP kernel where it applies (fp32, b <= 128 wide)."""
    if a.dtype == torch.float32 and a.is_cuda and b.shape[1] <= 128 and a.shape[0] >= 1024:
        return ops.linear_wgrad(a, b, want_bias=want_colsum)
    return a.t() @ b, (a.sum(0) if want_colsum else None)


class LULinearPermuteFn(torch.autograd.Function):
    """LULinearPermute (mixing.py:535-563).  direction 0 = .inverse (density), 1 = .forward (sample)."""

    @staticmethod
    def forward(ctx, x, perm, lower_entries, upper_entries, udiag_raw, bias, eps, direction):
        y, ld = ops.lu_linear_permute(x, perm, lower_entries, upper_entries, udiag_raw, bias, direction, eps=eps)
        ctx.save_for_backward(x, y, perm, lower_entries, upper_entries, udiag_raw, bias)
        ctx.eps, ctx.direction = eps, direction
        return y, ld

    @staticmethod
    def backward(ctx, gy, gld):
        x, y, perm, lower_entries, upper_entries, udiag_raw, bias = ctx.saved_tensors
        Lm, Um, diag, li, ui = _assemble_lu(lower_entries, upper_entries, udiag_raw, ctx.eps)
        sig = torch.sigmoid(udiag_raw)
        sig = torch.where(udiag_raw > 20, torch.ones_like(sig), sig)  # softplus threshold
        if gy is None:
            gy = torch.zeros_like(y)
        gl_sum = gld.sum() if gld is not None else torch.zeros((), dtype=x.dtype, device=x.device)
        if ctx.direction == 0:
            # y = L (U x_p) + b ; logdet = sum log diag
            xp = x.index_select(1, perm)
            u = xp @ Um.t()
            gu = gy @ Lm            # d/du
            gxp = gu @ Um           # d/dx_p
            gx = torch.empty_like(x)
            gx.index_copy_(1, perm, gxp)
            gL, g_bias = _batch_outer(gy, u, want_colsum=True)
            gU, _ = _batch_outer(gu, xp)
            gdiag = torch.diagonal(gU) + gl_sum / diag
        else:
            # y[:, perm] = t,  U t = u,  L u = x - b ; logdet = -sum log diag
            t = y.index_select(1, perm)
            u = t @ Um.t()
            gt = gy.index_select(1, perm)
            gu = torch.linalg.solve_triangular(Um.t(), gt.t(), upper=False).t()      # U^T gu = gt
            gv = torch.linalg.solve_triangular(Lm.t(), gu.t(), upper=True, unitriangular=True).t()  # L^T gv = gu
            gx = gv
            gL, g_bias = _batch_outer(gv, u, want_colsum=True)
            gU, _ = _batch_outer(gu, t)
            gL, gU, g_bias = -gL, -gU, -g_bias
            gdiag = torch.diagonal(gU) - gl_sum / diag
        g_lower = gL[li[0], li[1]]
        g_upper = gU[ui[0], ui[1]]
        g_udiag = gdiag * sig
        return gx, None, g_lower, g_upper, g_udiag, g_bias, None, None


class DiagGaussianLogProbFn(torch.autograd.Function):
    """DiagGaussian.log_prob (distributions/base.py:94-103)."""

    @staticmethod
    def forward(ctx, z, loc, log_scale, shift):
        out = ops.diag_gaussian_log_prob(z, loc, log_scale, shift)
        ctx.save_for_backward(z, loc, log_scale)
        ctx.shift = shift
        return out

    @staticmethod
    def backward(ctx, g):
        z, loc, log_scale = ctx.saved_tensors
        ls = log_scale + ctx.shift
        q = (z - loc) / torch.exp(ls)
        gq = g.view(-1, *([1] * (z.dim() - 1)))
        gz = -(q / torch.exp(ls)) * gq
        gloc = (-gz).sum(0, keepdim=True)
        gls = ((q * q - 1.0) * gq).sum(0, keepdim=True)
        return gz, gloc.view_as(loc), gls.view_as(log_scale), None


class LinearFn(torch.autograd.Function):
    """y = x W^T + b with the weight / bias gradients on the split-K HIP kernel: at the training batch sizes of the
    path (K = 65 536 rows, 128 x 128 outputs) the library GEMM runs at a few percent of peak."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gx = gw = gb = None
        gy = gy.contiguous()
        if ctx.needs_input_grad[0]:
            gx = gy @ weight
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            gw, gb = ops.linear_wgrad(gy, x, want_bias=ctx.has_bias)
        return gx, gw, gb


def linear(x, weight, bias):
    """F.linear, routed through LinearFn where the custom weight-gradient kernel applies."""
    if (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and weight.shape[1] <= 128 and x.shape[0] >= 1024
            and torch.is_grad_enabled() and (weight.requires_grad or (bias is not None and bias.requires_grad))):
        return LinearFn.apply(x, weight, bias)
    return torch.nn.functional.linear(x, weight, bias)
