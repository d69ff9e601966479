"""torch.autograd.Function wrappers: the training path.

The reference trains with `loss = model.forward_kld(x); loss.backward()` (core.py:87-102, examples); every layer is
differentiated by PyTorch autograd.  Here the forward of each Function is the same HIP kernel as in inference; the
backward is
  * a HIP kernel for the spline transform (nf_rqs_coupling_bwd, csrc/rqs_bwd.hip),
  * library GEMMs / triangular solves through torch for the batch reductions of LULinearPermute's parameter
    gradients (dL = tril(gy^T u), dU = triu(gu^T x_p): plain GEMMs over the batch),
  * closed-form elementwise expressions for DiagGaussian.
The conditioner networks: the benchmark layer's ResidualNet through CouplingTrainFn / ResidualBlockFn / autograd.linear; MADE, wider
ResidualNets and GlowBlock's ConvNet2d through MadeFn / ConvNetFn (csrc/made_fwd.hip EPI 3, csrc/made_bwd.hip, csrc/conv_rows.hip);
anything else (contexts, batch norm, other activations) stays an ordinary torch module differentiated by autograd itself.  Layers
switch to these Functions only when gradients are needed (`needs_grad`); under torch.no_grad() the fused inference kernels run.
"""
import contextlib
import math

import torch
from torch.autograd.function import once_differentiable

from . import _gradbuf
from . import _sidestream
from . import _lib as L
from . import config as _config
from . import ops


def refuse_grad(what, *tensors_or_modules):
    """Layers without a training path fail loudly instead of silently cutting the autograd graph."""
    if needs_grad(*tensors_or_modules):
        raise NotImplementedError("%s has no differentiable path in normflows_amd (inference kernels only): call it under "
                                  "torch.no_grad() or freeze its parameters and inputs" % what)


def _stamp(*tensors):
    """Identity + version of the tensors a layer-owned weight image (packed blob, transposed / padded weights, LU factors) was
    built from.  The training Functions keep those images in buffers OWNED BY THE LAYER (they are rewritten once per step, not
    per call, and a multi-layer pack launch writes them through a cached pointer table), so autograd's saved-tensor version
    check cannot see them being overwritten: a forward pass records the stamp in the layer's holder and in its ctx, the
    backward refuses to run when the holder has moved on (the same layer ran forward again with OTHER weights before this
    graph's backward: retain_graph across an optimizer step, a two-optimizer schedule, checkpointing with an update in
    between).  Two forwards on the SAME weights (gradient accumulation over micro-batches) leave the stamp unchanged."""
    return tuple((id(t), t._version) for t in tensors if t is not None)


def _check_stamp(ctx, what):
    holder = getattr(ctx, "holder", None)
    if holder is not None and holder.get("stamp") != ctx.stamp:
        raise RuntimeError("%s: this layer ran forward again -- with modified parameters, or through its other training path (the "
                           "one-launch and the layer-wise path keep DIFFERENT images in the same layer-owned buffers; the stamp "
                           "carries the path) -- before the backward of an earlier forward; the weight images that backward needs "
                           "now belong to the later call.  Run backward before the parameters (or config.set_train_full / a "
                           "layer's use_fused_train) change, or run the forward again." % what)


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


class CouplingDensityFn(torch.autograd.Function):
    """The whole density-direction coupling transform (nsf/coupling.py:71-98) on full (B, D) rows as ONE forward and ONE
    backward launch: transform columns with the per-element parameters `cond`, identity columns with the batch-shared
    unconditional spline (or copied), split / merge inside the kernels."""

    @staticmethod
    def forward(ctx, x, cond, uw, uh, ud, iidx, tidx, K, kw):
        y, ld = ops.rqs_coupling(x, cond.contiguous(), uw, uh, ud, iidx, tidx, K, L.RQS_DENSITY, **kw)
        ctx.save_for_backward(x, cond, uw, uh, ud, iidx, tidx)
        ctx.K, ctx.kw = K, kw
        return y, ld

    @staticmethod
    def backward(ctx, gy, gld):
        x, cond, uw, uh, ud, iidx, tidx = ctx.saved_tensors
        if gy is None:
            gy = torch.zeros_like(x)
        if gld is None:
            gld = torch.zeros(x.shape[0], dtype=x.dtype, device=x.device)
        gx, gcond, guw, guh, gud = ops.rqs_coupling_bwd(x, gy, gld, cond, uw, uh, ud, iidx, tidx, ctx.K, L.RQS_DENSITY,
                                                        **ctx.kw)
        return gx, gcond, guw, guh, gud, None, None, None, None


class FinalSplineDensityFn(torch.autograd.Function):
    """The conditioner's final Linear (nets/resnet.py:104) + the whole density-direction coupling transform
    (nsf/coupling.py:83-98) as ONE forward launch of the fused kernel's training variant (nf_rqs_fused_train_fwd): the
    736-wide conditioner output never makes the library-GEMM round trip through HBM -- it is written once, in the lanes' own
    24-float rows, for the backward (nf_rqs_coupling_bwd_p24).  The final layer's input gradient is a library GEMM on the
    padded rows, its weight / bias gradients the split-K kernel; pad rows are dropped on the way out.
    Shape of the benchmark layer only: D = 64, hidden = 128, 8 bins, linear tails."""

    @staticmethod
    def forward(ctx, x, h2, wf, bf, uw, uh, ud, iidx, tidx, blob, parity, nblocks, kw, wpad, ld_acc=None, acc=1):
        ops.rqs_fused_pack_final(blob, wf.detach(), bf.detach(), uw.detach(), uh.detach(), ud.detach(), nblocks,
                                 tail_bound=kw["tail_bound"], min_bin_width=kw["min_bin_width"],
                                 min_bin_height=kw["min_bin_height"], min_derivative=kw["min_derivative"])
        # the caller's running log-density is updated inside the launch (no (B) add per layer); autograd sees an in-place op
        y, ld, cond24 = ops.rqs_fused_train_fwd(x, h2, blob, parity, nblocks, tail_bound=kw["tail_bound"],
                                                min_bin_width=kw["min_bin_width"], min_bin_height=kw["min_bin_height"],
                                                min_derivative=kw["min_derivative"], logdet=ld_acc,
                                                acc=None if ld_acc is None else (L.LD_ADD if acc > 0 else L.LD_SUB))
        if ld_acc is not None:
            ctx.mark_dirty(ld_acc)
        # final-layer weight on the 24-row layout of cond24 (pad row of every feature zero): `wpad` (nT, 24, H) is a zero
        # buffer OWNED BY THE LAYER (only its 23 real rows per feature are ever written), one strided copy per step, read by
        # the backward's input-gradient GEMM
        nT, H = cond24.shape[1], wf.shape[1]
        wpad[:, :23].copy_(wf.detach().view(nT, 23, H))
        wpad[:, 23].zero_()       # (the whole-layer path's pack leaves another image of the final weight in this buffer)
        ctx.save_for_backward(x, h2, wf, cond24, uw, uh, ud, iidx, tidx)
        ctx.kw, ctx.wpad, ctx.acc, ctx.has_acc = kw, wpad, acc, ld_acc is not None
        ctx.holder, ctx.stamp = kw.get("holder"), ("final",) + _stamp(wf, bf, uw, uh, ud)
        if ctx.holder is not None:
            ctx.holder["stamp"] = ctx.stamp
        return y, ld

    @staticmethod
    def backward(ctx, gy, gld):
        x, h2, wf, cond24, uw, uh, ud, iidx, tidx = ctx.saved_tensors
        _check_stamp(ctx, "FinalSplineDensityFn")
        kw = ctx.kw
        if gy is None:
            gy = torch.zeros_like(x)
        if gld is None:
            gld = torch.zeros(x.shape[0], dtype=x.dtype, device=x.device)
        gld_own = -gld if (ctx.has_acc and ctx.acc < 0) else gld
        gx, gcond24, guw, guh, gud = ops.rqs_coupling_bwd_p24(x, gy, gld_own, cond24, uw, uh, ud, iidx, tidx,
                                                              tail_bound=kw["tail_bound"], min_bin_width=kw["min_bin_width"],
                                                              min_bin_height=kw["min_bin_height"],
                                                              min_derivative=kw["min_derivative"], wh_div=kw["wh_div"])
        B, nT, H = x.shape[0], cond24.shape[1], wf.shape[1]
        g2 = gcond24.view(B, nT * 24)
        gh2 = g2 @ ctx.wpad.view(nT * 24, H)                 # input gradient of the final layer (library GEMM, padded rows)
        gwf, gbf = ops.linear_wgrad(g2, h2, want_bias=True, skip_every=24)   # pad rows dropped in the reduction
        return gx, gh2, gwf, gbf, guw, guh, gud, None, None, None, None, None, None, None, (gld if ctx.has_acc else None), None


_tri_cache = {}


def _tri_indices(D, dev):
    """(strictly-lower, strictly-upper) index pairs of a D x D matrix on `dev`, built once (two launches each otherwise)."""
    key = (D, str(dev))
    if key not in _tri_cache:
        _tri_cache[key] = (torch.tril_indices(D, D, -1, device=dev), torch.triu_indices(D, D, 1, device=dev))
    return _tri_cache[key]


def _assemble_lu(lower_entries, upper_entries, udiag_raw, eps):
    D = udiag_raw.numel()
    dev, dt = udiag_raw.device, udiag_raw.dtype
    li, ui = _tri_indices(D, dev)
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
    def forward(ctx, x, perm, lower_entries, upper_entries, udiag_raw, bias, eps, direction, ld_acc=None, acc=1, factors_out=None,
                holder=None):
        D = x.shape[1]
        ctx.acc, ctx.has_acc = acc, ld_acc is not None
        if ld_acc is not None:
            ctx.mark_dirty(ld_acc)      # the caller's running log-density, updated in place (inside the launch where possible)
        if direction == 0 and x.is_cuda and x.dtype == torch.float32 and D <= 128:
            # density direction on the fp32-MFMA row mat-vec kernel: u = U x[perm] (kept for the backward), y = L u + b with
            # the constant log-det in the same launch -- two 13 us launches against 64 us for the LDS-tile kernel (D = 128: two
            # nf_rows_matvec_affine launches against 3 ms for that kernel at B = 65 536)
            with torch.no_grad():
                if factors_out is not None:    # assembled for every layer of the model by one launch (_prepack.py)
                    Lm, Um, Up, diag, lad, LT, UpT = ops.lu_factors_views(factors_out, D)
                else:
                    Lm, Um, Up, diag, lad, LT, UpT = ops.lu_factors(perm, lower_entries.detach(), upper_entries.detach(),
                                                                    udiag_raw.detach(), eps=eps)      # one launch (nf_lu_factors)
                # u = U x[perm] (kept for the backward) and y = L u + b with the constant log-det: one launch (nf_rows_matvec2)
                from . import config
                lacc = None if ld_acc is None else (L.LD_ADD if acc > 0 else L.LD_SUB)
                if config.lu_bwd_fused and D == 64 and x.shape[0] % 64 == 0 and x.shape[0] >= 1024:
                    u, y, ld = ops.lu_fwd(x, UpT, LT, bias.detach(), lad, +1.0, logdet=ld_acc, acc=lacc)   # LDS-DMA tiles (nf_lu_fwd)
                elif config.lu_matvec2 and D <= 64:
                    u, y, ld = ops.rows_matvec2(x, Up, Lm, bias.detach(), lad, +1.0, logdet=ld_acc, acc=lacc)
                else:
                    u = ops.rows_matvec(x, Up)
                    y, ld = ops.rows_matvec_affine(u, Lm, bias.detach(), lad, +1.0, logdet=ld_acc, acc=lacc)
            ctx.save_for_backward(x, y, perm, lower_entries, upper_entries, udiag_raw, bias, u)
            ctx.eps, ctx.direction = eps, direction
            ctx.factors = (Lm, Um, diag, Up, LT, UpT)      # assembled once per step: the backward reuses them
            # (views of a LAYER-OWNED buffer when a multi-layer launch assembled them: see _stamp)
            ctx.holder = holder if factors_out is not None else None
            ctx.stamp = _stamp(lower_entries, upper_entries, udiag_raw)
            if ctx.holder is not None:
                ctx.holder["stamp"] = ctx.stamp
            return y, ld
        y, ld = ops.lu_linear_permute(x, perm, lower_entries, upper_entries, udiag_raw, bias, direction, eps=eps)
        ctx.save_for_backward(x, y, perm, lower_entries, upper_entries, udiag_raw, bias, None)
        ctx.eps, ctx.direction = eps, direction
        if ld_acc is not None:
            ld = ld_acc.add_(ld) if acc > 0 else ld_acc.sub_(ld)
        return y, ld

    @staticmethod
    def backward(ctx, gy, gld):
        x, y, perm, lower_entries, upper_entries, udiag_raw, bias, u_saved = ctx.saved_tensors
        fac = getattr(ctx, "factors", None)
        _check_stamp(ctx, "LULinearPermuteFn")
        D_ = x.shape[1]
        g_acc = gld if ctx.has_acc else None            # the running log-density passes its cotangent straight through
        if ctx.has_acc and ctx.acc < 0 and gld is not None:
            gld = -gld
        if fac is not None and ctx.direction == 0:
            # the density direction of the training step, all on hand-written kernels: three row mat-vecs, two split-K batch
            # reductions, one launch for the packed parameter gradients
            Lm, Um, diag, Up, LT, UpT = fac
            gy = torch.zeros_like(y) if gy is None else gy.contiguous()
            from . import config
            if config.lu_bwd_fused and D_ == 64 and gy.shape[0] % 64 == 0 and gy.shape[0] >= 1024 and u_saved is not None:
                # both row products and both batch reductions in one pass over the rows (nf_lu_bwd)
                # (gradients go straight to their destinations: views of dp.FlatParameters' buffer when registered, _gradbuf.py)
                gx, gL, g_bias, gUx = ops.lu_bwd(gy, u_saved, x, Lm, Up, db_out=_gradbuf.out(bias))
                g_lower, g_upper, g_udiag = ops.lu_param_grads(gL, gUx, gld, udiag_raw.detach(), lower_entries.numel(),
                                                               eps=ctx.eps, sign=1.0, perm=perm,
                                                               out=(_gradbuf.out(lower_entries), _gradbuf.out(upper_entries),
                                                                    _gradbuf.out(udiag_raw)))
                return gx, None, g_lower, g_upper, g_udiag, g_bias, None, None, g_acc, None, None, None
            if config.lu_matvec2 and D_ <= 64:
                gu, gx, _ = ops.rows_matvec2(gy, LT, UpT)   # d/du = L^T gy, d/dx = P (U^T gu): one launch
            else:
                gu = ops.rows_matvec(gy, LT)
                gx = ops.rows_matvec(gu, UpT)
            # gy^T u (+ the column sums of gy = the bias gradient) and gu^T x as ONE pair launch; gU = (gu^T x)[:, perm] is
            # taken through perm inside nf_lu_param_grads
            from . import config
            if config.wgrad_pair and gy.shape[0] >= 1024:
                gL, g_bias, gUx, _ = ops.linear_wgrad_pair(gy, u_saved, gu, x)
            else:
                gL, g_bias = _batch_outer(gy, u_saved, want_colsum=True)
                gUx, _ = _batch_outer(gu, x)
            g_lower, g_upper, g_udiag = ops.lu_param_grads(gL, gUx, gld, udiag_raw.detach(), lower_entries.numel(),
                                                           eps=ctx.eps, sign=1.0, perm=perm)
            return gx, None, g_lower, g_upper, g_udiag, g_bias, None, None, g_acc, None, None, None
        li, ui = _tri_indices(D_, x.device)
        if fac is not None:
            Lm, Um, diag, Up_saved = fac[:4]
        else:
            Lm, Um, diag, li, ui = _assemble_lu(lower_entries, upper_entries, udiag_raw, ctx.eps)
            Up_saved = None
        sig = torch.sigmoid(udiag_raw)
        sig = torch.where(udiag_raw > 20, torch.ones_like(sig), sig)  # softplus threshold
        if gy is None:
            gy = torch.zeros_like(y)
        gy = gy.contiguous()
        gl_sum = gld.sum() if gld is not None else torch.zeros((), dtype=x.dtype, device=x.device)
        D = x.shape[1]
        # Row-wise products with the D x D factors run on nf_rows_matvec (exact-fp32 MFMA, csrc/rows_matvec.hip) -- no library
        # GEMM / triangular solve touches the batch; the permutation is folded into the matrices:
        # Up[:, perm[j]] = U[:, j]  =>  Up x = U x[perm],  Up^T g = scatter of U^T g back through perm.
        hip_rows = x.is_cuda and D <= 64 and x.dtype == torch.float32

        def rows(v, W):   # r_b = W v_b for every row b
            return ops.rows_matvec(v, W) if hip_rows else v @ W.t()

        if Up_saved is not None:
            Up = Up_saved
        else:
            Up = torch.zeros_like(Um)
            Up[:, perm] = Um
        if ctx.direction == 0:
            # y = L (U x_p) + b ; logdet = sum log diag
            u = u_saved if u_saved is not None else rows(x, Up)     # U x[perm]
            gu = rows(gy, Lm.t())           # d/du = L^T gy
            gx = rows(gu, Up.t())           # d/dx = P (U^T gu)
            gL, g_bias = _batch_outer(gy, u, want_colsum=True)
            gU, _ = _batch_outer(gu, x.index_select(1, perm))
            gdiag = torch.diagonal(gU) + gl_sum / diag
        else:
            # y[:, perm] = t,  U t = u,  L u = x - b ; logdet = -sum log diag.  The D x D inverses are parameter-side work
            # (float64, like the reference's torch.inverse in mixing.py); the batch only sees mat-vec kernels.
            eye = torch.eye(D, dtype=torch.float64, device=x.device)
            Linv = torch.linalg.solve_triangular(Lm.double(), eye, upper=False, unitriangular=True)
            Uinv = torch.linalg.solve_triangular(Um.double(), eye, upper=True)
            t = y.index_select(1, perm)
            u = rows(y, Up)                                       # U t
            gu = rows(gy.index_select(1, perm), Uinv.t().to(x.dtype))     # U^T gu = gt
            gv = rows(gu, Linv.t().to(x.dtype))                   # L^T gv = gu
            gx = gv
            gL, g_bias = _batch_outer(gv, u, want_colsum=True)
            gU, _ = _batch_outer(gu, t)
            gL, gU, g_bias = -gL, -gU, -g_bias
            gdiag = torch.diagonal(gU) - gl_sum / diag
        g_lower = gL[li[0], li[1]]
        g_upper = gU[ui[0], ui[1]]
        g_udiag = gdiag * sig
        return gx, None, g_lower, g_upper, g_udiag, g_bias, None, None, g_acc, None, None, None


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
    """y = x W^T + b with the weight / bias gradients on the split-K HIP kernel (nf_linear_wgrad): at the training batch
    sizes of the path (K = 65 536 rows, 128 x 128 outputs) the library runs that reduction at a few percent of peak.  The
    forward and input-gradient products of a LONE Linear stay library GEMMs: measured on MI355X (tools/kernel_bench.py)
    hipBLASLt does the 65 536 x 128 x 128 product in 28 us (77 TFLOP/s) and the 736-row final layer at 0.55-0.85 of the fp32
    MFMA peak; round 2's stand-alone row-panel kernel lost to it on every shape and was removed -- what beats the library is
    keeping intermediates on chip (ResidualBlockFn, CouplingTrainFn, nf_final_bwd).  Benchmark-shaped NSF layers never come
    here: their whole forward and backward run on the one-pass kernels."""

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
            if weight.shape[1] <= 128:
                gw, gb = ops.linear_wgrad(gy, x, want_bias=ctx.has_bias)
            else:
                gw, gb = gy.t() @ x, (gy.sum(0) if ctx.has_bias else None)
        return gx, gw, gb


def linear(x, weight, bias):
    """F.linear, routed through LinearFn where the custom kernels apply."""
    if (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.shape[0] >= 1024 and torch.is_grad_enabled()
            and weight.shape[1] <= 128
            and (x.requires_grad or weight.requires_grad or (bias is not None and bias.requires_grad))):
        return LinearFn.apply(x, weight, bias)
    return torch.nn.functional.linear(x, weight, bias)


class CouplingTrainFn(torch.autograd.Function):
    """A whole CoupledRationalQuadraticSpline layer of the benchmark shape under autograd (density direction): ONE forward launch
    (nf_rqs_fused_train_full_fwd: initial layer, residual blocks, final layer, spline -- exactly the inference kernel -- writing
    the intermediates the backward needs), and a backward orchestrated here from the same kernels the per-module Functions use
    (spline backward on the 24-float rows, the final layer's input gradient as a library GEMM, one nf_rows_block per residual
    block, pair / ring weight-gradient launches, the initial layer's input gradient accumulated into the transform's with one
    addmm).  Replaces IdentLinearFn + ResidualBlockFn x blocks + FinalSplineDensityFn: four launches fewer forward, no
    per-module autograd nodes."""

    @staticmethod
    def forward(ctx, x, w0, b0, wf, bf, uw, uh, ud, iidx, tidx, blob, parity, kw, wfull, wpad, ld_acc, acc, *blk):
        nb = len(blk) // 4
        wb = [blk[4 * i + j].detach() for i in range(nb) for j in (0, 2)]
        bb = [blk[4 * i + j].detach() for i in range(nb) for j in (1, 3)]
        fk = dict(tail_bound=kw["tail_bound"], min_bin_width=kw["min_bin_width"], min_bin_height=kw["min_bin_height"],
                  min_derivative=kw["min_derivative"])
        # the same launch leaves wfull / wpad (the zero-padded weight images of the backward's products) current
        if not kw.get("prepacked"):
            ops.rqs_fused_pack_all(blob, w0.detach(), b0.detach(), wb, bb, wf.detach(), bf.detach(), uw.detach(), uh.detach(),
                                   ud.detach(), wfull=wfull, wpad=wpad, identity_idx=iidx, **fk)
        y, ld, cond24, acts = ops.rqs_fused_train_full_fwd(x, blob, parity, nb, logdet=ld_acc,
                                                           acc=None if ld_acc is None else (L.LD_ADD if acc > 0 else L.LD_SUB), **fk)
        if ld_acc is not None:
            ctx.mark_dirty(ld_acc)
        ctx.save_for_backward(x, cond24, acts, w0, wf, uw, uh, ud, iidx, tidx, *blk)
        ctx.kw, ctx.wfull, ctx.wpad, ctx.acc, ctx.has_acc, ctx.nb = kw, wfull, wpad, acc, ld_acc is not None, nb
        ctx.biases = (b0, bf)       # (identity only: where their gradients are written, _gradbuf.out)
        ctx.blob, ctx.parity = blob, parity
        ctx.holder, ctx.stamp = kw.get("holder"), ("full",) + _stamp(w0, b0, wf, bf, uw, uh, ud, *blk)
        if ctx.holder is not None:
            ctx.holder["stamp"] = ctx.stamp
        return y, ld

    @staticmethod
    def backward(ctx, gy, gld):
        x, cond24, acts, w0, wf, uw, uh, ud, iidx, tidx, *blk = ctx.saved_tensors
        _check_stamp(ctx, "CouplingTrainFn")
        kw, nb = ctx.kw, ctx.nb
        if gy is None:
            gy = torch.zeros_like(x)
        if gld is None:
            gld = torch.zeros(x.shape[0], dtype=x.dtype, device=x.device)
        gld_own = -gld if (ctx.has_acc and ctx.acc < 0) else gld
        B, nT, H = x.shape[0], cond24.shape[1], wf.shape[1]
        fk = dict(tail_bound=kw["tail_bound"], min_bin_width=kw["min_bin_width"], min_bin_height=kw["min_bin_height"],
                  min_derivative=kw["min_derivative"])
        if (_config.train_bwd_onecall and _config.final_bwd_fused and _config.resblock_bwd and 1 <= nb <= 5 and B % 64 == 0
                and H == 128 and x.shape[1] == 64):
            # round 6: the layer's whole backward behind one call -- four passes over the rows, ONE reduction launch, every gradient
            # written straight to its destination (a view of dp.FlatParameters' flat buffer when the parameter is registered there)
            b0, bf = ctx.biases
            dest = dict(w0=_gradbuf.out(w0), b0=_gradbuf.out(b0), wf=_gradbuf.out(wf), bf=_gradbuf.out(bf), uw=_gradbuf.out(uw),
                        uh=_gradbuf.out(uh), ud=_gradbuf.out(ud), blocks=[_gradbuf.out(p_) for p_ in blk])
            gx = ops.coupling_train_bwd(x, gy, gld_own, cond24, acts, ctx.wpad, ctx.blob, ctx.wfull,
                                        [blk[4 * b + j].detach() for b in range(nb) for j in (0, 2)], uw.detach(), uh.detach(),
                                        ud.detach(), kw["col_map"], iidx.numel(), ctx.parity, nb, dest, **fk)
            return (gx, dest["w0"], dest["b0"], dest["wf"], dest["bf"], dest["uw"], dest["uh"], dest["ud"], None, None, None, None,
                    None, None, None, (gld if ctx.has_acc else None), None, *dest["blocks"])
        if _config.final_bwd_fused:
            # ONE pass over the rows: spline backward on the vector ALU, its gradient rows straight into the MFMAs of the final
            # layer's input gradient (nf_final_bwd; ctx.wpad = the transposed stage image the forward's pack launch left)
            gx, gcond24, gh, guw, guh, gud = ops.final_bwd(x, gy, gld_own, cond24, ctx.wpad, ctx.blob, uw, uh, ud, ctx.parity, nb, **fk)
            g2 = gcond24.view(B, nT * 24)
        else:
            gx, gcond24, guw, guh, gud = ops.rqs_coupling_bwd_p24(x, gy, gld_own, cond24, uw, uh, ud, iidx, tidx, wh_div=kw["wh_div"], **fk)
            g2 = gcond24.view(B, nT * 24)
            wrows = torch.zeros(nT, 24, H, dtype=wf.dtype, device=wf.device)
            wrows[:, :23].copy_(wf.detach().view(nT, 23, H))
            gh = g2 @ wrows.view(nT * 24, H)        # (round 2: library GEMM on the padded rows)
        gwf, gbf = ops.linear_wgrad(g2, acts[2 * nb], want_bias=True, skip_every=24)
        gblk = [None] * (4 * nb)
        fused = _config.resblock_bwd and nb > 0 and B % 64 == 0 and H == 128 and x.shape[1] == 64
        for b in range(nb - 1, -1, -1):
            w1, w2 = blk[4 * b].detach(), blk[4 * b + 2].detach()
            h_in, t = acts[2 * b], acts[2 * b + 1]
            if fused:
                # one pass over the rows per block: both input-gradient products and both weight gradients
                # (nf_resblock_bwd); behind the first block also the initial layer's (gx += gh0 @ wfull, dW0, db0)
                if b == 0:
                    _, gw1, gb1, gw2, gb2, gw0, gb0 = ops.resblock_bwd(gh, t, h_in, w1, w2, x=x, wfull=ctx.wfull, gx=gx,
                                                                       col_map=kw["col_map"], n_cols=iidx.numel())
                else:
                    gh, gw1, gb1, gw2, gb2 = ops.resblock_bwd(gh, t, h_in, w1, w2)
                gblk[4 * b:4 * b + 4] = [gw1, gb1, gw2, gb2]
                continue
            gt, gh_in = ops.rows_block(gh, w2, None, w1, None, trans=True, mask1=t, mask2=h_in, relu=False)
            gw2, gb2, gw1, gb1 = ops.linear_wgrad_pair(gh, t, gt, h_in, relu_x=True)
            gblk[4 * b:4 * b + 4] = [gw1, gb1, gw2, gb2]
            gh = gh_in
        if not fused:
            gx.addmm_(gh, ctx.wfull.t())                           # + the conditioner's input gradient on the identity columns
            gw0f, gb0 = ops.linear_wgrad(gh, x, want_bias=True)
            gw0 = gw0f.index_select(1, iidx)
        return (gx, gw0, gb0, gwf, gbf, guw, guh, gud, None, None, None, None, None, None, None,
                (gld if ctx.has_acc else None), None, *gblk)


class PairTrainFn(torch.autograd.Function):
    """A benchmark-shaped [CoupledRationalQuadraticSpline, LULinearPermute] pair in the density direction under autograd (round 6):
    ONE forward launch for LULinearPermute.inverse + the whole coupling layer (nf_rqs_fused_train_pair_fwd: the inference kernel's
    pair fusion with the training variant's saved rows), and a backward of six launches: nf_coupling_train_bwd's five, then the
    composed LU's backward in one pass (nf_lu_bwd_composed: gx = g W_d, dW_d = g^T x; no second product, no intermediate u) with the
    factors' gradients on the parameter side (nf_lu_param_grads_composed).  Replaces LULinearPermuteFn (nf_lu_fwd 17 us + nf_lu_bwd
    27 us + two small launches per layer at the benchmark batch) + CouplingTrainFn.  Only inside core.run_chain, whose per-step
    multi-layer packs (_prepack.py) leave the blob's LU stage, W_d and the dense factors current."""

    @staticmethod
    def forward(ctx, x, perm, lower, upper, udiag, lbias, lu_eps, lu_fbuf, lu_wd, w0, b0, wf, bf, uw, uh, ud, iidx, tidx, blob, parity, kw,
                wfull, wpad, ld_acc, acc, *blk):
        nb = len(blk) // 4
        fk = dict(tail_bound=kw["tail_bound"], min_bin_width=kw["min_bin_width"], min_bin_height=kw["min_bin_height"],
                  min_derivative=kw["min_derivative"])
        xlu, y, ld, cond24, acts = ops.rqs_fused_train_pair_fwd(x, blob, parity, nb, logdet=ld_acc,
                                                                acc=None if ld_acc is None else (L.LD_ADD if acc > 0 else L.LD_SUB), **fk)
        if ld_acc is not None:
            ctx.mark_dirty(ld_acc)
        ctx.save_for_backward(x, xlu, cond24, acts, perm, lower, upper, udiag, w0, wf, uw, uh, ud, iidx, *blk)
        ctx.kw, ctx.wfull, ctx.wpad, ctx.acc, ctx.has_acc, ctx.nb = kw, wfull, wpad, acc, ld_acc is not None, nb
        ctx.blob, ctx.parity, ctx.biases = blob, parity, (b0, bf, lbias)
        ctx.lu = (lu_eps, lu_fbuf, lu_wd)
        ctx.holder, ctx.stamp = kw.get("holder"), ("pair",) + _stamp(lower, upper, udiag, lbias, w0, b0, wf, bf, uw, uh, ud, *blk)
        if ctx.holder is not None:
            ctx.holder["stamp"] = ctx.stamp
        return y, ld

    @staticmethod
    def backward(ctx, gy, gld):
        x, xlu, cond24, acts, perm, lower, upper, udiag, w0, wf, uw, uh, ud, iidx, *blk = ctx.saved_tensors
        _check_stamp(ctx, "PairTrainFn")
        kw, nb = ctx.kw, ctx.nb
        if gy is None:
            gy = torch.zeros_like(x)
        if gld is None:
            gld = torch.zeros(x.shape[0], dtype=x.dtype, device=x.device)
        gld_own = -gld if (ctx.has_acc and ctx.acc < 0) else gld
        fk = dict(tail_bound=kw["tail_bound"], min_bin_width=kw["min_bin_width"], min_bin_height=kw["min_bin_height"],
                  min_derivative=kw["min_derivative"])
        b0, bf, lbias = ctx.biases
        lu_eps, lu_fbuf, lu_wd = ctx.lu
        dest = dict(w0=_gradbuf.out(w0), b0=_gradbuf.out(b0), wf=_gradbuf.out(wf), bf=_gradbuf.out(bf), uw=_gradbuf.out(uw),
                    uh=_gradbuf.out(uh), ud=_gradbuf.out(ud), blocks=[_gradbuf.out(p_) for p_ in blk], lower=_gradbuf.out(lower),
                    upper=_gradbuf.out(upper), udiag=_gradbuf.out(udiag), lbias=_gradbuf.out(lbias))
        D = x.shape[1]
        Lm, Um = lu_fbuf[:D * D].view(D, D), lu_fbuf[D * D:2 * D * D].view(D, D)
        # The last two of the seven launches only produce parameter gradients: on the side stream when nobody can read them before the
        # join at the end of this backward pass (_sidestream.py) -- every gradient goes into a registered buffer that autograd will
        # adopt as .grad without a kernel (no existing .grad to accumulate into) and no tensor hook looks at it on the way.
        side = None
        if _config.train_reduce_async:
            params = (w0, b0, wf, bf, uw, uh, ud, lower, upper, udiag, lbias) + tuple(blk)
            if all(_gradbuf.target(p_) is not None and _sidestream.nobody_reads_early(p_) for p_ in params):
                side = _sidestream.stream(x.device)
        # ONE C-ABI call, seven launches: the coupling's four passes, the composed LU's pass, one reduction for both, the LU's factors
        # (two calls when the last two launches go to the side stream)
        gx = ops.pair_train_bwd(x, xlu, gy, gld_own, cond24, acts, ctx.wpad, ctx.blob, ctx.wfull,
                                [blk[4 * b + j].detach() for b in range(nb) for j in (0, 2)], uw.detach(), uh.detach(), ud.detach(),
                                kw["col_map"], iidx.numel(), ctx.parity, nb, lu_wd, Lm, Um, perm, udiag.detach(), lu_eps, dest, side=side,
                                **fk)
        if side is not None:
            _sidestream.mark(x.device)
            _sidestream.queue_join()
        g_lower, g_upper, g_udiag, g_lbias = dest["lower"], dest["upper"], dest["udiag"], dest["lbias"]
        return (gx, None, g_lower, g_upper, g_udiag, g_lbias, None, None, None, dest["w0"], dest["b0"], dest["wf"], dest["bf"],
                dest["uw"], dest["uh"], dest["ud"], None, None, None, None, None, None, None, (gld if ctx.has_acc else None), None,
                *dest["blocks"])


class IdentLinearFn(torch.autograd.Function):
    """The conditioner's initial Linear on the identity columns of a full-width row (nsf/coupling.py:71-76 `inputs[:,
    identity_features]` + nets/resnet.py:92): y = x[:, iidx] W^T + b computed as x Wfull^T + b with W scattered into a
    zero (H, D) matrix kept by the layer.  No (B, nI) gather forward, and the backward's input gradient comes out full-width
    from the GEMM (zero in the transform columns) instead of a zero fill + index_add over the batch."""

    @staticmethod
    def forward(ctx, x, weight, bias, iidx, wfull):
        # wfull (H, D): a zero buffer OWNED BY THE LAYER; only its identity columns are ever written
        wfull.index_copy_(1, iidx, weight.detach())
        ctx.save_for_backward(x, iidx)
        ctx.wfull = wfull
        return torch.addmm(bias.detach(), x, wfull.t())

    @staticmethod
    def backward(ctx, gy):
        x, iidx = ctx.saved_tensors
        gy = gy.contiguous()
        gx = gy @ ctx.wfull if ctx.needs_input_grad[0] else None
        gwf, gb = ops.linear_wgrad(gy, x, want_bias=True)
        return gx, gwf.index_select(1, iidx), gb, None, None


class ResidualBlockFn(torch.autograd.Function):
    """The plain pre-activation block y = x + W2 relu(W1 relu(x) + b1) + b2 (resnet.py:37-50 without batch norm, dropout,
    context) as ONE launch forward and ONE backward (nf_rows_block, csrc/rows_linear.hip): both weight panels in LDS, the
    intermediate goes from the first product's accumulators straight into the second product (it is written once, for the
    backward, never read back), ReLUs on registers, biases / residual / ReLU masks in the epilogues -- the library path is
    two GEMMs plus three element-wise kernels each way.  Weight and bias gradients on nf_linear_wgrad (ReLU of its second
    operand applied on load)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        x = x.contiguous()
        t, y = ops.rows_block(x, w1.detach(), b1.detach(), w2.detach(), b2.detach(), trans=False, relu=True)
        ctx.save_for_backward(x, t, w1, w2)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, t, w1, w2 = ctx.saved_tensors
        gy = gy.contiguous()
        # gt = (gy W2) * (t > 0);  gx = gy + (gt W1) * (x > 0)
        gt, gx = ops.rows_block(gy, w2.detach(), None, w1.detach(), None, trans=True, mask1=t, mask2=x, relu=False)
        from . import config
        if config.wgrad_pair:
            gw2, gb2, gw1, gb1 = ops.linear_wgrad_pair(gy, t, gt, x, relu_x=True)     # both layers: one launch + one reduction
        else:
            gw2, gb2 = ops.linear_wgrad(gy, t, want_bias=True, relu_x=True)
            gw1, gb1 = ops.linear_wgrad(gt, x, want_bias=True, relu_x=True)
        return gx, gw1, gb1, gw2, gb2


def residual_block_fused_ok(block, x):
    """True when `block` (nets.ResidualBlock) is the plain ReLU block on float32 device rows that ResidualBlockFn covers."""
    lin = block.linear_layers
    return (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and x.shape[0] >= 1024 and torch.is_grad_enabled()
            and not block.use_batch_norm and block.dropout.p == 0.0
            and (block.activation is torch.nn.functional.relu or isinstance(block.activation, torch.nn.ReLU))
            and all(l.bias is not None and l.weight.dtype == torch.float32 for l in lin)
            and lin[0].weight.shape[0] % 4 == 0 and lin[0].weight.shape[1] % 4 == 0 and lin[0].weight.shape[1] <= 128
            and lin[0].weight.shape[0] <= 128 and lin[0].weight.shape[0] == lin[0].weight.shape[1]
            and tuple(lin[1].weight.shape) == tuple(lin[0].weight.shape)
            and (x.requires_grad or any(p.requires_grad for p in block.parameters())))


# ---- affine / Glow layers: HIP forward and HIP backward (csrc/affine_bwd.hip: closed-form vector-Jacobian products); the
# torch re-evaluation `_vjp` stays for the shapes the kernels do not take (and for the row-wise Gaussian / MAF element maps) ------
def _vjp(formula, inputs, cotangents):
    """Vector-Jacobian product of `formula(*inputs) -> (y, log_det)` (plain torch, reference arithmetic) -- backward only."""
    with torch.enable_grad():
        leaves = [None if t is None else t.detach().requires_grad_(t.is_floating_point()) for t in inputs]
        outs = formula(*leaves)
        pairs = [(o, c) for o, c in zip(outs, cotangents) if c is not None and o.requires_grad]
        wanted = [t for t in leaves if t is not None and t.requires_grad]
        grads = torch.autograd.grad([o for o, _ in pairs], wanted, [c for _, c in pairs], allow_unused=True)
    it = iter(grads)
    return [None if (t is None or not t.requires_grad) else next(it) for t in leaves]


def _sum_rows(t):
    return t.reshape(t.shape[0], -1).sum(1)


class MaskedAffineFn(torch.autograd.Function):
    """MaskedAffineFlow (affine/coupling.py:209-229) on given s(b z), t(b z)."""

    @staticmethod
    def forward(ctx, z, b, s, t, direction):
        y, ld = ops.masked_affine(z, b, None if s is None else s.contiguous(), None if t is None else t.contiguous(),
                                  direction)
        ctx.save_for_backward(z, b, s, t)
        ctx.direction = direction
        return y, ld

    @staticmethod
    def backward(ctx, gy, gld):
        z, b, s, t = ctx.saved_tensors
        if gy is None:
            gy = torch.zeros_like(z)
        gz, gs, gt = ops.masked_affine_bwd(z, b, s, t, gy, gld, ctx.direction)     # csrc/affine_bwd.hip
        return gz, None, gs, gt, None


def _coupling_formula(z, param, c1, flip, scale_map, direction):
    """Split -> AffineCoupling -> Merge for the channel modes (coupling.py:117-171, reshape.py:30-85); c1 = 0: z is z2."""
    C = z.shape[1]
    if c1 == 0:
        z1, z2 = None, z
    elif not flip:
        z1, z2 = z[:, :c1], z[:, c1:]
    else:
        z1, z2 = z[:, C - c1:], z[:, :C - c1]
    if scale_map is None:
        y2 = z2 + param if direction == 0 else z2 - param
        ld = torch.zeros(z.shape[0], dtype=z.dtype, device=z.device) + 0.0 * _sum_rows(param)
    else:
        shift, s_ = param[:, 0::2, ...], param[:, 1::2, ...]
        if scale_map == "exp":
            if direction == 0:
                y2, ld = z2 * torch.exp(s_) + shift, _sum_rows(s_)
            else:
                y2, ld = (z2 - shift) * torch.exp(-s_), -_sum_rows(s_)
        else:
            scale = torch.sigmoid(s_ + 2)
            mult = (scale_map == "sigmoid_inv") == (direction == 0)   # True: multiply by scale
            if direction == 0:
                y2 = (z2 * scale if mult else z2 / scale) + shift
            else:
                y2 = (z2 - shift) * scale if mult else (z2 - shift) / scale
            ld = _sum_rows(torch.log(scale)) * (1.0 if mult else -1.0)
    if z1 is None:
        return y2, ld
    return (torch.cat([z1, y2], 1) if not flip else torch.cat([y2, z1], 1)), ld


def _leaf_fork(device, params, keep=()):
    """The side stream for launches that only feed `params`' gradients (config.train_leaf_async, _sidestream.fork), or None: switched
    off, not a GPU, or one of `params` would have its gradient read on the current stream before the join (an existing .grad to
    accumulate into, hooks)."""
    if not _config.train_leaf_async or device.type != "cuda":
        return None
    # (a non-leaf's gradient travels on through autograd nodes that run on the current stream: leaves only)
    if not all(p_.is_leaf and _sidestream.nobody_reads_early(p_) for p_ in params if p_ is not None and p_.requires_grad):
        return None
    return _sidestream.fork(device, keep=[t for t in keep if t is not None])


def _on(side):
    return contextlib.nullcontext() if side is None else torch.cuda.stream(side)


class AffineCouplingFn(torch.autograd.Function):
    """nf_affine_coupling (coupling.py:117-171 with the channel split / merge folded in) on a given `param`."""

    @staticmethod
    def forward(ctx, z, param, c1, flip, scale_map, direction):
        y, ld = ops.affine_coupling(z, param.contiguous(), c1, flip, scale_map, direction)
        ctx.save_for_backward(z, param)
        ctx.cfg = (c1, flip, scale_map, direction)
        return y, ld

    @staticmethod
    def backward(ctx, gy, gld):
        z, param = ctx.saved_tensors
        c1, flip, scale_map, direction = ctx.cfg
        if gy is None:
            gy = torch.zeros_like(z)
        if z.dim() < 2 or scale_map not in L.SCALE:
            gz, gp = _vjp(lambda z_, p_: _coupling_formula(z_, p_, *ctx.cfg), (z, param), (gy, gld))
        else:
            gz, gp = ops.affine_coupling_bwd(z, param, gy, gld, c1, flip, scale_map, direction)   # csrc/affine_bwd.hip
        return gz, gp, None, None, None, None


class ActNormFn(torch.autograd.Function):
    """AffineConstFlow / ActNorm (coupling.py:38-54) with per-channel s, t; log-det returned per sample."""

    @staticmethod
    def forward(ctx, z, s, t, direction):
        ld = torch.empty(z.shape[0], dtype=z.dtype, device=z.device)        # (written, not accumulated: no zero fill per call)
        y, _ = ops.actnorm(z, s.detach(), t.detach(), direction, logdet=ld, acc=L.LD_WRITE, want_scalar=False)
        ctx.save_for_backward(z, s, t)
        ctx.direction = direction
        return y, ld

    @staticmethod
    def backward(ctx, gy, gld):
        z, s, t = ctx.saved_tensors
        if gy is None:
            gy = torch.zeros_like(z)
        gz, gs, gt = ops.actnorm_bwd(z, s.detach(), t.detach(), gy, gld, ctx.direction)   # csrc/affine_bwd.hip
        return gz, gs.view_as(s), gt.view_as(t), None


class Inv1x1WeightFn(torch.autograd.Function):
    """(W, per-pixel log|det|) of Invertible1x1Conv's LU parametrisation in the density direction (mixing.py:88-104: W = P (tril(L, -1) +
    I) (triu(U, 1) + diag(sign_S exp(log_S)))): nf_inv1x1_assemble forward, nf_inv1x1_lu_grads backward -- two launches where torch
    autograd needs ~25 on C x C matrices."""

    @staticmethod
    def forward(ctx, P, Lm, U, sign_S, log_S):
        W, ldu = ops.inv1x1_assemble(P, Lm.detach(), U.detach(), sign_S, log_S.detach(), inverse=False)
        ctx.save_for_backward(P, Lm, U, sign_S, log_S)
        ctx.set_materialize_grads(False)
        return W, ldu

    @staticmethod
    @once_differentiable          # (the backward is a set of kernels, not a differentiable graph: double backward raises)
    def backward(ctx, gW, gl):
        P, Lm, U, sign_S, log_S = ctx.saved_tensors
        if gW is None:
            gW = torch.zeros_like(Lm)
        # (gW / gl may come from Inv1x1Fn's side-stream launch: stay on that stream, or join before reading them)
        side = _leaf_fork(Lm.device, (Lm, U, log_S), keep=(gW, gl))
        if side is None:
            _sidestream.join()
        with _on(side):
            gL, gU, gs = ops.inv1x1_lu_grads(P, Lm.detach(), U.detach(), sign_S, log_S.detach(), gW, gl)
        return None, gL, gU, None, gs


class LdFoldFn(torch.autograd.Function):
    """ld <- (((ld +- t_0) +- t_1) ...) in place, one launch (nf_ld_fold_multi): the `ld += log_det` statements of a chain of layers
    under autograd, deferred and issued together (flows/affine.lazy_ld) -- same order, same bits."""

    @staticmethod
    def forward(ctx, ld, negate, *terms):
        ops.ld_fold_multi(ld, [t.detach() for t in terms], negate)
        ctx.mark_dirty(ld)
        ctx.negate = negate
        return ld

    @staticmethod
    def backward(ctx, g):
        gn = -g if any(ctx.negate) else None
        return (g, None) + tuple(gn if s else g for s in ctx.negate)


class Inv1x1WeightsFn(torch.autograd.Function):
    """Inv1x1WeightFn for ALL the Invertible1x1Convs of a Glow level at once (round 6, late): forward(n, P_0, L_0, U_0, sign_S_0,
    log_S_0, P_1, ...) -> (W_0, ld_0, W_1, ld_1, ...) in one nf_inv1x1_assemble_multi launch per 32 layers; autograd runs the backward
    when every block's (gW, gl) exists -- the end of the level's backward -- as one nf_inv1x1_lu_grads_multi launch.  The matrices
    depend on parameters only, so nothing is gained or lost in precision or order: the same kernels' bodies per layer."""

    @staticmethod
    def forward(ctx, n, *tensors):
        layers = [tuple(t.detach() for t in tensors[5 * i:5 * i + 5]) for i in range(n)]
        out = ops.inv1x1_assemble_multi(layers)
        ctx.save_for_backward(*tensors)
        ctx.n = n
        ctx.set_materialize_grads(False)
        return tuple(t for pair in out for t in pair)

    @staticmethod
    @once_differentiable
    def backward(ctx, *grads):
        tensors = ctx.saved_tensors
        n = ctx.n
        layers = [tuple(t.detach() for t in tensors[5 * i:5 * i + 5]) for i in range(n)]
        _sidestream.join()          # (a gW may come from Inv1x1Fn's side-stream launch, config.train_leaf_async)
        gWs = [grads[2 * i] if grads[2 * i] is not None else torch.zeros_like(layers[i][1]) for i in range(n)]
        gls = [grads[2 * i + 1] for i in range(n)]
        res = ops.inv1x1_lu_grads_multi(layers, gWs, gls)
        out = [None]
        for gL, gU, gs in res:
            out += [None, gL, gU, None, gs]
        return tuple(out)


_ZERO0 = {}


def _zero_scalar(device, dtype):
    """A cached device scalar 0 (no fill launch per block and step).  Created EAGERLY, never under hipGraph capture: a
    tensor first allocated while capturing lives in the graph's private pool and its fill is only a captured node -- later eager
    users would read memory that is zeroed on replay only.  Under capture a fresh (captured) zero is returned and nothing is cached."""
    key = (device, dtype)
    zero = _ZERO0.get(key)
    if zero is None:
        if torch.cuda.is_current_stream_capturing():
            return torch.zeros((), dtype=dtype, device=device)
        zero = _ZERO0[key] = torch.zeros((), dtype=dtype, device=device)
    return zero


class Inv1x1Fn(torch.autograd.Function):
    """Per-pixel C x C product (mixing.py:106-133) with a given matrix W and per-pixel log|det| `ldu` (0-dim)."""

    @staticmethod
    def forward(ctx, z, W, ldu):
        if z.is_cuda:
            _zero_scalar(z.device, z.dtype)       # (the backward's cached zero exists before any capture of a backward can start)
        ld = torch.empty(z.shape[0], dtype=z.dtype, device=z.device)
        y, _ = ops.inv1x1_conv(z, W.detach().contiguous(), ldu.detach(), logdet=ld, acc=L.LD_WRITE, want_scalar=False)
        ctx.save_for_backward(z, W, ldu)
        return y, ld

    @staticmethod
    def backward(ctx, gy, gld):
        z, W, ldu = ctx.saved_tensors
        if gy is None:
            gy = torch.zeros_like(z)
        gz = gW = gl = None
        if z.shape[1] <= 64:
            if ctx.needs_input_grad[0]:      # gz = W^T gy per pixel: the forward kernel reading W transposed (no copy)
                gz = ops.inv1x1_conv_t(gy, W.detach())
            if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
                # feeds only W's side (Inv1x1WeightFn.backward, which follows onto the side stream, or W.grad itself)
                # -- but not W's of any other origin (slogdet's backward etc. run on the current stream)
                fn = W.grad_fn
                mine = W.is_leaf and not ldu.requires_grad
                mine = mine or (fn is not None and fn is ldu.grad_fn and fn.name().startswith("Inv1x1Weight"))       # (Inv1x1WeightFn / Inv1x1WeightsFn: both join or follow)
                side = _leaf_fork(z.device, (W,) if W.is_leaf else (), keep=(z, gy, gld)) if mine else None
                with _on(side):
                    gW, gl = ops.inv1x1_wgrad(z, gy, gld)                             # csrc/affine_bwd.hip
            return gz, gW, gl
        hw = z[0, 0].numel() if z.dim() > 2 else 1

        def formula(z_, W_, l_):
            y = torch.einsum("oc,bc...->bo...", W_, z_)
            return y, hw * l_ * torch.ones(z_.shape[0], dtype=z_.dtype, device=z_.device)

        gz, gW, gl = _vjp(formula, (z, W, ldu), (gy, gld))
        return gz, gW, gl


class SqueezeFn(torch.autograd.Function):
    """Squeeze (reshape.py:116-128) is a permutation: the backward is the opposite direction on the cotangent."""

    @staticmethod
    def forward(ctx, z, direction):
        ctx.direction = direction
        return ops.squeeze(z, direction)

    @staticmethod
    def backward(ctx, gy):
        return ops.squeeze(gy.contiguous(), 1 - ctx.direction), None


class GaussianRowsLogProbFn(torch.autograd.Function):
    """nf_diag_gaussian_log_prob_rows (base.py:326-345): per-sample (loc, log_scale) rows, picked by `idx` or row b."""

    @staticmethod
    def forward(ctx, z, loc_rows, ls_rows, idx, shift):
        ctx.save_for_backward(z, loc_rows, ls_rows, idx)
        ctx.shift = shift
        return ops.diag_gaussian_log_prob_rows(z, loc_rows.detach(), ls_rows.detach(), idx, shift)

    @staticmethod
    def backward(ctx, g):
        z, loc_rows, ls_rows, idx = ctx.saved_tensors
        shift = ctx.shift
        d = z[0].numel()

        def formula(z_, loc_, ls_):
            zz = z_.reshape(z_.shape[0], -1)
            lo = loc_ if idx is None else loc_.index_select(0, idx)
            ls = (ls_ if idx is None else ls_.index_select(0, idx)) + shift
            return (-0.5 * d * math.log(2 * math.pi) - torch.sum(ls + 0.5 * torch.pow((zz - lo) / torch.exp(ls), 2), 1),)

        gz, gl, gs = _vjp(formula, (z, loc_rows, ls_rows), (g,))
        return gz, gl, gs, None, None


class MadeFn(torch.autograd.Function):
    """MADE.forward (nets/made.py:296-304; every linear F.linear(x, weight * mask, bias), :80-81) under autograd: forward =
    nf_made_forward_train (the one-launch forward + saved pre-activations / ReLU signs), backward = nf_made_backward (input-gradient
    chain on the transposed masked weights) + nf_made_wgrad (all weight / bias gradients, one launch + a fixed-order reduction).
    `fwd` / `bwd`: the layer's packs as built at forward time (new tensors are built when a parameter changes, so the ones held here
    stay what this graph's forward used); params = weight, bias of the initial layer, the blocks' linears, the final layer.
    A plain ReLU ResidualNet (nets/resnet.py:53-104) is the same network without masks and runs through the same three kernels
    (flows/made_pack.pack_resnet_forward / pack_resnet_backward)."""

    @staticmethod
    def forward(ctx, fwd, bwd, x, *params):
        blob, table, hp = fwd[:3]
        x = x.contiguous()
        out, save, bits = ops.made_forward_train(x, blob, table, hp, bwd["MD"], bwd["NB"])
        ctx.save_for_backward(x, save, bits)
        ctx.bwd = bwd
        ctx.shapes = [tuple(p.shape) for p in params]
        return out

    @staticmethod
    @once_differentiable          # (the backward is a set of kernels, not a differentiable graph: double backward raises)
    def backward(ctx, gout):
        x, save, bits = ctx.saved_tensors
        bwd = ctx.bwd
        gout = gout.contiguous()
        gx, G = ops.made_backward(gout, bits, bwd["blob"], bwd["table"], x.shape[1], bwd["Hp"], bwd["NB"])
        grads = [None] * len(ctx.shapes)
        if any(ctx.needs_input_grad[3:]):
            flat = ops.made_wgrad(gout, x, G, save, bwd["wtable"], bwd["stable"], bwd["mask"], bwd["ntiles"], bwd["nflat"],
                                  bwd["Mp"], bwd["Dx"])
            for k, (woff, shape, boff, n) in enumerate(bwd["offsets"]):
                grads[2 * k] = flat[woff:woff + shape[0] * shape[1]].view(shape)
                grads[2 * k + 1] = flat[boff:boff + n]
        return (None, None, gx if ctx.needs_input_grad[2] else None) + tuple(grads)


class ConvNetFn(torch.autograd.Function):
    """GlowBlock's conditioner ConvNet2d([Cin, hidden, hidden, Cout], kernels (3, 1, 3), LeakyReLU(0); nets/cnn.py:5-63) under
    autograd (flows/affine/glow.py:10-100 inside core.py:87-102) without the convolution library: pixels are rows, the 3x3
    convolutions a gather in front (nf_conv3x3_gather) and a neighbour sum behind (nf_conv3x3_gather_sum) of a per-pixel MLP
    9 Cin -> hidden -> hidden -> 9 Cout, which runs -- forward, input-gradient chain, weight gradients -- on the MADE training kernels
    in plain-MLP mode (csrc/made_fwd.hip EPI 3, csrc/made_bwd.hip; packs: flows/made_pack.pack_mlp_*)."""

    @staticmethod
    def forward(ctx, fwd, bwd, x, w1, b1, w2, b2, w3, b3):
        B, Cin, H, W = x.shape
        Cout = w3.shape[0]
        R = B * H * W
        col = ops.conv3x3_gather(x, ld=bwd["Dx"])                 # (R up to 64, 9 Cin up to 128): also the weight gradient's operand
        P, save, bits = ops.made_forward_train(col, fwd[0], fwd[1], fwd[2], 9 * Cout, 1, rows=R, features=9 * Cin)
        out = ops.conv3x3_gather_sum(P, b3.detach(), (B, Cout, H, W))
        ctx.save_for_backward(col, save, bits)
        ctx.bwd, ctx.shape = bwd, (B, Cin, H, W)
        ctx.wshapes = (tuple(w1.shape), tuple(w2.shape), tuple(w3.shape))
        ctx.params = (w1, b1, w2, b2, w3, b3)           # (asked at backward time whether anything reads their gradients early)
        return out

    @staticmethod
    @once_differentiable          # (the backward is a set of kernels, not a differentiable graph: double backward raises)
    def backward(ctx, gout):
        col, save, bits = ctx.saved_tensors
        bwd = ctx.bwd
        B, Cin, H, W = ctx.shape
        hid, Cout = ctx.wshapes[0][0], ctx.wshapes[2][0]
        R = B * H * W
        gout = gout.contiguous()
        gP = ops.conv3x3_gather(gout, flip=True, ld=bwd["Mp"])
        gcol, G = ops.made_backward(gP, bits, bwd["blob"], bwd["table"], 9 * Cin, bwd["Hp"], 1, rows=R, out_features=9 * Cout,
                                    ld_out=bwd["Dx"])
        gx = ops.conv3x3_gather_sum(gcol, None, (B, Cin, H, W), flip=True) if ctx.needs_input_grad[2] else None
        if not any(ctx.needs_input_grad[3:]):        # frozen conditioner: no weight-gradient launch
            return (None, None, gx) + (None,) * 6
        # weight / bias gradients: nothing downstream of this block's backward waits for them (config.train_leaf_async)
        side = _leaf_fork(gout.device, ctx.params, keep=(gP, col, G, save, gout))
        with _on(side):
            flat = ops.made_wgrad(gP, col, G, save, bwd["wtable"], bwd["stable"], bwd["mask"], bwd["ntiles"], bwd["nflat"], bwd["Mp"],
                                  bwd["Dx"], rows=R)
            gb3 = ops.channel_sum(gout)
        (o0, s0, c0, n0), (o1, s1, c1, n1), (o2, s2, _, _) = bwd["offsets"]
        # (the reduction scatters straight into the conv parameters' own (o, c, ky, kx) layouts: made_pack.convnet_train_structure)
        gw1 = flat[o0:o0 + s0[0] * s0[1]].view(hid, Cin, 3, 3)
        gw2 = flat[o1:o1 + s1[0] * s1[1]].view(hid, hid, 1, 1)
        gw3 = flat[o2:o2 + s2[0] * s2[1]].view(Cout, hid, 3, 3)
        return None, None, gx, gw1, flat[c0:c0 + n0], gw2, flat[c1:c1 + n1], gw3, gb3


class MafInverseFn(torch.autograd.Function):
    """MaskedAffineAutoregressive.inverse (affine/autoregressive.py:29-38 + :114-128: the DENSITY direction of the reference's MAF, D
    sequential MADE passes) under autograd WITHOUT differentiating through the D passes.  x = T^-1(z) satisfies T(x; theta) = z with
    T the single-pass direction (z = s(x) x + t(x), log|det| = sum log s), so for the cotangents (g_x, g_ld) of (x, -sum log s):

        v solves   v s + J^T g_p(v, g_ld) = g_x      (J = dMADE/dx; g_p(a, c) = the affine transform's parameter cotangent for
                                                      cotangent a on z and c on sum log s: nf_maf_affine_bwd)
        g_z = v,   g_theta = MADE^T-weight-gradient of g_p(-v, -g_ld)

    J^T is strictly upper triangular in the feature order (feature i only feeds features > i), so the iteration v <- (g_x - J^T g_p(v,
    g_ld)) / s is EXACT after at most D sweeps (component i is final once the components > i are) and usually stops changing -- bit for
    bit, the kernels are deterministic -- much earlier.  Forward: the one-pass inverse kernel (nf_maf_inverse_h); backward: one
    nf_made_forward_train at x, one nf_made_backward per sweep, ONE nf_made_wgrad: memory of a single MADE pass instead of D."""

    @staticmethod
    def forward(ctx, inv, fwd, bwd, z, *params):
        z = z.contiguous()
        if isinstance(inv, dict):       # round 5: format-0 inverse that leaves its ReLU masks + the transposed pack of the one-pass solve
            keep = inv.get("fcols") is not None and _config.maf_solve_grads and any(ctx.needs_input_grad[4:])
            # round 6: the pass stores MADE's output at the solution itself (nf_maf_inverse_h_train) when the backward will read the
            # scratches in place -- otherwise it is recomputed there from the last hidden tensor (round 5)
            pw = inv.get("pw") if _config.maf_wgrad_in_place else None
            want_p = keep and pw is not None and z.shape[0] % 64 == 0 and pw["positions"] == inv["hp"]
            out = ops.maf_inverse_bits(z, inv["blob"], inv["table"], inv["hp"], inv["nb"], inv["tiles"],
                                       table_host=inv.get("table_host"), return_scratch=True, want_params=want_p)
            x, ld, bits, scratch = out[:4]
            ctx.prm = out[4] if want_p else None
            ctx.save_for_backward(x, bits, *([params[-1]] if params else []))
            ctx.tpack = (inv["tblob"], inv["ttable"], inv["hp"], inv["nb"], inv.get("gcols"))
            ctx.tth = inv.get("ttable_host")
            ctx.pw = inv.get("pw") if _config.maf_wgrad_in_place else None
            # the pass's own activations (its scratch) are the inputs of MADE's linears at x: kept for the weight-gradient launch, so
            # the backward does not run MADE forward again (671 MB per config-5 layer at B = 65 536 instead of a transient of that size)
            # (the masked final weight in the training kernels' column order: only where the backward rebuilds MADE's output itself)
            wf_t = None
            if keep and not want_p:
                wf_t = inv.get("wf_t")
                if wf_t is None and inv.get("wf_src") is not None:
                    wf_t = ops.pack_gather(list(params), inv["wf_src"].view(-1)).view(inv["wf_src"].shape)
            ctx.fpack = (scratch, inv["fcols"], wf_t) if keep else None
        else:
            x, ld = ops.maf_inverse(z, inv[0], inv[1], inv[2], num_blocks=inv[3], table_host=inv[4] if len(inv) > 4 else None)
            ctx.save_for_backward(x, *([params[-1]] if params else []))
            ctx.tpack = None
        ctx.fwd, ctx.bwd = fwd, bwd
        ctx.nparams = len(params)
        # (the final layer's bias -- the last of [w, b] per linear -- goes through save_for_backward: an in-place update between forward
        # and backward trips autograd's version check instead of mixing new bias with old weights, ADVICE r05.  The kept scratch
        # (ctx.fpack: 671 MB per config-5 layer at B = 65 536, for every MAF layer from forward to backward) is what config.
        # set_maf_solve_grads(False) trades for one more MADE forward + chain per layer in the backward.)
        ctx.has_bias_f = bool(params)
        ctx.set_materialize_grads(False)
        return x, ld

    @staticmethod
    @once_differentiable
    def backward(ctx, gx, gld):
        x = ctx.saved_tensors[0]
        fwd, bwd = ctx.fwd, ctx.bwd
        B, D = x.shape
        gx = torch.zeros_like(x) if gx is None else gx.contiguous()
        gld = torch.zeros(B, dtype=x.dtype, device=x.device) if gld is None else gld.contiguous()
        fpack = getattr(ctx, "fpack", None)
        if ctx.tpack is not None and fpack is not None and ctx.tpack[4] is not None:
            # round 5: nothing of MADE runs forward again -- `save` (the linears' inputs) from the inverse pass's own scratch, the
            # parameters p = h_NB Wf^T + bf from its last hidden tensor by one library product, the hidden gradients from the solve's
            tb, tt, hp, nb, gcols = ctx.tpack
            fscratch, fcols, wf_t = fpack
            ctx.fpack = None
            bias_f = ctx.saved_tensors[-1].detach() if ctx.has_bias_f else None
            pw = getattr(ctx, "pw", None)
            if pw is not None and B % 64 == 0 and pw["positions"] == hp:
                # round 6: the weight-gradient launch reads both scratches where they are (nf_made_wgrad_pos: problems, tiles and
                # scatter maps over scratch positions) -- only the last hidden tensor is still laid out in rows, for MADE's output
                p = getattr(ctx, "prm", None)
                ctx.prm = None
                if p is None and wf_t is not None:
                    h_last = ops.maf_scratch_layer(fscratch, fcols, B, nb, hp, 2 * nb)
                    p = torch.nn.functional.linear(h_last[:B], wf_t, bias_f)
                elif p is None:       # (a switch of config.py flipped between forward and backward: one MADE pass at the solution)
                    p = ops.made_forward_train(x, fwd[0], fwd[1], fwd[2], 2 * D, bwd["NB"])[0]
                v, scratch = ops.maf_solve_t(x, p, gx, gld, ctx.saved_tensors[1], tb, tt, hp, nb, return_scratch=True, table_host=getattr(ctx, "tth", None))
                MafInverseFn.last_sweeps = 1
                return MafInverseFn._finish(ctx, x, p, v, gld, None, None, None, pos=(scratch, fscratch, pw))
            save = ops.maf_scratch_rows(fscratch, fcols, B, nb, hp)
            p = torch.nn.functional.linear(save[2 * nb, :B], wf_t, bias_f)
            v, scratch = ops.maf_solve_t(x, p, gx, gld, ctx.saved_tensors[1], tb, tt, hp, nb, return_scratch=True, table_host=getattr(ctx, "tth", None))
            G = ops.maf_scratch_rows(scratch, gcols, B, nb, hp, sign=-1.0, reverse_layers=True)
            MafInverseFn.last_sweeps = 1
            return MafInverseFn._finish(ctx, x, p, v, gld, save, None, G)
        p, save, bits = ops.made_forward_train(x, fwd[0], fwd[1], fwd[2], 2 * D, bwd["NB"])
        if ctx.tpack is not None:
            # ONE launch: back-substitution of v s + J^T g_p(v, g_ld) = g_x on the transposed pack (nf_maf_solve_t), the ReLU masks from
            # the forward inverse's own pass; no sweeps, no host read-back (hipGraph-capturable)
            tb, tt, hp, nb, gcols = ctx.tpack
            G = None
            if gcols is not None and _config.maf_solve_grads and any(ctx.needs_input_grad[4:]):
                # the solve's activation scratch IS the input-gradient chain at the solution: rearranged into the weight-gradient
                # launch's G (sign: the parameter cotangent below is g_p(-v, -g_ld)) instead of one more nf_made_backward pass
                v, scratch = ops.maf_solve_t(x, p, gx, gld, ctx.saved_tensors[1], tb, tt, hp, nb, return_scratch=True, table_host=getattr(ctx, "tth", None))
                G = ops.maf_scratch_rows(scratch, gcols, B, nb, hp, sign=-1.0, reverse_layers=True)
            else:
                v = ops.maf_solve_t(x, p, gx, gld, ctx.saved_tensors[1], tb, tt, hp, nb, table_host=getattr(ctx, "tth", None))
            MafInverseFn.last_sweeps = 1
            return MafInverseFn._finish(ctx, x, p, v, gld, save, bits, G)
        v = torch.empty_like(x)
        gp = torch.empty_like(p)
        changed = torch.zeros(1, dtype=torch.int32, device=x.device)
        rtol = _config.maf_implicit_rtol
        gxm, sweeps = None, 0
        # Under stream capture (a whole training step recorded into one hipGraph) nothing may be read back: the sweep count is
        # then FIXED at its exact bound D (J^T is strictly triangular: D sweeps are exact whatever the data) -- slower than the
        # early exit, but capturable and host-synchronisation free (autograd.py:890).
        capturing = torch.cuda.is_current_stream_capturing()
        # sweep 0 = the start value v = g_x / scale; then one element-wise launch (update of v, the next cotangent g_p, "did v move?")
        # and one chain per sweep; the flag is read back every other sweep (a read is a host synchronisation)
        while sweeps <= D:
            v_prev = v.clone() if rtol > 0.0 and sweeps > 0 and not capturing else None
            ops.maf_implicit_sweep(x, p, gx, gld, gxm, v, gp, changed)
            if capturing:
                pass
            elif sweeps > 0 and (sweeps % 2 == 0 or sweeps == D):
                if rtol > 0.0:
                    done = bool((v - v_prev).abs().max() <= rtol * v.abs().max())
                else:
                    done = int(changed.item()) == 0
                if done:
                    break
                changed.zero_()
            elif sweeps > 0 and sweeps % 2 == 1:
                changed.zero_()          # (only the latest sweep's verdict counts)
            gxm, _ = ops.made_backward(gp, bits, bwd["blob"], bwd["table"], D, bwd["Hp"], bwd["NB"], want_G=False)
            sweeps += 1
        MafInverseFn.last_sweeps = sweeps        # (debug / bench read-out only; not used by any computation)
        return MafInverseFn._finish(ctx, x, p, v, gld, save, bits)

    @staticmethod
    def _finish(ctx, x, p, v, gld, save, bits, G=None, pos=None):
        """g_z = v; g_theta = MADE's weight gradients for the parameter cotangent g_p(-v, -g_ld): one chain (unless the one-pass solve
        left its hidden gradients: G, or both scratches are read in place: pos) + ONE weight-gradient launch."""
        bwd = ctx.bwd
        D = x.shape[1]
        _, gp = ops.maf_affine_bwd(x, p, -v, -gld, 0)
        grads = [None] * ctx.nparams
        if any(ctx.needs_input_grad[4:]):
            if pos is not None:
                gscratch, fscratch, pw = pos
                flat = ops.made_wgrad_pos(gp, x, gscratch, fscratch, pw["wtable"], pw["stable"], bwd["mask"], pw["ntiles"], bwd["nflat"],
                                          bwd["Mp"], bwd["Dx"], pw["NL"], pw["positions"])
            else:
                if G is None:
                    _, G = ops.made_backward(gp, bits, bwd["blob"], bwd["table"], D, bwd["Hp"], bwd["NB"])
                flat = ops.made_wgrad(gp, x, G, save, bwd["wtable"], bwd["stable"], bwd["mask"], bwd["ntiles"], bwd["nflat"], bwd["Mp"],
                                      bwd["Dx"])
            for k, (woff, shape, boff, n) in enumerate(bwd["offsets"]):
                grads[2 * k] = flat[woff:woff + shape[0] * shape[1]].view(shape)
                grads[2 * k + 1] = flat[boff:boff + n]
        return (None, None, None, v if ctx.needs_input_grad[3] else None) + tuple(grads)


class ArInverseImplicitFn(torch.autograd.Function):
    """`Autoregressive.inverse` (autoregressive.py:29-40: D passes of the autoregressive net, each recorded by autograd in the
    reference) under autograd for ANY element-wise transform -- the autoregressive spline layer's sampling direction
    (neural_spline/autoregressive.py:94-134 through wrapper.py:140-155), the circular variant, MAF structures the one-pass kernels do
    not take -- differentiated IMPLICITLY like MafInverseFn, on the layer's own density-direction graph instead of special kernels:

        forward   x, ld = layer.inverse(z) without a graph (one launch where nf_arnsf_inverse / nf_maf_inverse_h apply)
        backward  with F(x, W) = f(x; theta(x, W)) (= z) and l(x, W) = log|dF/dx| (ld = -l) rebuilt ONCE at the final x
                  (one autoregressive-net pass + the element-wise transform, parameters frozen):
                  A^T v = g_x - g_ld dl/dx,  A^T = diag(s) + N^T,  s = df/dx at fixed theta,  N = (df/dtheta)(dtheta/dx)
                  by v <- v + (g_x - VJP_x[(F, l); (v, g_ld)]) / s: N is strictly triangular in the feature order, the iteration is
                  EXACT after at most D updates and stops earlier when v no longer moves (config.maf_implicit_rtol; checked every
                  other sweep); then g_z = v and g_W = VJP_W[(F, l); (-v, -g_ld)] on a second pass with the parameters attached.

    One backward of the net per sweep and ONE weight-gradient pass instead of D forward + D backward passes and D saved graphs."""
    last_sweeps = 0

    @staticmethod
    def eligible(layer, inputs, context):
        if context is not None or inputs.dim() != 2 or not _config.ar_implicit:
            return False
        for m in layer.modules():
            if m.training and (isinstance(m, torch.nn.modules.batchnorm._BatchNorm) or (isinstance(m, torch.nn.Dropout) and m.p > 0)):
                return False      # (the graph rebuilt in backward must be THE function the forward inverted)
        return True

    @staticmethod
    def forward(ctx, layer, z, *params):
        with torch.no_grad():
            x, ld = layer.inverse(z)
        ctx.layer, ctx.params = layer, params
        ctx.save_for_backward(x, *params)          # (the parameters: autograd's in-place-modification check between forward and backward)
        ctx.set_materialize_grads(False)
        return x, ld

    @staticmethod
    @once_differentiable
    def backward(ctx, gx, gld):
        x = ctx.saved_tensors[0]
        layer, params = ctx.layer, ctx.params
        D = x.shape[1]
        net = layer.autoregressive_net
        capturing = x.is_cuda and torch.cuda.is_current_stream_capturing()
        rtol = _config.maf_implicit_rtol
        with torch.enable_grad():
            xa = x.detach().requires_grad_(True)          # the transform's own argument
            xb = x.detach().requires_grad_(True)          # the autoregressive net's input
            frozen = {n: p.detach() for n, p in net.named_parameters()}
            theta = torch.func.functional_call(net, frozen, (xb,))
            zf, lf = layer._elementwise(xa, theta, 0)
            (s,) = torch.autograd.grad([zf], [xa], [torch.ones_like(zf)], retain_graph=True)
            rhs = torch.zeros_like(x) if gx is None else gx
            if gld is not None:       # the log-determinant's direct dependence on x: independent of v, taken once (none for affine maps)
                (dl,) = torch.autograd.grad([lf], [xa], [gld], retain_graph=True, allow_unused=True)
                if dl is not None:
                    rhs = rhs - dl
            outs, v, sweeps = ([zf] if gld is None else [zf, lf]), rhs / s, 1
            # v_new = (rhs - N^T v - g_ld dl/dx_b) / s reads v only through the net (components > i of v for component i): a component
            # is final -- and equal to back-substitution's -- one sweep after the components it depends on are
            while sweeps < D + 1:
                (gb,) = torch.autograd.grad(outs, [xb], [v] if gld is None else [v, gld], retain_graph=True, allow_unused=True)
                if gb is None:
                    break
                v_new = (rhs - gb) / s
                sweeps += 1
                done = False
                if not capturing and (sweeps % 2 == 0 or sweeps == D + 1):
                    if rtol > 0.0:
                        done = float((v_new - v).abs().max()) <= rtol * float(v_new.abs().max())
                    else:
                        done = bool(torch.equal(v_new, v))
                v = v_new
                if done:
                    break
            ArInverseImplicitFn.last_sweeps = sweeps
            del zf, lf, theta, outs, xa, xb
            grads = [None] * len(params)
            want = [i for i, p in enumerate(params) if ctx.needs_input_grad[2 + i]]
            if want:
                zf, lf = layer.forward(x.detach())
                got = torch.autograd.grad([zf] if gld is None else [zf, lf], [params[i] for i in want],
                                          [-v] if gld is None else [-v, -gld], allow_unused=True)
                for i, g in zip(want, got):
                    grads[i] = g
        return (None, v if ctx.needs_input_grad[1] else None) + tuple(grads)


class MafAffineFn(torch.autograd.Function):
    """nf_maf_affine (affine/autoregressive.py:98-128) on given MADE output `params` (B, 2D); backward = nf_maf_affine_bwd."""

    @staticmethod
    def forward(ctx, x, params, direction):
        y, ld = ops.maf_affine(x, params.contiguous(), direction)
        ctx.save_for_backward(x, params)
        ctx.direction = direction
        ctx.set_materialize_grads(False)
        return y, ld

    @staticmethod
    @once_differentiable          # (the backward is a set of kernels, not a differentiable graph: double backward raises)
    def backward(ctx, gy, gld):
        x, params = ctx.saved_tensors
        gx, gp = ops.maf_affine_bwd(x, params, gy, gld, ctx.direction)
        return gx, gp, None
