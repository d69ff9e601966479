"""Tensor-level wrappers: one Python function per C-ABI entry point of include/nf_mi355x.h.

Each wrapper validates devices/dtypes, allocates outputs with torch.empty on the input's device and
enqueues exactly one kernel on torch's current HIP stream.  No arithmetic happens in Python.
"""
import ctypes as C
import math

import torch

from . import _lib as L
from ._lib import f64, i32, i64, ptr


def _ld_buffer(logdet, B, like):
    if logdet is None:
        return torch.empty(B, dtype=like.dtype, device=like.device), L.LD_WRITE
    return logdet, None


def rqs_spline(x, w, h, d, inverse=False, tails="linear", tail_bound=1.0, left=0.0, right=1.0, bottom=0.0, top=1.0,
               min_bin_width=1e-3, min_bin_height=1e-3, min_derivative=1e-3, wh_div=1.0):
    """utils/splines.py:16-97 / :100-219.  x (...,), w/h (..., K), d (..., K-1|K|K+1); last-dim strided views of one
    parameter block are accepted (row stride taken from .stride(-2))."""
    L.require_device(x, w, h, d)
    K = w.shape[-1]
    xs = x.contiguous().view(-1)
    N = xs.numel()

    def rows(a):
        a2 = a.reshape(N, a.shape[-1]) if N > 0 else a.reshape(0, a.shape[-1])
        if a2.stride(-1) != 1:
            a2 = a2.contiguous()
        return a2

    w2, h2, d2 = rows(w), rows(h), rows(d)
    y = torch.empty_like(xs)
    lad = torch.empty_like(xs)
    rc = L.lib().nf_rqs_spline(ptr_any(xs), ptr_any(w2), i64(w2.stride(0) if N else K), ptr_any(h2),
                               i64(h2.stride(0) if N else K), ptr_any(d2), i64(d2.stride(0) if N else 1), ptr_any(y),
                               ptr_any(lad), i64(N), i32(K), i32(L.TAILS[tails]), f64(tail_bound), f64(left), f64(right),
                               f64(bottom), f64(top), f64(min_bin_width), f64(min_bin_height), f64(min_derivative),
                               f64(wh_div), i32(int(inverse)), i32(L.dtype_code(x)), L.stream())
    L.check(rc, "nf_rqs_spline")
    from . import config
    if config.debug_checks and N > 0:      # device-side flags, read back only in debug mode (config.set_debug_checks)
        flags = torch.zeros(1, dtype=torch.int32, device=x.device)
        rc = L.lib().nf_rqs_spline_check(ptr_any(xs), ptr_any(y), i64(N), i32(L.TAILS[tails]), f64(tail_bound), f64(left), f64(right),
                                         f64(bottom), f64(top), i32(int(inverse)), i32(L.dtype_code(x)), ptr_any(flags), L.stream())
        L.check(rc, "nf_rqs_spline_check")
        f = int(flags.item())
        if f & 1:
            raise RuntimeError("rational_quadratic_spline: input outside the domain with tails=None (the reference's gather fails on "
                               "bin index -1 / K, utils/splines.py:154-160)")
        if f & 2:
            raise AssertionError("rational_quadratic_spline: negative discriminant in the inverse direction (utils/splines.py:181)")
    return y.view(x.shape), lad.view(x.shape)


def ptr_any(t):
    """Device pointer of a tensor whose rows may be strided (last dim unit stride)."""
    import ctypes
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def rqs_coupling(x, cond, uw, uh, ud, identity_idx, transform_idx, K, mode, y=None, logdet=None, acc=None,
                 tails="linear", tail_bound=3.0, min_bin_width=1e-3, min_bin_height=1e-3, min_derivative=1e-3,
                 wh_div=1.0, tails_t=None, bound_t=None, tails_i=None, bound_i=None):
    """nsf/coupling.py:71-128 given the conditioner output `cond` (B, nT*M) or (B, nT, M).  tails="feature" with
    int32 tensors tails_t / tails_i (1 linear, 2 circular) and / or per-feature bound tensors bound_t / bound_i select
    the per-feature variant (nf_rqs_coupling_ft, utils/splines.py:48-66)."""
    L.require_device(x, cond, uw, uh, ud, identity_idx, transform_idx, tails_t, bound_t, tails_i, bound_i)
    B, D = x.shape
    x = x.contiguous()
    if y is None:
        y = torch.empty_like(x)
    if logdet is None:
        logdet = torch.empty(B, dtype=x.dtype, device=x.device)
        acc = L.LD_WRITE
    elif acc is None:
        acc = L.LD_ADD
    per_feature = tails == "feature" or bound_t is not None or bound_i is not None
    if not per_feature:
        rc = L.lib().nf_rqs_coupling(ptr(x), ptr(y), ptr(logdet), ptr(cond), ptr(uw), ptr(uh), ptr(ud),
                                     ptr(identity_idx), i32(identity_idx.numel()), ptr(transform_idx),
                                     i32(transform_idx.numel()), i64(B), i32(D), i32(K), i32(L.TAILS[tails]),
                                     f64(tail_bound), f64(min_bin_width), f64(min_bin_height), f64(min_derivative),
                                     f64(wh_div), i32(mode), i32(acc), i32(L.dtype_code(x)), L.stream())
        L.check(rc, "nf_rqs_coupling")
        return y, logdet
    fix_b = lambda t: None if t is None else t.to(device=x.device, dtype=x.dtype).contiguous()
    fix_t = lambda t: None if t is None else t.to(device=x.device, dtype=torch.int32).contiguous()
    bt, bi, tt, ti = fix_b(bound_t), fix_b(bound_i), fix_t(tails_t), fix_t(tails_i)
    code = 3 if tails == "feature" else L.TAILS[tails]
    scalar_bound = float(tail_bound) if not torch.is_tensor(tail_bound) else 1.0
    rc = L.lib().nf_rqs_coupling_ft(ptr(x), ptr(y), ptr(logdet), ptr(cond), ptr(uw), ptr(uh), ptr(ud),
                                    ptr(identity_idx), i32(identity_idx.numel()), ptr(transform_idx),
                                    i32(transform_idx.numel()), i64(B), i32(D), i32(K), i32(code), f64(scalar_bound),
                                    f64(min_bin_width), f64(min_bin_height), f64(min_derivative), f64(wh_div),
                                    i32(mode), i32(acc), i32(L.dtype_code(x)), ptr(tt), ptr(bt), ptr(ti), ptr(bi),
                                    L.stream())
    L.check(rc, "nf_rqs_coupling_ft")
    return y, logdet


def lu_linear_permute(x, perm, lower_entries, upper_entries, unconstrained_upper_diag, bias, direction, eps=1e-3,
                      logdet=None, acc=None):
    """mixing.py:535-563.  direction 0 = density (LULinearPermute.inverse), 1 = sample (.forward)."""
    L.require_device(x, perm, lower_entries, upper_entries, unconstrained_upper_diag, bias)
    B, D = x.shape
    x = x.contiguous()
    y = torch.empty_like(x)
    if logdet is None:
        logdet = torch.empty(B, dtype=x.dtype, device=x.device)
        acc = L.LD_WRITE
    elif acc is None:
        acc = L.LD_ADD
    rc = L.lib().nf_lu_linear_permute(ptr(x), ptr(y), ptr(logdet), ptr(perm), ptr(lower_entries), ptr(upper_entries),
                                      ptr(unconstrained_upper_diag), ptr(bias), i64(B), i32(D), f64(eps),
                                      i32(direction), i32(acc), i32(L.dtype_code(x)), L.stream())
    L.check(rc, "nf_lu_linear_permute")
    return y, logdet


def masked_affine(z, b, s, t, direction, logdet=None, acc=None):
    """affine/coupling.py:209-229.  b broadcastable to z.shape[1:]; s, t same shape as z or None."""
    L.require_device(z, b, s, t)
    z = z.contiguous()
    B = z.shape[0]
    inner = z[0].numel() if B else int(math.prod(z.shape[1:]))
    bb = b.to(z.dtype)
    if bb.numel() != inner:
        bb = bb.expand((1,) + tuple(z.shape[1:]))
    bb = bb.contiguous().view(-1)
    y = torch.empty_like(z)
    if logdet is None:
        logdet = torch.empty(B, dtype=z.dtype, device=z.device)
        acc = L.LD_WRITE
    elif acc is None:
        acc = L.LD_ADD
    s = None if s is None else s.contiguous()
    t = None if t is None else t.contiguous()
    rc = L.lib().nf_masked_affine(ptr(z), ptr(bb), ptr(s), ptr(t), ptr(y), ptr(logdet), i64(B), i64(inner),
                                  i32(direction), i32(acc), i32(L.dtype_code(z)), L.stream())
    L.check(rc, "nf_masked_affine")
    return y, logdet


def affine_coupling(z, param, c1, flip, scale_map, direction, logdet=None, acc=None, param_bias=None):
    """affine/coupling.py:117-171 + channel Split/Merge.  z (B, C, *spatial), param (B, P, *spatial); param_bias (P)
    is added to param inside the kernel (bias of a bias-free last convolution)."""
    L.require_device(z, param, param_bias)
    z = z.contiguous()
    param = param.contiguous()
    B, Cc = z.shape[:2]
    HW = int(math.prod(z.shape[2:])) if z.dim() > 2 else 1
    y = torch.empty_like(z)
    if logdet is None:
        logdet = torch.empty(B, dtype=z.dtype, device=z.device)
        acc = L.LD_WRITE
    elif acc is None:
        acc = L.LD_ADD
    rc = L.lib().nf_affine_coupling_pb(ptr(z), ptr(param), ptr(None if param_bias is None else param_bias.contiguous()),
                                       ptr(y), ptr(logdet), i64(B), i32(Cc), i32(c1), i32(int(flip)), i64(HW),
                                       i32(L.SCALE[scale_map]), i32(direction), i32(acc), i32(L.dtype_code(z)), L.stream())
    L.check(rc, "nf_affine_coupling_pb")
    return y, logdet


def actnorm(z, s, t, direction, logdet=None, acc=None, want_scalar=True):
    """affine/coupling.py:38-54 with s, t of C elements (shape (1,C,1,..,1) flattened)."""
    L.require_device(z, s, t)
    z = z.contiguous()
    B, Cc = z.shape[:2]
    HW = int(math.prod(z.shape[2:])) if z.dim() > 2 else 1
    y = torch.empty_like(z)
    lds = torch.empty((), dtype=z.dtype, device=z.device) if want_scalar else None
    if logdet is not None and acc is None:
        acc = L.LD_ADD
    rc = L.lib().nf_actnorm(ptr(z), ptr(s.contiguous().view(-1)), ptr(t.contiguous().view(-1)), ptr(y), ptr(lds),
                            ptr(logdet), i64(B), i32(Cc), i64(HW), i32(direction), i32(acc or 0), i32(L.dtype_code(z)),
                            L.stream())
    L.check(rc, "nf_actnorm")
    return y, lds


def actnorm_stats(z):
    L.require_device(z)
    z = z.contiguous()
    B, Cc = z.shape[:2]
    HW = int(math.prod(z.shape[2:])) if z.dim() > 2 else 1
    mean = torch.empty(Cc, dtype=z.dtype, device=z.device)
    std = torch.empty(Cc, dtype=z.dtype, device=z.device)
    rc = L.lib().nf_actnorm_stats(ptr(z), ptr(mean), ptr(std), i64(B), i32(Cc), i64(HW), i32(L.dtype_code(z)),
                                  L.stream())
    L.check(rc, "nf_actnorm_stats")
    return mean, std


def actnorm_init(mean, std, s_out, t_out, direction):
    L.require_device(mean, std, s_out, t_out)
    rc = L.lib().nf_actnorm_init(ptr(mean), ptr(std), ptr(s_out), ptr(t_out), i32(mean.numel()), i32(direction),
                                 i32(L.dtype_code(mean)), L.stream())
    L.check(rc, "nf_actnorm_init")


def inv1x1_assemble(P, Lm, U, sign_S, log_S, inverse):
    L.require_device(P, Lm, U, sign_S, log_S)
    Cc = Lm.shape[0]
    W = torch.empty((Cc, Cc), dtype=Lm.dtype, device=Lm.device)
    ldu = torch.empty((), dtype=Lm.dtype, device=Lm.device)
    rc = L.lib().nf_inv1x1_assemble(ptr(P.contiguous()), ptr(Lm.contiguous()), ptr(U.contiguous()),
                                    ptr(sign_S.contiguous()), ptr(log_S.contiguous()), ptr(W), ptr(ldu), i32(Cc),
                                    i32(int(inverse)), i32(L.dtype_code(Lm)), L.stream())
    L.check(rc, "nf_inv1x1_assemble")
    return W, ldu


def inv1x1_lu_grads(P, Lm, U, sign_S, log_S, gW, gl):
    """(gL, gU, g_log_S) of Invertible1x1Conv's LU parametrisation in the density direction (nf_inv1x1_lu_grads)."""
    L.require_device(P, Lm, U, sign_S, log_S, gW, gl)
    Cc = Lm.shape[0]
    gL, gU, gs = torch.empty_like(Lm), torch.empty_like(U), torch.empty_like(log_S)
    rc = L.lib().nf_inv1x1_lu_grads(ptr(P.contiguous()), ptr(Lm.contiguous()), ptr(U.contiguous()), ptr(sign_S.contiguous()),
                                    ptr(log_S.contiguous()), ptr(gW.contiguous()), ptr(None if gl is None else gl.contiguous()),
                                    ptr(gL), ptr(gU), ptr(gs), i32(Cc), i32(L.dtype_code(Lm)), L.stream())
    L.check(rc, "nf_inv1x1_lu_grads")
    return gL, gU, gs


def _ptr_array(tensors):
    """A ctypes array of device pointers (NULL for None) for the *_multi entry points; the tensors must stay alive through the call."""
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


def ld_fold_multi(ld, terms, negate):
    """nf_ld_fold_multi: ld (B) float32 updated IN PLACE with the (B) terms in order (negate[i]: subtracted)."""
    L.require_device(ld, *terms)
    if ld.dtype != torch.float32 or not ld.is_contiguous() or any(t.dtype != torch.float32 or t.shape != ld.shape for t in terms):
        raise NotImplementedError("ld_fold_multi: contiguous float32 vectors of one length")
    terms = [t.contiguous() for t in terms]
    neg = (C.c_int * len(terms))(*[1 if x else 0 for x in negate])
    rc = L.lib().nf_ld_fold_multi(ptr(ld), _ptr_array(terms), neg, i32(len(terms)), i64(ld.numel()), L.stream())
    L.check(rc, "nf_ld_fold_multi")
    return ld


def inv1x1_assemble_multi(layers):
    """nf_inv1x1_assemble_multi: [(W, per-pixel log|det|)] for `layers` = [(P, L, U, sign_S, log_S)] of ONE size, float32, density
    direction -- one launch per 32 layers."""
    n = len(layers)
    Cc = layers[0][1].shape[0]
    for tup in layers:
        L.require_device(*tup)
        if tup[1].shape != (Cc, Cc) or tup[1].dtype != torch.float32:
            raise NotImplementedError("inv1x1_assemble_multi: float32 layers of one size")
    dev = layers[0][1].device
    cols = [[t.contiguous() for t in tup] for tup in layers]
    W = torch.empty(n, Cc, Cc, dtype=torch.float32, device=dev)
    ld = torch.empty(n, dtype=torch.float32, device=dev)
    Ws, lds = list(W.unbind(0)), list(ld.unbind(0))
    arrs = [_ptr_array([c[k] for c in cols]) for k in range(5)]
    rc = L.lib().nf_inv1x1_assemble_multi(*arrs, _ptr_array(Ws), _ptr_array(lds), i32(n), i32(Cc), L.stream())
    L.check(rc, "nf_inv1x1_assemble_multi")
    return list(zip(Ws, lds))


def inv1x1_lu_grads_multi(layers, gWs, gls):
    """nf_inv1x1_lu_grads_multi: [(gL, gU, g_log_S)] for `layers` = [(P, L, U, sign_S, log_S)] of one size with cotangents gWs[i]
    (C x C) and gls[i] (0-dim or None)."""
    n = len(layers)
    Cc = layers[0][1].shape[0]
    dev = layers[0][1].device
    cols = [[t.contiguous() for t in tup] for tup in layers]
    gWs = [g.contiguous() for g in gWs]
    gls = [None if g is None else g.contiguous() for g in gls]
    L.require_device(*gWs)
    gL = torch.empty(n, Cc, Cc, dtype=torch.float32, device=dev)
    gU = torch.empty(n, Cc, Cc, dtype=torch.float32, device=dev)
    gs = torch.empty(n, Cc, dtype=torch.float32, device=dev)
    gLs, gUs, gss = list(gL.unbind(0)), list(gU.unbind(0)), list(gs.unbind(0))
    arrs = [_ptr_array([c[k] for c in cols]) for k in range(5)]
    rc = L.lib().nf_inv1x1_lu_grads_multi(*arrs, _ptr_array(gWs), _ptr_array(gls), _ptr_array(gLs), _ptr_array(gUs), _ptr_array(gss),
                                          i32(n), i32(Cc), L.stream())
    L.check(rc, "nf_inv1x1_lu_grads_multi")
    return list(zip(gLs, gUs, gss))


def inv1x1_conv(z, W, logdet_unit, logdet=None, acc=None, want_scalar=True, bias=None):
    L.require_device(z, W, logdet_unit, bias)
    z = z.contiguous()
    B, Cc = z.shape[:2]
    HW = int(math.prod(z.shape[2:]))
    y = torch.empty_like(z)
    lds = torch.empty((), dtype=z.dtype, device=z.device) if want_scalar else None
    if logdet is not None and acc is None:
        acc = L.LD_ADD
    rc = L.lib().nf_inv1x1_conv_affine(ptr(z), ptr(W.contiguous()), ptr(None if bias is None else bias.contiguous()),
                                       ptr(logdet_unit), ptr(y), ptr(lds), ptr(logdet), i64(B), i32(Cc), i64(HW),
                                       i32(acc or 0), i32(L.dtype_code(z)), L.stream())
    L.check(rc, "nf_inv1x1_conv_affine")
    return y, lds


def inv1x1_conv_t(z, W):
    """y = W^T z per pixel (nf_inv1x1_conv_t): the 1x1 convolution's input gradient without a transposed copy of W."""
    L.require_device(z, W)
    z = z.contiguous()
    B, Cc = z.shape[:2]
    HW = int(math.prod(z.shape[2:]))
    y = torch.empty_like(z)
    L.check(L.lib().nf_inv1x1_conv_t(ptr(z), ptr(W.contiguous()), ptr(y), i64(B), i32(Cc), i64(HW), i32(L.dtype_code(z)), L.stream()),
            "nf_inv1x1_conv_t")
    return y


# ---- backward of the affine family (csrc/affine_bwd.hip): closed-form vector-Jacobian products ------------------------
def masked_affine_bwd(z, b, s, t, gy, gld, direction):
    """(gz, gs, gt) of nf_masked_affine for cotangents gy (like z) and gld (B) -- coupling.py:209-229 under autograd."""
    L.require_device(z, b, s, t, gy, gld)
    z, gy = z.contiguous(), gy.contiguous()
    B = z.shape[0]
    inner = z[0].numel() if B else int(math.prod(z.shape[1:]))
    bb = b.to(z.dtype)
    if bb.numel() != inner:
        bb = bb.expand((1,) + tuple(z.shape[1:]))
    bb = bb.contiguous().view(-1)
    gz = torch.empty_like(z)
    gs = None if s is None else torch.empty_like(z)
    gt = None if t is None else torch.empty_like(z)
    rc = L.lib().nf_masked_affine_bwd(ptr(z), ptr(bb), ptr(None if s is None else s.contiguous()),
                                      ptr(None if t is None else t.contiguous()), ptr(gy),
                                      ptr(None if gld is None else gld.contiguous()), ptr(gz), ptr(gs), ptr(gt), i64(B),
                                      i64(inner), i32(direction), i32(L.dtype_code(z)), L.stream())
    L.check(rc, "nf_masked_affine_bwd")
    return gz, gs, gt


def affine_coupling_bwd(z, param, gy, gld, c1, flip, scale_map, direction):
    """(gz, gparam) of nf_affine_coupling -- coupling.py:117-171 with the channel split / merge under autograd."""
    L.require_device(z, param, gy, gld)
    z, param, gy = z.contiguous(), param.contiguous(), gy.contiguous()
    B, Cc = z.shape[:2]
    HW = int(math.prod(z.shape[2:])) if z.dim() > 2 else 1
    gz, gp = torch.empty_like(z), torch.empty_like(param)
    rc = L.lib().nf_affine_coupling_bwd(ptr(z), ptr(param), ptr(gy), ptr(None if gld is None else gld.contiguous()), ptr(gz),
                                        ptr(gp), i64(B), i32(Cc), i32(c1), i32(int(flip)), i64(HW), i32(L.SCALE[scale_map]),
                                        i32(direction), i32(L.dtype_code(z)), L.stream())
    L.check(rc, "nf_affine_coupling_bwd")
    return gz, gp


def actnorm_bwd(z, s, t, gy, gld, direction):
    """(gz, gs (C), gt (C)) of nf_actnorm with the log-det returned per sample -- coupling.py:38-54 under autograd."""
    L.require_device(z, s, t, gy, gld)
    z, gy = z.contiguous(), gy.contiguous()
    B, Cc = z.shape[:2]
    HW = int(math.prod(z.shape[2:])) if z.dim() > 2 else 1
    gz = torch.empty_like(z)
    gs = torch.empty(Cc, dtype=z.dtype, device=z.device)
    gt = torch.empty(Cc, dtype=z.dtype, device=z.device)
    if B == 0:
        return gz, gs.zero_(), gt.zero_()
    import ctypes
    lib = L.lib()
    lib.nf_actnorm_bwd_scratch_doubles.restype = ctypes.c_int64
    scratch = torch.empty(int(lib.nf_actnorm_bwd_scratch_doubles(i64(B), i32(Cc))), dtype=torch.float64, device=z.device)
    rc = lib.nf_actnorm_bwd(ptr(z), ptr(s.contiguous().view(-1)), ptr(t.contiguous().view(-1)), ptr(gy),
                            ptr(None if gld is None else gld.contiguous()), ptr(gz), ptr(gs), ptr(gt), ptr(scratch), i64(B),
                            i32(Cc), i64(HW), i32(direction), i32(L.dtype_code(z)), L.stream())
    L.check(rc, "nf_actnorm_bwd")
    return gz, gs, gt


def rows_matvec(x, W):
    """y_b = W x_b for every row of x (B, D) float32, D <= 128 (nf_rows_matvec, csrc/rows_matvec.hip)."""
    L.require_device(x, W)
    if x.dtype != torch.float32 or x.dim() != 2 or x.shape[1] > 128:
        raise NotImplementedError("rows_matvec: (B, D <= 128) float32")
    x = x.contiguous()
    y = torch.empty_like(x)
    rc = L.lib().nf_rows_matvec(ptr(x), ptr(W.to(torch.float32).contiguous()), ptr(y), i64(x.shape[0]), i32(x.shape[1]), L.stream())
    L.check(rc, "nf_rows_matvec")
    return y


def rows_block(x, M1, c1, M2, c2, trans=False, mask1=None, mask2=None, relu=True):
    """(out1, out2) of nf_rows_block: out1 = mask1(M1 pre(x) + c1), out2 = x + mask2(M2 pre(out1) + c2); (B, H <= 128)
    float32.  trans: M1 / M2 are used transposed (the block's backward); relu: pre = ReLU on both products."""
    L.require_device(x, M1, c1, M2, c2, mask1, mask2)
    x = x.contiguous()
    B, H = x.shape
    if x.dtype != torch.float32 or H > 128 or H % 4:
        raise NotImplementedError("rows_block: (B, H <= 128, H % 4 == 0) float32")
    M1, M2 = M1.contiguous(), M2.contiguous()
    assert tuple(M1.shape) == (H, H) and tuple(M2.shape) == (H, H)
    out1, out2 = torch.empty_like(x), torch.empty_like(x)
    c = lambda t: None if t is None else t.contiguous()   # noqa: E731
    rc = L.lib().nf_rows_block(ptr(x), i64(H), ptr(M1), i64(H), i32(int(trans)), ptr(c(c1)), ptr(c(mask1)), i64(H), ptr(out1),
                               i64(H), ptr(M2), i64(H), i32(int(trans)), ptr(c(c2)), ptr(c(mask2)), i64(H), ptr(out2), i64(H),
                               i64(B), i32(H), i32(int(relu)), i32(int(relu)), L.stream())
    L.check(rc, "nf_rows_block")
    return out1, out2


def lu_compose(perm, lower_entries, upper_entries, unconstrained_upper_diag, bias, eps=1e-3):
    """LULinearPermute as dense matrices (nf_lu_compose): (Wd, Ws, bias_d, bias_s, log|det| (1-element)) views of one
    buffer; float32, D <= 64."""
    L.require_device(perm, lower_entries, upper_entries, unconstrained_upper_diag, bias)
    D = bias.numel()
    out = torch.empty(2 * D * D + 2 * D + 1, dtype=torch.float32, device=bias.device)
    rc = L.lib().nf_lu_compose(ptr(perm), ptr(lower_entries.contiguous()), ptr(upper_entries.contiguous()),
                               ptr(unconstrained_upper_diag.contiguous()), ptr(bias.contiguous()), f64(eps), ptr(out), i32(D),
                               L.stream())
    L.check(rc, "nf_lu_compose")
    N = D * D
    return out[:N].view(D, D), out[N:2 * N].view(D, D), out[2 * N:2 * N + D], out[2 * N + D:2 * N + 2 * D], out[2 * N + 2 * D:]


def lu_factors(perm, lower_entries, upper_entries, unconstrained_upper_diag, eps=1e-3):
    """(L, U, Up, diag, log|det| (1-element), L^T, Up^T) of LULinearPermute's factors in one launch (nf_lu_factors); float32."""
    L.require_device(perm, lower_entries, upper_entries, unconstrained_upper_diag)
    D = unconstrained_upper_diag.numel()
    out = torch.empty(5 * D * D + D + 1, dtype=torch.float32, device=unconstrained_upper_diag.device)
    rc = L.lib().nf_lu_factors(ptr(perm), ptr(lower_entries.contiguous()), ptr(upper_entries.contiguous()),
                               ptr(unconstrained_upper_diag.contiguous()), f64(eps), ptr(out), i32(D), L.stream())
    L.check(rc, "nf_lu_factors")
    return lu_factors_views(out, D)


def lu_factors_views(out, D):
    """The seven views (L, U, Up, diag, log|det|, L^T, Up^T) of a (5 D^2 + D + 1) factor buffer written by nf_lu_factors[_multi]."""
    N = D * D
    e = 3 * N + D + 1
    return (out[:N].view(D, D), out[N:2 * N].view(D, D), out[2 * N:3 * N].view(D, D), out[3 * N:3 * N + D], out[3 * N + D:e],
            out[e:e + N].view(D, D), out[e + N:e + 2 * N].view(D, D))


def lu_factors_multi(table, n_layers, eps, D):
    """nf_lu_factors for n_layers layers in one launch; table: (n_layers x 5) int64 device pointers (perm, lower, upper, udiag, out)."""
    L.require_device(table)
    L.check(L.lib().nf_lu_factors_multi(ptr(table), i32(n_layers), f64(eps), i32(D), L.stream()), "nf_lu_factors_multi")


def rqs_fused_pack_all_multi(table, n_layers, num_blocks, tail_bound=3.0, min_bin_width=1e-3, min_bin_height=1e-3,
                             min_derivative=1e-3):
    """nf_rqs_fused_pack_all for n_layers layers in one launch; table: (n_layers x (11 + 4 num_blocks)) int64 device pointers."""
    L.require_device(table)
    rc = L.lib().nf_rqs_fused_pack_all_multi(ptr(table), i32(n_layers), i32(128), i32(num_blocks), i32(8), f64(tail_bound),
                                             f64(min_bin_width), f64(min_bin_height), f64(min_derivative), L.stream())
    L.check(rc, "nf_rqs_fused_pack_all_multi")


def lu_param_grads(gL, gU, gld, unconstrained_upper_diag, n_tri, eps=1e-3, sign=1.0, perm=None, out=None):
    """(g_lower, g_upper, g_udiag) from the dense factor gradients (nf_lu_param_grads); float32.  gld: the (B) log-det
    cotangent (summed inside the launch) or None; perm: gU's columns are read through it (gU = (gu^T x)[:, perm]);
    out: (g_lower, g_upper, g_udiag) destinations written in place (contiguous float32) instead of new tensors."""
    L.require_device(gL, gU, gld, unconstrained_upper_diag, perm)
    if gld is not None:
        gld = gld.contiguous()
    D = unconstrained_upper_diag.numel()
    dev = gL.device
    if out is not None:
        g_lower, g_upper, g_udiag = out
        if (g_lower.numel() != n_tri or g_upper.numel() != n_tri or g_udiag.numel() != D
                or any(t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev for t in out)):
            raise ValueError("lu_param_grads: out = contiguous float32 (n_tri), (n_tri), (D) tensors on the inputs' device")
    else:
        g_lower = torch.empty(n_tri, dtype=torch.float32, device=dev)
        g_upper = torch.empty(n_tri, dtype=torch.float32, device=dev)
        g_udiag = torch.empty(D, dtype=torch.float32, device=dev)
    rc = L.lib().nf_lu_param_grads(ptr(gL.contiguous()), ptr(gU.contiguous()), ptr(perm), ptr(gld),
                                   i64(0 if gld is None else gld.numel()), ptr(unconstrained_upper_diag.contiguous()),
                                   f64(eps), f64(sign), ptr(g_lower), ptr(g_upper), ptr(g_udiag), i32(D), L.stream())
    L.check(rc, "nf_lu_param_grads")
    return g_lower, g_upper, g_udiag


def rows_matvec_affine(x, W, bias, ld_const=None, ld_sign=1.0, logdet=None, acc=None):
    """y_b = W x_b + bias and logdet[b] (acc) ld_sign * ld_const (nf_rows_matvec_affine); (B, D <= 128) float32."""
    L.require_device(x, W, bias, ld_const, logdet)
    x = x.contiguous()
    B, D = x.shape
    y = torch.empty_like(x)
    if ld_const is not None and logdet is None:
        logdet = torch.empty(B, dtype=x.dtype, device=x.device)
        acc = L.LD_WRITE
    elif acc is None:
        acc = L.LD_ADD
    rc = L.lib().nf_rows_matvec_affine(ptr(x), ptr(W), ptr(bias), ptr(y), ptr(logdet if ld_const is not None else None),
                                       ptr(ld_const), f64(ld_sign), i32(acc), i64(B), i32(D), L.stream())
    L.check(rc, "nf_rows_matvec_affine")
    return y, logdet


def rows_matvec2(x, W1, W2, bias=None, ld_const=None, ld_sign=1.0, logdet=None, acc=None, want_u=True):
    """(u, y, logdet): u_b = W1 x_b, y_b = W2 u_b + bias in one launch (nf_rows_matvec2); (B, D <= 64) float32."""
    L.require_device(x, W1, W2, bias, ld_const, logdet)
    x = x.contiguous()
    B, D = x.shape
    u = torch.empty_like(x) if want_u else None
    y = torch.empty_like(x)
    if ld_const is not None and logdet is None:
        logdet = torch.empty(B, dtype=x.dtype, device=x.device)
        acc = L.LD_WRITE
    elif acc is None:
        acc = L.LD_ADD
    rc = L.lib().nf_rows_matvec2(ptr(x), ptr(W1.contiguous()), ptr(W2.contiguous()), ptr(bias), ptr(u), ptr(y),
                                 ptr(logdet if ld_const is not None else None), ptr(ld_const), f64(ld_sign), i32(acc), i64(B),
                                 i32(D), L.stream())
    L.check(rc, "nf_rows_matvec2")
    return u, y, logdet


def inv1x1_wgrad(z, gy, gld):
    """(gW (C, C), g log|det|-per-pixel (0-dim)) of the per-pixel product y = W z (mixing.py:106-133): gW = sum over
    pixels of gy z^T, partial sums per group of images added in a fixed order."""
    import ctypes
    L.require_device(z, gy, gld)
    z, gy = z.contiguous(), gy.contiguous()
    B, Cc = z.shape[:2]
    HW = int(math.prod(z.shape[2:])) if z.dim() > 2 else 1
    lib = L.lib()
    lib.nf_inv1x1_wgrad_scratch_elems.restype = ctypes.c_int64
    n = lib.nf_inv1x1_wgrad_scratch_elems(i64(B), i32(Cc))
    if n < 0:
        raise NotImplementedError("inv1x1_wgrad: C <= 64")
    scratch = torch.empty(int(n), dtype=z.dtype, device=z.device)
    gW = torch.empty(Cc, Cc, dtype=z.dtype, device=z.device)
    gl = torch.empty((), dtype=z.dtype, device=z.device)
    rc = lib.nf_inv1x1_wgrad(ptr(z), ptr(gy), ptr(None if gld is None else gld.contiguous()), ptr(gW), ptr(gl), ptr(scratch),
                             i64(B), i32(Cc), i64(HW), i32(L.dtype_code(z)), L.stream())
    L.check(rc, "nf_inv1x1_wgrad")
    return gW, gl


def diag_gaussian_log_prob(z, loc, log_scale, log_scale_shift=0.0, out=None, acc=None):
    """distributions/base.py:94-103."""
    L.require_device(z, loc, log_scale)
    z = z.contiguous()
    B = z.shape[0]
    d = int(math.prod(z.shape[1:]))
    if out is None:
        out = torch.empty(B, dtype=z.dtype, device=z.device)
        acc = L.LD_WRITE
    elif acc is None:
        acc = L.LD_ADD
    rc = L.lib().nf_diag_gaussian_log_prob(ptr(z), ptr(loc.contiguous().view(-1)), ptr(log_scale.contiguous().view(-1)),
                                           f64(log_scale_shift), ptr(out), i64(B), i64(d), i32(acc),
                                           i32(L.dtype_code(z)), L.stream())
    L.check(rc, "nf_diag_gaussian_log_prob")
    return out


def squeeze(z, direction):
    """flows/reshape.py:116-128.  direction 0 = Squeeze.forward, 1 = Squeeze.inverse."""
    L.require_device(z)
    z = z.contiguous()
    B, Cc, H, W = z.shape
    shape = (B, Cc // 4, 2 * H, 2 * W) if direction == 0 else (B, 4 * Cc, H // 2, W // 2)
    y = torch.empty(shape, dtype=z.dtype, device=z.device)
    rc = L.lib().nf_squeeze(ptr(z), ptr(y), i64(B), i32(Cc), i32(H), i32(W), i32(direction), i32(L.dtype_code(z)),
                            L.stream())
    L.check(rc, "nf_squeeze")
    return y


# ---- fused NSF coupling layer (conditioner on MFMA + spline epilogue) ------------------------------------------
def rqs_fused_supported(nI, nT, hidden, num_blocks, K):
    return L.lib().nf_rqs_fused_pack_size(i32(nI), i32(nT), i32(hidden), i32(num_blocks), i32(K)) > 0


def rqs_fused_pack(w_init, b_init, w_blocks, b_blocks, w_final, b_final, uw, uh, ud, K, tail_bound, min_bin_width,
                   min_bin_height, min_derivative):
    """Re-lay-out one layer's weights in MFMA operand order (nf_rqs_fused_pack).  Returns the packed device blob."""
    import ctypes
    L.require_device(w_init, b_init, w_final, b_final, uw, uh, ud, *w_blocks, *b_blocks)
    hidden, nI = w_init.shape
    nT = uw.shape[0]
    nb = len(w_blocks) // 2
    lib = L.lib()
    lib.nf_rqs_fused_pack_size.restype = ctypes.c_int64
    size = lib.nf_rqs_fused_pack_size(i32(nI), i32(nT), i32(hidden), i32(nb), i32(K))
    if size <= 0:
        raise NotImplementedError("nf_rqs_fused: shape not supported")
    blob = torch.empty(size // 4, dtype=torch.float32, device=w_init.device)
    keep = [t.contiguous() for t in (w_init, b_init, w_final, b_final, uw, uh, ud)]
    wb = [t.contiguous() for t in w_blocks]
    bb = [t.contiguous() for t in b_blocks]
    wp = (ctypes.c_void_p * max(len(wb), 1))(*[t.data_ptr() for t in wb])
    bp = (ctypes.c_void_p * max(len(bb), 1))(*[t.data_ptr() for t in bb])
    rc = lib.nf_rqs_fused_pack(ptr(blob), ptr(keep[0]), ptr(keep[1]), wp, bp, ptr(keep[2]), ptr(keep[3]), ptr(keep[4]),
                               ptr(keep[5]), ptr(keep[6]), i32(nI), i32(nT), i32(hidden), i32(nb), i32(K),
                               f64(tail_bound), f64(min_bin_width), f64(min_bin_height), f64(min_derivative), L.stream())
    L.check(rc, "nf_rqs_fused_pack")
    return blob


def rqs_fused_pack_lu(blob, num_blocks, perm, lower_entries, upper_entries, unconstrained_upper_diag, bias, eps=1e-3, K=8):
    """Add the layer's LULinearPermute (composed dense 64 x 64 matrices, both directions) to a packed blob of K bins."""
    L.require_device(blob, perm, lower_entries, upper_entries, unconstrained_upper_diag, bias)
    D = bias.numel()
    rc = L.lib().nf_rqs_fused_pack_lu(ptr(blob), i32(num_blocks), ptr(perm), ptr(lower_entries.contiguous()),
                                      ptr(upper_entries.contiguous()), ptr(unconstrained_upper_diag.contiguous()),
                                      ptr(bias.contiguous()), i32(D), f64(eps), i32(K), L.stream())
    L.check(rc, "nf_rqs_fused_pack_lu")
    return blob


def rqs_fused(x, blob, mask_parity, hidden, num_blocks, K, direction, logdet=None, acc=None, tail_bound=3.0,
              min_bin_width=1e-3, min_bin_height=1e-3, min_derivative=1e-3, fuse_lu=False, live_d=None):
    """One launch for a whole CoupledRationalQuadraticSpline layer (+ its LULinearPermute when fuse_lu).
    direction 0 = density, 1 = sample."""
    L.require_device(x, blob)
    if x.dtype != torch.float32:
        raise TypeError("nf_rqs_fused is fp32 only")
    x = x.contiguous()
    B, D = x.shape
    y = torch.empty_like(x)
    if logdet is None:
        logdet = torch.empty(B, dtype=x.dtype, device=x.device)
        acc = L.LD_WRITE
    elif acc is None:
        acc = L.LD_ADD
    if D != 64:
        raise ValueError("nf_rqs_fused: rows of 64 columns (narrower layers: padded by the caller, live_d = columns in use)")
    rc = L.lib().nf_rqs_fused(ptr(x), ptr(y), ptr(logdet), ptr(blob), i32(mask_parity), i32(int(fuse_lu)), i64(B),
                              i32(D if live_d is None else live_d), i32(hidden),
                              i32(num_blocks), i32(K), f64(tail_bound), f64(min_bin_width), f64(min_bin_height),
                              f64(min_derivative), i32(direction), i32(acc), L.stream())
    L.check(rc, "nf_rqs_fused")
    return y, logdet


def rqs_coupling_bwd(x, grad_y, grad_logdet, cond, uw, uh, ud, identity_idx, transform_idx, K, mode, tails="linear",
                     tail_bound=3.0, min_bin_width=1e-3, min_bin_height=1e-3, min_derivative=1e-3, wh_div=1.0,
                     tails_t=None, bound_t=None, tails_i=None, bound_i=None):
    """Vector-Jacobian product of rqs_coupling (nf_rqs_coupling_bwd[_ft]).  Returns (gx, gcond, guw, guh, gud)."""
    L.require_device(x, grad_y, grad_logdet, cond, uw, uh, ud, identity_idx, transform_idx, tails_t, bound_t, tails_i,
                     bound_i)
    B, D = x.shape
    x, grad_y, grad_logdet = x.contiguous(), grad_y.contiguous(), grad_logdet.contiguous()
    # density mode owns (writes) every column of gx; the two sampling modes own one half and leave the rest zero
    gx = torch.empty_like(x) if mode == L.RQS_DENSITY else torch.zeros_like(x)
    gcond = torch.empty_like(cond) if cond is not None else None
    guw = torch.zeros_like(uw) if uw is not None else None
    guh = torch.zeros_like(uh) if uh is not None else None
    gud = torch.zeros_like(ud) if ud is not None else None
    if tails == "feature" or bound_t is not None or bound_i is not None:
        fix_b = lambda t: None if t is None else t.to(device=x.device, dtype=x.dtype).contiguous()
        fix_t = lambda t: None if t is None else t.to(device=x.device, dtype=torch.int32).contiguous()
        bt, bi, tt, ti = fix_b(bound_t), fix_b(bound_i), fix_t(tails_t), fix_t(tails_i)
        code = 3 if tails == "feature" else L.TAILS[tails]
        scalar_bound = float(tail_bound) if not torch.is_tensor(tail_bound) else 1.0
        rc = L.lib().nf_rqs_coupling_bwd_ft(ptr(x), ptr(grad_y), ptr(grad_logdet), ptr(cond), ptr(uw), ptr(uh), ptr(ud),
                                            ptr(identity_idx), i32(identity_idx.numel()), ptr(transform_idx),
                                            i32(transform_idx.numel()), i64(B), i32(D), i32(K), i32(code),
                                            f64(scalar_bound), f64(min_bin_width), f64(min_bin_height),
                                            f64(min_derivative), f64(wh_div), i32(mode), ptr(gx), ptr(gcond), ptr(guw),
                                            ptr(guh), ptr(gud), i32(L.dtype_code(x)), ptr(tt), ptr(bt), ptr(ti), ptr(bi),
                                            L.stream())
        L.check(rc, "nf_rqs_coupling_bwd_ft")
        return gx, gcond, guw, guh, gud
    rc = L.lib().nf_rqs_coupling_bwd(ptr(x), ptr(grad_y), ptr(grad_logdet), ptr(cond), ptr(uw), ptr(uh), ptr(ud),
                                     ptr(identity_idx), i32(identity_idx.numel()), ptr(transform_idx),
                                     i32(transform_idx.numel()), i64(B), i32(D), i32(K), i32(L.TAILS[tails]),
                                     f64(tail_bound), f64(min_bin_width), f64(min_bin_height), f64(min_derivative),
                                     f64(wh_div), i32(mode), ptr(gx), ptr(gcond), ptr(guw), ptr(guh), ptr(gud),
                                     i32(L.dtype_code(x)), L.stream())
    L.check(rc, "nf_rqs_coupling_bwd")
    return gx, gcond, guw, guh, gud


# ---- training forward of the fused layer's last stage (csrc/rqs_fused.hip, TRAIN variant) ------------------------------
def rqs_fused_train_blob(num_blocks, device):
    """Empty packed-blob buffer of the fused layer (filled by rqs_fused_pack_final)."""
    import ctypes
    lib = L.lib()
    lib.nf_rqs_fused_pack_size.restype = ctypes.c_int64
    size = lib.nf_rqs_fused_pack_size(i32(32), i32(32), i32(128), i32(num_blocks), i32(8))
    if size <= 0:
        raise NotImplementedError("nf_rqs_fused: shape not supported")
    return torch.zeros(size // 4, dtype=torch.float32, device=device)


def rqs_fused_pack_final(blob, w_final, b_final, uw, uh, ud, num_blocks, tail_bound=3.0, min_bin_width=1e-3,
                         min_bin_height=1e-3, min_derivative=1e-3):
    L.require_device(blob, w_final, b_final, uw, uh, ud)
    rc = L.lib().nf_rqs_fused_pack_final(ptr(blob), ptr(w_final.contiguous()), ptr(b_final.contiguous()), ptr(uw.contiguous()),
                                         ptr(uh.contiguous()), ptr(ud.contiguous()), i32(128), i32(num_blocks), i32(8),
                                         f64(tail_bound), f64(min_bin_width), f64(min_bin_height), f64(min_derivative), L.stream())
    L.check(rc, "nf_rqs_fused_pack_final")
    return blob


def rqs_fused_train_fwd(x, h2, blob, mask_parity, num_blocks, tail_bound=3.0, min_bin_width=1e-3, min_bin_height=1e-3,
                        min_derivative=1e-3, logdet=None, acc=None):
    """(y, logdet, cond24) of nf_rqs_fused_train_fwd: final Linear + density-direction coupling transform in one launch;
    cond24 (B, 32, 24) is the conditioner output kept for rqs_coupling_bwd_p24.  logdet given: folded into it per acc."""
    L.require_device(x, h2, blob)
    x, h2 = x.contiguous(), h2.contiguous()
    B = x.shape[0]
    y = torch.empty_like(x)
    if logdet is None:
        ld, acc = torch.empty(B, dtype=x.dtype, device=x.device), L.LD_WRITE
    else:
        ld, acc = logdet, (L.LD_ADD if acc is None else acc)
    cond = torch.empty(B, 32, 24, dtype=x.dtype, device=x.device)
    rc = L.lib().nf_rqs_fused_train_fwd(ptr(x), ptr(h2), ptr(y), ptr(ld), ptr(cond), ptr(blob), i32(mask_parity), i64(B), i32(64),
                                        i32(128), i32(num_blocks), i32(8), f64(tail_bound), f64(min_bin_width),
                                        f64(min_bin_height), f64(min_derivative), i32(acc), L.stream())
    L.check(rc, "nf_rqs_fused_train_fwd")
    return y, ld, cond


def rqs_fused_pack_all(blob, w_init, b_init, w_blocks, b_blocks, w_final, b_final, uw, uh, ud, tail_bound=3.0, min_bin_width=1e-3,
                       min_bin_height=1e-3, min_derivative=1e-3, wfull=None, wpad=None, identity_idx=None):
    """The whole layer's blob (no LU) in one launch (nf_rqs_fused_pack_all); hidden 128, 8 bins."""
    import ctypes
    L.require_device(blob, w_init, b_init, w_final, b_final, uw, uh, ud, wfull, wpad, identity_idx, *w_blocks, *b_blocks)
    n = len(w_blocks)
    wp = (ctypes.c_void_p * max(n, 1))(*[w.data_ptr() for w in w_blocks])
    bp = (ctypes.c_void_p * max(n, 1))(*[b.data_ptr() for b in b_blocks])
    rc = L.lib().nf_rqs_fused_pack_all(ptr(blob), ptr(w_init), ptr(b_init), wp, bp, ptr(w_final), ptr(b_final), ptr(uw), ptr(uh),
                                       ptr(ud), i32(128), i32(n // 2), i32(8), f64(tail_bound), f64(min_bin_width),
                                       f64(min_bin_height), f64(min_derivative), ptr(wfull), ptr(wpad), ptr(identity_idx),
                                       L.stream())
    L.check(rc, "nf_rqs_fused_pack_all")
    return blob


def rqs_fused_train_full_fwd(x, blob, mask_parity, num_blocks, tail_bound=3.0, min_bin_width=1e-3, min_bin_height=1e-3,
                             min_derivative=1e-3, logdet=None, acc=None):
    """(y, logdet, cond24, acts) of nf_rqs_fused_train_full_fwd: the whole conditioner + coupling transform in one launch;
    acts (2 num_blocks + 1, B, 128) = h0, then (t, h) per residual block."""
    L.require_device(x, blob)
    x = x.contiguous()
    B = x.shape[0]
    y = torch.empty_like(x)
    if logdet is None:
        ld, acc = torch.empty(B, dtype=x.dtype, device=x.device), L.LD_WRITE
    else:
        ld, acc = logdet, (L.LD_ADD if acc is None else acc)
    cond = torch.empty(B, 32, 24, dtype=x.dtype, device=x.device)
    acts = torch.empty(2 * num_blocks + 1, B, 128, dtype=x.dtype, device=x.device)
    rc = L.lib().nf_rqs_fused_train_full_fwd(ptr(x), ptr(y), ptr(ld), ptr(cond), ptr(acts), ptr(blob), i32(mask_parity), i64(B),
                                             i32(64), i32(128), i32(num_blocks), i32(8), f64(tail_bound), f64(min_bin_width),
                                             f64(min_bin_height), f64(min_derivative), i32(acc), L.stream())
    L.check(rc, "nf_rqs_fused_train_full_fwd")
    return y, ld, cond, acts


def rqs_coupling_bwd_p24(x, grad_y, grad_logdet, cond24, uw, uh, ud, identity_idx, transform_idx, tail_bound=3.0,
                         min_bin_width=1e-3, min_bin_height=1e-3, min_derivative=1e-3, wh_div=1.0):
    """rqs_coupling_bwd (density direction) on the 24-float rows of rqs_fused_train_fwd.  Returns (gx, gcond24, guw, guh, gud)."""
    L.require_device(x, grad_y, grad_logdet, cond24, uw, uh, ud, identity_idx, transform_idx)
    B, D = x.shape
    x, grad_y, grad_logdet = x.contiguous(), grad_y.contiguous(), grad_logdet.contiguous()
    gx = torch.empty_like(x)
    gcond = torch.empty_like(cond24)
    nw, nh = uw.numel(), uh.numel()                        # one zero fill for the three atomically-accumulated outputs
    gz = torch.zeros(nw + nh + ud.numel(), dtype=uw.dtype, device=uw.device)
    guw, guh, gud = gz[:nw].view_as(uw), gz[nw:nw + nh].view_as(uh), gz[nw + nh:].view_as(ud)
    rc = L.lib().nf_rqs_coupling_bwd_p24(ptr(x), ptr(grad_y), ptr(grad_logdet), ptr(cond24), ptr(uw), ptr(uh), ptr(ud),
                                         ptr(identity_idx), i32(identity_idx.numel()), ptr(transform_idx),
                                         i32(transform_idx.numel()), i64(B), i32(D), f64(tail_bound), f64(min_bin_width),
                                         f64(min_bin_height), f64(min_derivative), f64(wh_div), ptr(gx), ptr(gcond), ptr(guw),
                                         ptr(guh), ptr(gud), L.stream())
    L.check(rc, "nf_rqs_coupling_bwd_p24")
    return gx, gcond, guw, guh, gud


def final_bwd(x, grad_y, grad_logdet, cond24, w_t, blob, uw, uh, ud, mask_parity, num_blocks, tail_bound=3.0, min_bin_width=1e-3,
              min_bin_height=1e-3, min_derivative=1e-3):
    """(gx, gcond24, gh, guw, guh, gud) of nf_final_bwd + nf_final_bwd_reduce: the coupling transform's backward and the final
    Linear's input gradient in one pass over the rows; the batch-shared parameters' gradients by a fixed-order reduction."""
    L.require_device(x, grad_y, grad_logdet, cond24, w_t, blob, uw, uh, ud)
    B = x.shape[0]
    x, grad_y, grad_logdet = x.contiguous(), grad_y.contiguous(), grad_logdet.contiguous()
    gx = torch.empty_like(x)
    gcond = torch.empty_like(cond24)
    gh = torch.empty(B, 128, dtype=x.dtype, device=x.device)
    lib = L.lib()
    nparts = lib.nf_final_bwd_partials(i64(B))
    part = torch.empty(max(nparts, 1) * 768, dtype=x.dtype, device=x.device)
    gz = torch.empty(uw.numel() + uh.numel() + ud.numel(), dtype=uw.dtype, device=uw.device)
    nw, nh = uw.numel(), uh.numel()
    guw, guh, gud = gz[:nw].view_as(uw), gz[nw:nw + nh].view_as(uh), gz[nw + nh:].view_as(ud)
    kw = (f64(tail_bound), f64(min_bin_width), f64(min_bin_height), f64(min_derivative))
    rc = lib.nf_final_bwd(ptr(x), ptr(grad_y), ptr(grad_logdet), ptr(cond24), ptr(w_t), ptr(blob), ptr(gx), ptr(gcond), ptr(gh),
                          ptr(part), i32(mask_parity), i64(B), i32(64), i32(128), i32(num_blocks), i32(8), *kw, L.stream())
    L.check(rc, "nf_final_bwd")
    rc = lib.nf_final_bwd_reduce(ptr(part), i32(nparts), ptr(uw.contiguous()), ptr(uh.contiguous()), ptr(ud.contiguous()), ptr(guw),
                                 ptr(guh), ptr(gud), i32(8), *kw, L.stream())
    L.check(rc, "nf_final_bwd_reduce")
    return gx, gcond, gh, guw, guh, gud


def pair_train_bwd(x_in, xlu, grad_y, grad_logdet, cond24, acts, w_t, blob, wfull_t, w_blocks, uw, uh, ud, col_map, n_cols, mask_parity,
                   num_blocks, Wd, Lm, Um, perm, udiag, lu_eps, dest, tail_bound=3.0, min_bin_width=1e-3, min_bin_height=1e-3,
                   min_derivative=1e-3, side=None):
    """The whole backward of a [CoupledRQS, LULinearPermute] pair in one C-ABI call (nf_pair_train_bwd: seven launches).  dest: as
    coupling_train_bwd plus lower, upper, udiag, lbias (the LU's gradient destinations).  Returns the pair's input gradient.
    side (a torch.cuda.Stream): the last two launches -- the reduction of the partial tiles and the LU's factor gradients, which only
    produce parameter gradients -- go to that stream, forked from the current one by an event (nf_pair_train_bwd_head / _tail): they
    run under whatever the current stream does next.  The CALLER joins (`current_stream().wait_stream(side)`) before a gradient in
    `dest` is read; the scratch and the tensors the tail reads are kept from reuse through record_stream."""
    L.require_device(x_in, xlu, grad_y, grad_logdet, cond24, acts, w_t, blob, wfull_t, uw, uh, ud, col_map, Wd, Lm, Um, perm, udiag,
                     *w_blocks)
    B = x_in.shape[0]
    x_in, xlu, grad_y, grad_logdet = x_in.contiguous(), xlu.contiguous(), grad_y.contiguous(), grad_logdet.contiguous()
    lib = L.lib()
    lib.nf_pair_train_bwd_scratch_floats.restype = C.c_int64
    n = int(lib.nf_pair_train_bwd_scratch_floats(i64(B), i32(num_blocks)))
    if n <= 0:
        raise NotImplementedError("pair_train_bwd: batch a multiple of 64, 1 <= num_blocks <= 5")
    keys = ("w0", "b0", "wf", "bf", "uw", "uh", "ud", "lower", "upper", "udiag", "lbias")
    tensors = [dest[k] for k in keys] + list(dest["blocks"])
    if len(w_blocks) != 2 * num_blocks or len(dest["blocks"]) != 4 * num_blocks:
        raise ValueError("pair_train_bwd: 2 weights and 4 gradient destinations per residual block")
    L.require_device(*tensors)
    if any(t.dtype != torch.float32 or not t.is_contiguous() for t in tensors):
        raise ValueError("pair_train_bwd: gradient destinations must be contiguous float32")
    scratch = torch.empty(n, dtype=torch.float32, device=x_in.device)
    gx = torch.empty_like(x_in)
    wb = [w.contiguous() for w in w_blocks]
    wp = (C.c_void_p * len(wb))(*[w.data_ptr() for w in wb])
    gp = (C.c_void_p * len(dest["blocks"]))(*[t.data_ptr() for t in dest["blocks"]])
    uw_, uh_, ud_, Lm_, Um_, udiag_ = uw.contiguous(), uh.contiguous(), ud.contiguous(), Lm.contiguous(), Um.contiguous(), udiag.contiguous()
    args = (ptr(x_in), ptr(xlu), ptr(grad_y), ptr(grad_logdet), ptr(cond24), ptr(acts), ptr(w_t), ptr(blob),
            ptr(wfull_t), wp, ptr(uw_), ptr(uh_), ptr(ud_), ptr(col_map),
            i32(int(n_cols)), ptr(Wd), ptr(Lm_), ptr(Um_), ptr(perm), ptr(udiag_),
            f64(lu_eps), ptr(gx), ptr(dest["lower"]), ptr(dest["upper"]), ptr(dest["udiag"]), ptr(dest["lbias"]),
            ptr(dest["w0"]), ptr(dest["b0"]), ptr(dest["wf"]), ptr(dest["bf"]), ptr(dest["uw"]), ptr(dest["uh"]),
            ptr(dest["ud"]), gp, ptr(scratch), i32(mask_parity), i64(B), i32(64), i32(128), i32(num_blocks), i32(8),
            f64(tail_bound), f64(min_bin_width), f64(min_bin_height), f64(min_derivative))
    if side is None:
        L.check(lib.nf_pair_train_bwd(*args, L.stream()), "nf_pair_train_bwd")
        return gx
    tail = C.create_string_buffer(2048)                       # NF_PAIR_TAIL_BYTES
    L.check(lib.nf_pair_train_bwd_head(*args, tail, L.stream()), "nf_pair_train_bwd_head")
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)                                     # fork: an event recorded here, behind the five launches
    L.check(lib.nf_pair_train_bwd_tail(tail, C.c_void_p(side.cuda_stream)), "nf_pair_train_bwd_tail")
    # what the side stream reads or writes must not go back to the allocator (or be rewritten on the current stream) before it is done
    # (parameters, the prepacked factors and the flat gradient buffer outlive the step; the caller's join comes before they change)
    for t in (scratch, grad_logdet):
        t.record_stream(side)
    return gx


def lu_pack_train_multi(table, n_layers, num_blocks, eps, D=64):
    """The LU stage of n training blobs + the composed matrices for the backward in one launch (nf_lu_pack_train_multi)."""
    L.check(L.lib().nf_lu_pack_train_multi(ptr(table), i32(n_layers), i32(num_blocks), i32(D), f64(eps), L.stream()),
            "nf_lu_pack_train_multi")


def rqs_fused_train_pair_fwd(x, blob, mask_parity, num_blocks, tail_bound=3.0, min_bin_width=1e-3, min_bin_height=1e-3,
                             min_derivative=1e-3, logdet=None, acc=None):
    """(xlu, y, logdet, cond24, acts) of nf_rqs_fused_train_pair_fwd: LULinearPermute.inverse + the whole coupling layer in one
    launch; xlu (B, 64) = the LU's output (the coupling's input), the rest as rqs_fused_train_full_fwd."""
    L.require_device(x, blob)
    x = x.contiguous()
    B = x.shape[0]
    y, xlu = torch.empty_like(x), torch.empty_like(x)
    if logdet is None:
        ld, acc = torch.empty(B, dtype=x.dtype, device=x.device), L.LD_WRITE
    else:
        ld, acc = logdet, (L.LD_ADD if acc is None else acc)
    cond = torch.empty(B, 32, 24, dtype=x.dtype, device=x.device)
    acts = torch.empty(2 * num_blocks + 1, B, 128, dtype=x.dtype, device=x.device)
    rc = L.lib().nf_rqs_fused_train_pair_fwd(ptr(x), ptr(xlu), ptr(y), ptr(ld), ptr(cond), ptr(acts), ptr(blob), i32(mask_parity),
                                             i64(B), i32(64), i32(128), i32(num_blocks), i32(8), f64(tail_bound), f64(min_bin_width),
                                             f64(min_bin_height), f64(min_derivative), i32(acc), L.stream())
    L.check(rc, "nf_rqs_fused_train_pair_fwd")
    return xlu, y, ld, cond, acts


def lu_bwd_composed(g, x, Wd, db_out=None):
    """(gx, dWd, db) of the composed LULinearPermute's backward, D = 64 (nf_lu_bwd_composed): gx = g Wd, dWd = g^T x, db = colsum(g)."""
    L.require_device(g, x, Wd, db_out)
    g, x, Wd = g.contiguous(), x.contiguous(), Wd.contiguous()
    B, D = g.shape
    lib = L.lib()
    lib.nf_lu_bwd_composed_scratch_floats.restype = C.c_int64
    n = int(lib.nf_lu_bwd_composed_scratch_floats(i64(B)))
    if n <= 0 or D != 64 or g.dtype != torch.float32:
        raise NotImplementedError("lu_bwd_composed: float32, D = 64, batch a multiple of 64")
    scratch = torch.empty(n, dtype=torch.float32, device=g.device)
    dWd = torch.empty(D, D, dtype=torch.float32, device=g.device)
    if db_out is None:
        db_out = torch.empty(D, dtype=torch.float32, device=g.device)
    elif db_out.numel() != D or db_out.dtype != torch.float32 or not db_out.is_contiguous():
        raise ValueError("lu_bwd_composed: db_out = a contiguous float32 (D) tensor")
    gx = torch.empty_like(g)
    rc = lib.nf_lu_bwd_composed(ptr(g), ptr(x), ptr(Wd), ptr(gx), ptr(dWd), ptr(db_out), ptr(scratch), i64(B), i32(D), L.stream())
    L.check(rc, "nf_lu_bwd_composed")
    return gx, dWd, db_out


def lu_param_grads_composed(dWd, Lm, Um, perm, gld, unconstrained_upper_diag, n_tri, eps=1e-3, out=None):
    """(g_lower, g_upper, g_udiag) from the composed matrix's gradient (nf_lu_param_grads_composed); out: destinations."""
    L.require_device(dWd, Lm, Um, perm, gld, unconstrained_upper_diag)
    D = unconstrained_upper_diag.numel()
    dev = dWd.device
    if gld is not None:
        gld = gld.contiguous()
    if out is not None:
        g_lower, g_upper, g_udiag = out
        if (g_lower.numel() != n_tri or g_upper.numel() != n_tri or g_udiag.numel() != D
                or any(t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev for t in out)):
            raise ValueError("lu_param_grads_composed: out = contiguous float32 (n_tri), (n_tri), (D) tensors")
    else:
        g_lower = torch.empty(n_tri, dtype=torch.float32, device=dev)
        g_upper = torch.empty(n_tri, dtype=torch.float32, device=dev)
        g_udiag = torch.empty(D, dtype=torch.float32, device=dev)
    rc = L.lib().nf_lu_param_grads_composed(ptr(dWd.contiguous()), ptr(Lm.contiguous()), ptr(Um.contiguous()), ptr(perm), ptr(gld),
                                            i64(0 if gld is None else gld.numel()), ptr(unconstrained_upper_diag.contiguous()),
                                            f64(eps), ptr(g_lower), ptr(g_upper), ptr(g_udiag), i32(D), L.stream())
    L.check(rc, "nf_lu_param_grads_composed")
    return g_lower, g_upper, g_udiag


def coupling_train_bwd(x, grad_y, grad_logdet, cond24, acts, w_t, blob, wfull_t, w_blocks, uw, uh, ud, col_map, n_cols, mask_parity,
                       num_blocks, dest, tail_bound=3.0, min_bin_width=1e-3, min_bin_height=1e-3, min_derivative=1e-3):
    """The whole backward of a benchmark-shaped coupling layer in one C-ABI call (nf_coupling_train_bwd): four passes over the rows
    and ONE reduction launch.  w_blocks: [W1, W2] per residual block; dest: the gradient destinations, written in place --
    dict(w0, b0, wf, bf, uw, uh, ud, blocks=[gW1, gb1, gW2, gb2 per block]) of contiguous float32 tensors of the parameters'
    shapes (any addresses: fresh tensors or views of one flat gradient buffer).  Returns grad_x."""
    L.require_device(x, grad_y, grad_logdet, cond24, acts, w_t, blob, wfull_t, uw, uh, ud, col_map, *w_blocks)
    B = x.shape[0]
    x, grad_y, grad_logdet = x.contiguous(), grad_y.contiguous(), grad_logdet.contiguous()
    lib = L.lib()
    lib.nf_coupling_train_bwd_scratch_floats.restype = C.c_int64
    n = int(lib.nf_coupling_train_bwd_scratch_floats(i64(B), i32(num_blocks)))
    if n <= 0:
        raise NotImplementedError("coupling_train_bwd: batch a multiple of 64, 1 <= num_blocks <= 5")
    tensors = [dest[k] for k in ("w0", "b0", "wf", "bf", "uw", "uh", "ud")] + list(dest["blocks"])
    if len(w_blocks) != 2 * num_blocks or len(dest["blocks"]) != 4 * num_blocks:
        raise ValueError("coupling_train_bwd: 2 weights and 4 gradient destinations per residual block")
    L.require_device(*tensors)
    if any(t.dtype != torch.float32 or not t.is_contiguous() for t in tensors):
        raise ValueError("coupling_train_bwd: gradient destinations must be contiguous float32")
    scratch = torch.empty(n, dtype=torch.float32, device=x.device)
    gx = torch.empty_like(x)
    wb = [w.contiguous() for w in w_blocks]
    wp = (C.c_void_p * len(wb))(*[w.data_ptr() for w in wb])
    gp = (C.c_void_p * len(dest["blocks"]))(*[t.data_ptr() for t in dest["blocks"]])
    rc = lib.nf_coupling_train_bwd(ptr(x), ptr(grad_y), ptr(grad_logdet), ptr(cond24), ptr(acts), ptr(w_t), ptr(blob), ptr(wfull_t), wp,
                                   ptr(uw.contiguous()), ptr(uh.contiguous()), ptr(ud.contiguous()), ptr(col_map), i32(int(n_cols)),
                                   ptr(gx), ptr(dest["w0"]), ptr(dest["b0"]), ptr(dest["wf"]), ptr(dest["bf"]), ptr(dest["uw"]),
                                   ptr(dest["uh"]), ptr(dest["ud"]), gp, ptr(scratch), i32(mask_parity), i64(B), i32(64), i32(128),
                                   i32(num_blocks), i32(8), f64(tail_bound), f64(min_bin_width), f64(min_bin_height),
                                   f64(min_derivative), L.stream())
    L.check(rc, "nf_coupling_train_bwd")
    return gx


# ---- bf16x3 (error-compensated split-bf16 MFMA) variant of the fused layer ----------------------------------------
def rqs_fused_x3_pack(f32_blob, num_blocks, has_lu, nI=32, nT=32, hidden=128, K=8):
    """Derive the split-bf16 weight blob from an rqs_fused_pack blob of the same layer (nf_rqs_fused_x3_pack)."""
    import ctypes
    L.require_device(f32_blob)
    lib = L.lib()
    lib.nf_rqs_fused_x3_pack_size.restype = ctypes.c_int64
    size = lib.nf_rqs_fused_x3_pack_size(i32(nI), i32(nT), i32(hidden), i32(num_blocks), i32(K))
    if size <= 0:
        raise NotImplementedError("nf_rqs_fused_x3: shape not supported")
    blob = torch.zeros((size + 3) // 4, dtype=torch.float32, device=f32_blob.device)
    rc = lib.nf_rqs_fused_x3_pack(ptr(blob), ptr(f32_blob), i32(num_blocks), i32(int(has_lu)), L.stream())
    L.check(rc, "nf_rqs_fused_x3_pack")
    return blob


def rqs_fused_x3(x, blob, mask_parity, hidden, num_blocks, K, direction, logdet=None, acc=None, tail_bound=3.0,
                 min_bin_width=1e-3, min_bin_height=1e-3, min_derivative=1e-3, fuse_lu=False):
    L.require_device(x, blob)
    if x.dtype != torch.float32:
        raise TypeError("nf_rqs_fused_x3 is fp32 only")
    x = x.contiguous()
    B, D = x.shape
    y = torch.empty_like(x)
    if logdet is None:
        logdet = torch.empty(B, dtype=x.dtype, device=x.device)
        acc = L.LD_WRITE
    elif acc is None:
        acc = L.LD_ADD
    rc = L.lib().nf_rqs_fused_x3(ptr(x), ptr(y), ptr(logdet), ptr(blob), i32(mask_parity), i32(int(fuse_lu)), i64(B),
                                 i32(D), i32(hidden), i32(num_blocks), i32(K), f64(tail_bound), f64(min_bin_width),
                                 f64(min_bin_height), f64(min_derivative), i32(direction), i32(acc), L.stream())
    L.check(rc, "nf_rqs_fused_x3")
    return y, logdet


def maf_affine(x, params, direction, logdet=None, acc=None, want_logdet=True):
    """affine/autoregressive.py:98-128.  params (B, D*2) or (B, D, 2).  direction 0 = forward, 1 = inverse."""
    L.require_device(x, params)
    x = x.contiguous()
    params = params.contiguous()
    B, D = x.shape
    y = torch.empty_like(x)
    if logdet is None and want_logdet:
        logdet = torch.empty(B, dtype=x.dtype, device=x.device)
        acc = L.LD_WRITE
    elif acc is None:
        acc = L.LD_ADD
    rc = L.lib().nf_maf_affine(ptr(x), ptr(params), ptr(y), ptr(logdet), i64(B), i32(D), i32(direction), i32(acc or 0),
                               i32(L.dtype_code(x)), L.stream())
    L.check(rc, "nf_maf_affine")
    return y, logdet


def conv3x3_gather(x, flip=False, ld=None):
    """col (B H W, 9 C) of an NCHW float32 tensor: col[r][tap C + c] = x[b][c][y + dy][x + dx] (nf_conv3x3_gather); flip negates the
    offsets.  ld: row length of the returned buffer (>= 9 C; rows rounded up to 64: the layout the MADE training kernels' weight
    gradients read directly) -- columns beyond 9 C are not written, rows beyond B H W are zero."""
    L.require_device(x)
    if x.dtype != torch.float32 or x.dim() != 4:
        raise NotImplementedError("conv3x3_gather: (B, C, H, W) float32")
    B, C, H, W = x.shape
    if x.stride()[1:] != (H * W, W, 1) or x.stride(0) < C * H * W:       # (a channel split is read in place through its batch stride)
        x = x.contiguous()
    R = B * H * W
    if ld is None:
        col = torch.empty(R, 9 * C, dtype=x.dtype, device=x.device)
    else:
        Rp = (R + 63) // 64 * 64
        col = (torch.empty if Rp == R else torch.zeros)(Rp, ld, dtype=x.dtype, device=x.device)
    rc = L.lib().nf_conv3x3_gather(ptr_any(x), ptr(col), i64(B), i32(C), i32(H), i32(W), i32(col.shape[1]), i32(1 if flip else 0),
                                   i64(x.stride(0) if B > 1 else C * H * W), L.stream())
    L.check(rc, "nf_conv3x3_gather")
    return col


def conv3x3_gather_sum(P, bias, shape, flip=False):
    """(B, C, H, W) from per-pixel tap products P (B H W, 9 C): out[b][c][y][x] = bias[c] + sum_tap P[(b, y + dy, x + dx)][tap C + c]
    (nf_conv3x3_gather_sum); flip negates the offsets."""
    L.require_device(P, bias)
    B, C, H, W = shape
    P = P.contiguous()
    if P.dtype != torch.float32 or P.dim() != 2 or P.shape[0] < B * H * W or P.shape[1] < 9 * C:
        raise ValueError("conv3x3_gather_sum: P must be (>= B H W, >= 9 C) float32")
    out = torch.empty(B, C, H, W, dtype=P.dtype, device=P.device)
    rc = L.lib().nf_conv3x3_gather_sum(ptr(P), ptr(None if bias is None else bias.contiguous()), ptr(out), i64(B), i32(C), i32(H),
                                       i32(W), i32(P.shape[1]), i32(1 if flip else 0), L.stream())
    L.check(rc, "nf_conv3x3_gather_sum")
    return out


def channel_sum(g):
    """(C) = g (B, C, H, W).sum((0, 2, 3)) in one launch with a fixed summation order (nf_channel_sum)."""
    L.require_device(g)
    g = g.contiguous()
    if g.dtype != torch.float32 or g.dim() != 4:
        raise ValueError("channel_sum: a float32 (B, C, H, W) tensor")
    B, C, H, W = g.shape
    out = torch.empty(C, dtype=g.dtype, device=g.device)
    L.check(L.lib().nf_channel_sum(ptr(g), ptr(out), i64(B), i32(C), i64(H * W), L.stream()), "nf_channel_sum")
    return out


def maf_affine_bwd(x, params, gy, gld, direction):
    """Backward of maf_affine (nf_maf_affine_bwd): (g_x (B, D), g_params shaped like params); gy / gld may be None."""
    L.require_device(x, params)
    x = x.contiguous()
    params = params.contiguous()
    B, D = x.shape
    gx = torch.empty_like(x)
    gparams = torch.empty_like(params)
    gy = None if gy is None else gy.contiguous()
    gld = None if gld is None else gld.contiguous()
    rc = L.lib().nf_maf_affine_bwd(ptr(x), ptr(params), ptr(gy), ptr(gld), ptr(gx), ptr(gparams), i64(B), i32(D), i32(direction),
                                   i32(L.dtype_code(x)), L.stream())
    L.check(rc, "nf_maf_affine_bwd")
    return gx, gparams


def maf_implicit_sweep(x, params, gx, gld, gxm, v, gp, changed):
    """nf_maf_implicit_sweep: v, gp updated in place; `changed` (int32 scalar tensor) set when v moved."""
    L.require_device(x, params, gx, gld, gxm, v, gp, changed)
    B, D = x.shape
    rc = L.lib().nf_maf_implicit_sweep(ptr(x), ptr(params), ptr(gx), ptr(gld), ptr(gxm), ptr(v), ptr(gp), ptr(changed), i64(B), i32(D),
                                       i32(L.dtype_code(x)), L.stream())
    L.check(rc, "nf_maf_implicit_sweep")


def rqs_fused_chain(x, blobs, parities, hidden, num_blocks, K, direction, logdet=None, acc=None, tail_bound=3.0,
                    min_bin_width=1e-3, min_bin_height=1e-3, min_derivative=1e-3, fuse_lu=True, live_d=None):
    """Up to 64 fused layers of identical shape in ONE persistent launch (nf_rqs_fused_chain).  `blobs` / `parities`
    are in processing order."""
    import ctypes
    L.require_device(x, *blobs)
    if x.dtype != torch.float32:
        raise TypeError("nf_rqs_fused_chain is fp32 only")
    x = x.contiguous()
    B, D = x.shape
    y = torch.empty_like(x)
    if logdet is None:
        logdet = torch.empty(B, dtype=x.dtype, device=x.device)
        acc = L.LD_WRITE
    elif acc is None:
        acc = L.LD_ADD
    n = len(blobs)
    bp = (ctypes.c_void_p * n)(*[b.data_ptr() for b in blobs])
    pp = (ctypes.c_int * n)(*[int(v) for v in parities])
    if D != 64:
        raise ValueError("nf_rqs_fused_chain: rows of 64 columns (narrower layers: padded by the caller, live_d = columns in use)")
    rc = L.lib().nf_rqs_fused_chain(ptr(x), ptr(y), ptr(logdet), bp, pp, i32(n), i32(int(fuse_lu)), i64(B),
                                    i32(D if live_d is None else live_d),
                                    i32(hidden), i32(num_blocks), i32(K), f64(tail_bound), f64(min_bin_width),
                                    f64(min_bin_height), f64(min_derivative), i32(direction), i32(acc), L.stream())
    L.check(rc, "nf_rqs_fused_chain")
    return y, logdet


def rqs_fused_x3_chain(x, blobs, parities, hidden, num_blocks, K, direction, logdet=None, acc=None, tail_bound=3.0,
                       min_bin_width=1e-3, min_bin_height=1e-3, min_derivative=1e-3, fuse_lu=True, live_d=None):
    """Up to 64 fused layers of identical shape on the split-bf16 matrix path in ONE persistent launch
    (nf_rqs_fused_x3_chain).  `blobs` (rqs_fused_x3_pack) / `parities` are in processing order."""
    import ctypes
    L.require_device(x, *blobs)
    if x.dtype != torch.float32:
        raise TypeError("nf_rqs_fused_x3_chain is fp32 only")
    x = x.contiguous()
    B, D = x.shape
    y = torch.empty_like(x)
    if logdet is None:
        logdet = torch.empty(B, dtype=x.dtype, device=x.device)
        acc = L.LD_WRITE
    elif acc is None:
        acc = L.LD_ADD
    n = len(blobs)
    bp = (ctypes.c_void_p * n)(*[b.data_ptr() for b in blobs])
    pp = (ctypes.c_int * n)(*[int(v) for v in parities])
    rc = L.lib().nf_rqs_fused_x3_chain(ptr(x), ptr(y), ptr(logdet), bp, pp, i32(n), i32(int(fuse_lu)), i64(B), i32(D),
                                       i32(hidden), i32(num_blocks), i32(K), f64(tail_bound), f64(min_bin_width),
                                       f64(min_bin_height), f64(min_derivative), i32(direction), i32(acc), L.stream())
    L.check(rc, "nf_rqs_fused_x3_chain")
    return y, logdet


def nsf_wide_tables(uw, uh, ud, K, tail_bound, min_bin_width=1e-3, min_bin_height=1e-3, min_derivative=1e-3):
    """Knot tables (n_identity, 3 (K + 1)) of the batch-shared spline for nf_nsf_wide (nsf/coupling.py:170-259)."""
    L.require_device(uw, uh, ud)
    tabs = torch.empty(uw.shape[0], 3 * (K + 1), dtype=torch.float32, device=uw.device)
    rc = L.lib().nf_nsf_wide_tables(ptr(uw.contiguous()), ptr(uh.contiguous()), ptr(ud.contiguous()), ptr(tabs), i32(uw.shape[0]),
                                    i32(K), f64(float(tail_bound)), f64(min_bin_width), f64(min_bin_height), f64(min_derivative),
                                    L.stream())
    L.check(rc, "nf_nsf_wide_tables")
    return tabs


def nsf_wide(x, blob, table, tabs, hidden_padded, direction, tail_bound, min_bin_width=1e-3, min_bin_height=1e-3,
             min_derivative=1e-3, logdet=None, acc=None, lu_logdet=None, K=8):
    """CoupledRationalQuadraticSpline beyond the benchmark kernel's shapes as one launch (nf_nsf_wide_k); blob / table from
    flows/nsf_wide_pack.pack_nsf_wide, tabs from nsf_wide_tables (both for K bins: 4 | 8 | 16); lu_logdet: device scalar of the
    LULinearPermute packed with it."""
    L.require_device(x, blob, table, tabs, lu_logdet)
    if x.dtype != torch.float32:
        raise NotImplementedError("nsf_wide: float32 only")
    B, D = x.shape
    x = x.contiguous()
    y = torch.empty_like(x)
    if logdet is None:
        logdet = torch.empty(B, dtype=x.dtype, device=x.device)
        acc = L.LD_WRITE
    elif acc is None:
        acc = L.LD_ADD
    rc = L.lib().nf_nsf_wide_k(ptr(x), ptr(y), ptr(logdet), ptr(blob), ptr(table), ptr(tabs), ptr(lu_logdet), i64(B), i32(D), i32(hidden_padded),
                               i32(int(K)), i32(direction), i32(acc), f64(float(tail_bound)), f64(min_bin_width), f64(min_bin_height),
                               f64(min_derivative), L.stream())
    L.check(rc, "nf_nsf_wide_k")
    return y, logdet


def made_forward_affine(x, blob, table, hidden_padded, logdet=None, acc=None):
    """MaskedAffineAutoregressive.forward (autoregressive.py:24-27, :101-110 over nets/made.py:296-304) as one launch
    (nf_made_forward_affine); blob / table from flows/made_pack.pack_made_forward."""
    L.require_device(x, blob, table)
    if x.dtype != torch.float32:
        raise NotImplementedError("made_forward_affine: float32 only")
    B, D = x.shape
    x = x.contiguous()
    y = torch.empty_like(x)
    if logdet is None:
        logdet = torch.empty(B, dtype=x.dtype, device=x.device)
        acc = L.LD_WRITE
    elif acc is None:
        acc = L.LD_ADD
    rc = L.lib().nf_made_forward_affine(ptr(x), ptr(y), ptr(logdet), ptr(blob), ptr(table), i64(B), i32(D), i32(hidden_padded),
                                        i32(acc), L.stream())
    L.check(rc, "nf_made_forward_affine")
    return y, logdet


def made_forward_spline(x, blob, table, hidden_padded, tail_bound, min_bin_width=1e-3, min_bin_height=1e-3, min_derivative=1e-3,
                        logdet=None, acc=None):
    """neural_spline/autoregressive.py:94-134 density direction (MADE + 8-bin spline with linear tails) as one launch
    (nf_made_forward_spline); blob / table from flows/made_pack.pack_made_forward(made, 23, spline=True)."""
    L.require_device(x, blob, table)
    if x.dtype != torch.float32:
        raise NotImplementedError("made_forward_spline: float32 only")
    B, D = x.shape
    x = x.contiguous()
    y = torch.empty_like(x)
    if logdet is None:
        logdet = torch.empty(B, dtype=x.dtype, device=x.device)
        acc = L.LD_WRITE
    elif acc is None:
        acc = L.LD_ADD
    rc = L.lib().nf_made_forward_spline(ptr(x), ptr(y), ptr(logdet), ptr(blob), ptr(table), i64(B), i32(D), i32(hidden_padded),
                                        i32(acc), f64(float(tail_bound)), f64(min_bin_width), f64(min_bin_height),
                                        f64(min_derivative), L.stream())
    L.check(rc, "nf_made_forward_spline")
    return y, logdet


def made_forward(x, blob, table, hidden_padded, mult):
    """MADE.forward (nets/made.py:296-304) as one launch (nf_made_forward): (B, mult D) parameters."""
    L.require_device(x, blob, table)
    if x.dtype != torch.float32:
        raise NotImplementedError("made_forward: float32 only")
    B, D = x.shape
    x = x.contiguous()
    params = torch.empty(B, mult * D, dtype=x.dtype, device=x.device)
    rc = L.lib().nf_made_forward(ptr(x), ptr(params), ptr(blob), ptr(table), i64(B), i32(D), i32(hidden_padded), i32(mult),
                                 L.stream())
    L.check(rc, "nf_made_forward")
    return params


_ZERO1 = {}


def pack_gather(params, src):
    """The packed weight streams from the current parameters (nf_pack_gather): flat = [0, params flattened ...], out = flat[src]."""
    L.require_device(src, *params)
    zero = _ZERO1.get(src.device)
    if zero is None:
        zero = _ZERO1[src.device] = torch.zeros(1, dtype=torch.float32, device=src.device)
    out = torch.empty(src.numel(), dtype=torch.float32, device=src.device)
    if len(params) <= 16 and all(p.dtype == torch.float32 and p.is_contiguous() for p in params):
        # round 6: straight from the parameter tensors (no torch.cat per module and step)
        pp = (C.c_void_p * len(params))(*[p.data_ptr() for p in params])
        nn_ = (C.c_int64 * len(params))(*[p.numel() for p in params])
        rc = L.lib().nf_pack_gather_multi(pp, nn_, i32(len(params)), ptr(src), ptr(out), i64(src.numel()), L.stream())
        L.check(rc, "nf_pack_gather_multi")
        return out
    flat = torch.cat([zero] + [p.detach().reshape(-1) for p in params])
    rc = L.lib().nf_pack_gather(ptr(flat), ptr(src), ptr(out), i64(src.numel()), L.stream())
    L.check(rc, "nf_pack_gather")
    return out


def pack_gather_batch(param_lists, src):
    """nf_pack_gather_batch: [flat_m[src]] for modules m of ONE structure (param_lists[m] = its <= 8 contiguous float32 parameter
    tensors, the same shapes for every module) -- one launch per 32 modules."""
    L.require_device(src, *[p for pl in param_lists for p in pl])
    n_mod, n_par = len(param_lists), len(param_lists[0])
    if n_par > 8 or any(len(pl) != n_par for pl in param_lists):
        raise NotImplementedError("pack_gather_batch: <= 8 parameters per module, the same number for all")
    shapes = [tuple(p.shape) for p in param_lists[0]]
    for pl in param_lists:
        if [tuple(p.shape) for p in pl] != shapes or any(p.dtype != torch.float32 or not p.is_contiguous() for p in pl):
            raise NotImplementedError("pack_gather_batch: contiguous float32 parameters of one structure")
    out = torch.empty(n_mod, src.numel(), dtype=torch.float32, device=src.device)
    outs = list(out.unbind(0))
    pp = _ptr_array([p for pl in param_lists for p in pl])
    nn_ = (C.c_int64 * n_par)(*[p.numel() for p in param_lists[0]])
    rc = L.lib().nf_pack_gather_batch(pp, nn_, i32(n_par), ptr(src), _ptr_array(outs), i64(src.numel()), i32(n_mod), L.stream())
    L.check(rc, "nf_pack_gather_batch")
    return outs


def made_forward_train(x, blob, table, hidden_padded, out_features, num_blocks, rows=None, features=None):
    """MADE.forward / ResidualNet.forward under autograd (nf_made_forward_train): (params (B, out_features), save (2 NB + 1, Bp, Hp)
    pre-activations, bits (Bp / 64, 2 NB, 2, 512) ReLU signs), Bp = B rounded up to 64 -- the operands of made_backward / made_wgrad.
    out_features = mult D for a MADE (the table's hdr[12])."""
    L.require_device(x, blob, table)
    if x.dtype != torch.float32:
        raise NotImplementedError("made_forward_train: float32 only")
    B, D = x.shape
    if rows is not None:           # x is a padded buffer (rows >= B, row stride in the table's hdr[14]): the conv path
        B, D = rows, features
    x = x.contiguous()
    Bp = (B + 63) // 64 * 64
    params = torch.empty(B, out_features, dtype=x.dtype, device=x.device)
    save = torch.empty(2 * num_blocks + 1, Bp, hidden_padded, dtype=x.dtype, device=x.device)
    bits = torch.empty(max(Bp // 64, 1), 2 * num_blocks, 2, 512, dtype=torch.int32, device=x.device)
    rc = L.lib().nf_made_forward_train(ptr(x), ptr(params), ptr(save), ptr(bits), ptr(blob), ptr(table), i64(B), i32(D),
                                       i32(hidden_padded), i32(max(1, out_features // D)), L.stream())
    L.check(rc, "nf_made_forward_train")
    return params, save, bits


def made_backward(g_params, bits, blob, table, D, hidden_padded, num_blocks, rows=None, out_features=None, ld_out=None, want_G=True):
    """The input-gradient chain of MADE (nf_made_backward): g_x (B, D) and every layer's output gradient G (2 NB + 1, Bp, Hp).
    rows / out_features / ld_out: g_params is a padded buffer (row strides in the table's hdr[14], hdr[15]) and so is the returned
    g_x (rows, ld_out): the conv path."""
    L.require_device(g_params, bits, blob, table)
    B = g_params.shape[0] if rows is None else rows
    md = g_params.shape[1] if out_features is None else out_features
    g_params = g_params.contiguous()
    Bp = (B + 63) // 64 * 64
    gx = torch.empty(B, D if ld_out is None else ld_out, dtype=g_params.dtype, device=g_params.device)
    G = torch.empty(2 * num_blocks + 1, Bp, hidden_padded, dtype=g_params.dtype, device=g_params.device) if want_G else None
    rc = L.lib().nf_made_backward(ptr(g_params), ptr(bits), ptr(gx), ptr(G), ptr(blob), ptr(table), i64(B), i32(D),
                                  i32(hidden_padded), i32(max(1, md // D)), L.stream())
    L.check(rc, "nf_made_backward")
    return gx, G


def _pad_rows_cols(t, rows, cols):
    if t.shape[0] == rows and t.shape[1] == cols:
        return t.contiguous()
    return torch.nn.functional.pad(t, (0, cols - t.shape[1], 0, rows - t.shape[0])).contiguous()


def made_wgrad(g_params, x, G, save, wtable, stable, mask, ntiles, nflat, Mp, Dx, rows=None):
    """Every weight / bias gradient of the MADE (nf_made_wgrad): the flat vector in flows/made_pack.pack_made_backward's layout,
    masked entries zero."""
    L.require_device(g_params, x, G, save, wtable, stable, mask)
    B = g_params.shape[0] if rows is None else rows          # (rows: the operands are already padded buffers)
    Bp = G.shape[1]
    gp_pad = _pad_rows_cols(g_params, Bp, Mp)
    x_pad = _pad_rows_cols(x, Bp, Dx)
    grads = torch.zeros(nflat, dtype=torch.float32, device=x.device)
    lib = L.lib()
    n = int(lib.nf_made_wgrad_scratch_floats(i64(B), i32(ntiles)))
    if n < 0:
        L.check(n, "nf_made_wgrad_scratch_floats")
    part = torch.empty(max(n, 1), dtype=torch.float32, device=x.device)
    rc = lib.nf_made_wgrad(ptr(gp_pad), ptr(x_pad), ptr(G), ptr(save), ptr(grads), ptr(mask), ptr(part), ptr(wtable), ptr(stable),
                           i32(ntiles), i64(B), L.stream())
    L.check(rc, "nf_made_wgrad")
    return grads


def made_wgrad_pos(g_params, x, gscratch, fscratch, wtable, stable, mask, ntiles, nflat, Mp, Dx, num_layers, positions):
    """nf_made_wgrad_pos (round 6): nf_made_wgrad with the hidden operands read from the one-pass kernels' scratches in place --
    gscratch from maf_solve_t(return_scratch=True), fscratch from maf_inverse_bits(return_scratch=True); tables from
    flows/maf_pack.position_wgrad_tables; mask / nflat / Mp / Dx of the MADE's ordinary backward pack (same flat layout).  The batch must
    be a multiple of 64 rows."""
    L.require_device(g_params, x, gscratch, fscratch, wtable, stable, mask)
    B = g_params.shape[0]
    if B % 64:
        raise NotImplementedError("made_wgrad_pos: a multiple of 64 rows")
    gp_pad = _pad_rows_cols(g_params, B, Mp)
    x_pad = _pad_rows_cols(x, B, Dx)
    grads = torch.zeros(nflat, dtype=torch.float32, device=x.device)
    lib = L.lib()
    n = int(lib.nf_made_wgrad_scratch_floats(i64(B), i32(ntiles)))
    if n < 0:
        L.check(n, "nf_made_wgrad_scratch_floats")
    part = torch.empty(max(n, 1), dtype=torch.float32, device=x.device)
    rc = lib.nf_made_wgrad_pos(ptr(gp_pad), ptr(x_pad), ptr(gscratch), ptr(fscratch), ptr(grads), ptr(mask), ptr(part), ptr(wtable),
                               ptr(stable), i32(ntiles), i64(B), i32(num_layers), i32(positions), L.stream())
    L.check(rc, "nf_made_wgrad_pos")
    return grads


def maf_inverse(z, blob, table, hidden_padded, logdet=None, acc=None, num_blocks=2, table_host=None):
    """autoregressive.py:29-38 + :114-128 in one pass; blob/table from flows/maf_pack.pack_made.  config.maf_halves (default):
    nf_maf_inverse_h (32 samples per wave, 1..3 residual blocks); otherwise round 2's nf_maf_inverse (two blocks only).
    `table_host`: the host (numpy int32) copy of a FORMAT-1 table (pack_made(tri=True)) -> nf_maf_inverse_h_tri; a format-1 pack
    must come with it (the format-0 entry points cannot read its regular tiles)."""
    L.require_device(z, blob, table)
    if z.dtype != torch.float32:
        raise NotImplementedError("maf_inverse: float32 only")
    from . import config
    B, D = z.shape
    z = z.contiguous()
    y = torch.empty_like(z)
    if logdet is None:
        logdet = torch.empty(B, dtype=z.dtype, device=z.device)
        acc = L.LD_WRITE
    elif acc is None:
        acc = L.LD_ADD
    lib = L.lib()
    if table_host is not None:
        import numpy as np
        th = np.ascontiguousarray(table_host, dtype=np.int32)
        if int(th[7]) != 1:
            raise ValueError("maf_inverse: table_host is given for format-1 packs only")
        n = lib.nf_maf_inverse_h_scratch_floats(i64(B), i32(D), i32(hidden_padded), i32(num_blocks))
        scratch = torch.empty(max(int(n), 1), dtype=torch.float32, device=z.device)
        rc = lib.nf_maf_inverse_h_tri(ptr(z), ptr(y), ptr(logdet), ptr(blob), ptr(table), C.c_void_p(th.ctypes.data), ptr(scratch),
                                      i64(B), i32(D), i32(hidden_padded), i32(num_blocks), i32(acc), L.stream())
        L.check(rc, "nf_maf_inverse_h_tri")
        return y, logdet
    if config.maf_halves or num_blocks != 2:
        n = lib.nf_maf_inverse_h_scratch_floats(i64(B), i32(D), i32(hidden_padded), i32(num_blocks))
        scratch = torch.empty(max(int(n), 1), dtype=torch.float32, device=z.device)
        rc = lib.nf_maf_inverse_h(ptr(z), ptr(y), ptr(logdet), ptr(blob), ptr(table), ptr(scratch), i64(B), i32(D),
                                  i32(hidden_padded), i32(num_blocks), i32(acc), L.stream())
        L.check(rc, "nf_maf_inverse_h")
        return y, logdet
    n = lib.nf_maf_inverse_scratch_floats(i64(B), i32(D), i32(hidden_padded))
    scratch = torch.empty(max(int(n), 1), dtype=torch.float32, device=z.device)
    rc = lib.nf_maf_inverse(ptr(z), ptr(y), ptr(logdet), ptr(blob), ptr(table), ptr(scratch), i64(B), i32(D),
                            i32(hidden_padded), i32(acc), L.stream())
    L.check(rc, "nf_maf_inverse")
    return y, logdet


def maf_inverse_bits(z, blob, table, hidden_padded, num_blocks, tiles, table_host=None, return_scratch=False, want_params=False):
    """nf_maf_inverse_h_bits / nf_maf_inverse_h_tri_bits (table_host = the host copy of a format-1 table): the one-pass inverse that also
    leaves the pass's ReLU masks (uint32 words per 32-row wave, tile and lane, in the pack's positions) for maf_solve_t on a
    transposed pack of the same format.  Returns (y, logdet, bits [, scratch]); want_params (round 6, nf_maf_inverse_h_train): MADE's
    output at the solution, (B, 2 D), is appended."""
    L.require_device(z, blob, table)
    if z.dtype != torch.float32:
        raise NotImplementedError("maf_inverse_bits: float32 only")
    B, D = z.shape
    z = z.contiguous()
    y = torch.empty_like(z)
    logdet = torch.empty(B, dtype=z.dtype, device=z.device)
    lib = L.lib()
    n = lib.nf_maf_inverse_h_scratch_floats(i64(B), i32(D), i32(hidden_padded), i32(num_blocks))
    scratch = torch.empty(max(int(n), 1), dtype=torch.float32, device=z.device)
    bits = torch.empty(max((B + 31) // 32 * tiles * 64 * num_blocks, 1), dtype=torch.int32, device=z.device)
    if want_params:
        import numpy as np
        prm = torch.empty(B, 2 * D, dtype=torch.float32, device=z.device)
        th = None if table_host is None else np.ascontiguousarray(table_host, dtype=np.int32)
        rc = lib.nf_maf_inverse_h_train(ptr(z), ptr(y), ptr(logdet), ptr(blob), ptr(table), C.c_void_p(None if th is None else th.ctypes.data),
                                        ptr(scratch), ptr(bits), ptr(prm), i64(B), i32(D), i32(hidden_padded), i32(num_blocks),
                                        i32(L.LD_WRITE), L.stream())
        L.check(rc, "nf_maf_inverse_h_train")
        return (y, logdet, bits, scratch, prm) if return_scratch else (y, logdet, bits, prm)
    if table_host is not None:
        import numpy as np
        th = np.ascontiguousarray(table_host, dtype=np.int32)
        rc = lib.nf_maf_inverse_h_tri_bits(ptr(z), ptr(y), ptr(logdet), ptr(blob), ptr(table), C.c_void_p(th.ctypes.data), ptr(scratch),
                                           ptr(bits), i64(B), i32(D), i32(hidden_padded), i32(num_blocks), i32(L.LD_WRITE), L.stream())
        L.check(rc, "nf_maf_inverse_h_tri_bits")
        return (y, logdet, bits, scratch) if return_scratch else (y, logdet, bits)
    rc = lib.nf_maf_inverse_h_bits(ptr(z), ptr(y), ptr(logdet), ptr(blob), ptr(table), ptr(scratch), ptr(bits), i64(B), i32(D),
                                   i32(hidden_padded), i32(num_blocks), i32(L.LD_WRITE), L.stream())
    L.check(rc, "nf_maf_inverse_h_bits")
    return (y, logdet, bits, scratch) if return_scratch else (y, logdet, bits)


def maf_solve_t(x, params, gx, gld, bits, blob, table, hidden_padded, num_blocks, return_scratch=False, table_host=None):
    """nf_maf_solve_t: v with  v s + J^T g_p(v, g_ld) = g_x  in one pass (the implicit backward of the MAF inverse);
    blob / table from flows/maf_pack.pack_made_transposed.  return_scratch: (v, scratch) -- the activation scratch for
    maf_scratch_rows."""
    L.require_device(x, params, gx, gld, bits, blob, table)
    if x.dtype != torch.float32:
        raise NotImplementedError("maf_solve_t: float32 only")
    B, D = x.shape
    x, params, gx = x.contiguous(), params.contiguous(), gx.contiguous()
    v = torch.empty_like(x)
    lib = L.lib()
    n = lib.nf_maf_solve_t_scratch_floats(i64(B), i32(D), i32(hidden_padded), i32(num_blocks))
    scratch = torch.empty(max(int(n), 1), dtype=torch.float32, device=x.device)
    from . import config
    if table_host is not None and config.maf_solve_fast:      # round 6: the regular-8 tiles on the statically unrolled sequential part
        import numpy as np
        th = np.ascontiguousarray(table_host, dtype=np.int32)
        rc = lib.nf_maf_solve_t_tri(ptr(x), ptr(params), ptr(gx), ptr(None if gld is None else gld.contiguous()), ptr(bits), ptr(v),
                                    ptr(blob), ptr(table), C.c_void_p(th.ctypes.data), ptr(scratch), i64(B), i32(D), i32(hidden_padded),
                                    i32(num_blocks), L.stream())
        L.check(rc, "nf_maf_solve_t_tri")
        return (v, scratch) if return_scratch else v
    rc = lib.nf_maf_solve_t(ptr(x), ptr(params), ptr(gx), ptr(None if gld is None else gld.contiguous()), ptr(bits), ptr(v), ptr(blob),
                            ptr(table), ptr(scratch), i64(B), i32(D), i32(hidden_padded), i32(num_blocks), L.stream())
    L.check(rc, "nf_maf_solve_t")
    return (v, scratch) if return_scratch else v


def maf_scratch_rows(scratch, pos_of_col, B, num_blocks, hidden_padded, sign=1.0, reverse_layers=False):
    """nf_maf_scratch_rows: the one-pass kernels' activation scratch as (2 num_blocks + 1, Bp, len(pos_of_col)) row-major tensors."""
    L.require_device(scratch, pos_of_col)
    ldo = pos_of_col.numel()
    Bp = (B + 63) // 64 * 64
    out = torch.empty(2 * num_blocks + 1, Bp, ldo, dtype=torch.float32, device=scratch.device)
    rc = L.lib().nf_maf_scratch_rows(ptr(scratch), ptr(pos_of_col), ptr(out), i64(B), i32(num_blocks), i32(hidden_padded), i32(ldo),
                                     f64(sign), i32(int(reverse_layers)), L.stream())
    L.check(rc, "nf_maf_scratch_rows")
    return out


def maf_scratch_layer(scratch, pos_of_col, B, num_blocks, hidden_padded, layer):
    """nf_maf_scratch_layer: ONE layer of a one-pass kernel's activation scratch as a (Bp, len(pos_of_col)) row-major tensor."""
    L.require_device(scratch, pos_of_col)
    ldo = pos_of_col.numel()
    Bp = (B + 63) // 64 * 64
    out = torch.empty(Bp, ldo, dtype=torch.float32, device=scratch.device)
    rc = L.lib().nf_maf_scratch_layer(ptr(scratch), ptr(pos_of_col), ptr(out), i64(B), i32(num_blocks), i32(hidden_padded), i32(ldo),
                                      i32(layer), L.stream())
    L.check(rc, "nf_maf_scratch_layer")
    return out


def arnsf_inverse(z, blob, table, hidden_padded, K, tails, tail_bound, min_bin_width=1e-3, min_bin_height=1e-3,
                  min_derivative=1e-3, logdet=None, acc=None):
    """neural_spline/autoregressive.py:94-134 inverse over affine/autoregressive.py:29-38 in one pass
    (nf_arnsf_inverse); blob/table from flows/maf_pack.pack_made(made, mult, rows=True)."""
    L.require_device(z, blob, table)
    if z.dtype != torch.float32:
        raise NotImplementedError("arnsf_inverse: float32 only")
    B, D = z.shape
    z = z.contiguous()
    y = torch.empty_like(z)
    if logdet is None:
        logdet = torch.empty(B, dtype=z.dtype, device=z.device)
        acc = L.LD_WRITE
    elif acc is None:
        acc = L.LD_ADD
    n = L.lib().nf_maf_inverse_scratch_floats(i64(B), i32(D), i32(hidden_padded))
    scratch = torch.empty(max(int(n), 1), dtype=torch.float32, device=z.device)
    rc = L.lib().nf_arnsf_inverse(ptr(z), ptr(y), ptr(logdet), ptr(blob), ptr(table), ptr(scratch), i64(B), i32(D),
                                  i32(hidden_padded), i32(K), i32(L.TAILS[tails]), f64(tail_bound),
                                  f64(min_bin_width), f64(min_bin_height), f64(min_derivative), i32(acc), L.stream())
    L.check(rc, "nf_arnsf_inverse")
    return y, logdet


GLOW_CONV_WIDE, GLOW_CONV_SMALL, GLOW_CONV_TINY = 0, 1, 2


def glow_convnet_layout(B, H, W):
    """Which nf_glow_convnet kernel takes (B, *, H, W) inputs: GLOW_CONV_WIDE, _SMALL, _TINY or None."""
    code = L.lib().nf_glow_convnet_layout(i64(B), i32(H), i32(W))
    return code if code >= 0 else None


def glow_convnet_pack(w1, b1, w2, b2, w3, b3, layout=GLOW_CONV_WIDE):
    """Packed weights of a GlowBlock conditioner for glow_convnet (nf_glow_convnet_pack); None for unsupported shapes."""
    import ctypes
    L.require_device(w1, b1, w2, b2, w3, b3)
    Cin, Cout, hidden = w1.shape[1], w3.shape[0], w1.shape[0]
    lib = L.lib()
    lib.nf_glow_convnet_pack_size.restype = ctypes.c_int64
    size = lib.nf_glow_convnet_pack_size(i32(Cin), i32(Cout), i32(hidden))
    if size <= 0 or (layout == GLOW_CONV_SMALL and Cout > 48):
        return None
    blob = torch.empty(size // 4, dtype=torch.float32, device=w1.device)
    rc = lib.nf_glow_convnet_pack(ptr(blob), ptr(w1.contiguous()), ptr(b1.contiguous()), ptr(w2.contiguous()),
                                  ptr(b2.contiguous()), ptr(w3.contiguous()), ptr(b3.contiguous()), i32(Cin), i32(Cout),
                                  i32(hidden), i32(layout), L.stream())
    L.check(rc, "nf_glow_convnet_pack")
    return blob


def glow_convnet(x, blob, Cout, slope, layout=GLOW_CONV_WIDE, hidden=256):
    """cnn.py:5-63 for the GlowBlock network in one launch (nf_glow_convnet).  x: (B, Cin, H, W), possibly a channel
    slice of a contiguous NCHW tensor (planes contiguous, arbitrary image stride); blob packed for the same layout."""
    L.require_device(x, blob)
    B, Cin, H, W = x.shape
    if x.dtype != torch.float32 or x.stride(3) != 1 or x.stride(2) != W or x.stride(1) != H * W:
        raise NotImplementedError("glow_convnet: float32 with contiguous (H, W) planes")
    out = torch.empty(B, Cout, H, W, dtype=x.dtype, device=x.device)
    rc = L.lib().nf_glow_convnet(ptr_any(x), i64(x.stride(0) if B > 1 else Cin * H * W), ptr(out), ptr(blob), i64(B),
                                 i32(Cin), i32(H), i32(W), i32(Cout), i32(hidden), f64(slope), i32(layout), L.stream())
    L.check(rc, "nf_glow_convnet")
    return out


def glow_block_table(entries, device):
    """DEVICE pointer table of nf_glow_level: `entries` = [(blob, mix_w, mix_b, mix_logdet), ...] in processing order (all
    float32, contiguous, on `device`).  Returns (int64 tensor of 4 n pointers, the entries -- keep both alive while a launch
    or a recorded graph may use the table)."""
    ptrs = []
    for blob, w, b_, l in entries:
        L.require_device(blob, w, b_, l)
        for t in (blob, w, b_, l):
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise ValueError("glow_block_table: float32 contiguous tensors")
            ptrs.append(t.data_ptr())
    return torch.tensor(ptrs, dtype=torch.int64).to(device), entries


def glow_level(in0, in1, in_squeezed, C, H, W, table, nblocks, layout, slope, scale_map, direction, cout0=None,
               out_squeezed=False, logdet=None, acc=None, hidden=256):
    """`nblocks` GlowBlocks of one shape in ONE persistent launch (nf_glow_level) with the level's glue folded in.
    Input: in_squeezed -> in0 is (B, C/4, 2H, 2W) read through Squeeze.inverse; else channels of in0 (B, cin0, H, W) then
    in1 (B, C - cin0, H, W) (in1 None: in0 has all C channels).  Output: out_squeezed -> (B, C/4, 2H, 2W) through
    Squeeze.forward; cout0 < C -> two tensors split after cout0 channels; else one (B, C, H, W) tensor.
    Returns (out0, out1 or None, logdet)."""
    L.require_device(in0, in1, table)
    if in0.dtype != torch.float32 or (in1 is not None and in1.dtype != torch.float32):
        raise NotImplementedError("glow_level: float32 only")
    in0 = in0.contiguous()
    in1 = None if in1 is None else in1.contiguous()
    B = in0.shape[0]
    cin0 = C if in_squeezed else in0.shape[1]
    if in_squeezed:
        assert tuple(in0.shape[1:]) == (C // 4, 2 * H, 2 * W), (in0.shape, C, H, W)
    else:
        assert tuple(in0.shape[2:]) == (H, W) and (cin0 == C or (in1 is not None and in1.shape[1] == C - cin0))
    if cout0 is None or out_squeezed:
        cout0 = C
    if out_squeezed:
        out0, out1 = torch.empty(B, C // 4, 2 * H, 2 * W, dtype=in0.dtype, device=in0.device), None
    else:
        out0 = torch.empty(B, cout0, H, W, dtype=in0.dtype, device=in0.device)
        out1 = torch.empty(B, C - cout0, H, W, dtype=in0.dtype, device=in0.device) if cout0 < C else None
    if logdet is None:
        logdet = torch.empty(B, dtype=in0.dtype, device=in0.device)
        acc = L.LD_WRITE
    elif acc is None:
        acc = L.LD_ADD
    rc = L.lib().nf_glow_level(ptr(in0), ptr(in1), i32(cin0), i32(1 if in_squeezed else 0), ptr(out0), ptr(out1), i32(cout0),
                               i32(1 if out_squeezed else 0), ptr(logdet), ptr(table), i32(nblocks), i64(B), i32(C), i32(H),
                               i32(W), i32(hidden), f64(slope), i32(L.SCALE[scale_map]), i32(direction), i32(acc), i32(layout),
                               L.stream())
    L.check(rc, "nf_glow_level")
    return out0, out1, logdet


def glow_block(z, blob, layout, mix_w, mix_b, mix_logdet, slope, scale_map, direction, logdet=None, acc=None, hidden=256,
               table=None):
    """GlowBlock.forward (direction 0) / .inverse (1) in one launch (nf_glow_level with one block): channel-split affine
    coupling with the packed conditioner `blob`, and [Invertible1x1Conv, ActNorm] as the per-pixel affine map (mix_w,
    mix_b) with log|det| per pixel mix_logdet (0-dim device tensor).  `table`: a cached glow_block_table of the block."""
    if table is None:
        table = glow_block_table([(blob, mix_w.contiguous(), mix_b.contiguous(), mix_logdet.to(z.dtype).contiguous())],
                                 z.device)[0]
    B, C, H, W = z.shape
    y, _, logdet = glow_level(z, None, False, C, H, W, table, 1, layout, slope, scale_map, direction, logdet=logdet, acc=acc,
                              hidden=hidden)
    return y, logdet


def logit(z, alpha, direction, logdet=None, acc=None):
    """transforms.py:8-47.  direction 0 = Logit.forward (sigmoid side), 1 = Logit.inverse (logit side)."""
    L.require_device(z)
    z = z.contiguous()
    B = z.shape[0]
    inner = z[0].numel() if B else int(math.prod(z.shape[1:]))
    y = torch.empty_like(z)
    if logdet is None:
        logdet = torch.empty(B, dtype=z.dtype, device=z.device)
        acc = L.LD_WRITE
    elif acc is None:
        acc = L.LD_ADD
    rc = L.lib().nf_logit(ptr(z), ptr(y), ptr(logdet), i64(B), i64(inner), f64(alpha), i32(direction), i32(acc),
                          i32(L.dtype_code(z)), L.stream())
    L.check(rc, "nf_logit")
    return y, logdet


def diag_gaussian_log_prob_rows(z, loc_rows, log_scale_rows, row_idx=None, ls_shift=0.0, out=None, acc=None):
    """distributions/base.py:326-345: one (loc, log_scale) row per sample, picked by `row_idx` (class labels) or row b."""
    L.require_device(z, loc_rows, log_scale_rows, row_idx)
    B = z.shape[0]
    z = z.contiguous()
    d = z[0].numel() if B else int(math.prod(z.shape[1:]))
    loc_rows = loc_rows.contiguous().view(-1, d)
    log_scale_rows = log_scale_rows.contiguous().view(-1, d)
    if out is None:
        out = torch.empty(B, dtype=z.dtype, device=z.device)
        acc = L.LD_WRITE
    elif acc is None:
        acc = L.LD_ADD
    if row_idx is not None:
        row_idx = row_idx.to(torch.long).contiguous()
    rc = L.lib().nf_diag_gaussian_log_prob_rows(ptr(z), ptr(loc_rows), ptr(log_scale_rows), ptr(row_idx),
                                                i64(loc_rows.shape[0]), f64(ls_shift), ptr(out), i64(B), i64(d), i32(acc),
                                                i32(L.dtype_code(z)), L.stream())
    L.check(rc, "nf_diag_gaussian_log_prob_rows")
    return out


def linear_wgrad(dy, x, want_bias=True, relu_x=False, skip_every=0):
    """dW = dy^T x (x -> relu(x) with relu_x), db = dy.sum(0) for a Linear layer (nf_linear_wgrad[_act|_skip], split-K fp32
    MFMA, deterministic reduction).  skip_every > 1: every skip_every-th column of dy is padding and has no output row."""
    L.require_device(dy, x)
    if dy.dtype != torch.float32 or x.dtype != torch.float32:
        raise NotImplementedError("linear_wgrad: float32 only")
    dy, x = dy.contiguous(), x.contiguous()
    B, M = dy.shape
    N = x.shape[1]
    Mo = M - M // skip_every if skip_every else M
    dW = torch.empty(Mo, N, dtype=torch.float32, device=dy.device)
    db = torch.empty(Mo, dtype=torch.float32, device=dy.device) if want_bias else None
    n = L.lib().nf_linear_wgrad_scratch_floats(i64(B), i32(M), i32(N))
    scratch = torch.empty(max(int(n), 1), dtype=torch.float32, device=dy.device)
    rc = L.lib().nf_linear_wgrad_skip(ptr(dy), ptr(x), ptr(dW), ptr(db), ptr(scratch), i64(B), i32(M), i32(N), i32(0),
                                      i32(int(relu_x)), i32(int(skip_every)), L.stream())
    L.check(rc, "nf_linear_wgrad_skip")
    return dW, db


def linear_wgrad_pair(dy0, x0, dy1, x1, relu_x=False):
    """(dW0, db0, dW1, db1) of two same-shape Linear layers in one partial launch + one reduction (nf_linear_wgrad_pair)."""
    L.require_device(dy0, x0, dy1, x1)
    dy0, x0, dy1, x1 = dy0.contiguous(), x0.contiguous(), dy1.contiguous(), x1.contiguous()
    if dy0.shape != dy1.shape or x0.shape != x1.shape or any(t.dtype != torch.float32 for t in (dy0, x0, dy1, x1)):
        raise ValueError("linear_wgrad_pair: two float32 problems of the same shape")
    B, M = dy0.shape
    N = x0.shape[1]
    out = torch.empty(2, M * N + M, dtype=torch.float32, device=dy0.device)     # dW | db per problem
    n = int(L.lib().nf_linear_wgrad_scratch_floats(i64(B), i32(M), i32(N)))
    scratch = torch.empty(max(2 * n, 1), dtype=torch.float32, device=dy0.device)
    w0, b0, w1, b1 = out[0, :M * N], out[0, M * N:], out[1, :M * N], out[1, M * N:]
    rc = L.lib().nf_linear_wgrad_pair(ptr(dy0), ptr(x0), ptr(w0), ptr(b0), ptr(dy1), ptr(x1), ptr(w1), ptr(b1), ptr(scratch),
                                      i64(B), i32(M), i32(N), i32(0), i32(int(relu_x)), L.stream())
    L.check(rc, "nf_linear_wgrad_pair")
    return w0.view(M, N), b0, w1.view(M, N), b1


def mfma_clock_mhz(device, iters=20000):
    """Shader clock (MHz) under fp32-MFMA load (nf_mfma_clock_probe): what the matrix pipe runs at while a kernel keeps it busy."""
    out = torch.zeros(2, dtype=torch.int64, device=device)
    sink = torch.zeros(1, dtype=torch.float32, device=device)
    L.check(L.lib().nf_mfma_clock_probe(ptr(out), ptr(sink), i32(iters), L.stream()), "nf_mfma_clock_probe")
    c, w = out.tolist()
    return 100.0 * c / max(w, 1)


def lu_fwd(x, UpT, LT, bias=None, ld_const=None, ld_sign=1.0, logdet=None, acc=None):
    """(u, y, logdet): u = U x[perm], y = L u + bias per row with the constant log-det, D = 64, LDS-DMA tiles (nf_lu_fwd); the
    arguments of rows_matvec2 with the TRANSPOSED factor images (UpT, LT of lu_factors)."""
    L.require_device(x, UpT, LT, bias, ld_const, logdet)
    x = x.contiguous()
    B, D = x.shape
    u, y = torch.empty_like(x), torch.empty_like(x)
    if ld_const is not None and logdet is None:
        logdet, acc = torch.empty(B, dtype=x.dtype, device=x.device), L.LD_WRITE
    elif acc is None:
        acc = L.LD_ADD
    rc = L.lib().nf_lu_fwd(ptr(x), ptr(UpT.contiguous()), ptr(LT.contiguous()), ptr(bias), ptr(u), ptr(y),
                           ptr(logdet if ld_const is not None else None), ptr(ld_const), f64(ld_sign), i32(acc), i64(B), i32(D),
                           L.stream())
    L.check(rc, "nf_lu_fwd")
    return u, y, logdet


def lu_bwd(gy, u, x, Lm, Up, db_out=None):
    """(gx, dL, db, dUp) of LULinearPermute's batch side in the density direction, D = 64, one pass over the rows (nf_lu_bwd);
    db_out: where the bias gradient (D floats) is written instead of a new tensor."""
    L.require_device(gy, u, x, Lm, Up)
    gy, u, x, Lm, Up = gy.contiguous(), u.contiguous(), x.contiguous(), Lm.contiguous(), Up.contiguous()
    B, D = gy.shape
    lib = L.lib()
    lib.nf_lu_bwd_scratch_floats.restype = C.c_int64
    n = int(lib.nf_lu_bwd_scratch_floats(i64(B)))
    if n <= 0 or D != 64 or gy.dtype != torch.float32:
        raise NotImplementedError("lu_bwd: float32, D = 64, batch a multiple of 64")
    scratch = torch.empty(n, dtype=torch.float32, device=gy.device)
    out = torch.empty(2 * D * D + D, dtype=torch.float32, device=gy.device)
    dL, dUp, db = out[:D * D], out[D * D:2 * D * D], out[2 * D * D:]
    if db_out is not None:
        if db_out.numel() != D or db_out.dtype != torch.float32 or not db_out.is_contiguous() or db_out.device != gy.device:
            raise ValueError("lu_bwd: db_out = a contiguous float32 (D) tensor on the inputs' device")
        db = db_out
    gx = torch.empty_like(gy)
    rc = lib.nf_lu_bwd(ptr(gy), ptr(u), ptr(x), ptr(Lm), ptr(Up), ptr(gx), ptr(dL), ptr(db), ptr(dUp), ptr(scratch), i64(B),
                       i32(D), L.stream())
    L.check(rc, "nf_lu_bwd")
    return gx, dL.view(D, D), db, dUp.view(D, D)


def resblock_bwd(gh, t, h_in, w1, w2, x=None, wfull=None, gx=None, col_map=None, n_cols=0):
    """Backward of one residual block (hidden 128) in one pass over the rows (nf_resblock_bwd): returns
    (gh_in, dW1, db1, dW2, db2); with x / wfull / gx also the initial Linear layer behind the block:
    gx += gh_in @ wfull.t() in place (wfull (64, 128): the layer's weight transposed on full rows), and
    (None, dW1, db1, dW2, db2, dW0 (128, 64), db0) is returned; col_map (64 int32, -1 = drop):
    dW0 is compacted to the (128, n_cols) columns it names."""
    L.require_device(gh, t, h_in, w1, w2, x, wfull, gx, col_map)
    gh, t, h_in, w1, w2 = gh.contiguous(), t.contiguous(), h_in.contiguous(), w1.contiguous(), w2.contiguous()
    if any(v.dtype != torch.float32 for v in (gh, t, h_in, w1, w2)):
        raise ValueError("resblock_bwd: float32 only")
    B, H = gh.shape
    init = x is not None
    lib = L.lib()
    lib.nf_resblock_bwd_scratch_floats.restype = C.c_int64
    n = int(lib.nf_resblock_bwd_scratch_floats(i64(B), i32(int(init))))
    if n <= 0 or H != 128:
        raise NotImplementedError("resblock_bwd: hidden 128, batch a multiple of 64")
    scratch = torch.empty(n, dtype=torch.float32, device=gh.device)
    out = torch.empty(2, H * H + H, dtype=torch.float32, device=gh.device)      # (dW2 | db2), (dW1 | db1)
    w2g, b2g, w1g, b1g = out[0, :H * H], out[0, H * H:], out[1, :H * H], out[1, H * H:]
    if init:
        if not (x.is_contiguous() and gx.is_contiguous() and wfull.is_contiguous()) or x.shape[1] != 64 or \
                tuple(wfull.shape) != (64, 128):
            raise ValueError("resblock_bwd: contiguous x / gx (B, 64) and wfull (64, 128)")
        nc = int(n_cols) if col_map is not None else 64
        out0 = torch.empty(H * nc + H, dtype=torch.float32, device=gh.device)
        gh_in, w0g, b0g = None, out0[:H * nc], out0[H * nc:]
    else:
        gh_in, w0g, b0g = torch.empty_like(gh), None, None
    rc = lib.nf_resblock_bwd(ptr(gh), ptr(t), ptr(h_in), ptr(w1), ptr(w2), ptr(gh_in), ptr(w1g), ptr(b1g), ptr(w2g), ptr(b2g),
                             ptr(x), ptr(wfull), ptr(gx), ptr(w0g), ptr(b0g), ptr(col_map), i32(int(n_cols)), ptr(scratch),
                             i64(B), i32(H), i32(64), L.stream())
    L.check(rc, "nf_resblock_bwd")
    if init:
        return None, w1g.view(H, H), b1g, w2g.view(H, H), b2g, w0g.view(H, nc), b0g
    return gh_in, w1g.view(H, H), b1g, w2g.view(H, H), b2g


def bias_leaky_relu_(y, bias, negative_slope):
    """In place y = leaky_relu(y + bias[c]) on a contiguous NCHW tensor (nf_bias_leaky_relu)."""
    L.require_device(y, bias)
    if not y.is_contiguous():
        raise ValueError("bias_leaky_relu_: contiguous NCHW tensor required")
    B, Cc = y.shape[0], y.shape[1]
    HW = int(math.prod(y.shape[2:]))
    rc = L.lib().nf_bias_leaky_relu(ptr(y), ptr(bias.contiguous()), i64(B), i32(Cc), i64(HW), f64(negative_slope),
                                    i32(L.dtype_code(y)), L.stream())
    L.check(rc, "nf_bias_leaky_relu")
    return y


def realnvp_chain(z, blob, d, hmax, direction, logdet=None, acc=None):
    """A stack of MaskedAffineFlow(MLP s, MLP t) / ActNorm layers in one launch (nf_realnvp_chain)."""
    L.require_device(z, blob)
    if z.dtype != torch.float32:
        raise NotImplementedError("realnvp_chain: float32 only")
    z = z.contiguous()
    B = z.shape[0]
    y = torch.empty_like(z)
    if logdet is None:
        logdet = torch.empty(B, dtype=z.dtype, device=z.device)
        acc = L.LD_WRITE
    elif acc is None:
        acc = L.LD_ADD
    rc = L.lib().nf_realnvp_chain(ptr(z), ptr(y), ptr(logdet), ptr(blob), i64(B), i32(d), i32(hmax), i32(direction),
                                  i32(acc), L.stream())
    L.check(rc, "nf_realnvp_chain")
    return y, logdet
