"""ctypes/numpy front-end of the CPU oracle (oracle/libnf_oracle.so).

TEST INFRASTRUCTURE ONLY -- see the header of oracle/nf_oracle.c.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg import this module; the product package (normalizing-flows_amd/) never does.

Every function takes and returns numpy arrays (float32 or float64, C-contiguous) and mirrors one entry
point of include/nf_mi355x.h; `OracleNSF` chains them into the NormalizingFlow.log_prob / sample loops of
normflows/core.py:167-197 for a stack of [CoupledRationalQuadraticSpline, LULinearPermute] layers.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libnf_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("nf_oracle.c", "nf_oracle_impl.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
    return _LIB


def _sfx(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "_f32"
    if dtype == np.float64:
        return "_f64"
    raise TypeError("oracle supports float32/float64, got %s" % dtype)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _c(a, dtype):
    return None if a is None else np.ascontiguousarray(a, dtype=dtype)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


_TAILS = {None: 0, "linear": 1, "circular": 2}


def rqs_spline(x, w, h, d, inverse=False, tails="linear", tail_bound=1.0, left=0.0, right=1.0, bottom=0.0, top=1.0,
               min_bin_width=1e-3, min_bin_height=1e-3, min_derivative=1e-3, wh_div=1.0):
    """utils/splines.py:16-97 / :100-219 on arrays x (...,), w,h (..., K), d (..., K-1|K|K+1)."""
    dt = x.dtype
    K = w.shape[-1]
    xs = _c(x, dt).reshape(-1)
    N = xs.size
    w2, h2, d2 = (_c(a, dt).reshape(N, -1) for a in (w, h, d))
    y = np.empty_like(xs)
    lad = np.empty_like(xs)
    f = getattr(lib(), "nfo_rqs_spline" + _sfx(dt))
    f(_p(xs), _p(w2), C.c_int64(w2.shape[1]), _p(h2), C.c_int64(h2.shape[1]), _p(d2), C.c_int64(d2.shape[1]), _p(y),
      _p(lad), C.c_int64(N), C.c_int(K), C.c_int(_TAILS[tails]), C.c_double(tail_bound), C.c_double(left),
      C.c_double(right), C.c_double(bottom), C.c_double(top), C.c_double(min_bin_width), C.c_double(min_bin_height),
      C.c_double(min_derivative), C.c_double(wh_div), C.c_int(int(inverse)))
    return y.reshape(x.shape), lad.reshape(x.shape)


def rqs_coupling(x, cond, uw, uh, ud, identity_idx, transform_idx, K, mode, y=None, logdet=None, acc=0,
                 tails="linear", tail_bound=3.0, min_bin_width=1e-3, min_bin_height=1e-3, min_derivative=1e-3,
                 wh_div=1.0):
    dt = x.dtype
    x = _c(x, dt)
    B, D = x.shape
    ii, ti = _i64(identity_idx), _i64(transform_idx)
    y = x.copy() if y is None else y
    logdet = np.zeros(B, dt) if logdet is None else logdet
    cond, uw, uh, ud = (_c(a, dt) for a in (cond, uw, uh, ud))
    f = getattr(lib(), "nfo_rqs_coupling" + _sfx(dt))
    f(_p(x), _p(y), _p(logdet), _p(cond), _p(uw), _p(uh), _p(ud), _p(ii), C.c_int(ii.size), _p(ti), C.c_int(ti.size),
      C.c_int64(B), C.c_int(D), C.c_int(K), C.c_int(_TAILS[tails]), C.c_double(tail_bound), C.c_double(min_bin_width),
      C.c_double(min_bin_height), C.c_double(min_derivative), C.c_double(wh_div), C.c_int(mode), C.c_int(acc))
    return y, logdet


def resnet_mlp(x, idx, w_init, b_init, w_blocks, b_blocks, w_final, b_final):
    """nets/resnet.py:92-104; w_blocks/b_blocks are flat lists [blk0.lin0, blk0.lin1, blk1.lin0, ...]."""
    dt = x.dtype
    x = _c(x, dt)
    B, ldx = x.shape
    hidden, in_f = w_init.shape
    out_f = w_final.shape[0]
    assert hidden <= 1024
    idx_a = None if idx is None else _i64(idx)
    ws = [_c(a, dt) for a in w_blocks]
    bs = [_c(a, dt) for a in b_blocks]
    wp = (C.c_void_p * len(ws))(*[a.ctypes.data for a in ws])
    bp = (C.c_void_p * len(bs))(*[a.ctypes.data for a in bs])
    out = np.empty((B, out_f), dt)
    w_init, b_init, w_final, b_final = (_c(a, dt) for a in (w_init, b_init, w_final, b_final))
    f = getattr(lib(), "nfo_resnet_mlp" + _sfx(dt))
    f(_p(x), C.c_int64(ldx), _p(idx_a), C.c_int(in_f), _p(w_init), _p(b_init), wp, bp, C.c_int(len(ws) // 2),
      _p(w_final), _p(b_final), C.c_int(hidden), C.c_int(out_f), _p(out), C.c_int64(B))
    return out


def lu_linear_permute(x, perm, lower_entries, upper_entries, udiag_raw, bias, direction, eps=1e-3, logdet=None, acc=0):
    dt = x.dtype
    x = _c(x, dt)
    B, D = x.shape
    assert D <= 1024
    y = np.empty_like(x)
    logdet = np.zeros(B, dt) if logdet is None else logdet
    pm = _i64(perm)
    lo, up, ud, bi = (_c(a, dt) for a in (lower_entries, upper_entries, udiag_raw, bias))
    f = getattr(lib(), "nfo_lu_linear_permute" + _sfx(dt))
    f(_p(x), _p(y), _p(logdet), _p(pm), _p(lo), _p(up), _p(ud), _p(bi), C.c_int64(B), C.c_int(D), C.c_double(eps),
      C.c_int(direction), C.c_int(acc))
    return y, logdet


def masked_affine(z, b, s, t, direction, logdet=None, acc=0):
    dt = z.dtype
    z = _c(z, dt)
    B = z.shape[0]
    inner = int(np.prod(z.shape[1:]))
    y = np.empty_like(z)
    logdet = np.zeros(B, dt) if logdet is None else logdet
    b, s, t = _c(np.broadcast_to(b, z.shape[1:]) if b is not None else None, dt), _c(s, dt), _c(t, dt)
    f = getattr(lib(), "nfo_masked_affine" + _sfx(dt))
    f(_p(z), _p(b), _p(s), _p(t), _p(y), _p(logdet), C.c_int64(B), C.c_int64(inner), C.c_int(direction), C.c_int(acc))
    return y, logdet


_SCALE = {"exp": 0, "sigmoid": 1, "sigmoid_inv": 2, None: 3}


def affine_coupling(z, param, c1, flip, scale_map, direction, logdet=None, acc=0):
    dt = z.dtype
    z = _c(z, dt)
    B, Cc = z.shape[:2]
    HW = int(np.prod(z.shape[2:])) if z.ndim > 2 else 1
    y = np.empty_like(z)
    logdet = np.zeros(B, dt) if logdet is None else logdet
    param = _c(param, dt)
    f = getattr(lib(), "nfo_affine_coupling" + _sfx(dt))
    f(_p(z), _p(param), _p(y), _p(logdet), C.c_int64(B), C.c_int(Cc), C.c_int(c1), C.c_int(int(flip)), C.c_int64(HW),
      C.c_int(_SCALE[scale_map]), C.c_int(direction), C.c_int(acc))
    return y, logdet


def actnorm(z, s, t, direction, logdet=None, acc=0):
    dt = z.dtype
    z = _c(z, dt)
    B, Cc = z.shape[:2]
    HW = int(np.prod(z.shape[2:])) if z.ndim > 2 else 1
    y = np.empty_like(z)
    lds = np.zeros(1, dt)
    s, t = _c(s, dt).reshape(-1), _c(t, dt).reshape(-1)
    f = getattr(lib(), "nfo_actnorm" + _sfx(dt))
    f(_p(z), _p(s), _p(t), _p(y), _p(lds), _p(logdet), C.c_int64(B), C.c_int(Cc), C.c_int64(HW), C.c_int(direction),
      C.c_int(acc))
    return y, lds[0]


def actnorm_stats(z):
    dt = z.dtype
    z = _c(z, dt)
    B, Cc = z.shape[:2]
    HW = int(np.prod(z.shape[2:])) if z.ndim > 2 else 1
    mean, std = np.empty(Cc, dt), np.empty(Cc, dt)
    getattr(lib(), "nfo_actnorm_stats" + _sfx(dt))(_p(z), _p(mean), _p(std), C.c_int64(B), C.c_int(Cc), C.c_int64(HW))
    return mean, std


def actnorm_init(mean, std, direction):
    dt = mean.dtype
    s, t = np.empty_like(mean), np.empty_like(mean)
    getattr(lib(), "nfo_actnorm_init" + _sfx(dt))(_p(_c(mean, dt)), _p(_c(std, dt)), _p(s), _p(t), C.c_int(mean.size),
                                                 C.c_int(direction))
    return s, t


def inv1x1_assemble(P, L, U, sign_S, log_S, inverse):
    dt = L.dtype
    Cc = L.shape[0]
    W = np.empty((Cc, Cc), dt)
    ldu = np.zeros(1, dt)
    P, L, U, sign_S, log_S = (_c(a, dt) for a in (P, L, U, sign_S, log_S))
    getattr(lib(), "nfo_inv1x1_assemble" + _sfx(dt))(_p(P), _p(L), _p(U), _p(sign_S), _p(log_S), _p(W), _p(ldu),
                                                    C.c_int(Cc), C.c_int(int(inverse)))
    return W, ldu


def inv1x1_conv(z, W, logdet_unit):
    dt = z.dtype
    z = _c(z, dt)
    B, Cc = z.shape[:2]
    HW = int(np.prod(z.shape[2:]))
    y = np.empty_like(z)
    lds = np.zeros(1, dt)
    getattr(lib(), "nfo_inv1x1_conv" + _sfx(dt))(_p(z), _p(_c(W, dt)), _p(_c(logdet_unit, dt)), _p(y), _p(lds), None,
                                                C.c_int64(B), C.c_int(Cc), C.c_int64(HW), C.c_int(0))
    return y, lds[0]


def convnet2d(x, weights, biases, leaky=0.0):
    """nets/cnn.py:5-63: Conv2d(padding = k // 2) + LeakyReLU(leaky) for every layer but the last, which has no
    activation.  weights[i] (Cout_i, Cin_i, k_i, k_i), biases[i] (Cout_i)."""
    dt = x.dtype
    h = _c(x, dt)
    for i, (w, b) in enumerate(zip(weights, biases)):
        w, b = _c(w, dt), _c(b, dt)
        B, Cin, H, W = h.shape
        y = np.empty((B, w.shape[0], H, W), dt)
        real = C.c_float if dt == np.float32 else C.c_double
        getattr(lib(), "nfo_conv2d_same" + _sfx(dt))(_p(h), _p(w), _p(b), _p(y), C.c_int64(B), C.c_int(Cin), C.c_int(H),
                                                     C.c_int(W), C.c_int(w.shape[0]), C.c_int(w.shape[2]),
                                                     C.c_int(1 if i + 1 < len(weights) else 0), real(leaky))
        h = y
    return h


def glow_block(st, z, inverse, leaky=0.0, scale_map="sigmoid", init_actnorm=False):
    """GlowBlock.forward / .inverse (affine/glow.py:72-84) = [AffineCouplingBlock(split "channel", ConvNet2d), Invertible1x1Conv,
    ActNorm] composed from the oracle's layer functions; `st` = the block's state_dict as numpy arrays (ActNorm's s / t are
    (re)initialised from `z` when init_actnorm is set and the call is the inverse: normalization.py:30-39 on the block input).
    Returns (z', log_det (B,), st)."""
    dt = z.dtype
    z = _c(z, dt)
    B, Cc = z.shape[:2]
    HW = int(np.prod(z.shape[2:]))
    ld = np.zeros(B, dt)
    net = "flows.0.flows.1.param_map.net."
    ws = [st[net + "%d.weight" % i] for i in (0, 2, 4)]
    bs = [st[net + "%d.bias" % i] for i in (0, 2, 4)]
    c1 = (Cc + 1) // 2

    def coupling(v, direction):
        param = convnet2d(np.ascontiguousarray(v[:, :c1]), ws, bs, leaky)
        y, l = affine_coupling(v, param, c1, False, scale_map, direction)
        return y, l

    if inverse:
        if init_actnorm:
            mean, std = actnorm_stats(z)
            s_, t_ = actnorm_init(mean, std, 1)
            st = dict(st, **{"flows.2.s": s_.reshape(1, Cc, 1, 1), "flows.2.t": t_.reshape(1, Cc, 1, 1)})
        z, l = actnorm(z, st["flows.2.s"], st["flows.2.t"], 1)
        ld += l
        W, ldu = inv1x1_assemble(st["flows.1.P"], st["flows.1.L"], st["flows.1.U"], st["flows.1.sign_S"], st["flows.1.log_S"], False)
        z, l = inv1x1_conv(z, W, ldu)
        ld += l
        z, l = coupling(z, 1)
        ld += l
    else:
        z, l = coupling(z, 0)
        ld += l
        W, ldu = inv1x1_assemble(st["flows.1.P"], st["flows.1.L"], st["flows.1.U"], st["flows.1.sign_S"], st["flows.1.log_S"], True)
        z, l = inv1x1_conv(z, W, ldu)
        ld += l
        z, l = actnorm(z, st["flows.2.s"], st["flows.2.t"], 0)
        ld += l
    return z, ld, st


def diag_gaussian_log_prob(z, loc, log_scale, ls_shift=0.0, out=None, acc=0):
    dt = z.dtype
    z = _c(z, dt)
    B = z.shape[0]
    d = int(np.prod(z.shape[1:]))
    out = np.zeros(B, dt) if out is None else out
    loc, log_scale = _c(loc, dt).reshape(-1), _c(log_scale, dt).reshape(-1)
    getattr(lib(), "nfo_diag_gaussian_log_prob" + _sfx(dt))(_p(z), _p(loc), _p(log_scale), C.c_double(ls_shift),
                                                           _p(out), C.c_int64(B), C.c_int64(d), C.c_int(acc))
    return out


def diag_gaussian_log_prob_rows(z, loc_rows, log_scale_rows, row_idx=None, ls_shift=0.0):
    dt = z.dtype
    z = _c(z, dt)
    B = z.shape[0]
    d = int(np.prod(z.shape[1:]))
    out = np.zeros(B, dt)
    idx = None if row_idx is None else np.ascontiguousarray(row_idx, dtype=np.int64)
    getattr(lib(), "nfo_diag_gaussian_log_prob_rows" + _sfx(dt))(
        _p(z), _p(_c(loc_rows, dt).reshape(-1, d)), _p(_c(log_scale_rows, dt).reshape(-1, d)),
        _p(idx) if idx is not None else None, C.c_double(ls_shift), _p(out), C.c_int64(B), C.c_int64(d))
    return out


def logit(z, alpha, direction):
    dt = z.dtype
    z = _c(z, dt)
    B = z.shape[0]
    inner = int(np.prod(z.shape[1:]))
    y = np.empty_like(z)
    ld = np.empty(B, dt)
    getattr(lib(), "nfo_logit" + _sfx(dt))(_p(z), _p(y), _p(ld), C.c_int64(B), C.c_int64(inner), C.c_double(alpha),
                                           C.c_int(direction))
    return y, ld


def squeeze(z, direction):
    dt = z.dtype
    z = _c(z, dt)
    B, Cc, H, W = z.shape
    y = np.empty((B, Cc // 4, 2 * H, 2 * W) if direction == 0 else (B, 4 * Cc, H // 2, W // 2), dt)
    getattr(lib(), "nfo_squeeze" + _sfx(dt))(_p(z), _p(y), C.c_int64(B), C.c_int(Cc), C.c_int(H), C.c_int(W),
                                            C.c_int(direction))
    return y


def maf_affine(x, params, direction):
    dt = x.dtype
    x = _c(x, dt)
    B, D = x.shape
    y = np.empty_like(x)
    ld = np.empty(B, dt)
    getattr(lib(), "nfo_maf_affine" + _sfx(dt))(_p(x), _p(_c(params, dt)), _p(y), _p(ld), C.c_int64(B), C.c_int(D),
                                               C.c_int(direction))
    return y, ld


def made_forward(st, x, prefix="autoregressive_net."):
    """nets/made.py:292-304 with residual blocks (:196-214), ReLU, masked weights W * mask (:80-81); numpy."""
    def lin(name, v):
        return v @ (st[prefix + name + ".weight"] * st[prefix + name + ".mask"]).T + st[prefix + name + ".bias"]
    h = lin("initial_layer", x)
    b = 0
    while (prefix + "blocks.%d.linear_layers.0.weight" % b) in st:
        t = lin("blocks.%d.linear_layers.0" % b, np.maximum(h, 0))
        t = lin("blocks.%d.linear_layers.1" % b, np.maximum(t, 0))
        h = h + t
        b += 1
    return lin("final_layer", h)


def maf_layer(st, x, inverse, prefix="autoregressive_net."):
    """Autoregressive.forward / .inverse (affine/autoregressive.py:24-38)."""
    if not inverse:
        return maf_affine(x, made_forward(st, x, prefix).astype(x.dtype), 0)
    out = np.zeros_like(x)
    ld = None
    for _ in range(x.shape[1]):
        out, ld = maf_affine(x, made_forward(st, out, prefix).astype(x.dtype), 1)
    return out, ld


def arnsf_transform(st, x, inverse, K, tails="linear", tail_bound=3.0, prefix="mprqat.autoregressive_net."):
    """MaskedPiecewiseRationalQuadraticAutoregressive.forward / .inverse (neural_spline/autoregressive.py:94-140 over
    affine/autoregressive.py:24-38): MADE -> (B, D, 3K-1|3K|3K+1) -> element-wise spline, row-summed log-det; no
    sqrt(hidden) scaling (the reference's MADE has no `hidden_features` attribute, :107-109)."""
    def elementwise(inp, params, inv):
        B, D = inp.shape
        prm = params.reshape(B, D, -1).astype(inp.dtype)
        y, lad = rqs_spline(inp, prm[..., :K], prm[..., K:2 * K], prm[..., 2 * K:], inverse=inv, tails=tails,
                            tail_bound=tail_bound)
        return y, lad.sum(1)
    if not inverse:
        return elementwise(x, made_forward(st, x, prefix), False)
    out = np.zeros_like(x)
    ld = None
    for _ in range(x.shape[1]):
        out, ld = elementwise(x, made_forward(st, out, prefix), True)
    return out, ld


# ---------------------------------------------------------------------------------------------------------
class OracleNSF:
    """NormalizingFlow([CoupledRationalQuadraticSpline, LULinearPermute] * L, DiagGaussian) on the CPU oracle.

    `state` is the normflows state_dict as {name: numpy array}; layer i lives under "flows.{i}.".
    log_prob follows core.py:182-197, sample follows core.py:167-180 (given the base noise eps).
    """

    def __init__(self, state, num_layers, K=8, tail_bound=3.0, hidden=None):
        self.st = {k: np.asarray(v) for k, v in state.items()}
        self.n = num_layers
        self.K = K
        self.tail_bound = tail_bound

    def _coupling_params(self, i):
        p = "flows.%d.prqct." % i
        s = self.st
        nb = 0
        while (p + "transform_net.blocks.%d.linear_layers.0.weight" % nb) in s:
            nb += 1
        wb, bb = [], []
        for b in range(nb):
            for l in range(2):
                wb.append(s[p + "transform_net.blocks.%d.linear_layers.%d.weight" % (b, l)])
                bb.append(s[p + "transform_net.blocks.%d.linear_layers.%d.bias" % (b, l)])
        return dict(ii=s[p + "identity_features"], ti=s[p + "transform_features"],
                    w0=s[p + "transform_net.initial_layer.weight"], b0=s[p + "transform_net.initial_layer.bias"],
                    wb=wb, bb=bb, wf=s[p + "transform_net.final_layer.weight"],
                    bf=s[p + "transform_net.final_layer.bias"], uw=s[p + "unconditional_transform.unnormalized_widths"],
                    uh=s[p + "unconditional_transform.unnormalized_heights"],
                    ud=s[p + "unconditional_transform.unnormalized_derivatives"])

    def _lu_params(self, i):
        p = "flows.%d." % i
        s = self.st
        return dict(perm=s[p + "permutation._permutation"], lo=s[p + "linear.lower_entries"],
                    up=s[p + "linear.upper_entries"], ud=s[p + "linear.unconstrained_upper_diag"],
                    bias=s[p + "linear.bias"])

    def _is_coupling(self, i):
        return ("flows.%d.prqct.identity_features" % i) in self.st

    def coupling(self, i, z, direction, logq, acc):
        c = self._coupling_params(i)
        hidden = c["w0"].shape[0]
        kw = dict(K=self.K, tail_bound=self.tail_bound, wh_div=float(np.sqrt(hidden)))
        if direction == 0:  # density: wrapper.inverse -> prqct.forward
            cond = resnet_mlp(z, c["ii"], c["w0"], c["b0"], c["wb"], c["bb"], c["wf"], c["bf"])
            y, _ = rqs_coupling(z, cond, c["uw"], c["uh"], c["ud"], c["ii"], c["ti"], mode=0, logdet=logq, acc=acc, **kw)
        else:  # sample: wrapper.forward -> prqct.inverse
            y, _ = rqs_coupling(z, None, c["uw"], c["uh"], c["ud"], c["ii"], c["ti"], mode=1, logdet=logq, acc=acc, **kw)
            cond = resnet_mlp(y, c["ii"], c["w0"], c["b0"], c["wb"], c["bb"], c["wf"], c["bf"])
            y, _ = rqs_coupling(z, cond, c["uw"], c["uh"], c["ud"], c["ii"], c["ti"], mode=2, y=y, logdet=logq, acc=acc,
                                **kw)
        return y

    def lu(self, i, z, direction, logq, acc):
        l = self._lu_params(i)
        y, _ = lu_linear_permute(z, l["perm"], l["lo"], l["up"], l["ud"], l["bias"], direction, logdet=logq, acc=acc)
        return y

    def log_prob_whole(self, x):
        """log_prob through the single C entry point nfo_nsf_log_prob (row chunks in parallel under OpenMP, every
        chunk runs the whole layer chain) -- the timed CPU baseline of bench.py.  Requires alternating
        [CoupledRQS, LULinearPermute] layers."""
        x = np.ascontiguousarray(x)
        dt = x.dtype
        B, D = x.shape
        L = self.n // 2
        keep, ptrs = [], []

        def add(a, dtype=None):
            a = np.ascontiguousarray(a, dtype=dtype or dt)
            keep.append(a)
            ptrs.append(a.ctypes.data)

        nblk = hidden = None
        for l in range(L):
            c, u = self._coupling_params(2 * l), self._lu_params(2 * l + 1)
            nblk, hidden = len(c["wb"]) // 2, c["w0"].shape[0]
            add(c["ii"], np.int64), add(c["ti"], np.int64), add(c["w0"]), add(c["b0"])
            for w, b in zip(c["wb"], c["bb"]):
                add(w), add(b)
            add(c["wf"]), add(c["bf"]), add(c["uw"]), add(c["uh"]), add(c["ud"])
            add(u["perm"], np.int64), add(u["lo"]), add(u["up"]), add(u["ud"]), add(u["bias"])
        arr = (C.c_void_p * len(ptrs))(*ptrs)
        out = np.empty(B, dt)
        loc, ls = _c(self.st["q0.loc"], dt).reshape(-1), _c(self.st["q0.log_scale"], dt).reshape(-1)
        f = getattr(lib(), "nfo_nsf_log_prob" + _sfx(dt))
        f(_p(x), _p(out), C.c_int64(B), C.c_int(D), C.c_int(L), arr, C.c_int(nblk), C.c_int(hidden), C.c_int(self.K),
          C.c_double(self.tail_bound), _p(loc), _p(ls), C.c_double(1e-3))
        return out

    def log_prob(self, x):
        z = np.ascontiguousarray(x)
        logq = np.zeros(z.shape[0], z.dtype)
        for i in range(self.n - 1, -1, -1):
            z = (self.coupling if self._is_coupling(i) else self.lu)(i, z, 0, logq, +1)
        loc, ls = self.st["q0.loc"], self.st["q0.log_scale"]
        diag_gaussian_log_prob(z, loc, ls, out=logq, acc=+1)
        return logq

    def sample_from(self, eps):
        """sample() of core.py:167-180 with the base noise given: z0 = loc + exp(log_scale) * eps."""
        loc, ls = self.st["q0.loc"].reshape(1, -1), self.st["q0.log_scale"].reshape(1, -1)
        dt = eps.dtype
        z = np.ascontiguousarray(loc + np.exp(ls) * eps, dtype=dt)
        d = z.shape[1]
        logq = (-0.5 * d * np.log(2 * np.pi) - np.sum(ls + 0.5 * eps.astype(dt) ** 2, axis=1)).astype(dt)
        for i in range(self.n):
            z = (self.coupling if self._is_coupling(i) else self.lu)(i, z, 1, logq, -1)
        return z, logq
