#!/usr/bin/env python3
"""Generates tests/golden/*.npz by running the REAL reference (normflows 1.7.3 at /root/reference, PyTorch CPU).

Run in the build container only (the GPU box has no /root/reference):
    python tests/golden/make_golden.py
The fixtures pin the CPU oracle (tests/test_oracle_golden.py) and the HIP kernels (tests/test_gpu_parity.py).
The reference's own tests contain no value-level golden vectors for this path; these
files are outputs of the reference itself on seeded inputs, which is the strongest pin available.
Every case stores its inputs, the parameters / conditioner outputs involved and the reference outputs, in
float32 and (where cheap) float64.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
import normflows as nf  # noqa: E402
from normflows.utils import splines  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(4)


def npz(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print("wrote %-34s %6.1f KB" % (name + ".npz", os.path.getsize(os.path.join(OUT, name + ".npz")) / 1024))


def sd(module, prefix=""):
    return {prefix + k.replace(".", "__"): v for k, v in module.state_dict().items()}


# ---------------------------------------------------------------------------------------------------------------
def gen_splines():
    """utils/splines.py: bounded (tails None), linear and circular tails, K in {2, 8, 10, 16}, edge inputs."""
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        g = torch.Generator().manual_seed(11)
        for K in (2, 8, 10, 16):
            shape = [6, 5]
            w = torch.randn(*shape, K, generator=g, dtype=dt)
            h = torch.randn(*shape, K, generator=g, dtype=dt)
            # bounded spline on [0, 1]: inputs strictly inside
            d_none = torch.randn(*shape, K + 1, generator=g, dtype=dt)
            x01 = torch.rand(*shape, generator=g, dtype=dt) * 0.998 + 0.001
            y, lad = splines.rational_quadratic_spline(x01, w.clone(), h.clone(), d_none.clone(), inverse=False)
            xi, ladi = splines.rational_quadratic_spline(y, w.clone(), h.clone(), d_none.clone(), inverse=True)
            # linear tails, bound 3: heavy-tailed inputs + explicit edge list
            d_lin = torch.randn(*shape, K - 1, generator=g, dtype=dt)
            xl = 4 * torch.randn(*shape, generator=g, dtype=dt)
            edge = torch.tensor([3.0, -3.0, 3.0 * (1 + 2 ** -23), -3.0 * (1 + 2 ** -23), 3.0 * (1 - 2 ** -23), 0.0,
                                 float("nan"), float("inf"), float("-inf"), 2.9999, -2.9999, 1e-30], dtype=dt)
            xl.view(-1)[: edge.numel()] = edge
            yl, ladl = splines.unconstrained_rational_quadratic_spline(xl, w.clone(), h.clone(), d_lin.clone(),
                                                                       inverse=False, tails="linear", tail_bound=3.0)
            xli, ladli = splines.unconstrained_rational_quadratic_spline(xl, w.clone(), h.clone(), d_lin.clone(),
                                                                         inverse=True, tails="linear", tail_bound=3.0)
            # circular tails
            d_cir = torch.randn(*shape, K, generator=g, dtype=dt)
            yc, ladc = splines.unconstrained_rational_quadratic_spline(xl, w.clone(), h.clone(), d_cir.clone(),
                                                                       inverse=False, tails="circular", tail_bound=2.5)
            yci, ladci = splines.unconstrained_rational_quadratic_spline(xl, w.clone(), h.clone(), d_cir.clone(),
                                                                         inverse=True, tails="circular", tail_bound=2.5)
            npz("spline_K%d_%s" % (K, tag), w=w, h=h, d_none=d_none, x01=x01, y01=y, lad01=lad, x01_inv=xi, lad01_inv=ladi,
                d_lin=d_lin, xl=xl, yl=yl, ladl=ladl, yl_inv=xli, ladl_inv=ladli, d_cir=d_cir, yc=yc, ladc=ladc,
                yc_inv=yci, ladc_inv=ladci)
        # inputs exactly on interior knots (K = 8, linear tails): bin k, maps to knot_y_k
        K = 8
        w = torch.randn(4, K, generator=g, dtype=dt)
        h = torch.randn(4, K, generator=g, dtype=dt)
        dl = torch.randn(4, K - 1, generator=g, dtype=dt)
        wn = 1e-3 + (1 - 1e-3 * K) * torch.softmax(w, -1)
        cw = torch.nn.functional.pad(torch.cumsum(wn, -1), (1, 0)) * 6.0 - 3.0
        cw[..., 0] = -3.0
        cw[..., -1] = 3.0
        xk = cw[:, 1:K].contiguous()  # (4, 7) interior knots
        wk = w[:, None, :].expand(4, K - 1, K).contiguous()
        hk = h[:, None, :].expand(4, K - 1, K).contiguous()
        dk = dl[:, None, :].expand(4, K - 1, K - 1).contiguous()
        yk, ladk = splines.unconstrained_rational_quadratic_spline(xk, wk.clone(), hk.clone(), dk.clone(), inverse=False,
                                                                   tails="linear", tail_bound=3.0)
        npz("spline_knots_%s" % tag, w=wk, h=hk, d=dk, x=xk, y=yk, lad=ladk)


def perturb(module, sigma, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in module.parameters():
            p.add_(sigma * torch.randn(p.shape, generator=g, dtype=p.dtype))


def gen_nsf_layers():
    """CoupledRationalQuadraticSpline (wrapper.py) and LULinearPermute (mixing.py), per layer, both directions."""
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        for d, hidden, K in ((2, 16, 8), (5, 16, 4), (64, 32, 8), (7, 24, 10)):
            torch.manual_seed(100 + d)
            layer = nf.flows.CoupledRationalQuadraticSpline(d, 2, hidden, num_bins=K, init_identity=False,
                                                            reverse_mask=(d == 5))
            perturb(layer, 0.3, 5)
            with torch.no_grad():
                layer.prqct.unconditional_transform.unnormalized_widths.normal_()
                layer.prqct.unconditional_transform.unnormalized_heights.normal_()
                layer.prqct.unconditional_transform.unnormalized_derivatives.normal_()
            layer = layer.to(dt)
            g = torch.Generator().manual_seed(d)
            x = 2.0 * torch.randn(24, d, generator=g, dtype=dt)
            x[0, 0] = 3.0
            x[1, -1] = -3.0
            x[2, 0] = 5.5
            with torch.no_grad():
                # conditioner outputs seen by each direction (for kernel-level tests)
                ident = x[:, layer.prqct.identity_features]
                cond_density = layer.prqct.transform_net(ident)
                z_inv, ld_inv = layer.inverse(x)       # density direction
                z_fwd, ld_fwd = layer.forward(x)       # sampling direction
                ident_s = z_fwd[:, layer.prqct.identity_features]
                cond_sample = layer.prqct.transform_net(ident_s)
            npz("crqs_d%d_%s" % (d, tag), x=x, cond_density=cond_density, cond_sample=cond_sample, z_inv=z_inv,
                ld_inv=ld_inv, z_fwd=z_fwd, ld_fwd=ld_fwd, hidden=hidden, K=K, **sd(layer, "sd__"))
        for d in (3, 4, 64):
            torch.manual_seed(200 + d)
            layer = nf.flows.LULinearPermute(d, identity_init=False)
            perturb(layer, 0.2, 6)
            layer = layer.to(dt)
            g = torch.Generator().manual_seed(50 + d)
            x = torch.randn(19, d, generator=g, dtype=dt)
            with torch.no_grad():
                z_inv, ld_inv = layer.inverse(x)
                z_fwd, ld_fwd = layer.forward(x)
            npz("lulinear_d%d_%s" % (d, tag), x=x, z_inv=z_inv, ld_inv=ld_inv, z_fwd=z_fwd, ld_fwd=ld_fwd,
                **sd(layer, "sd__"))


def gen_affine():
    """MaskedAffineFlow, AffineCouplingBlock, ActNorm, Invertible1x1Conv, GlowBlock, DiagGaussian, Squeeze."""
    dt = torch.float32
    for d in (2, 7):
        torch.manual_seed(300 + d)
        b = torch.tensor([1.0 if i % 2 == 0 else 0.0 for i in range(d)])
        s = nf.nets.MLP([d, 2 * d, d])
        t = nf.nets.MLP([d, 2 * d, d])
        layer = nf.flows.MaskedAffineFlow(b, t, s)
        g = torch.Generator().manual_seed(d)
        z = torch.randn(9, d, generator=g)
        with torch.no_grad():
            sv, tv = s(b * z), t(b * z)
            zf, ldf = layer.forward(z)
            zi, ldi = layer.inverse(z)
        npz("masked_affine_d%d" % d, z=z, b=b, s=sv, t=tv, z_fwd=zf, ld_fwd=ldf, z_inv=zi, ld_inv=ldi, **sd(layer, "sd__"))
    # non-finite s / t handling (coupling.py:212-215)
    z = torch.randn(4, 2)
    b = torch.tensor([0.0, 1.0])

    class Const(torch.nn.Module):
        def __init__(self, v):
            super().__init__()
            self.v = v

        def forward(self, x):
            return self.v

    sv = torch.tensor([[0.1, 0.2], [float("inf"), 0.0], [0.3, float("nan")], [-0.2, 0.5]])
    tv = torch.tensor([[0.0, 1.0], [1.0, 1.0], [2.0, 0.0], [float("-inf"), 0.1]])
    layer = nf.flows.MaskedAffineFlow(b, Const(tv), Const(sv))
    zf, ldf = layer.forward(z)
    zi, ldi = layer.inverse(z)
    npz("masked_affine_nonfinite", z=z, b=b, s=sv, t=tv, z_fwd=zf, ld_fwd=ldf, z_inv=zi, ld_inv=ldi)

    for C, split, smap, scale in ((4, "channel", "exp", True), (5, "channel", "sigmoid", True),
                                  (5, "channel_inv", "sigmoid_inv", True), (3, "channel_inv", "exp", False),
                                  (4, "checkerboard", "sigmoid", True), (6, "checkerboard_inv", "exp", True)):
        torch.manual_seed(400 + C)
        if "checkerboard" in split:
            ch = (C, 8, 8, (2 if scale else 1) * C)
        elif split == "channel":
            ch = ((C + 1) // 2, 8, 8, (2 if scale else 1) * (C // 2))
        else:
            ch = (C // 2, 8, 8, (2 if scale else 1) * ((C + 1) // 2))
        net = nf.nets.ConvNet2d(ch, (3, 1, 3), 0.0, init_zeros=False)
        layer = nf.flows.AffineCouplingBlock(net, scale, smap, split)
        g = torch.Generator().manual_seed(C)
        z = torch.randn(3, C, 4, 4, generator=g)
        with torch.no_grad():
            [z1, z2], _ = layer.flows[0](z)
            param = net(z1)
            zf, ldf = layer.forward(z)
            zi, ldi = layer.inverse(z)
        npz("affine_block_C%d_%s_%s" % (C, split, smap if scale else "noscale"), z=z, param=param, z_fwd=zf, ld_fwd=ldf,
            z_inv=zi, ld_inv=ldi, **sd(layer, "sd__"))
    # 2-D AffineCouplingBlock with an MLP (real_nvp_colab / README path)
    torch.manual_seed(410)
    layer = nf.flows.AffineCouplingBlock(nf.nets.MLP([1, 16, 16, 2], init_zeros=False))
    z = torch.randn(11, 2)
    with torch.no_grad():
        param = layer.flows[1].param_map(z[:, :1])
        zf, ldf = layer.forward(z)
        zi, ldi = layer.inverse(z)
    npz("affine_block_2d", z=z, param=param, z_fwd=zf, ld_fwd=ldf, z_inv=zi, ld_inv=ldi, **sd(layer, "sd__"))

    # ActNorm: data-dependent init, forward-first and inverse-first (normalization.py:19-39)
    for shape, zshape in (((6, 1, 1), (5, 6, 4, 4)), ((3,), (17, 3))):
        g = torch.Generator().manual_seed(len(zshape))
        z = torch.randn(*zshape, generator=g) * 1.7 + 0.4
        z2 = torch.randn(*zshape, generator=g)
        a = nf.flows.ActNorm(shape)
        with torch.no_grad():
            zf, ldf = a.forward(z)
            s_f, t_f = a.s.clone(), a.t.clone()
            zf2, ldf2 = a.forward(z2)     # second call: no re-init
            zi2, ldi2 = a.inverse(z2)
        bnew = nf.flows.ActNorm(shape)
        with torch.no_grad():
            zi, ldi = bnew.inverse(z)
            s_i, t_i = bnew.s.clone(), bnew.t.clone()
        npz("actnorm_%dd" % len(zshape), z=z, z2=z2, z_fwd=zf, ld_fwd=ldf, s_fwd=s_f, t_fwd=t_f, z2_fwd=zf2, ld2_fwd=ldf2,
            z2_inv=zi2, ld2_inv=ldi2, z_inv=zi, ld_inv=ldi, s_inv=s_i, t_inv=t_i)

    # Invertible1x1Conv (LU and plain), mixing.py:57-133
    for C, use_lu in ((3, True), (4, True), (12, True), (48, True), (4, False)):
        torch.manual_seed(500 + C)
        layer = nf.flows.Invertible1x1Conv(C, use_lu)
        perturb(layer, 0.1, 9)
        g = torch.Generator().manual_seed(C)
        z = torch.randn(2, C, 3, 5, generator=g)
        with torch.no_grad():
            zf, ldf = layer.forward(z)
            zi, ldi = layer.inverse(z)
            extra = {}
            if use_lu:
                extra = dict(W_inv_dir=layer._assemble_W(), W_fwd_dir=layer._assemble_W(inverse=True))
        npz("inv1x1_C%d_%s" % (C, "lu" if use_lu else "plain"), z=z, z_fwd=zf, ld_fwd=ldf, z_inv=zi, ld_inv=ldi,
            **extra, **sd(layer, "sd__"))

    # GlowBlock end to end (glow.py:11-84); first call initialises ActNorm
    for C, split, use_lu in ((4, "channel", True), (5, "channel_inv", True), (3, "checkerboard", True)):
        torch.manual_seed(600 + C)
        layer = nf.flows.GlowBlock(C, 8, split_mode=split, use_lu=use_lu, init_zeros=False)
        g = torch.Generator().manual_seed(C)
        z = torch.randn(4, C, 4, 4, generator=g)
        sd0 = sd(layer, "sd0__")
        with torch.no_grad():
            zi, ldi = layer.inverse(z)            # density direction, triggers inverse-first ActNorm init
            zf, ldf = layer.forward(zi)           # round trip with initialised ActNorm
        npz("glowblock_C%d_%s" % (C, split), z=z, z_inv=zi, ld_inv=ldi, z_fwd=zf, ld_fwd=ldf, **sd0, **sd(layer, "sd__"))

    # DiagGaussian.log_prob (base.py:94-103)
    torch.manual_seed(700)
    q = nf.distributions.DiagGaussian((3, 2, 2))
    with torch.no_grad():
        q.loc.normal_()
        q.log_scale.normal_(std=0.3)
        z = torch.randn(6, 3, 2, 2)
        lp = q.log_prob(z)
        q.temperature = 0.7
        lpt = q.log_prob(z)
    npz("diag_gaussian", z=z, loc=q.loc, log_scale=q.log_scale, log_prob=lp, log_prob_t07=lpt)

    # Squeeze (reshape.py:103-128)
    z = torch.randn(2, 8, 4, 6)
    s = nf.flows.Squeeze()
    npz("squeeze", z=z, fwd=s.forward(z)[0], inv=s.inverse(z)[0])


def gen_models():
    """Whole-model fixtures: NormalizingFlow.log_prob / sample (core.py:167-197), MultiscaleFlow.log_prob."""
    sys.path.insert(0, os.path.join(OUT, "..", ".."))
    from bench import build_c2_model, c2_inputs

    # C2-mini: 4 x [CoupledRQS(16, 2, 32) + LULinearPermute(16)], sigma 0.05
    m = build_c2_model(num_layers=4, dim=16, hidden=32, seed=0, sigma=0.05, lib=nf)
    x = c2_inputs(64, 16)
    torch.manual_seed(3)
    eps = torch.randn(64, 16)
    with torch.no_grad():
        lp = m.log_prob(x)
        z, logq = eps.clone(), None
        z0 = m.q0.loc + torch.exp(m.q0.log_scale) * eps
        logq = -0.5 * 16 * np.log(2 * np.pi) - torch.sum(m.q0.log_scale + 0.5 * eps ** 2, 1)
        z = z0
        for f in m.flows:
            z, ld = f(z)
            logq = logq - ld
    npz("model_c2mini", x=x, log_prob=lp, eps=eps, sample=z, sample_logq=logq, **sd(m, "sd__"))

    # C2 at full width, first 128 rows of the benchmark batch: pins the bench model itself.  Only outputs are
    # stored (the 21.8 MB of weights are reproduced by seeded construction, tests/test_state_dict_compat.py).
    m = build_c2_model(lib=nf)
    x = c2_inputs(65536, 64)[:128]
    with torch.no_grad():
        lp = m.log_prob(x)
    npz("model_c2_head", x=x, log_prob=lp)

    # C1: 4 x [MaskedAffineFlow(MLP[2,4,2]) + ActNorm(2)] on TwoMoons, B = 1024 (BASELINE config 1)
    torch.manual_seed(0)
    b = torch.tensor([1.0, 0.0])
    fl = []
    for i in range(4):
        s = nf.nets.MLP([2, 4, 2], init_zeros=True)
        t = nf.nets.MLP([2, 4, 2], init_zeros=True)
        fl += [nf.flows.MaskedAffineFlow(b if i % 2 == 0 else 1 - b, t, s), nf.flows.ActNorm(2)]
    m = nf.NormalizingFlow(nf.distributions.DiagGaussian(2), fl)
    g = torch.Generator().manual_seed(1234)
    with torch.no_grad():
        for p in m.parameters():
            p.add_(0.1 * torch.randn(p.shape, generator=g))
    torch.manual_seed(0)
    x = nf.distributions.TwoMoons().sample(1024)
    sd0 = sd(m, "sd0__")
    with torch.no_grad():
        lp = m.log_prob(x)       # triggers inverse-first ActNorm init
        lp2 = m.log_prob(x)
        torch.manual_seed(5)
        eps = torch.randn(1024, 2)
        z = m.q0.loc + torch.exp(m.q0.log_scale) * eps
        logq = -0.5 * 2 * np.log(2 * np.pi) - torch.sum(m.q0.log_scale + 0.5 * eps ** 2, 1)
        for f in m.flows:
            z, ld = f(z)
            logq = logq - ld
    npz("model_c1_realnvp", x=x, log_prob=lp, log_prob_second=lp2, eps=eps, sample=z, sample_logq=logq, **sd0,
        **sd(m, "sd__"))

    # C4-mini: Glow multiscale L=2, K=2, hidden 16, 8x8x3 images, B=6, DiagGaussian bases (class_cond False)
    torch.manual_seed(0)
    L_, K_, hidden = 2, 2, 16
    input_shape = (3, 8, 8)
    channels = 3
    q0, merges, flows = [], [], []
    for i in range(L_):
        flows_ = []
        for j in range(K_):
            flows_ += [nf.flows.GlowBlock(channels * 2 ** (L_ + 1 - i), hidden, split_mode="channel", scale=True,
                                          init_zeros=False)]
        flows_ += [nf.flows.Squeeze()]
        flows += [flows_]
        if i > 0:
            merges += [nf.flows.Merge()]
            latent_shape = (input_shape[0] * 2 ** (L_ - i), input_shape[1] // 2 ** (L_ - i),
                            input_shape[2] // 2 ** (L_ - i))
        else:
            latent_shape = (input_shape[0] * 2 ** (L_ + 1), input_shape[1] // 2 ** L_, input_shape[2] // 2 ** L_)
        q0 += [nf.distributions.DiagGaussian(latent_shape)]
    m = nf.MultiscaleFlow(q0, flows, merges, class_cond=False)
    g = torch.Generator().manual_seed(9)
    x = torch.rand(6, 3, 8, 8, generator=g)
    sd0 = sd(m, "sd0__")
    with torch.no_grad():
        lp = m.log_prob(x)
        lp2 = m.log_prob(x)
    npz("model_c4mini_glow", x=x, log_prob=lp, log_prob_second=lp2, **sd0, **sd(m, "sd__"))


def gen_grads():
    """Gradients from the reference's autograd (training path): per layer with random
    cotangents (both directions) and the forward-KL training loss of core.py:87-102 on the C2-mini model."""
    def layer_grads(layer, x, tag):
        out = {}
        g = torch.Generator().manual_seed(77)
        cz = torch.randn(x.shape, generator=g, dtype=x.dtype)
        cl = torch.randn(x.shape[0], generator=g, dtype=x.dtype)
        for name, fn in (("inv", layer.inverse), ("fwd", layer.forward)):
            xx = x.clone().requires_grad_(True)
            layer.zero_grad()
            z, ld = fn(xx)
            loss = (z * cz).sum() + (ld * cl).sum()
            loss.backward()
            out["gx_" + name] = xx.grad.clone()
            for k, p_ in layer.named_parameters():
                out["g_%s__%s" % (name, k.replace(".", "__"))] = p_.grad.clone()
        npz(tag, x=x, cz=cz, cl=cl, **out, **sd(layer, "sd__"))

    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        for d, hidden, K in ((6, 16, 8), (64, 32, 8)):
            torch.manual_seed(900 + d)
            layer = nf.flows.CoupledRationalQuadraticSpline(d, 2, hidden, num_bins=K, init_identity=False)
            perturb(layer, 0.2, 5)
            with torch.no_grad():
                u = layer.prqct.unconditional_transform
                u.unnormalized_widths.normal_()
                u.unnormalized_heights.normal_()
                u.unnormalized_derivatives.normal_()
            layer = layer.to(dt)
            x = 1.5 * torch.randn(12, d, generator=torch.Generator().manual_seed(d), dtype=dt)
            x[0, 0] = 4.0   # outside the tails
            layer_grads(layer, x, "grad_crqs_d%d_%s" % (d, tag))
        for d in (5, 64):
            torch.manual_seed(950 + d)
            layer = nf.flows.LULinearPermute(d, identity_init=False)
            perturb(layer, 0.1, 6)
            layer = layer.to(dt)
            x = torch.randn(9, d, generator=torch.Generator().manual_seed(d), dtype=dt)
            layer_grads(layer, x, "grad_lulinear_d%d_%s" % (d, tag))
    # training loss on the C2-mini model (trainable base distribution)
    sys.path.insert(0, os.path.join(OUT, "..", ".."))
    from bench import c2_inputs, perturb_
    torch.manual_seed(0)
    flows = []
    for _ in range(4):
        flows += [nf.flows.CoupledRationalQuadraticSpline(16, 2, 32, num_bins=8), nf.flows.LULinearPermute(16)]
    m = nf.NormalizingFlow(nf.distributions.DiagGaussian(16, trainable=True), flows)
    perturb_(m, 0.05)
    x = c2_inputs(48, 16)
    loss = m.forward_kld(x)
    loss.backward()
    grads = {"g__" + k.replace(".", "__"): p_.grad for k, p_ in m.named_parameters()}
    npz("grad_model_c2mini", x=x, loss=loss.detach(), **grads, **sd(m, "sd__"))


def gen_train_c2_w64h128():
    """The training step at the BENCHMARK layer shape (core.py:87-102 forward_kld + loss.backward() over
    wrapper.py:14-85, nets/resnet.py:53-104, mixing.py:535-563): 2 x [CoupledRationalQuadraticSpline(64, 2, 128) +
    LULinearPermute(64)], sigma = 0.05, B = 1024 rows -- the smallest batch from which our layers take the one-launch training
    forward (nf_rqs_fused_train_full_fwd) and the one-pass backward kernels (nf_final_bwd, nf_resblock_bwd, nf_lu_bwd), so those
    kernels are pinned to the reference's autograd and not only to our own layer-wise path.  float32 leg + float64 leg (the same
    weights cast up); the model is rebuilt from its seed on the test side (bench.build_c2_model, bit-identical construction is
    tests/test_host.py's business), the fixture carries per-parameter checksums instead of 1.4 MB of weights."""
    sys.path.insert(0, os.path.join(OUT, "..", ".."))
    from bench import build_c2_model, c2_inputs
    m = build_c2_model(num_layers=2, dim=64, hidden=128, seed=0, sigma=0.05, lib=nf)
    x = c2_inputs(1024, 64, seed=4321)
    out = {"x": x}
    for k, p_ in m.named_parameters():
        out["chk__" + k.replace(".", "__")] = np.array([p_.detach().double().sum().item(), p_.detach().double().abs().sum().item()])
    for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
        mm = build_c2_model(num_layers=2, dim=64, hidden=128, seed=0, sigma=0.05, lib=nf)   # fp32 weights ...
        mm = mm.to(dt)                                                                    # ... cast up for the fp64 leg
        xx = x.detach().clone().to(dt).requires_grad_(True)
        loss = mm.forward_kld(xx)
        loss.backward()
        out["loss_" + tag] = loss.detach().double()
        out["gx_" + tag] = xx.grad.float()
        for k, p_ in mm.named_parameters():
            out["g_%s__%s" % (tag, k.replace(".", "__"))] = p_.grad.float()       # fp64 gradients rounded to fp32: 6e-8 relative
        with torch.no_grad():
            out["log_prob_" + tag] = mm.log_prob(x.detach().clone().to(dt)).double()
    npz("grad_model_c2_w64h128", **out)


def gen_train_nsf_wide():
    """The training step (core.py:87-102 forward_kld + loss.backward()) of NSF models with conditioners beyond 128 hidden units
    (wrapper.py:20-35, nets/resnet.py:53-104): 2 x [CoupledRationalQuadraticSpline(D, 2, hidden) + LULinearPermute(D)], sigma 0.05,
    B = 200; loss, input gradient, of every parameter gradient a strided sample (every 37th element) + sum / absolute sum; float32
    and float64 legs; weights by seeded construction (bench.build_c2_model)."""
    sys.path.insert(0, os.path.join(OUT, "..", ".."))
    from bench import build_c2_model, c2_inputs
    STRIDE = 37
    for D, H in ((64, 256), (128, 160)):
        out = {"stride": np.array(STRIDE)}
        for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            mm = build_c2_model(num_layers=2, dim=D, hidden=H, seed=41 + D, sigma=0.05, lib=nf).to(dt)
            x = c2_inputs(200, D, seed=5 + H)
            xx = x.clone().to(dt).requires_grad_(True)
            loss = mm.forward_kld(xx)
            loss.backward()
            out["loss_" + tag] = loss.detach().double()
            out["gx_" + tag] = xx.grad
            for k, p_ in mm.named_parameters():
                key = k.replace(".", "__")
                gflat = (torch.zeros_like(p_) if p_.grad is None else p_.grad).reshape(-1)
                out["g_%s__%s" % (tag, key)] = gflat[::STRIDE].clone()
                out["chk_%s__%s" % (tag, key)] = torch.tensor([float(gflat.double().sum()), float(gflat.double().abs().sum())],
                                                              dtype=torch.float64)
            if dt == torch.float32:
                out["x"] = x
        npz("grad_model_nsf_wide_d%d_h%d" % (D, H), **out)


def gen_nsf_wide():
    """NSF models beyond the benchmark kernel's shapes (hidden 256, D = 128; wrapper.py:20-35 over nets/resnet.py:53-104): 3 x
    [CoupledRationalQuadraticSpline + LULinearPermute], sigma 0.05, 96 rows, log_prob and the sampling pass, fp32 and fp64 legs.
    Outputs only: the weights are reproduced by seeded construction (bench.build_c2_model with the same arguments)."""
    sys.path.insert(0, os.path.join(OUT, "..", ".."))
    from bench import build_c2_model, c2_inputs
    for D, H in ((64, 256), (128, 128), (128, 256), (96, 192)):
        out = {}
        for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            m = build_c2_model(num_layers=3, dim=D, hidden=H, seed=40 + D, sigma=0.05, lib=nf).to(dt)
            x = c2_inputs(96, D, seed=D + H).to(dt)
            eps = torch.randn(96, D, generator=torch.Generator().manual_seed(H)).to(dt)
            with torch.no_grad():
                lp = m.log_prob(x)
                z = m.q0.loc + torch.exp(m.q0.log_scale) * eps
                logq = -0.5 * D * np.log(2 * np.pi) - torch.sum(m.q0.log_scale + 0.5 * eps ** 2, 1)
                for f in m.flows:
                    z, ld = f(z)
                    logq = logq - ld
            out.update({"log_prob_" + tag: lp, "sample_" + tag: z, "sample_logq_" + tag: logq})
            if dt == torch.float32:
                out.update(x=x, eps=eps)
        npz("model_nsf_wide_d%d_h%d" % (D, H), **out)


def gen_made_train():
    """Reference autograd through the single-pass direction of the autoregressive layers at widths the one-launch MADE kernels are
    built for (affine/autoregressive.py:24-27, neural_spline/autoregressive.py:94-134 over nets/made.py:296-304): loss = sum(z * cz)
    + sum(log_det * cl), float32 and float64 legs.  Weights by seeded construction (not stored); of every parameter gradient a strided
    sample (every STRIDE-th element of the flattened tensor) plus its sum and absolute sum."""
    STRIDE = 37
    cases = (("grad_maf_d128_h512", lambda: nf.flows.MaskedAffineAutoregressive(128, 512, num_blocks=2), 128, 1128, 0.05, 96,
              "forward"),
             ("grad_arnsf_d32_h64", lambda: nf.flows.AutoregressiveRationalQuadraticSpline(32, 2, 64, num_bins=8, tail_bound=3,
                                                                                           init_identity=False), 32, 2032, 0.2, 70,
              "inverse"),
             # round 6: the DENSITY direction of config 5's layer (affine/autoregressive.py:29-38: D = 128 recorded MADE passes in the
             # reference) at a batch that takes the in-place weight gradients / the statically unrolled solve (64 rows, 512 positions)
             ("grad_maf_inv_d128_h512", lambda: nf.flows.MaskedAffineAutoregressive(128, 512, num_blocks=2), 128, 1129, 0.05, 64,
              "inverse"))
    only = os.environ.get("NF_GOLDEN_ONLY")
    for tag, make, D, seed, sigma, B, direction in cases:
        if only and tag != only:
            continue
        out = {}
        for dt, leg in ((torch.float32, "f32"), (torch.float64, "f64")):
            torch.manual_seed(seed)
            layer = make()
            perturb(layer, sigma, 8)
            layer = layer.to(dt)
            g = torch.Generator().manual_seed(seed + 1)
            x = (1.3 * torch.randn(B, D, generator=g)).to(dt)
            cz = torch.randn(x.shape, generator=g).to(dt)
            cl = torch.randn(B, generator=g).to(dt)
            xx = x.clone().requires_grad_(True)
            z, ld = getattr(layer, direction)(xx)
            ((z * cz).sum() + (ld * cl).sum()).backward()
            out.update({"z_" + leg: z.detach(), "ld_" + leg: ld.detach(), "gx_" + leg: xx.grad})
            for k, p_ in layer.named_parameters():
                gflat = p_.grad.reshape(-1)
                key = k.replace(".", "__")
                out["g_%s__%s" % (leg, key)] = gflat[::STRIDE].clone()
                out["chk_%s__%s" % (leg, key)] = torch.tensor([float(gflat.double().sum()), float(gflat.double().abs().sum())],
                                                              dtype=torch.float64)
            if dt == torch.float32:
                out.update(x=x, cz=cz, cl=cl)
        npz(tag, stride=np.array(STRIDE), **out)


def gen_maf():
    """MaskedAffineAutoregressive (affine/autoregressive.py): forward = one MADE pass, inverse = D passes."""
    for d, hidden, B in ((20, 40, 9), (128, 512, 4)):
        torch.manual_seed(1000 + d)
        layer = nf.flows.MaskedAffineAutoregressive(d, hidden, num_blocks=2)
        perturb(layer, 0.05, 8)
        x = torch.randn(B, d, generator=torch.Generator().manual_seed(d))
        with torch.no_grad():
            params = layer.autoregressive_net(x)
            zf, ldf = layer.forward(x)
            zi, ldi = layer.inverse(x)
        st = sd(layer, "sd__") if d == 20 else {}   # the d=128 weights are reproduced by seeded construction
        npz("maf_d%d" % d, x=x, params=params, z_fwd=zf, ld_fwd=ldf, z_inv=zi, ld_inv=ldi, **st)


def gen_arnsf():
    """AutoregressiveRationalQuadraticSpline (neural_spline/wrapper.py:188-245 over autoregressive.py:17-140):
    wrapper.forward = mprqat.inverse (D MADE passes), wrapper.inverse = one MADE pass + spline."""
    for name, d, hidden, K, B, ident in (("arnsf_d6", 6, 16, 8, 11, False), ("arnsf_d5_ident", 5, 12, 4, 7, True)):
        torch.manual_seed(2000 + d)
        layer = nf.flows.AutoregressiveRationalQuadraticSpline(d, 2, hidden, num_bins=K, tail_bound=3,
                                                               init_identity=ident)
        perturb(layer, 0.3 if not ident else 0.0, 9)
        x = 1.7 * torch.randn(B, d, generator=torch.Generator().manual_seed(d))
        x[0, 0], x[1, 1] = 3.5, -4.0   # outside the tails
        with torch.no_grad():
            zf, ldf = layer.forward(x)
            zi, ldi = layer.inverse(x)
        npz(name, x=x, z_fwd=zf, ld_fwd=ldf, z_inv=zi, ld_inv=ldi, **sd(layer, "sd__"))
    # the bare transform with tails=None on [0, 1] (autoregressive.py:87-93, :113-115)
    torch.manual_seed(77)
    t = nf.flows.neural_spline.autoregressive.MaskedPiecewiseRationalQuadraticAutoregressive(
        4, 10, num_bins=5, tails=None, num_blocks=2, init_identity=False)
    perturb(t, 0.3, 10)
    x = torch.rand(9, 4, generator=torch.Generator().manual_seed(4))
    with torch.no_grad():
        zf, ldf = t.forward(x)
        zi, ldi = t.inverse(x)
    npz("arnsf_notails", x=x, z_fwd=zf, ld_fwd=ldf, z_inv=zi, ld_inv=ldi, **sd(t, "sd__"))


def gen_glue():
    """Image-side glue: transforms.Logit, ClassCondDiagGaussian, and both inside a class-conditional
    MultiscaleFlow as in examples/glow.ipynb (reduced: L=2, K=2, hidden 8, 3x8x8)."""
    g = torch.Generator().manual_seed(21)
    t = nf.transforms.Logit(alpha=0.05)
    u = torch.rand(5, 3, 4, 4, generator=g)
    v = 3.0 * torch.randn(5, 3, 4, 4, generator=g)
    xi, ldi = t.inverse(u)
    xf, ldf = t.forward(v)
    npz("logit_transform", u=u, v=v, x_inv=xi, ld_inv=ldi, x_fwd=xf, ld_fwd=ldf)
    q = nf.distributions.ClassCondDiagGaussian((3, 2, 2), 4)
    with torch.no_grad():
        q.loc.copy_(torch.randn(q.loc.shape, generator=g))
        q.log_scale.copy_(0.3 * torch.randn(q.log_scale.shape, generator=g))
    z = torch.randn(7, 3, 2, 2, generator=g)
    y = torch.tensor([0, 3, 1, 1, 2, 0, 3])
    ysoft = torch.softmax(torch.randn(7, 4, generator=g), 1)
    with torch.no_grad():
        lp = q.log_prob(z, y)
        lps = q.log_prob(z, ysoft)
        q.temperature = 0.7
        lpt = q.log_prob(z, y)
        q.temperature = None
    npz("class_cond_gauss", z=z, y=y, ysoft=ysoft, log_prob=lp, log_prob_soft=lps, log_prob_temp=lpt, **sd(q, "sd__"))
    torch.manual_seed(5)
    L_, K_, hidden, input_shape, ncls = 2, 2, 8, (3, 8, 8), 3
    q0, merges, flows = [], [], []
    for i in range(L_):
        fl = [nf.flows.GlowBlock(3 * 2 ** (L_ + 1 - i), hidden, split_mode="channel", scale=True) for _ in range(K_)]
        fl += [nf.flows.Squeeze()]
        flows += [fl]
        if i > 0:
            merges += [nf.flows.Merge()]
            latent = (input_shape[0] * 2 ** (L_ - i), input_shape[1] // 2 ** (L_ - i), input_shape[2] // 2 ** (L_ - i))
        else:
            latent = (input_shape[0] * 2 ** (L_ + 1), input_shape[1] // 2 ** L_, input_shape[2] // 2 ** L_)
        q0 += [nf.distributions.ClassCondDiagGaussian(latent, ncls)]
    m = nf.MultiscaleFlow(q0, flows, merges, transform=nf.transforms.Logit(0.05), class_cond=True)
    perturb(m, 0.05, 11)
    x = torch.rand(6, *input_shape, generator=g) * 0.98 + 0.01
    yl = torch.tensor([0, 1, 2, 2, 1, 0])
    sd0 = sd(m, "sd0__")                      # before the data-dependent ActNorm init
    with torch.no_grad():
        lp = m.log_prob(x, yl)
        lp2 = m.log_prob(x, yl)
    npz("model_glow_classcond", x=x, y=yl, log_prob=lp, log_prob_second=lp2, **sd0, **sd(m, "sd__"))


def gen_misc():
    """Remaining classes of the hot-path files: CCAffineConst (affine/coupling.py:57-97), InvertibleAffine
    (mixing.py:136-207), BatchNorm flow (normalization.py:42-62), GlowBase (distributions/base.py:348-471)."""
    g = torch.Generator().manual_seed(31)
    cc = nf.flows.affine.coupling.CCAffineConst((3, 1, 1), 4)
    with torch.no_grad():
        for p_ in cc.parameters():
            p_.copy_(0.3 * torch.randn(p_.shape, generator=g))
    z = torch.randn(5, 3, 2, 2, generator=g)
    y = torch.nn.functional.one_hot(torch.tensor([0, 3, 1, 2, 3]), 4).float()
    with torch.no_grad():
        zf, ldf = cc.forward(z, y)
        zi, ldi = cc.inverse(z, y)
    npz("cc_affine_const", z=z, y=y, z_fwd=zf, ld_fwd=ldf, z_inv=zi, ld_inv=ldi, **sd(cc, "sd__"))
    torch.manual_seed(8)
    for use_lu in (True, False):
        ia = nf.flows.InvertibleAffine(7, use_lu=use_lu)
        perturb(ia, 0.1, 12)
        z = torch.randn(6, 7, generator=g)
        with torch.no_grad():
            zf, ldf = ia.forward(z)
            zi, ldi = ia.inverse(z)
        npz("invertible_affine_lu%d" % int(use_lu), z=z, z_fwd=zf, ld_fwd=ldf, z_inv=zi, ld_inv=ldi, **sd(ia, "sd__"))
    bn = nf.flows.BatchNorm()
    z = 2.0 * torch.randn(9, 5, generator=g) + 1.0
    with torch.no_grad():
        zf, ldf = bn.forward(z)
    npz("batchnorm_flow", z=z, z_fwd=zf, ld_fwd=ldf)
    for ncls in (None, 3):
        gb = nf.distributions.GlowBase((4, 2, 2), num_classes=ncls)
        with torch.no_grad():
            for p_ in gb.parameters():
                p_.copy_(0.2 * torch.randn(p_.shape, generator=g))
        z = torch.randn(6, 4, 2, 2, generator=g)
        yl = torch.tensor([0, 2, 1, 1, 0, 2])
        with torch.no_grad():
            lp = gb.log_prob(z, yl) if ncls else gb.log_prob(z)
            gb.temperature = 0.8
            lpt = gb.log_prob(z, yl) if ncls else gb.log_prob(z)
            gb.temperature = None
        npz("glow_base_cc%d" % (ncls or 0), z=z, y=yl, log_prob=lp, log_prob_temp=lpt, **sd(gb, "sd__"))


def gen_circular():
    """Circular-coordinate spline layers (wrapper.py:88-185, 247-330): per-feature tails (utils/splines.py:48-57),
    scalar and tensor tail bounds (:61-66), periodic conditioner features (utils/nn.py:64-129)."""
    g = torch.Generator().manual_seed(41)
    for name, tb in (("circ_coupled_scalar", 3.0), ("circ_coupled_tensor", torch.tensor([3.0, np.pi, 2.0, np.pi, 3.5, 1.5]))):
        torch.manual_seed(9)
        layer = nf.flows.CircularCoupledRationalQuadraticSpline(6, 2, 16, ind_circ=[1, 3, 4], num_bins=5, tail_bound=tb,
                                                                init_identity=False)
        perturb(layer, 0.3, 13)
        bound = tb if torch.is_tensor(tb) else torch.full((6,), tb)
        x = (torch.rand(12, 6, generator=g) * 2 - 1) * bound * 0.98     # inside every interval
        x[0, 0], x[1, 2] = 50.0, -60.0                                    # linear features outside: the list branch zeroes them
        with torch.no_grad():
            zf, ldf = layer.forward(x)
            zi, ldi = layer.inverse(x)
        npz(name, x=x, z_fwd=zf, ld_fwd=ldf, z_inv=zi, ld_inv=ldi, **sd(layer, "sd__"))
    torch.manual_seed(10)
    layer = nf.flows.CircularAutoregressiveRationalQuadraticSpline(5, 2, 12, ind_circ=[0, 3], num_bins=4, tail_bound=2.5,
                                                                   permute_mask=False, init_identity=False)
    perturb(layer, 0.3, 14)
    x = (torch.rand(9, 5, generator=g) * 2 - 1) * 2.4
    with torch.no_grad():
        zf, ldf = layer.forward(x)
        zi, ldi = layer.inverse(x)
    npz("circ_autoregressive", x=x, z_fwd=zf, ld_fwd=ldf, z_inv=zi, ld_inv=ldi, **sd(layer, "sd__"))
    # string tails + tensor bound on the bare coupling transform
    torch.manual_seed(11)
    tbv = torch.tensor([2.0, 3.0, 1.5, 2.5])
    mask = nf.utils.masks.create_alternating_binary_mask(4, even=False)
    mk = lambda i, o: nf.nets.ResidualNet(i, o, hidden_features=8, num_blocks=1)
    t = nf.flows.neural_spline.coupling.PiecewiseRationalQuadraticCoupling(mask, mk, num_bins=4, tails="linear",
                                                                           tail_bound=tbv, apply_unconditional_transform=True)
    perturb(t, 0.3, 15)
    x = 1.5 * torch.randn(10, 4, generator=g)
    with torch.no_grad():
        zf, ldf = t.forward(x)
        zi, ldi = t.inverse(x)
    npz("coupling_tensor_bound", x=x, z_fwd=zf, ld_fwd=ldf, z_inv=zi, ld_inv=ldi, **sd(t, "sd__"))


def gen_conditional():
    """ConditionalNormalizingFlow (core.py:216-366): context reaches the conditioners (ResidualNet concat + GLU gate,
    MADE context layers) and the base distribution (ConditionalDiagGaussian, distributions/base.py:106-155)."""
    g = torch.Generator().manual_seed(51)
    torch.manual_seed(12)
    flows = []
    for _ in range(2):
        flows += [nf.flows.CoupledRationalQuadraticSpline(4, 1, 8, num_context_channels=3, num_bins=4, init_identity=False),
                  nf.flows.LULinearPermute(4)]
    flows += [nf.flows.AutoregressiveRationalQuadraticSpline(4, 1, 8, num_context_channels=3, num_bins=4,
                                                             init_identity=False)]
    q0 = nf.distributions.base.ConditionalDiagGaussian(4, torch.nn.Linear(3, 8))
    m = nf.ConditionalNormalizingFlow(q0, flows)
    perturb(m, 0.2, 16)
    x = 1.2 * torch.randn(9, 4, generator=g)
    c = torch.randn(9, 3, generator=g)
    with torch.no_grad():
        lp = m.log_prob(x, c)
        z, ld = m.inverse_and_log_det(x, c)
        xf, ldf = m.forward_and_log_det(x, c)
    npz("model_conditional_nsf", x=x, context=c, log_prob=lp, z_inv=z, ld_inv=ld, z_fwd=xf, ld_fwd=ldf, **sd(m, "sd__"))


def gen_reverse_kld():
    """reverse_kld (core.py:104-131): loss and gradients through the SAMPLING path (reparametrised base sample, layer
    .forward) for both gradient estimators; the base noise is the first torch.randn after the seed."""
    for score_fn in (True, False):
        torch.manual_seed(14)
        flows = []
        for _ in range(2):
            flows += [nf.flows.CoupledRationalQuadraticSpline(6, 1, 16, num_bins=4, init_identity=False),
                      nf.flows.LULinearPermute(6)]
        target = nf.distributions.DiagGaussian(6, trainable=False)
        target.loc.add_(0.5)
        target.log_scale.add_(-0.3)
        m = nf.NormalizingFlow(nf.distributions.DiagGaussian(6, trainable=True), flows, p=target)
        perturb(m, 0.1, 17)
        torch.manual_seed(99)
        eps = torch.randn(32, 6)
        torch.manual_seed(99)
        loss = m.reverse_kld(32, beta=0.7, score_fn=score_fn)
        loss.backward()
        grads = {"g__" + k.replace(".", "__"): p_.grad for k, p_ in m.named_parameters() if p_.grad is not None}
        npz("grad_reverse_kld_sf%d" % int(score_fn), eps=eps, loss=loss.detach(), **grads, **sd(m, "sd__"))


def gen_reverse_alpha_div():
    """reverse_alpha_div (core.py:133-165): loss and gradients, plain and doubly reparametrised estimator, alpha on both sides
    of 1; the model and the base noise of gen_reverse_kld."""
    for tag, alpha, dreg in (("a05", 0.5, False), ("a2", 2.0, False), ("a05_dreg", 0.5, True), ("a2_dreg", 2.0, True)):
        torch.manual_seed(14)
        flows = []
        for _ in range(2):
            flows += [nf.flows.CoupledRationalQuadraticSpline(6, 1, 16, num_bins=4, init_identity=False),
                      nf.flows.LULinearPermute(6)]
        target = nf.distributions.DiagGaussian(6, trainable=False)
        target.loc.add_(0.5)
        target.log_scale.add_(-0.3)
        m = nf.NormalizingFlow(nf.distributions.DiagGaussian(6, trainable=True), flows, p=target)
        perturb(m, 0.1, 17)
        torch.manual_seed(99)
        eps = torch.randn(32, 6)
        torch.manual_seed(99)
        loss = m.reverse_alpha_div(32, alpha=alpha, dreg=dreg)
        loss.backward()
        grads = {"g__" + k.replace(".", "__"): p_.grad for k, p_ in m.named_parameters() if p_.grad is not None}
        npz("grad_reverse_alpha_div_" + tag, eps=eps, loss=loss.detach(), alpha=np.float64(alpha), dreg=np.int64(dreg),
            **grads, **sd(m, "sd__"))


def gen_glow_grads():
    """Training step of the class-conditional Glow model of examples/glow.ipynb (reduced) and of the RealNVP model of
    examples/real_nvp.ipynb: forward_kld loss and every parameter gradient from the reference's autograd."""
    g = torch.Generator().manual_seed(61)
    torch.manual_seed(5)
    L_, K_, hidden, input_shape, ncls = 2, 2, 8, (3, 8, 8), 3
    q0, merges, flows = [], [], []
    for i in range(L_):
        fl = [nf.flows.GlowBlock(3 * 2 ** (L_ + 1 - i), hidden, split_mode="channel", scale=True) for _ in range(K_)]
        fl += [nf.flows.Squeeze()]
        flows += [fl]
        if i > 0:
            merges += [nf.flows.Merge()]
            latent = (input_shape[0] * 2 ** (L_ - i), input_shape[1] // 2 ** (L_ - i), input_shape[2] // 2 ** (L_ - i))
        else:
            latent = (input_shape[0] * 2 ** (L_ + 1), input_shape[1] // 2 ** L_, input_shape[2] // 2 ** L_)
        q0 += [nf.distributions.ClassCondDiagGaussian(latent, ncls)]
    m = nf.MultiscaleFlow(q0, flows, merges, class_cond=True)
    perturb(m, 0.05, 18)
    x = torch.rand(6, *input_shape, generator=g)
    yl = torch.tensor([0, 1, 2, 2, 1, 0])
    with torch.no_grad():
        m.log_prob(x, yl)                       # data-dependent ActNorm init
    m.zero_grad()
    loss = m.forward_kld(x, yl)
    loss.backward()
    grads = {"g__" + k.replace(".", "__"): p_.grad for k, p_ in m.named_parameters() if p_.grad is not None}
    npz("grad_glow_classcond", x=x, y=yl, loss=loss.detach(), **grads, **sd(m, "sd__"))
    # RealNVP (config 1 structure): MaskedAffineFlow with MLP s, t + ActNorm
    torch.manual_seed(6)
    b = torch.tensor([1.0, 0.0])
    flows = []
    for i in range(4):
        s_ = nf.nets.MLP([2, 8, 2], init_zeros=True)
        t_ = nf.nets.MLP([2, 8, 2], init_zeros=True)
        flows += [nf.flows.MaskedAffineFlow(b if i % 2 == 0 else 1 - b, t_, s_), nf.flows.ActNorm(2)]
    m = nf.NormalizingFlow(nf.distributions.DiagGaussian(2), flows)
    perturb(m, 0.2, 19)
    x = torch.randn(40, 2, generator=g) * 1.5
    with torch.no_grad():
        m.log_prob(x)
    m.zero_grad()
    loss = m.forward_kld(x)
    loss.backward()
    grads = {"g__" + k.replace(".", "__"): p_.grad for k, p_ in m.named_parameters() if p_.grad is not None}
    npz("grad_realnvp", x=x, loss=loss.detach(), **grads, **sd(m, "sd__"))


def layer_grads(layer, x, tag):
    """Reference autograd through one layer in both directions: loss = sum(z * cz) + sum(log_det * cl)."""
    out = {}
    g = torch.Generator().manual_seed(78)
    cz = torch.randn(x.shape, generator=g)
    cl = torch.randn(x.shape[0], generator=g)
    for name, fn in (("inv", layer.inverse), ("fwd", layer.forward)):
        xx = x.clone().requires_grad_(True)
        layer.zero_grad()
        z, ld = fn(xx)
        ((z * cz).sum() + (ld * cl).sum()).backward()
        out["gx_" + name] = xx.grad.clone()
        for k, p_ in layer.named_parameters():
            out["g_%s__%s" % (name, k.replace(".", "__"))] = torch.zeros_like(p_) if p_.grad is None else p_.grad.clone()
    npz(tag, x=x, cz=cz, cl=cl, **out, **sd(layer, "sd__"))


def gen_circular_grads():
    """Gradients through the per-feature-tails layers (utils/splines.py:48-66): circular coupled NSF with scalar and
    tensor bounds (some linear features outside their interval: zero output, zero gradient), circular AR-NSF, and the
    bare coupling transform with string tails + tensor bound."""
    g = torch.Generator().manual_seed(43)
    for name, tb in (("grad_circ_coupled_scalar", 3.0),
                     ("grad_circ_coupled_tensor", torch.tensor([3.0, np.pi, 2.0, np.pi, 3.5, 1.5]))):
        torch.manual_seed(12)
        layer = nf.flows.CircularCoupledRationalQuadraticSpline(6, 2, 16, ind_circ=[1, 3, 4], num_bins=5, tail_bound=tb,
                                                                init_identity=False)
        perturb(layer, 0.3, 16)
        bound = tb if torch.is_tensor(tb) else torch.full((6,), tb)
        x = (torch.rand(12, 6, generator=g) * 2 - 1) * bound * 0.98
        x[0, 0], x[1, 2], x[2, 5] = 50.0, -60.0, 7.0
        layer_grads(layer, x, name)
    torch.manual_seed(13)
    layer = nf.flows.CircularAutoregressiveRationalQuadraticSpline(5, 2, 12, ind_circ=[0, 3], num_bins=4, tail_bound=2.5,
                                                                   permute_mask=False, init_identity=False)
    perturb(layer, 0.3, 17)
    layer_grads(layer, (torch.rand(9, 5, generator=g) * 2 - 1) * 2.4, "grad_circ_autoregressive")
    torch.manual_seed(14)
    tbv = torch.tensor([2.0, 3.0, 1.5, 2.5])
    mask = nf.utils.masks.create_alternating_binary_mask(4, even=False)
    mk = lambda i, o: nf.nets.ResidualNet(i, o, hidden_features=8, num_blocks=1)
    t = nf.flows.neural_spline.coupling.PiecewiseRationalQuadraticCoupling(mask, mk, num_bins=4, tails="linear",
                                                                           tail_bound=tbv, apply_unconditional_transform=True)
    perturb(t, 0.3, 18)
    layer_grads(t, 1.5 * torch.randn(10, 4, generator=g), "grad_coupling_tensor_bound")


GLOW_CONVNET_CASES = [   # (name, seed, Cin, Cout, leaky, B, H, W)
    ("convnet_6_12_16x16", 31, 6, 12, 0.0, 3, 16, 16),
    ("convnet_12_24_8x8", 32, 12, 24, 0.1, 5, 8, 8),
    ("convnet_24_48_4x4", 33, 24, 48, 0.0, 17, 4, 4),
    ("convnet_3_5_4x8", 34, 3, 5, 0.2, 3, 4, 8),
]


def gen_glow_convnet():
    """GlowBlock conditioner at its real width (cnn.py:5-63 with 256 hidden channels, glow.py:41-62).  Only inputs and
    outputs are stored: the weights are the seeded default initialisation, which our ConvNet2d reproduces draw for
    draw (checked through the stored weight checksum)."""
    for name, seed, cin, cout, leaky, B, H, W in GLOW_CONVNET_CASES:
        torch.manual_seed(seed)
        net = nf.nets.ConvNet2d([cin, 256, 256, cout], [3, 1, 3], leaky, init_zeros=False)
        x = torch.randn(B, cin, H, W, generator=torch.Generator().manual_seed(seed + 100))
        with torch.no_grad():
            out = net(x)
        chk = torch.stack([p_.double().abs().sum() for p_ in net.parameters()])
        npz(name, x=x, out=out, weight_checksum=chk)


GLOW_BLOCK_CASES = [   # (name, seed, C, scale_map, leaky, B, H, W)
    ("glowblock256_C12_16x16", 41, 12, "sigmoid", 0.0, 3, 16, 16),
    ("glowblock256_C24_8x8", 42, 24, "exp", 0.1, 5, 8, 8),
    ("glowblock256_C48_4x4", 43, 48, "sigmoid_inv", 0.0, 19, 4, 4),
    ("glowblock256_C5_4x4", 44, 5, "sigmoid", 0.0, 6, 4, 4),
]


def gen_glow_block256():
    """GlowBlock at its real width (256 hidden channels, glow.py:11-84) in both directions; weights = the seeded default
    construction (checksums stored), the last conditioner layer scaled down so that `exp` scales stay moderate."""
    for name, seed, C, smap, leaky, B, H, W in GLOW_BLOCK_CASES:
        torch.manual_seed(seed)
        blk = nf.flows.GlowBlock(C, 256, scale_map=smap, leaky=leaky, init_zeros=False)
        with torch.no_grad():
            last = blk.flows[0].flows[1].param_map.net[-1]
            last.weight.mul_(0.2)
        x = torch.randn(B, C, H, W, generator=torch.Generator().manual_seed(seed + 100))
        with torch.no_grad():
            zi, ldi = blk.inverse(x)          # first call: ActNorm's data-dependent initialisation
            zi2, ldi2 = blk.inverse(x)
            zf, ldf = blk.forward(x)
        chk = torch.stack([p_.double().abs().sum() for p_ in blk.parameters()])
        npz(name, x=x, z_inv=zi2, ld_inv=ldi2, z_fwd=zf, ld_fwd=ldf, checksum=chk)


def build_glow_c4(nfmod, L_, K_, hidden, seed, scale_last=0.1):
    """BASELINE configs[4] architecture (examples/glow.ipynb cell 2) with a seeded default construction; the conditioners'
    last layers (zero-initialised in the notebook) get the default init scaled down so that the couplings are not
    trivial.  Works for the reference (`nfmod` = normflows) and for our package alike."""
    torch.manual_seed(seed)
    input_shape, channels = (3, 32, 32), 3
    q0, merges, flows = [], [], []
    for i in range(L_):
        fl = [nfmod.flows.GlowBlock(channels * 2 ** (L_ + 1 - i), hidden, split_mode="channel", scale=True, init_zeros=False)
              for _ in range(K_)]
        with torch.no_grad():
            for b in fl:
                b.flows[0].flows[1].param_map.net[-1].weight.mul_(scale_last)
        fl += [nfmod.flows.Squeeze()]
        flows += [fl]
        if i > 0:
            merges += [nfmod.flows.Merge()]
            latent = (input_shape[0] * 2 ** (L_ - i), input_shape[1] // 2 ** (L_ - i), input_shape[2] // 2 ** (L_ - i))
        else:
            latent = (input_shape[0] * 2 ** (L_ + 1), input_shape[1] // 2 ** L_, input_shape[2] // 2 ** L_)
        q0 += [nfmod.distributions.DiagGaussian(latent)]
    return nfmod.MultiscaleFlow(q0, flows, merges, class_cond=False)


def gen_glow_model256():
    """The config-4 model at its real width (L = 3, hidden 256, 32x32x3) with 2 blocks per level: log_prob of 12 images
    (after the data-dependent ActNorm initialisation on the same batch).  Only inputs / outputs / a weight checksum are
    stored; both sides construct the weights from the seed."""
    m = build_glow_c4(nf, 3, 2, 256, seed=61)
    x = torch.rand(12, 3, 32, 32, generator=torch.Generator().manual_seed(62))
    with torch.no_grad():
        lp0 = m.log_prob(x)
        lp = m.log_prob(x)
    chk = torch.stack([p_.double().abs().sum() for p_ in m.parameters()]).sum()
    npz("model_glow_c4_hidden256", x=x, log_prob_first=lp0, log_prob=lp, checksum=chk)


def gen_glow_model_full():
    """BASELINE configs[3] at FULL depth (L = 3, K = 32 blocks per level, hidden 256, 32x32x3; core.py:588-616 over
    affine/glow.py:72-84): reference log_prob of 8 images on the first call (data-dependent ActNorm initialisation of all
    96 blocks, normalization.py:19-39) and on the second call, and the sampling direction (core.py:553-586) on fixed
    per-level base noise: x and log_q.  Only inputs / outputs / a weight checksum are stored; both sides construct the
    weights from the seed."""
    m = build_glow_c4(nf, 3, 32, 256, seed=63)
    x = torch.rand(8, 3, 32, 32, generator=torch.Generator().manual_seed(64))
    with torch.no_grad():
        lp0 = m.log_prob(x)
        lp = m.log_prob(x)
        g = torch.Generator().manual_seed(65)
        eps = [torch.randn((8,) + tuple(q.shape), generator=g) for q in m.q0]
        for i, q in enumerate(m.q0):     # MultiscaleFlow.sample with the base noise given (distributions/base.py:80-92)
            z_ = q.loc + torch.exp(q.log_scale) * eps[i]
            lq_ = -0.5 * q.d * np.log(2 * np.pi) - torch.sum(q.log_scale + 0.5 * eps[i] ** 2, dim=(1, 2, 3))
            if i == 0:
                log_q, z = lq_, z_
            else:
                log_q = log_q + lq_
                z, _ = m.merges[i - 1]([z, z_])
            for flow in m.flows[i]:
                z, ld = flow(z)
                log_q = log_q - ld
        lps = m.log_prob(z)
    chk = torch.stack([p_.double().abs().sum() for p_ in m.parameters()]).sum()
    npz("model_glow_c4_full", x=x, log_prob_first=lp0, log_prob=lp, checksum=chk, eps0=eps[0], eps1=eps[1], eps2=eps[2],
        sample=z, sample_logq=log_q, log_prob_of_sample=lps)


def gen_maf_model_full():
    """BASELINE configs[4] as a MODEL: 10 x MaskedAffineAutoregressive(128, 512, num_blocks=2) under a DiagGaussian base
    (core.py:182-197 over affine/autoregressive.py:29-38, 98-128), 64 rows: `inverse` direction of every layer
    (= log_prob: one MADE pass per layer... the reference's NormalizingFlow.log_prob calls flow.inverse = the D-pass loop),
    the `forward` direction (sampling from fixed noise) and both log-densities.  Weights by seeded construction + the
    seeded perturbation below on both sides; the fixture carries a checksum."""
    torch.manual_seed(2000)
    flows = [nf.flows.MaskedAffineAutoregressive(128, 512, num_blocks=2) for _ in range(10)]
    m = nf.NormalizingFlow(nf.distributions.DiagGaussian(128, trainable=False), flows)
    perturb(m, 0.02, 9)
    g = torch.Generator().manual_seed(2001)
    x = torch.randn(64, 128, generator=g)
    eps = torch.randn(64, 128, generator=g)
    with torch.no_grad():
        z_inv, ld_inv = m.inverse_and_log_det(x)      # 10 x the 128-pass inverse
        lp = m.log_prob(x)
        z_fwd, ld_fwd = m.forward_and_log_det(eps)    # 10 x one MADE pass
        logq = m.q0.log_prob(eps) - ld_fwd
    chk = torch.stack([p_.double().abs().sum() for p_ in m.parameters()]).sum()
    # the same model evaluated by the reference in DOUBLE precision (same fp32 weights, widened): what the fp32 results are
    # rounded versions of -- lets the GPU test hold the 1e-4 bar against the value itself and bound the kernel's error by a
    # small multiple of the reference's own fp32 error instead of a looser absolute tolerance
    m64 = m.double()
    with torch.no_grad():
        z_inv64, ld_inv64 = m64.inverse_and_log_det(x.double())
        lp64 = m64.log_prob(x.double())
        z_fwd64, ld_fwd64 = m64.forward_and_log_det(eps.double())
    npz("model_maf_c5_full", x=x, eps=eps, z_inv=z_inv, ld_inv=ld_inv, log_prob=lp, z_fwd=z_fwd, ld_fwd=ld_fwd,
        sample_logq=logq, checksum=chk, z_inv_f64=z_inv64, ld_inv_f64=ld_inv64, log_prob_f64=lp64, z_fwd_f64=z_fwd64,
        ld_fwd_f64=ld_fwd64)


def gen_cdf():
    """Standalone PiecewiseRationalQuadraticCDF (nsf/coupling.py:170-259): 1-D and N-D parameter shapes, every tails
    variant, values and reference-autograd gradients in both directions."""
    CDF = nf.flows.neural_spline.coupling.PiecewiseRationalQuadraticCDF
    g = torch.Generator().manual_seed(51)
    cases = [("cdf_linear_d5", [5], dict(tails="linear", tail_bound=2.0), 2.5 * torch.randn(7, 5, generator=g)),
             ("cdf_none_img", [3, 4, 4], dict(tails=None), torch.rand(4, 3, 4, 4, generator=g)),
             ("cdf_list_2x3", [2, 3], dict(tails=["linear", "circular", "linear"],
                                          tail_bound=torch.tensor([2.0, 3.0, 1.5])),
              1.4 * torch.randn(6, 2, 3, generator=g))]
    for i, (name, shape, kw, x) in enumerate(cases):
        torch.manual_seed(30 + i)
        t = CDF(shape, num_bins=5, identity_init=False, **kw)
        with torch.no_grad():
            zf, ldf = t.forward(x)
            zi, ldi = t.inverse(x)
        npz(name, x=x, z_fwd=zf, ld_fwd=ldf, z_inv=zi, ld_inv=ldi, **sd(t, "sd__"))
        layer_grads(t, x, "grad_" + name)


def gen_ar_grads():
    """Gradients through the autoregressive layers (MAF, AR-NSF) in both directions and through GlowBase.log_prob."""
    torch.manual_seed(21)
    maf = nf.flows.MaskedAffineAutoregressive(5, 12, num_blocks=2)
    perturb(maf, 0.2, 20)
    layer_grads(maf, torch.randn(7, 5, generator=torch.Generator().manual_seed(1)), "grad_maf_d5")
    torch.manual_seed(22)
    ar = nf.flows.AutoregressiveRationalQuadraticSpline(4, 1, 10, num_bins=4, init_identity=False)
    perturb(ar, 0.2, 21)
    layer_grads(ar, 1.2 * torch.randn(6, 4, generator=torch.Generator().manual_seed(2)), "grad_arnsf_d4")
    g = torch.Generator().manual_seed(3)
    gb = nf.distributions.GlowBase((3, 2, 2), num_classes=2)
    with torch.no_grad():
        for p_ in gb.parameters():
            p_.copy_(0.2 * torch.randn(p_.shape, generator=g))
    z = torch.randn(5, 3, 2, 2, generator=g).requires_grad_(True)
    yl = torch.tensor([0, 1, 1, 0, 1])
    cl = torch.randn(5, generator=g)
    (gb.log_prob(z, yl) * cl).sum().backward()
    grads = {"g__" + k: p_.grad for k, p_ in gb.named_parameters()}
    npz("grad_glow_base", z=z.detach(), y=yl, cl=cl, gz=z.grad, **grads, **sd(gb, "sd__"))


def gen_image_coupling():
    """PiecewiseRationalQuadraticCoupling on NCHW inputs (nsf/coupling.py:150-160): channel mask, conv conditioner,
    per-pixel unconditional transform (img_shape)."""
    g = torch.Generator().manual_seed(71)
    torch.manual_seed(23)
    mask = nf.utils.masks.create_alternating_binary_mask(4, even=False)
    class CtxConv(torch.nn.Module):          # the reference ships no conv conditioner taking (x, context): thin wrapper
        def __init__(self, i, o):
            super().__init__()
            self.net = nf.nets.ConvNet2d([i, 8, o], [3, 3], init_zeros=False)

        def forward(self, x, context=None):
            return self.net(x)

    t = nf.flows.neural_spline.coupling.PiecewiseRationalQuadraticCoupling(
        mask, CtxConv, num_bins=4, tails="linear", tail_bound=3.0, apply_unconditional_transform=True, img_shape=[4, 4])
    perturb(t, 0.3, 22)
    x = 1.5 * torch.randn(3, 4, 4, 4, generator=g)
    x[0, 1, 0, 0] = 5.0
    with torch.no_grad():
        zf, ldf = t.forward(x)
        zi, ldi = t.inverse(x)
    npz("coupling_image", x=x, z_fwd=zf, ld_fwd=ldf, z_inv=zi, ld_inv=ldi, **sd(t, "sd__"))
    layer_grads(t, x, "grad_coupling_image")     # reference autograd through both directions of the same layer


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "image_coupling":
        gen_image_coupling()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "made_train":
        gen_made_train()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "train_nsf_wide":
        gen_train_nsf_wide()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "nsf_wide":
        gen_nsf_wide()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "ar_grads":
        gen_ar_grads()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "glow_grads":
        gen_glow_grads()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "reverse_kld":
        gen_reverse_kld()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "reverse_alpha_div":
        gen_reverse_alpha_div()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "conditional":
        gen_conditional()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "circular":
        gen_circular()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "maf_model_full":
        gen_maf_model_full()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "full_models":
        gen_glow_model_full()
        gen_maf_model_full()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "glow_model256":
        gen_glow_model256()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "glow_block256":
        gen_glow_block256()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "glow_convnet":
        gen_glow_convnet()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "cdf":
        gen_cdf()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "circular_grads":
        gen_circular_grads()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "misc":
        gen_misc()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "glue":
        gen_glue()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "arnsf":
        gen_arnsf()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "maf":
        gen_maf()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "train_c2":
        gen_train_c2_w64h128()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "grads":
        gen_grads()
        sys.exit(0)
    gen_splines()
    gen_nsf_layers()
    gen_affine()
    gen_models()
    gen_grads()
    gen_maf()
    gen_arnsf()
    gen_glue()
    gen_misc()
    gen_circular()
    gen_conditional()
    gen_reverse_kld()
    gen_reverse_alpha_div()
    gen_glow_grads()
    gen_ar_grads()
    gen_image_coupling()
    gen_circular_grads()
    gen_cdf()
    gen_glow_convnet()
    gen_glow_block256()
    gen_glow_model256()
    gen_glow_model_full()
    gen_maf_model_full()
    gen_train_c2_w64h128()
    gen_nsf_wide()
    gen_made_train()
    gen_train_nsf_wide()
