"""GPU tests of the training path: gradients of our autograd Functions (HIP forward,
HIP / GEMM backward) against gradients produced by the reference's autograd (tests/golden/grad_*.npz)."""
import os

import numpy as np
import pytest
import torch

from conftest import assert_close, golden_state, load_golden  # noqa: F401

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def nfa():
    import normflows_amd
    assert torch.cuda.is_available()
    return normflows_amd


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def N(t):
    return t.detach().cpu().numpy()


def load_layer(layer, state, dtype):
    layer.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()}, strict=True)
    return layer.to(dtype).to(DEV)


def check_layer_grads(layer, g, rtol, atol):
    cz, cl = T(g["cz"]), T(g["cl"])
    for name, fn in (("inv", layer.inverse), ("fwd", layer.forward)):
        x = T(g["x"]).requires_grad_(True)
        layer.zero_grad()
        z, ld = fn(x)
        assert z.requires_grad and ld.requires_grad
        ((z * cz).sum() + (ld * cl).sum()).backward()
        assert_close(N(x.grad), g["gx_" + name], what="gx_" + name, rtol=rtol, atol=atol)
        for k, p_ in layer.named_parameters():
            ref = g["g_%s__%s" % (name, k.replace(".", "__"))]
            scale = max(1.0, float(np.abs(ref).max()))
            got = np.zeros_like(ref) if p_.grad is None else N(p_.grad)
            assert_close(got, ref, what="%s grad %s" % (name, k), rtol=rtol, atol=atol * scale)


@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("d,hidden", [(6, 16), (64, 32)])
def test_coupled_rqs_gradients_vs_reference_autograd(nfa, d, hidden, tag):
    g = load_golden("grad_crqs_d%d_%s" % (d, tag))
    dt = torch.float32 if tag == "f32" else torch.float64
    layer = load_layer(nfa.flows.CoupledRationalQuadraticSpline(d, 2, hidden, num_bins=8, init_identity=False),
                       golden_state(g), dt)
    check_layer_grads(layer, g, rtol=2e-3 if tag == "f32" else 1e-8, atol=2e-4 if tag == "f32" else 1e-9)


@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("d", [5, 64])
def test_lu_linear_permute_gradients_vs_reference_autograd(nfa, d, tag):
    g = load_golden("grad_lulinear_d%d_%s" % (d, tag))
    dt = torch.float32 if tag == "f32" else torch.float64
    layer = load_layer(nfa.flows.LULinearPermute(d, identity_init=False), golden_state(g), dt)
    check_layer_grads(layer, g, rtol=2e-3 if tag == "f32" else 1e-8, atol=2e-4 if tag == "f32" else 1e-9)


def test_forward_kld_training_step_vs_reference(nfa):
    """loss = forward_kld(x); loss.backward() (core.py:87-102) on the C2-mini model: loss and every parameter
    gradient match the reference's autograd; one Adam step then lowers the loss."""
    g = load_golden("grad_model_c2mini")
    flows = []
    for _ in range(4):
        flows += [nfa.flows.CoupledRationalQuadraticSpline(16, 2, 32, num_bins=8), nfa.flows.LULinearPermute(16)]
    m = nfa.NormalizingFlow(nfa.distributions.DiagGaussian(16, trainable=True), flows)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in golden_state(g).items()}, strict=True)
    m = m.to(DEV)
    x = T(g["x"])
    loss = m.forward_kld(x)
    assert loss.requires_grad
    assert abs(float(loss) - float(g["loss"])) < 1e-4 * abs(float(g["loss"]))
    loss.backward()
    for k, p_ in m.named_parameters():
        ref = g["g__" + k.replace(".", "__")]
        scale = max(1.0, float(np.abs(ref).max()))
        assert_close(N(p_.grad), ref, what="grad " + k, rtol=5e-3, atol=5e-5 * scale)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    l0 = float(loss)
    for _ in range(5):
        opt.zero_grad()
        loss = m.forward_kld(x)
        loss.backward()
        opt.step()
    assert float(m.forward_kld(x)) < l0
    # the inference (fused / no_grad) path sees the updated parameters
    with torch.no_grad():
        lp = m.log_prob(x)
    assert abs(float(-lp.mean()) - float(m.forward_kld(x))) < 1e-4 * abs(l0)


def _c2_train_model_and_fixture(nfa):
    """2 x [CoupledRationalQuadraticSpline(64, 2, 128) + LULinearPermute(64)], sigma 0.05, rebuilt from its seed; the fixture's
    per-parameter checksums prove these are the weights the reference differentiated."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from bench import build_c2_model
    g = load_golden("grad_model_c2_w64h128")
    m = build_c2_model(num_layers=2, dim=64, hidden=128, seed=0, sigma=0.05)
    for k, p_ in m.named_parameters():
        chk = g["chk__" + k.replace(".", "__")]
        assert abs(float(p_.detach().double().sum()) - float(chk[0])) <= 1e-12 * float(chk[1]), k       # (fp64 sums: the
        assert abs(float(p_.detach().double().abs().sum()) - float(chk[1])) <= 1e-12 * float(chk[1]), k  # order may differ by host)
    return m.to(DEV), g


def test_benchmark_shape_training_step_vs_reference_autograd(train_workgroups, nfa, monkeypatch):
    """The kernels the training step of the BENCHMARK model runs -- the one-launch training forward of a [LU, coupling] pair
    (nf_rqs_fused_train_pair_fwd = rqs_fused_kernel<0,true,2>), nf_coupling_train_bwd (nf_final_bwd, the ring weight gradient,
    nf_resblock_bwd_partials, one reduction), nf_lu_bwd_composed -- against the REFERENCE's autograd (core.py:87-102 forward_kld + loss.backward()) at the benchmark layer shape
    (D = 64, hidden 128, un-padded) and B = 1024, the smallest batch that takes them (tests/golden/make_golden.py train_c2).
    Bars: loss 1e-4 relative; every gradient within 1e-3 of its scale (max |reference gradient|) of the reference's fp32 leg;
    against the fp64 leg, the 90th percentile of the entry errors within 4x the reference's OWN fp32-vs-fp64 90th percentile
    (floor 1e-6 of the scale: the reference's fp32 run happens to have no ReLU pre-activation within rounding of zero on these
    rows; a flipped mask moves single entries by O(1 / rows), which the max bar covers)."""
    from normflows_amd import ops
    m, g = _c2_train_model_and_fixture(nfa)
    calls = {}
    for name in ("rqs_fused_train_full_fwd", "rqs_fused_train_pair_fwd", "coupling_train_bwd", "pair_train_bwd", "final_bwd",
                 "resblock_bwd", "lu_fwd", "lu_bwd"):
        orig = getattr(ops, name)

        def spy(*a, _orig=orig, _name=name, **kw):
            calls[_name] = calls.get(_name, 0) + 1
            return _orig(*a, **kw)
        monkeypatch.setattr(ops, name, spy)
    x = T(g["x"]).requires_grad_(True)
    for f in m.flows[0::2]:
        assert f.prqct._train_full_ok(x, None, False)
    loss = m.forward_kld(x)
    loss.backward()
    # round 6: a [LU, coupling] pair is ONE forward launch (nf_rqs_fused_train_pair_fwd = rqs_fused_kernel<0,true,2>) and its whole
    # backward ONE C-ABI call (nf_pair_train_bwd: nf_final_bwd, the ring weight gradient, nf_resblock_bwd_partials per block,
    # nf_lu_bwd_composed_partials, one reduction launch, nf_lu_param_grads_composed); the kernel-by-kernel wrappers are not called
    assert calls.get("rqs_fused_train_pair_fwd") == 2 and calls.get("pair_train_bwd") == 2, calls
    assert not any(k in calls for k in ("rqs_fused_train_full_fwd", "coupling_train_bwd", "final_bwd", "resblock_bwd", "lu_fwd",
                                        "lu_bwd")), calls
    ref_loss = float(g["loss_f32"])
    assert abs(float(loss) - ref_loss) < 1e-4 * abs(ref_loss), (float(loss), ref_loss)
    assert abs(float(loss) - float(g["loss_f64"])) < 1e-4 * abs(ref_loss)

    def compare(xg, mdl):
        """[(name, max error vs the fp32 leg, our q90 vs the fp64 leg, the reference's own q90)] for x and every parameter; first
        violated bar or None."""
        report, bad = [], None
        items = [("x", N(xg), g["gx_f32"], g["gx_f64"])] + [(k, N(p_.grad), g["g_f32__" + k.replace(".", "__")],
                                                            g["g_f64__" + k.replace(".", "__")]) for k, p_ in mdl.named_parameters()]
        for name, got, r32, r64 in items:
            scale = max(float(np.abs(r64).max()), 1e-6)
            e32, e64, own = np.abs(got - r32) / scale, np.abs(got - r64) / scale, np.abs(r32 - r64) / scale
            q_ours, q_ref = float(np.quantile(e64, 0.9)), float(np.quantile(own, 0.9))
            report.append((name, float(e32.max()), q_ours, q_ref))
            if bad is None and not float(e32.max()) < 1e-3:
                bad = ("gradient of %s vs reference fp32 autograd" % name, float(e32.max()))
            if bad is None and not q_ours <= 4.0 * max(q_ref, 1e-6):
                bad = ("gradient of %s vs reference fp64 autograd" % name, q_ours, q_ref)
        return report, bad

    report, bad = compare(x.grad, m)
    if bad is not None:
        # Rows ON a kink.  A row whose input lies within float32 rounding of a spline knot (or whose ReLU pre-activation does) has two
        # one-sided gradients; which one an evaluation returns depends on the last bit of an intermediate.  Row 153 of this fixture is
        # such a row: scaling its input by 1 +- 2.4e-7 flips its gradient between two values 7 % of the scale apart, for the separate
        # layers and for the pair path alike (tools/kink_row_diag.py, profiles/r06_kink_row_diag.txt); the pair path evaluates the LU
        # as ONE float32 product with the matrix composed in float64, the reference as two float32 products, and they land on
        # opposite sides.  The bar then is: at most 2 such rows of 1024, and moving THOSE rows' inputs by at most 4 ulps gives a step
        # on which every gradient holds all bars (everything else is untouched: the reference values are those of the fixture).
        gx0 = N(x.grad)
        row_err = np.abs(gx0 - g["gx_f32"]).max(1) / max(float(np.abs(g["gx_f64"]).max()), 1e-6)
        kink = np.nonzero(row_err > 1e-3)[0]
        assert 1 <= len(kink) <= 2 and np.isfinite(gx0).all(), (bad, kink, row_err[kink])
        found = None
        for kk in (1, -1, 2, -2, 3, -3, 4, -4):
            x2 = T(g["x"])
            x2[torch.as_tensor(kink, device=x2.device)] *= (1.0 + kk * 1.2e-7)
            x2.requires_grad_(True)
            m.zero_grad(set_to_none=True)
            m.forward_kld(x2).backward()
            report, bad2 = compare(x2.grad, m)
            if bad2 is None:
                found = kk
                break
        assert found is not None, ("no input within 4 ulps of the kink rows %s satisfies the bars" % kink, bad)
        print("kink rows %s: bars hold with their inputs moved by %+d ulp(s)" % (kink.tolist(), found))
    worst = max(report, key=lambda r: r[1])
    print("benchmark-shape step vs reference autograd: worst max-normalised error %.2e (%s); q90 vs fp64 worst %.2e (reference's own %.2e)"
          % (worst[1], worst[0], max(r[2] for r in report), max(r[3] for r in report)))
    # the inference kernels on the same weights and rows (no_grad path) against the reference's log_prob
    with torch.no_grad():
        lp = N(m.log_prob(T(g["x"])))
    assert np.max(np.abs(lp - g["log_prob_f64"]) / np.maximum(1.0, np.abs(g["log_prob_f64"]))) < 1e-4


def test_spline_gradients_match_finite_differences_fp64(nfa):
    """Independent check of nf_rqs_coupling_bwd: central differences of the fp64 forward kernel."""
    torch.manual_seed(0)
    B, D, K = 7, 3, 5
    M = 3 * K - 1
    x = (2.0 * torch.randn(B, D, dtype=torch.float64, device=DEV)).requires_grad_(True)
    cond = torch.randn(B, D * M, dtype=torch.float64, device=DEV, requires_grad=True)
    from normflows_amd.autograd import SplineFn
    kw = dict(tails="linear", tail_bound=3.0, min_bin_width=1e-3, min_bin_height=1e-3, min_derivative=1e-3, wh_div=2.0)
    cz = torch.randn(B, D, dtype=torch.float64, device=DEV)
    cl = torch.randn(B, dtype=torch.float64, device=DEV)
    for inverse in (False, True):
        def f(xv, cv):
            y, ld = SplineFn.apply(xv, cv, None, None, None, K, inverse, kw)
            return (y * cz).sum() + (ld * cl).sum()
        x.grad = cond.grad = None
        f(x, cond).backward()
        eps = 1e-6
        for t, gr in ((x, x.grad), (cond, cond.grad)):
            flat = t.detach().clone().view(-1)
            for i in range(0, flat.numel(), max(1, flat.numel() // 25)):
                tp, tm = flat.clone(), flat.clone()
                tp[i] += eps
                tm[i] -= eps
                with torch.no_grad():
                    a = f(tp.view_as(t) if t is x else x.detach(), tp.view_as(t) if t is cond else cond.detach())
                    b = f(tm.view_as(t) if t is x else x.detach(), tm.view_as(t) if t is cond else cond.detach())
                fd = float(a - b) / (2 * eps)
                assert abs(fd - float(gr.view(-1)[i])) < 1e-5 * max(1.0, abs(fd)), (inverse, i, fd, float(gr.view(-1)[i]))


@pytest.mark.parametrize("B,M,N", [(65536, 128, 128), (65536, 128, 32), (65536, 736, 128), (4099, 70, 33), (1025, 5, 128),
                                   (65536, 768, 128), (4099, 300, 100), (777, 768, 128), (33, 256, 96)])
def test_linear_wgrad_kernel(nfa, B, M, N):
    """nf_linear_wgrad (split-K fp32 MFMA + fixed-order reduction) against fp64 matmul; ragged K and tile edges;
    bit-identical when repeated.  M >= 256 with N in [96, 128] takes the workgroup-tiled kernel (LDS operands)."""
    g = torch.Generator().manual_seed(B + M + N)
    dy = torch.randn(B, M, generator=g).to(DEV)
    x = torch.randn(B, N, generator=g).to(DEV)
    dW, db = nfa.ops.linear_wgrad(dy, x)
    ref = (dy.double().t() @ x.double())
    scale = float(ref.abs().max())
    assert float((dW.double() - ref).abs().max()) < 2e-5 * scale + 1e-3
    refb = dy.double().sum(0)
    assert float((db.double() - refb).abs().max()) < 1e-4 * float(refb.abs().max()) + 1e-3
    dW2, db2 = nfa.ops.linear_wgrad(dy, x)
    assert torch.equal(dW, dW2) and torch.equal(db, db2)


@pytest.mark.parametrize("B,M,N", [(65536, 128, 128), (4099, 64, 32), (3000, 768, 128)])
def test_linear_wgrad_pair_equals_two_single_launches(nfa, B, M, N):
    """nf_linear_wgrad_pair: same partial tiles and the same fixed-order reduction as two nf_linear_wgrad_act calls."""
    g = torch.Generator().manual_seed(B + M)
    dy0, dy1 = torch.randn(B, M, generator=g).to(DEV), torch.randn(B, M, generator=g).to(DEV)
    x0, x1 = torch.randn(B, N, generator=g).to(DEV), torch.randn(B, N, generator=g).to(DEV)
    for relu in (False, True):
        w0, b0, w1, b1 = nfa.ops.linear_wgrad_pair(dy0, x0, dy1, x1, relu_x=relu)
        rw0, rb0 = nfa.ops.linear_wgrad(dy0, x0, relu_x=relu)
        rw1, rb1 = nfa.ops.linear_wgrad(dy1, x1, relu_x=relu)
        assert torch.equal(w0, rw0) and torch.equal(b0, rb0) and torch.equal(w1, rw1) and torch.equal(b1, rb1)


def test_masked_residual_block_training_uses_masked_weights(nfa):
    """MADE's residual block (nets/made.py:140-214) on the one-launch kernel: the products run on weight * mask, the
    autoregressive structure survives (an output of degree d does not move with inputs of degree > d), gradients match the
    plain formula."""
    torch.manual_seed(3)
    made = nfa.nets.MADE(features=16, hidden_features=64, num_blocks=1, output_multiplier=2).to(DEV)
    blk = made.blocks[0]
    for l in blk.linear_layers:
        torch.nn.init.normal_(l.weight, std=0.3)
    x = torch.randn(2048, 64, device=DEV, requires_grad=True)
    assert nfa.autograd.residual_block_fused_ok(blk, x)
    y = blk(x)
    w = torch.randn_like(y)
    (y * w).sum().backward()
    got = [x.grad.clone()] + [p.grad.clone() for p in blk.parameters()]
    x.grad = None
    blk.zero_grad()
    l1, l2 = blk.linear_layers
    t = torch.nn.functional.linear(torch.relu(x), l1.weight * l1.mask, l1.bias)
    y2 = x + torch.nn.functional.linear(torch.relu(t), l2.weight * l2.mask, l2.bias)
    (y2 * w).sum().backward()
    ref = [x.grad.clone()] + [p.grad.clone() for p in blk.parameters()]
    assert _rel(N_(y), N_(y2)) < 1e-5
    for a, b in zip(got, ref):
        assert _rel(N_(a), N_(b)) < 2e-4 * max(1.0, float(b.abs().max())), (a.shape, _rel(N_(a), N_(b)))


def test_spline_backward_pipelined_kernel_full_batch(nfa):
    """The software-pipelined spline backward (counted vmcnt, LDS-DMA double buffering; even B, benchmark layer shape) at
    the full batch against the wave-private kernel it specialises (odd B takes that one): same rows, same gradients to rounding; and
    bit-identical row gradients run to run."""
    torch.manual_seed(11)
    B = 65536
    x = (torch.randn(B + 1, 64) * 1.5).to(DEV)
    gy = torch.randn(B + 1, 64).to(DEV)
    gld = torch.randn(B + 1).to(DEV)
    cond = torch.randn(B + 1, 32, 24).to(DEV)
    cond[:, :, 23] = 0.0
    uw, uh, ud = torch.randn(32, 8).to(DEV), torch.randn(32, 8).to(DEV), torch.randn(32, 7).to(DEV)
    iidx = torch.arange(0, 64, 2, device=DEV)
    tidx = torch.arange(1, 64, 2, device=DEV)
    kw = dict(tail_bound=3.0, wh_div=float(np.sqrt(128.0)))
    a = nfa.ops.rqs_coupling_bwd_p24(x[:B].contiguous(), gy[:B].contiguous(), gld[:B].contiguous(), cond[:B].contiguous(),
                                     uw, uh, ud, iidx, tidx, **kw)
    b = nfa.ops.rqs_coupling_bwd_p24(x[:B].contiguous(), gy[:B].contiguous(), gld[:B].contiguous(), cond[:B].contiguous(),
                                     uw, uh, ud, iidx, tidx, **kw)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    c = nfa.ops.rqs_coupling_bwd_p24(x, gy, gld, cond, uw, uh, ud, iidx, tidx, **kw)       # odd B: the general kernel
    for i in (0, 1):        # row gradients: the same arithmetic inlined into two kernels (fma contraction may differ by an ulp)
        ref = c[i][:B].double()
        err = (a[i].double() - ref).abs() / (1.0 + ref.abs())
        assert float(err.max()) < 1e-4 and float((err > 1e-6).double().mean()) < 1e-3, (i, float(err.max()))
    e = nfa.ops.rqs_coupling_bwd_p24(x[B:].contiguous(), gy[B:].contiguous(), gld[B:].contiguous(), cond[B:].contiguous(),
                                     uw, uh, ud, iidx, tidx, **kw)                         # the extra row alone
    for i in (2, 3, 4):     # shared parameters: fp32 sums over 65 536 rows in atomic order
        ref = c[i].double() - e[i].double()
        assert float((a[i].double() - ref).abs().max()) < 1e-3 * float(ref.abs().max()) + 1e-2


def test_linear_autograd_matches_torch(nfa):
    torch.manual_seed(0)
    lin = nfa.nets.Linear(48, 96).to(DEV)
    x = torch.randn(2048, 48, device=DEV, requires_grad=True)
    y = lin(x)
    w = torch.randn_like(y)
    (y * w).sum().backward()
    gw, gb, gx = lin.weight.grad.clone(), lin.bias.grad.clone(), x.grad.clone()
    lin.zero_grad(); x.grad = None
    y2 = torch.nn.functional.linear(x, lin.weight, lin.bias)
    (y2 * w).sum().backward()
    assert torch.allclose(gw, lin.weight.grad, rtol=1e-4, atol=1e-3)
    assert torch.allclose(gb, lin.bias.grad, rtol=1e-4, atol=1e-3)
    assert torch.allclose(gx, x.grad, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("score_fn", [True, False])
def test_reverse_kld_gradients_vs_reference(nfa, score_fn):
    """reverse_kld (core.py:104-131): loss and every parameter gradient through the sampling direction, for the plain
    and the frozen-parameter (score_fn=False) estimators, on the reference's base noise."""
    g = load_golden("grad_reverse_kld_sf%d" % int(score_fn))
    flows = []
    for _ in range(2):
        flows += [nfa.flows.CoupledRationalQuadraticSpline(6, 1, 16, num_bins=4, init_identity=False),
                  nfa.flows.LULinearPermute(6)]
    target = nfa.distributions.DiagGaussian(6, trainable=False)
    m = nfa.NormalizingFlow(nfa.distributions.DiagGaussian(6, trainable=True), flows, p=target)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in golden_state(g).items()}, strict=True)
    m = m.to(DEV)
    loss = m.reverse_kld(32, beta=0.7, score_fn=score_fn, eps=T(g["eps"]))
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"])) < 2e-4 * max(1.0, abs(float(g["loss"])))
    checked = 0
    for k, p_ in m.named_parameters():
        key = "g__" + k.replace(".", "__")
        if key not in g:
            continue
        ref = g[key]
        got = p_.grad.detach().cpu().numpy() if p_.grad is not None else np.zeros_like(ref)
        scale = max(1e-3, float(np.abs(ref).max()))
        assert float(np.abs(got - ref).max()) < 3e-3 * scale + 1e-5, (k, float(np.abs(got - ref).max()), scale)
        checked += 1
    assert checked >= 20


@pytest.mark.parametrize("tag", ["a05", "a2", "a05_dreg", "a2_dreg"])
def test_reverse_alpha_div_vs_reference(nfa, tag):
    """reverse_alpha_div (core.py:133-165): loss and every parameter gradient, plain and doubly reparametrised estimator,
    alpha = 0.5 and 2, on the reference's base noise (tests/golden/grad_reverse_alpha_div_*.npz)."""
    g = load_golden("grad_reverse_alpha_div_" + tag)
    flows = []
    for _ in range(2):
        flows += [nfa.flows.CoupledRationalQuadraticSpline(6, 1, 16, num_bins=4, init_identity=False),
                  nfa.flows.LULinearPermute(6)]
    target = nfa.distributions.DiagGaussian(6, trainable=False)
    m = nfa.NormalizingFlow(nfa.distributions.DiagGaussian(6, trainable=True), flows, p=target)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in golden_state(g).items()}, strict=True)
    m = m.to(DEV)
    loss = m.reverse_alpha_div(32, alpha=float(g["alpha"]), dreg=bool(int(g["dreg"])), eps=T(g["eps"]))
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"])) < 3e-4 * max(1.0, abs(float(g["loss"])))
    checked = 0
    for k, p_ in m.named_parameters():
        key = "g__" + k.replace(".", "__")
        if key not in g:
            continue
        ref = g[key]
        got = p_.grad.detach().cpu().numpy() if p_.grad is not None else np.zeros_like(ref)
        scale = max(1e-3, float(np.abs(ref).max()))
        assert float(np.abs(got - ref).max()) < 5e-3 * scale + 1e-5, (k, float(np.abs(got - ref).max()), scale)
        checked += 1
    assert checked >= 20


def _check_grads(m, g, rel=3e-3, min_checked=10):
    checked = 0
    for k, p_ in m.named_parameters():
        key = "g__" + k.replace(".", "__")
        if key not in g:
            continue
        ref = g[key]
        got = p_.grad.detach().cpu().numpy() if p_.grad is not None else np.zeros_like(ref)
        scale = max(1e-3, float(np.abs(ref).max()))
        err = float(np.abs(got - ref).max())
        assert err < rel * scale + 1e-5, (k, err, scale)
        checked += 1
    assert checked >= min_checked, checked


def test_glow_training_step_vs_reference(nfa):
    """forward_kld + backward of the class-conditional Glow model (examples/glow.ipynb, reduced): GlowBlock
    (AffineCouplingBlock + Invertible1x1Conv + ActNorm), Squeeze, Merge, ClassCondDiagGaussian -- HIP forward kernels,
    loss and all parameter gradients against the reference's autograd."""
    g = load_golden("grad_glow_classcond")
    L_, K_, hidden, input_shape, ncls = 2, 2, 8, (3, 8, 8), 3
    q0, merges, flows = [], [], []
    for i in range(L_):
        fl = [nfa.flows.GlowBlock(3 * 2 ** (L_ + 1 - i), hidden, split_mode="channel", scale=True) for _ in range(K_)]
        fl += [nfa.flows.Squeeze()]
        flows += [fl]
        if i > 0:
            merges += [nfa.flows.Merge()]
            latent = (input_shape[0] * 2 ** (L_ - i), input_shape[1] // 2 ** (L_ - i), input_shape[2] // 2 ** (L_ - i))
        else:
            latent = (input_shape[0] * 2 ** (L_ + 1), input_shape[1] // 2 ** L_, input_shape[2] // 2 ** L_)
        q0 += [nfa.distributions.ClassCondDiagGaussian(latent, ncls)]
    m = nfa.MultiscaleFlow(q0, flows, merges, class_cond=True)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in golden_state(g).items()}, strict=True)
    m = m.to(DEV)
    loss = m.forward_kld(T(g["x"]), torch.from_numpy(g["y"]).to(DEV))
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"])) < 2e-4 * abs(float(g["loss"]))
    _check_grads(m, g, min_checked=40)
    opt = torch.optim.Adamax(m.parameters(), lr=1e-3)   # the notebook's optimiser: one step runs and changes the loss
    opt.step()
    with torch.no_grad():
        loss2 = m.forward_kld(T(g["x"]), torch.from_numpy(g["y"]).to(DEV))
    assert torch.isfinite(loss2) and float(loss2) != float(loss.detach())


def test_realnvp_training_step_vs_reference(nfa):
    """forward_kld + backward of the RealNVP model of examples/real_nvp.ipynb (MaskedAffineFlow with MLP s, t + ActNorm)."""
    g = load_golden("grad_realnvp")
    b = torch.tensor([1.0, 0.0])
    flows = []
    for i in range(4):
        s_ = nfa.nets.MLP([2, 8, 2], init_zeros=True)
        t_ = nfa.nets.MLP([2, 8, 2], init_zeros=True)
        flows += [nfa.flows.MaskedAffineFlow(b if i % 2 == 0 else 1 - b, t_, s_), nfa.flows.ActNorm(2)]
    m = nfa.NormalizingFlow(nfa.distributions.DiagGaussian(2), flows)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in golden_state(g).items()}, strict=True)
    m = m.to(DEV)
    loss = m.forward_kld(T(g["x"]))
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"])) < 2e-4 * abs(float(g["loss"]))
    _check_grads(m, g, min_checked=30)


def test_maf_gradients_vs_reference_autograd(nfa, monkeypatch):
    """MaskedAffineAutoregressive, both directions (inverse = the D-pass loop under autograd); every MADE pass through the
    hand-written forward / backward kernels (1 single-pass call + 5 passes of the loop)."""
    g = load_golden("grad_maf_d5")
    layer = load_layer(nfa.flows.MaskedAffineAutoregressive(5, 12, num_blocks=2), golden_state(g), torch.float32)
    from normflows_amd.autograd import MafInverseFn
    calls = _spy_made(monkeypatch)
    check_layer_grads(layer, g, rtol=2e-3, atol=2e-4)
    # single-pass direction: one forward / chain / weight-gradient launch; density direction (autograd.MafInverseFn, implicit
    # differentiation): one forward at the solution, ONE nf_maf_solve_t launch for the linear system (round 5; round 4: one chain per
    # sweep, <= D = 5), NO chain for the weight gradients and NO forward at the solution any more (the solve's scratch is that chain,
    # the inverse pass's own scratch holds the linears' inputs: nf_maf_scratch_rows), one weight-gradient launch
    assert calls["fwd"] == 1 and calls["wgrad"] == 2 and calls["bwd"] == 1, calls
    assert MafInverseFn.last_sweeps == 1


def test_arnsf_gradients_vs_reference_autograd(nfa):
    g = load_golden("grad_arnsf_d4")
    layer = load_layer(nfa.flows.AutoregressiveRationalQuadraticSpline(4, 1, 10, num_bins=4, init_identity=False),
                       golden_state(g), torch.float32)
    check_layer_grads(layer, g, rtol=2e-3, atol=2e-4)


@pytest.mark.parametrize("name,tb", [("grad_circ_coupled_scalar", 3.0),
                                     ("grad_circ_coupled_tensor", [3.0, np.pi, 2.0, np.pi, 3.5, 1.5])])
def test_circular_coupled_gradients_vs_reference_autograd(nfa, name, tb):
    """Per-feature tails (utils/splines.py:48-66) through nf_rqs_coupling_bwd_ft: circular / linear features side by
    side, scalar and tensor bounds, linear features outside their interval (zero output, zero gradient)."""
    g = load_golden(name)
    tbv = torch.tensor(tb) if isinstance(tb, list) else tb
    layer = nfa.flows.CircularCoupledRationalQuadraticSpline(6, 2, 16, ind_circ=[1, 3, 4], num_bins=5, tail_bound=tbv,
                                                             init_identity=False)
    check_layer_grads(load_layer(layer, golden_state(g), torch.float32), g, rtol=2e-3, atol=2e-4)


def test_circular_autoregressive_gradients_vs_reference_autograd(nfa):
    g = load_golden("grad_circ_autoregressive")
    layer = nfa.flows.CircularAutoregressiveRationalQuadraticSpline(5, 2, 12, ind_circ=[0, 3], num_bins=4, tail_bound=2.5,
                                                                    permute_mask=False, init_identity=False)
    check_layer_grads(load_layer(layer, golden_state(g), torch.float32), g, rtol=2e-3, atol=2e-4)


def test_coupling_tensor_bound_gradients_vs_reference_autograd(nfa):
    g = load_golden("grad_coupling_tensor_bound")
    mask = nfa.utils.create_alternating_binary_mask(4, even=False)
    mk = lambda i, o: nfa.nets.ResidualNet(i, o, hidden_features=8, num_blocks=1)
    t = nfa.flows.PiecewiseRationalQuadraticCoupling(mask, mk, num_bins=4, tails="linear",
                                                     tail_bound=torch.tensor([2.0, 3.0, 1.5, 2.5]),
                                                     apply_unconditional_transform=True)
    check_layer_grads(load_layer(t, golden_state(g), torch.float32), g, rtol=2e-3, atol=2e-4)


def test_image_spline_coupling_gradients_vs_reference_autograd(nfa):
    """NCHW spline coupling (nsf/coupling.py:150-160) under autograd: conv conditioner, per-pixel unconditional
    transform whose gradient is summed over the batch."""
    import warnings
    g = load_golden("grad_coupling_image")
    mask = nfa.utils.create_alternating_binary_mask(4, even=False)

    class CtxConv(torch.nn.Module):
        def __init__(self, i, o):
            super().__init__()
            self.net = nfa.nets.ConvNet2d([i, 8, o], [3, 3], init_zeros=False)

        def forward(self, x, context=None):
            return self.net(x)

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        t = nfa.flows.PiecewiseRationalQuadraticCoupling(mask, CtxConv, num_bins=4, tails="linear", tail_bound=3.0,
                                                         apply_unconditional_transform=True, img_shape=[4, 4])
    check_layer_grads(load_layer(t, golden_state(g), torch.float32), g, rtol=2e-3, atol=2e-4)


@pytest.mark.parametrize("name,shape,kw", [
    ("cdf_linear_d5", [5], dict(tails="linear", tail_bound=2.0)),
    ("cdf_none_img", [3, 4, 4], dict(tails=None)),
    ("cdf_list_2x3", [2, 3], dict(tails=["linear", "circular", "linear"], tail_bound=[2.0, 3.0, 1.5]))])
def test_standalone_cdf_values_and_gradients_vs_reference(nfa, name, shape, kw):
    """PiecewiseRationalQuadraticCDF on its own (nsf/coupling.py:170-259): (B, *shape) inputs, every tails variant."""
    kw = dict(kw)
    if isinstance(kw.get("tail_bound"), list):
        kw["tail_bound"] = torch.tensor(kw["tail_bound"])
    g = load_golden(name)
    t = load_layer(nfa.flows.PiecewiseRationalQuadraticCDF(shape, num_bins=5, identity_init=False, **kw),
                   golden_state(g), torch.float32)
    with torch.no_grad():
        for d, fn in (("fwd", t.forward), ("inv", t.inverse)):
            z, ld = fn(T(g["x"]))
            assert z.shape == g["x"].shape
            assert_close(N(z), g["z_" + d], what="z_" + d, rtol=1e-4, atol=1e-5)
            assert_close(N(ld), g["ld_" + d], what="ld_" + d, rtol=1e-4, atol=1e-4)
    check_layer_grads(t, load_golden("grad_" + name), rtol=2e-3, atol=2e-4)


def test_glow_base_gradients_vs_reference_autograd(nfa):
    g = load_golden("grad_glow_base")
    gb = nfa.distributions.GlowBase((3, 2, 2), num_classes=2)
    gb.load_state_dict({k: torch.from_numpy(v) for k, v in golden_state(g).items()}, strict=True)
    gb = gb.to(DEV)
    z = T(g["z"]).requires_grad_(True)
    (gb.log_prob(z, torch.from_numpy(g["y"]).to(DEV)) * T(g["cl"])).sum().backward()
    assert_close(N(z.grad), g["gz"], what="gz", rtol=2e-3, atol=2e-4)
    for k, p_ in gb.named_parameters():
        assert_close(N(p_.grad), g["g__" + k], what="grad " + k, rtol=2e-3, atol=2e-4)


def test_example_nsf_density_trains(nfa):
    """examples/nsf_density.py (reference-style training loop on our layers): the loss goes down."""
    import importlib.util
    import sys as _sys
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "nsf_density.py")
    spec = importlib.util.spec_from_file_location("nsf_density_example", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    argv, _sys.argv = _sys.argv, ["nsf_density.py", "--steps", "40", "--batch", "1024", "--dim", "8", "--layers", "3"]
    try:
        first, last = mod.main()
    finally:
        _sys.argv = argv
    assert last < first - 0.5, (first, last)


def test_frozen_realnvp_stack_keeps_the_input_gradient(nfa):
    """A frozen RealNVP / ActNorm stack evaluated on an input that requires grad (reverse_kld(score_fn=False),
    core.py:104-131; the score grad_x log q(x)): the one-launch chain kernel has no autograd path, so run_chain must not
    take it -- d log_q / dx has to equal the gradient obtained with trainable parameters (the layer-by-layer autograd path)."""
    g = load_golden("grad_realnvp")
    b = torch.tensor([1.0, 0.0])
    flows = []
    for i in range(4):
        s_ = nfa.nets.MLP([2, 8, 2], init_zeros=True)
        t_ = nfa.nets.MLP([2, 8, 2], init_zeros=True)
        flows += [nfa.flows.MaskedAffineFlow(b if i % 2 == 0 else 1 - b, t_, s_), nfa.flows.ActNorm(2)]
    m = nfa.NormalizingFlow(nfa.distributions.DiagGaussian(2), flows)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in golden_state(g).items()}, strict=True)
    m = m.to(DEV)
    with torch.no_grad():
        m.log_prob(T(g["x"]))                      # ActNorm initialisation
    x1 = T(g["x"]).requires_grad_(True)
    m.log_prob(x1).sum().backward()                # trainable parameters: autograd path
    for p_ in m.parameters():
        p_.requires_grad_(False)
    x2 = T(g["x"]).requires_grad_(True)
    lp = m.log_prob(x2)
    assert lp.requires_grad, "frozen stack on an input that requires grad lost the graph"
    lp.sum().backward()
    assert float(x2.grad.abs().sum()) > 0
    assert_close(N(x2.grad), N(x1.grad), what="d log_q / dx", rtol=1e-5, atol=1e-5)
    # and the estimator that hits this path: reverse_kld(score_fn=False) on a trainable model yields finite, non-zero gradients
    for p_ in m.parameters():
        p_.requires_grad_(True)
    m.p = nfa.distributions.DiagGaussian(2, trainable=False).to(DEV)
    torch.manual_seed(0)
    loss = m.reverse_kld(num_samples=256, score_fn=False)
    m.zero_grad()
    loss.backward()
    gs = [p_.grad for p_ in m.flows.parameters() if p_.grad is not None]
    assert gs and all(torch.isfinite(g_).all() for g_ in gs) and sum(float(g_.abs().sum()) for g_ in gs) > 0


def test_layers_with_torch_formula_training_path(nfa):
    """InvertibleAffine / CCAffineConst (not on the hot path): inference kernels under no_grad, the reference's formulas as
    differentiable torch ops when a gradient is asked for -- same values, gradients reach the parameters.  Logit likewise."""
    lay = nfa.flows.InvertibleAffine(4).to(DEV)
    x = torch.randn(5, 4, device=DEV)
    for fn in (lay.forward, lay.inverse):
        with torch.no_grad():
            y0, l0 = fn(x)
        y1, l1 = fn(x)
        assert y1.requires_grad and l1.requires_grad
        assert_close(N(y1), N(y0), what="InvertibleAffine torch vs kernel", rtol=1e-5, atol=1e-5)
        assert_close(N(l1), N(l0), what="InvertibleAffine log-det", rtol=1e-5, atol=1e-6)
    (y1.square().sum() + l1).backward()
    assert all(p_.grad is not None and torch.isfinite(p_.grad).all() for p_ in lay.parameters())
    cc = nfa.flows.CCAffineConst((3, 1, 1), 4).to(DEV)
    with torch.no_grad():
        for p_ in cc.parameters():
            p_.normal_(0, 0.3)
    zc = torch.randn(6, 3, 4, 4, device=DEV)
    yc = torch.nn.functional.one_hot(torch.arange(6, device=DEV) % 4, 4).float()
    for fn in (cc.forward, cc.inverse):
        with torch.no_grad():
            y0, l0 = fn(zc, yc)
        y1, l1 = fn(zc, yc)
        assert_close(N(y1), N(y0), what="CCAffineConst torch vs kernel", rtol=1e-5, atol=1e-5)
        assert_close(N(l1), N(l0), what="CCAffineConst log-det", rtol=1e-5, atol=1e-5)
    (y1.sum() + l1.sum()).backward()
    assert all(p_.grad is not None for p_ in cc.parameters())
    lg = nfa.transforms.Logit(0.05)
    z = (torch.rand(3, 2, 4, 4, device=DEV) * 0.9 + 0.05)
    with torch.no_grad():
        y0, ld0 = lg.inverse(z)
    zz = z.clone().requires_grad_(True)
    y1, ld1 = lg.inverse(zz)
    assert y1.requires_grad
    assert_close(N(y1), N(y0), what="logit torch vs kernel", rtol=1e-5, atol=1e-5)
    assert_close(N(ld1), N(ld0), what="logit ld torch vs kernel", rtol=1e-5, atol=1e-4)


def N_(t):
    return t.detach().cpu().numpy()


def _rel(a, b):
    return np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))


def test_residual_block_and_linear_functions_vs_torch_autograd(nfa):
    """ResidualBlockFn / LinearFn (HIP forward, HIP input gradient, HIP weight gradients with ReLU on load) against PyTorch
    autograd of the reference block x + W2 relu(W1 relu(x) + b1) + b2 (resnet.py:37-50) and of the 736-row final layer."""
    from normflows_amd import autograd as ag
    torch.manual_seed(3)
    B, H = 4096, 128
    net = nfa.nets.ResidualNet(32, 736, H, num_blocks=2).to(DEV)
    with torch.no_grad():
        for p_ in net.parameters():
            p_.add_(0.05 * torch.randn_like(p_))
    x = torch.randn(B, 32, device=DEV)
    cy = torch.randn(B, 736, device=DEV)
    assert ag.residual_block_fused_ok(net.blocks[0], torch.randn(B, H, device=DEV))
    results = []
    xa = x.clone().requires_grad_(True)
    (net(xa) * cy).sum().backward()                                # our Functions (B >= 1024): blocks on nf_rows_block
    results.append([xa.grad.clone()] + [p_.grad.clone() for p_ in net.parameters()])
    net.zero_grad()
    xb = x.clone().requires_grad_(True)
    h = torch.nn.functional.linear(xb, net.initial_layer.weight, net.initial_layer.bias)
    for blk in net.blocks:
        l1, l2 = blk.linear_layers
        t = torch.nn.functional.linear(torch.relu(h), l1.weight, l1.bias)
        h = h + torch.nn.functional.linear(torch.relu(t), l2.weight, l2.bias)
    out = torch.nn.functional.linear(h, net.final_layer.weight, net.final_layer.bias)
    (out * cy).sum().backward()
    ref = [xb.grad] + [p_.grad for p_ in net.parameters()]
    with torch.no_grad():
        assert_close(N_(net(x)), N_(out), what="forward (inference path vs torch)", rtol=1e-4, atol=1e-4)
    # The two forwards are different fp32 summation orders: a pre-activation within rounding of zero may get the other ReLU
    # branch (a handful of the 2 M hidden values), which moves that row's gradients by O(1): every gradient is held to
    # 1e-4 of its scale at the median, 1e-3 at the 90th percentile and 2 % everywhere.
    for got in results:
        for i, (a, b) in enumerate(zip(got, ref)):
            scale = max(float(b.abs().max()), 1.0)
            err = np.abs(N_(a) - N_(b))
            assert np.median(err) < 1e-4 * scale and np.quantile(err, 0.9) < 1e-3 * scale, (i, np.median(err), scale)
            assert err.max() < 2e-2 * scale, (i, err.max(), scale)


@pytest.mark.parametrize("B,H", [(1024, 128), (1000, 64), (65536, 128), (3, 36)])
def test_rows_block_kernel_vs_torch(nfa, B, H):
    """nf_rows_block: the residual block forward and its backward form (transposed panels, ReLU masks) in one launch each,
    against float64 torch arithmetic; ragged batches, widths below 128."""
    torch.manual_seed(B + H)
    x = torch.randn(B, H, device=DEV)
    W1, W2 = torch.randn(H, H, device=DEV) / np.sqrt(H), torch.randn(H, H, device=DEV) / np.sqrt(H)
    b1, b2 = torch.randn(H, device=DEV), torch.randn(H, device=DEV)
    t, y = nfa.ops.rows_block(x, W1, b1, W2, b2)
    x64 = x.double()
    t64 = x64.clamp_min(0) @ W1.double().t() + b1.double()
    y64 = x64 + t64.clamp_min(0) @ W2.double().t() + b2.double()
    tol = dict(rtol=3e-5, atol=3e-5 * np.sqrt(H))
    assert_close(N_(t), N_(t64.float()), what="t", **tol)
    # y through OUR t (a t within rounding of zero may take the other ReLU branch in float64)
    y_ref = x64 + t.double().clamp_min(0) @ W2.double().t() + b2.double()
    assert_close(N_(y), N_(y_ref.float()), what="y", **tol)
    gy = torch.randn(B, H, device=DEV)
    gt, gx = nfa.ops.rows_block(gy, W2, None, W1, None, trans=True, mask1=t, mask2=x, relu=False)
    gt64 = (gy.double() @ W2.double()) * (t > 0)
    assert_close(N_(gt), N_(gt64.float()), what="gt", **tol)
    gx64 = gy.double() + (gt.double() @ W1.double()) * (x > 0)
    assert_close(N_(gx), N_(gx64.float()), what="gx", **tol)
    t2, y2 = nfa.ops.rows_block(x, W1, b1, W2, b2)
    assert torch.equal(t, t2) and torch.equal(y, y2)


@pytest.mark.parametrize("B,init", [(65536, False), (65536, True), (192, True), (4096 + 64, False)])
def test_resblock_backward_one_pass_kernel(nfa, B, init):
    """nf_resblock_bwd (both input-gradient products and both weight / bias gradients of a residual block in one pass over the
    rows; with init also the initial Linear layer behind it) against float64 torch on the same inputs, and bit-reproducible."""
    torch.manual_seed(B + init)
    gh, t, h = (torch.randn(B, 128, device=DEV) for _ in range(3))
    W1, W2 = 0.1 * torch.randn(128, 128, device=DEV), 0.1 * torch.randn(128, 128, device=DEV)
    x = torch.randn(B, 64, device=DEV)
    wfull = 0.1 * torch.randn(128, 64, device=DEV)
    wfull[:, 1::2] = 0
    gx0 = torch.randn(B, 64, device=DEV)
    d = lambda v: v.double()
    gt = (d(gh) @ d(W2)) * (t > 0)
    gh_in = d(gh) + (gt @ d(W1)) * (h > 0)
    ref = [gh_in, gt.t() @ d(h).clamp(min=0), gt.sum(0), d(gh).t() @ d(t).clamp(min=0), d(gh).sum(0)]
    if init:
        ref = [d(gx0) + gh_in @ d(wfull)] + ref[1:] + [gh_in.t() @ d(x), gh_in.sum(0)]

    def run():
        if not init:
            return list(nfa.ops.resblock_bwd(gh, t, h, W1, W2))
        gx = gx0.clone()
        return [gx] + list(nfa.ops.resblock_bwd(gh, t, h, W1, W2, x=x, wfull=wfull.t().contiguous(), gx=gx))[1:]

    out, out2 = run(), run()
    for nm, a, r in zip(["gh_in / gx", "dW1", "db1", "dW2", "db2", "dW0", "db0"], out, ref):
        scale = float(r.abs().max())
        assert float((a.double() - r).abs().max()) < 2e-5 * scale, (nm, float((a.double() - r).abs().max()), scale)
    assert all(torch.equal(a, b) for a, b in zip(out, out2))
    with pytest.raises(NotImplementedError):
        nfa.ops.resblock_bwd(gh[:100], t[:100], h[:100], W1, W2)        # rows: multiples of 64


@pytest.mark.parametrize("blocks,B", [(1, 2048), (3, 1088), (2, 1500), (0, 1024)])
def test_whole_layer_training_path_other_depths_and_batches(nfa, blocks, B):
    """The one-launch training forward + one-pass block backward with 1 / 3 / 0 residual blocks (the INIT variant behind the only
    block; no block at all: the trunk-free fallback) and a batch that is no multiple of 64 (the backward falls back to the
    separate kernels), against the layer-wise path on the same weights."""
    from bench import build_c2_model
    m = build_c2_model(num_layers=2, blocks=blocks, sigma=0.05).to(DEV)
    torch.manual_seed(blocks)
    x = 1.2 * torch.randn(B, 64, device=DEV)
    res = []
    for full in (True, False):
        nfa.config.set_train_full(full)
        for f in m.flows[0::2]:
            f.prqct.use_fused_train = full
        xa = x.clone().requires_grad_(True)
        m.zero_grad()
        lp = m.log_prob(xa)
        (-lp.mean()).backward()
        res.append((lp.detach().clone(), xa.grad.clone(), [p_.grad.clone() for p_ in m.parameters()]))
    nfa.config.set_train_full(True)
    for f in m.flows[0::2]:
        f.prqct.use_fused_train = True
    (lp_a, gx_a, gp_a), (lp_b, gx_b, gp_b) = res
    assert _rel(N(lp_a), N(lp_b)) < 2e-5
    # (an identity-column x within rounding of a knot of the batch-shared spline lands in the neighbouring bin in one of the two
    # backward kernels -- libm knot table vs exp2 knots; the spline is C1, so only the log-det term's x-derivative jumps there:
    # a handful of the 131 072 entries may differ by a few per cent)
    ga, gb = N(gx_a), N(gx_b)
    off = np.abs(ga - gb) > 2e-5 + 2e-3 * np.abs(gb)
    assert off.sum() <= 4 and np.abs(ga - gb).max() <= 0.1 * np.abs(gb).max(), (int(off.sum()), float(np.abs(ga - gb).max()))
    for (name, _), a, b in zip(m.named_parameters(), gp_a, gp_b):
        scale = max(float(b.abs().max()), 1e-6)
        assert float((a - b).abs().max()) < 2e-3 * scale, (name, float((a - b).abs().max()), scale)


def test_model_level_prepack_gives_identical_steps(nfa):
    """config.train_prepack (all layers' weights / LU factors packed by one launch per kind at the start of the density pass)
    against every layer packing for itself, over three Adam steps: before every step the prepack model takes the other model's
    parameters (in place: the cached plan and its pointer tables stay in use), then both compute loss and gradients from the
    SAME weights -- the forward is deterministic, so the losses are equal bit for bit and the gradients equal up to the
    summation order of the spline backward's atomics on the batch-shared parameters (~1e-6 relative).  A stale or wrong weight
    image (blob, wpad, wfull_t, LU factors) at step 2 or 3 shows up as a dense O(lr) difference.

    Two free-running trajectories are NOT comparable at this tolerance: tools/prepack_diag.py (profiles/r03_prepack_diag.log)
    shows prepack OFF vs OFF parting ways at step 3 by 7.6e-3 in one weight gradient -- the atomics' 1e-7 noise moves the
    parameters by 1e-8, which flips one of the ~5 M ReLU pre-activations that lie within 1e-7 of zero (round-2 GPUTEST failure).
    Then: a layer called on its own after a model step packs for itself (the token is gone); the pack images of the two modes
    are equal byte for byte; re-assigning ANY parameter rebuilds the plan."""
    import copy
    from bench import build_c2_model
    m0 = build_c2_model(num_layers=4, sigma=0.05).to(DEV)
    x = torch.randn(2048, 64, device=DEV)
    try:
        # (the fused-pair path of round 6 exists only behind the model-level packs: this test compares the PACK IMAGES and the steps of
        # one and the same set of kernels, so both models run the separate layers)
        nfa.config.set_train_pair(False)
        m_on, m_off = copy.deepcopy(m0), copy.deepcopy(m0)
        names = [n for n, _ in m_on.named_parameters()]
        o_on = torch.optim.Adam(m_on.parameters(), lr=1e-3)
        o_off = torch.optim.Adam(m_off.parameters(), lr=1e-3)
        losses = []
        for step in range(3):
            with torch.no_grad():
                for a, b in zip(m_on.parameters(), m_off.parameters()):
                    a.copy_(b)
            out = []
            for on, m, opt in ((True, m_on, o_on), (False, m_off, o_off)):
                nfa.config.set_train_prepack(on)
                opt.zero_grad(set_to_none=True)
                loss = m.forward_kld(x)
                loss.backward()
                out.append((float(loss.detach()), [p_.grad.clone() for p_ in m.parameters()]))
                opt.step()
            assert out[0][0] == out[1][0], (step, out[0][0], out[1][0])
            for n, a, b in zip(names, out[0][1], out[1][1]):
                scale = max(float(b.abs().max()), 1e-6)
                assert float((a - b).abs().max()) <= 1e-5 * scale, (step, n, float((a - b).abs().max()), scale)
            losses.append(out[1][0])
        assert losses[2] < losses[0]
        # the images the backward reads, multi-layer pack against per-layer pack of the same weights: byte for byte
        for fa, fb in zip(m_on.flows, m_off.flows):
            fb.load_state_dict(fa.state_dict())
        nfa.config.set_train_prepack(True)
        m_on.forward_kld(x).backward()
        nfa.config.set_train_prepack(False)
        m_off.forward_kld(x).backward()
        for fa, fb in zip(m_on.flows, m_off.flows):
            if hasattr(fa, "prqct"):
                assert torch.equal(fa.prqct.__dict__["_train_blob"], fb.prqct.__dict__["_train_blob"])
                for ta, tb in zip(fa.prqct.__dict__["_train_wbufs"], fb.prqct.__dict__["_train_wbufs"]):
                    assert torch.equal(ta, tb)
            else:
                D = fa.linear.features
                va = nfa.ops.lu_factors_views(fa.__dict__["_lu_fbuf"], D)
                vb = nfa.ops.lu_factors(fb.permutation._permutation, fb.linear.lower_entries.detach(),
                                        fb.linear.upper_entries.detach(), fb.linear.unconstrained_upper_diag.detach(),
                                        eps=fb.linear.eps)
                for ta, tb in zip(va, vb):
                    assert torch.equal(ta, tb)
        # re-assigning a parameter the old sentinel check did not look at: the plan is rebuilt, no stale row
        nfa.config.set_train_prepack(True)
        lin = m_on.flows[2].prqct.transform_net.blocks[1].linear_layers[0]
        lin.weight = torch.nn.Parameter(lin.weight.detach() * 0.5)
        m_off.flows[2].prqct.transform_net.blocks[1].linear_layers[0].weight.data.mul_(0.5)
        res = []
        for on, m in ((True, m_on), (False, m_off)):
            nfa.config.set_train_prepack(on)
            m.zero_grad(set_to_none=True)
            loss = m.forward_kld(x)
            loss.backward()
            res.append((float(loss.detach()), [p_.grad.clone() for p_ in m.parameters()]))
        assert res[0][0] == res[1][0]
        for n, a, b in zip(names, res[0][1], res[1][1]):
            assert float((a - b).abs().max()) <= 1e-5 * max(float(b.abs().max()), 1e-6), n
        # a layer on its own, after the weights moved: no stale blob
        nfa.config.set_train_prepack(True)
        m = copy.deepcopy(m0)
        m.forward_kld(x).backward()
        with torch.no_grad():
            for p_ in m.parameters():
                p_.add_(0.01 * torch.randn_like(p_))
        lay = m.flows[0]
        xa = x.clone().requires_grad_(True)
        z1, ld1 = lay.inverse(xa)
        nfa.config.set_train_prepack(False)
        z2, ld2 = lay.inverse(x.clone().requires_grad_(True))
        assert torch.equal(z1, z2) and torch.equal(ld1, ld2)
        lu = m.flows[1]
        nfa.config.set_train_prepack(True)
        z3, ld3 = lu.inverse(xa)
        nfa.config.set_train_prepack(False)
        z4, ld4 = lu.inverse(xa)
        assert torch.equal(z3, z4) and torch.equal(ld3, ld4)
    finally:
        nfa.config.set_train_prepack(True)
        nfa.config.set_train_pair(True)


@pytest.mark.parametrize("B", [65536, 1024, 4096 + 64])
def test_lu_backward_one_pass_kernel(nfa, B):
    """nf_lu_bwd (LULinearPermute's batch-side backward, D = 64: both row products and both batch reductions in one pass)
    against float64 torch, bit-reproducible; and the layer's gradients with / without it."""
    torch.manual_seed(B)
    gy, u, x = (torch.randn(B, 64, device=DEV) for _ in range(3))
    Lm = torch.tril(0.2 * torch.randn(64, 64, device=DEV), -1) + torch.eye(64, device=DEV)
    Up = 0.2 * torch.randn(64, 64, device=DEV)
    d = lambda v: v.double()
    gu = d(gy) @ d(Lm)
    ref = [gu @ d(Up), d(gy).t() @ d(u), d(gy).sum(0), gu.t() @ d(x)]
    out, out2 = nfa.ops.lu_bwd(gy, u, x, Lm, Up), nfa.ops.lu_bwd(gy, u, x, Lm, Up)
    for nm, a, r in zip(["gx", "dL", "db", "dUp"], out, ref):
        scale = float(r.abs().max())
        assert float((a.double() - r).abs().max()) < 2e-5 * scale, (nm, float((a.double() - r).abs().max()), scale)
    assert all(torch.equal(a, b) for a, b in zip(out, out2))
    # the forward on the same tiles (nf_lu_fwd) against nf_rows_matvec2: u, y and the accumulated constant log-det
    bias, lad = torch.randn(64, device=DEV), torch.tensor([0.37], device=DEV)
    ld_a, ld_b = torch.ones(B, device=DEV), torch.ones(B, device=DEV)
    u1, y1, _ = nfa.ops.lu_fwd(x, Up.t().contiguous(), Lm.t().contiguous(), bias, lad, +1.0, logdet=ld_a, acc=nfa._lib.LD_SUB)
    u2, y2, _ = nfa.ops.rows_matvec2(x, Up, Lm, bias, lad, +1.0, logdet=ld_b, acc=nfa._lib.LD_SUB)
    for nm, a, b in (("u", u1, u2), ("y", y1, y2), ("ld", ld_a, ld_b)):
        assert float((a - b).abs().max()) < 1e-5 * max(float(b.abs().max()), 1.0), (nm, float((a - b).abs().max()))
    layer = nfa.flows.LULinearPermute(64).to(DEV)
    with torch.no_grad():
        layer.linear.lower_entries.normal_(0, 0.1)
        layer.linear.upper_entries.normal_(0, 0.1)
    xin = torch.randn(B, 64, device=DEV)
    w = torch.randn(B, 64, device=DEV)
    res = []
    for fused in (True, False):
        nfa.config.set_lu_bwd_fused(fused)
        xa = xin.clone().requires_grad_(True)
        layer.zero_grad()
        z, ld = layer.inverse(xa)
        ((z * w).sum() / B + ld.mean()).backward()
        res.append([xa.grad.clone()] + [p_.grad.clone() for p_ in layer.parameters()])
    nfa.config.set_lu_bwd_fused(True)
    for a, b in zip(*res):
        scale = max(float(b.abs().max()), 1e-6)
        assert float((a - b).abs().max()) < 1e-4 * scale, (float((a - b).abs().max()), scale)


def test_fused_final_layer_and_spline_training_forward_vs_layerwise(nfa):
    """Training step of the benchmark-shaped layer: final Linear + coupling transform as ONE forward launch
    (FinalSplineDensityFn: nf_rqs_fused_train_fwd, conditioner output kept in 24-float rows, nf_rqs_coupling_bwd_p24)
    against the layer-wise path (library GEMM for the final layer, nf_rqs_coupling / nf_rqs_coupling_bwd): loss, input
    gradient and every parameter gradient; non-identity weights so that every bin and the tails are exercised."""
    from bench import build_c2_model
    torch.manual_seed(0)
    m = build_c2_model(num_layers=2, sigma=0.05).to(DEV)
    with torch.no_grad():
        for f in m.flows[0::2]:
            u = f.prqct.unconditional_transform
            u.unnormalized_widths.normal_()
            u.unnormalized_heights.normal_()
            f.prqct.transform_net.final_layer.weight.add_(0.05 * torch.randn_like(f.prqct.transform_net.final_layer.weight))
    x = 1.3 * torch.randn(3008, 64, device=DEV)      # a multiple of 64: the one-pass block backward is eligible
    x[:4, :4] = torch.tensor([3.0, -3.0, 3.5, 0.0], device=DEV)
    res = []
    for fused, full, rb in ((True, True, True), (True, True, False), (True, False, False), (False, False, False)):
        # whole layer in one forward launch (CouplingTrainFn), with / without the one-pass residual-block backward /
        # final Linear + spline in one launch / layer-wise
        nfa.config.set_train_full(full)
        nfa.config.set_resblock_bwd(rb)
        for f in m.flows[0::2]:
            f.prqct.use_fused_train = fused
        xa = x.clone().requires_grad_(True)
        m.zero_grad()
        lp = m.log_prob(xa)
        (-lp.mean()).backward()
        res.append((lp.detach().clone(), xa.grad.clone(), [p_.grad.clone() for p_ in m.parameters()]))
    nfa.config.set_train_full(True)
    nfa.config.set_resblock_bwd(True)
    for f in m.flows[0::2]:
        f.prqct.use_fused_train = True
    for what in ("whole-layer launch + one-pass block backward", "whole-layer launch"):
        (lp_w, gx_w, gp_w) = res.pop(0)
        assert _rel(N(lp_w), N(res[-1][0])) < 2e-5, (what, _rel(N(lp_w), N(res[-1][0])))
        assert_close(N(gx_w), N(res[-1][1]), what="input gradient (%s)" % what, rtol=2e-3, atol=2e-5)
        for (name, _), a, b in zip(m.named_parameters(), gp_w, res[-1][2]):
            scale = max(float(b.abs().max()), 1e-6)
            assert float((a - b).abs().max()) < 2e-3 * scale, (what, name, float((a - b).abs().max()), scale)
    (lp_f, gx_f, gp_f), (lp_u, gx_u, gp_u) = res
    assert _rel(N(lp_f), N(lp_u)) < 2e-5, _rel(N(lp_f), N(lp_u))
    assert_close(N(gx_f), N(gx_u), what="input gradient", rtol=2e-3, atol=2e-5)
    for (name, _), a, b in zip(m.named_parameters(), gp_f, gp_u):
        scale = max(float(b.abs().max()), 1e-6)
        assert float((a - b).abs().max()) < 2e-3 * scale, (name, float((a - b).abs().max()), scale)


@pytest.mark.parametrize("B,parity_reverse", [(65536, False), (4096, True), (1100, False), (37, True)])
def test_final_layer_backward_one_pass_kernel(nfa, B, parity_reverse):
    """nf_final_bwd + nf_final_bwd_reduce (the coupling transform's backward and the final Linear's input gradient in one pass,
    csrc/final_bwd.hip) against round 2's separate kernels on the same saved tensors: nf_rqs_coupling_bwd_p24 (the stand-alone
    spline backward) and a float64 product of its gradient rows with the final weight; both mask parities, batches off the
    128-row tile, rows on / outside the interval bounds and NaN; run-to-run bit equality (fixed-order reductions, no atomics)."""
    torch.manual_seed(B)
    layer = nfa.flows.CoupledRationalQuadraticSpline(64, 2, 128, num_bins=8, reverse_mask=parity_reverse).to(DEV)
    c = layer.prqct
    with torch.no_grad():
        for p_ in c.parameters():
            p_.add_(0.15 * torch.randn_like(p_))
        c.transform_net.final_layer.weight.add_(0.3 * torch.randn_like(c.transform_net.final_layer.weight))
    x = 1.6 * torch.randn(B, 64, device=DEV)
    x.view(-1)[:6] = torch.tensor([3.0, -3.0, 3.0000002, float("nan"), float("inf"), 0.0], device=DEV)
    assert c._fused_eligible(x, None)
    net, u = c.transform_net, c.unconditional_transform
    fk = dict(tail_bound=float(c.tail_bound), min_bin_width=c.min_bin_width, min_bin_height=c.min_bin_height,
              min_derivative=c.min_derivative)
    blob = c._train_blob_for(x)
    _, wpad, _, wfull_t = c._train_buffers(x)
    lin = [l for blk in net.blocks for l in blk.linear_layers]
    d = lambda t: t.detach()
    nfa.ops.rqs_fused_pack_all(blob, d(net.initial_layer.weight), d(net.initial_layer.bias), [d(l.weight) for l in lin],
                               [d(l.bias) for l in lin], d(net.final_layer.weight), d(net.final_layer.bias), d(u.unnormalized_widths),
                               d(u.unnormalized_heights), d(u.unnormalized_derivatives), wfull=wfull_t, wpad=wpad,
                               identity_idx=c.identity_features, **fk)
    y, ld, cond24, acts = nfa.ops.rqs_fused_train_full_fwd(x, blob, c._fused_parity, len(net.blocks), **fk)
    gy, gld = torch.randn(B, 64, device=DEV), torch.randn(B, device=DEV)
    uw, uh, ud = d(u.unnormalized_widths), d(u.unnormalized_heights), d(u.unnormalized_derivatives)
    ref = nfa.ops.rqs_coupling_bwd_p24(x, gy, gld, cond24, uw, uh, ud, c.identity_features, c.transform_features,
                                       wh_div=float(np.sqrt(128.0)), **fk)
    wrows = torch.zeros(32, 24, 128, dtype=torch.float64, device=DEV)
    wrows[:, :23] = d(net.final_layer.weight).double().view(32, 23, 128)
    out = nfa.ops.final_bwd(x, gy, gld, cond24, wpad, blob, uw, uh, ud, c._fused_parity, len(net.blocks), **fk)
    out2 = nfa.ops.final_bwd(x, gy, gld, cond24, wpad, blob, uw, uh, ud, c._fused_parity, len(net.blocks), **fk)
    gx, gcond, gh, guw, guh, gud = out
    assert all(torch.equal(a, b) or (torch.isnan(a) == torch.isnan(b)).all() and torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))
               for a, b in zip(out, out2))

    def close(a, b, rtol, atol, what):
        a, b = a.double(), b.double()
        fin = torch.isfinite(b)
        assert (torch.isfinite(a) == fin).all(), what
        if not fin.any():
            return
        scale = max(float(b[fin].abs().max()), 1e-6)
        bad = (a[fin] - b[fin]).abs() > rtol * b[fin].abs() + atol * scale
        assert not bad.any(), (what, int(bad.sum()), float((a[fin] - b[fin]).abs().max()), scale)

    # (the same register routine compiled into two kernels: contraction differences only, amplified in ill-conditioned elements)
    close(gcond, ref[1], 1e-3, 1e-6, "gradient rows")
    tcols, icols = c.transform_features, c.identity_features
    close(gx[:, tcols], ref[0][:, tcols], 1e-3, 1e-6, "gx, transform columns")
    close(gx[:, icols], ref[0][:, icols], 2e-3, 2e-5, "gx, identity columns (libm knot table vs exp2 knots)")
    gh_ref = gcond.double().view(B, 768) @ wrows.view(768, 128)
    close(gh, gh_ref, 1e-4, 1e-5, "gh = g W_final vs float64 on the same rows")
    for a, b, nm in ((guw, ref[2], "widths"), (guh, ref[3], "heights"), (gud, ref[4], "derivatives")):
        close(a, b, 1e-3, 2e-4, "batch-shared %s gradient (knot-space sums vs per-row chain + atomics)" % nm)


def test_training_step_fused_final_backward_vs_separate_kernels(nfa):
    """forward_kld + backward of a 4-pair model with config.final_bwd_fused on / off: every parameter gradient and the input
    gradient agree (same forward launch; the backward differs only in summation order)."""
    from bench import build_c2_model
    m = build_c2_model(num_layers=4, sigma=0.05).to(DEV)
    x = torch.randn(4096, 64, device=DEV)
    res = []
    try:
        for on in (True, False):
            nfa.config.set_final_bwd_fused(on)
            m.zero_grad(set_to_none=True)
            xa = x.clone().requires_grad_(True)
            loss = m.forward_kld(xa)
            loss.backward()
            res.append((float(loss.detach()), xa.grad.clone(), [p_.grad.clone() for p_ in m.parameters()]))
    finally:
        nfa.config.set_final_bwd_fused(True)
    assert res[0][0] == res[1][0]
    assert float((res[0][1] - res[1][1]).abs().max()) <= 1e-5 * float(res[1][1].abs().max())
    for (n, _), a, b in zip(m.named_parameters(), res[0][2], res[1][2]):
        assert float((a - b).abs().max()) <= 2e-5 * max(float(b.abs().max()), 1e-6), n


def test_one_call_layer_backward_is_bit_identical_to_kernel_by_kernel(nfa):
    """Round 6: nf_coupling_train_bwd (four passes over the rows + ONE reduction launch for all of the layer's partial tiles) against
    the same kernels issued one by one, each followed by its own reduction (config.set_train_bwd_onecall(False), rounds 3-5): the
    summation order of every output element is unchanged, so loss, input gradient and EVERY parameter gradient are bit-identical;
    1, 2 and 3 residual blocks, both mask parities, a batch that leaves the last workgroups' tile ranges ragged."""
    from bench import build_c2_model
    for blocks, B in ((2, 4096), (1, 1024), (3, 8192 + 64)):
        m = build_c2_model(num_layers=3, sigma=0.05, blocks=blocks).to(DEV)      # 3 layers: both mask parities
        x = torch.randn(B, 64, device=DEV)
        res = []
        try:
            nfa.config.set_train_pair(False)       # (the fused-pair path needs the one-call backward: compare like with like)
            for on in (True, False):
                nfa.config.set_train_bwd_onecall(on)
                m.zero_grad(set_to_none=True)
                xa = x.clone().requires_grad_(True)
                loss = m.forward_kld(xa)
                loss.backward()
                res.append((float(loss.detach()), xa.grad.clone(), [p_.grad.clone() for p_ in m.parameters()]))
        finally:
            nfa.config.set_train_bwd_onecall(True)
            nfa.config.set_train_pair(True)
        assert res[0][0] == res[1][0]
        assert torch.equal(res[0][1], res[1][1]), "input gradient"
        for (n, _), a, b in zip(m.named_parameters(), res[0][2], res[1][2]):
            assert torch.equal(a, b), (blocks, B, n, float((a - b).abs().max()))


def test_pair_training_path_vs_separate_layers(nfa, monkeypatch):
    """Round 6: [CoupledRQS, LULinearPermute] pairs under autograd as autograd.PairTrainFn -- ONE forward launch with the LU fused in
    front of the coupling (nf_rqs_fused_train_pair_fwd), the composed LU's one-product backward (nf_lu_bwd_composed) and its factor
    gradients on the parameter side (nf_lu_param_grads_composed) -- against LULinearPermuteFn + CouplingTrainFn
    (config.set_train_pair(False)): loss to 1e-6 relative, input-gradient rows to 2e-5 of scale except rows on a kink (W_d is
    composed in float64 and rounded once instead of two float32 products per row), parameter gradients to 5e-4 of their scale (a kink row moves them by O(1 / B)).  The pair
    kernels ran (spy) and nf_lu_fwd / nf_lu_bwd did not; 1 and 2 residual blocks; deterministic."""
    from bench import build_c2_model
    from normflows_amd import ops
    calls = {}
    for name in ("rqs_fused_train_pair_fwd", "pair_train_bwd", "lu_fwd", "lu_bwd"):
        orig = getattr(ops, name)

        def spy(*a, _orig=orig, _name=name, **kw):
            calls[_name] = calls.get(_name, 0) + 1
            return _orig(*a, **kw)
        monkeypatch.setattr(ops, name, spy)
    torch.manual_seed(11)
    for blocks, B in ((2, 4096), (1, 1024)):
        m = build_c2_model(num_layers=3, sigma=0.05, blocks=blocks).to(DEV)
        x = torch.randn(B, 64, device=DEV)
        res = []
        try:
            for on in (True, True, False):
                nfa.config.set_train_pair(on)
                calls.clear()
                m.zero_grad(set_to_none=True)
                xa = x.clone().requires_grad_(True)
                loss = m.forward_kld(xa)
                loss.backward()
                if on:
                    assert calls.get("rqs_fused_train_pair_fwd") == 3 and calls.get("pair_train_bwd") == 3, calls
                    assert "lu_fwd" not in calls and "lu_bwd" not in calls, calls
                else:
                    assert calls.get("lu_fwd") == 3 and calls.get("lu_bwd") == 3 and "rqs_fused_train_pair_fwd" not in calls, calls
                res.append((float(loss.detach()), xa.grad.clone(), [p_.grad.clone() for p_ in m.parameters()]))
        finally:
            nfa.config.set_train_pair(True)
        assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1])
        assert all(torch.equal(a, b) for a, b in zip(res[0][2], res[1][2])), "deterministic"
        assert abs(res[0][0] - res[2][0]) <= 1e-6 * abs(res[2][0]), (res[0][0], res[2][0])
        # rows on a kink of the network (a ReLU pre-activation / knot within float32 rounding: ~1 per million pre-activations) may take
        # the other branch in one of the two evaluations (tools/kink_row_diag.py): at most 1 row per 1000 beyond 2e-5 of scale, each of
        # them moving a parameter gradient by O(1 / B) of its scale
        row_err = (res[0][1] - res[2][1]).abs().amax(1) / float(res[2][1].abs().max())
        assert int((row_err > 2e-5).sum()) <= max(1, B // 1000), (blocks, int((row_err > 2e-5).sum()), float(row_err.max()))
        assert float(row_err.median()) < 2e-6
        for (n, _), a, b in zip(m.named_parameters(), res[0][2], res[2][2]):
            assert float((a - b).abs().max()) <= 5e-4 * max(float(b.abs().max()), 1e-6), (blocks, n, float((a - b).abs().max()), float(b.abs().max()))


def test_flat_parameters_training_step_on_the_benchmark_kernels(nfa):
    """dp.FlatParameters on the benchmark-shaped model (round 6): after backward every parameter's .grad IS its slice of the one flat
    gradient buffer (the one-call layer backward and LULinearPermute's backward wrote there: sync() has nothing to copy), the
    gradients equal those of the ordinary per-tensor run bit for bit, Adam(fused) on the ONE flat tensor gives the parameters Adam
    on the 76 tensors gives, and the inference path sees the stepped weights (packed-weight caches follow the flat step)."""
    import copy
    from bench import build_c2_model
    m = build_c2_model(num_layers=4, sigma=0.05).to(DEV)
    ref = copy.deepcopy(m)
    x = torch.randn(2048, 64, device=DEV)
    with torch.no_grad():
        lp0 = m.log_prob(x).clone()
    flat = nfa.dp.FlatParameters(m)
    opt = torch.optim.Adam(flat.parameters(), lr=1e-3, fused=True)
    opt_ref = torch.optim.Adam(ref.parameters(), lr=1e-3, fused=True)
    for step in range(2):
        flat.zero_grad()
        loss = m.forward_kld(x)
        loss.backward()
        views = {id(p_): v for p_, v in flat.views}
        assert all(p_.grad is not None and p_.grad.data_ptr() == views[id(p_)].data_ptr() for p_ in m.parameters())
        assert flat.sync() == 0
        ref.zero_grad(set_to_none=True)
        loss_ref = ref.forward_kld(x)
        loss_ref.backward()
        assert float(loss.detach()) == float(loss_ref.detach())
        for (n, p_), q in zip(m.named_parameters(), ref.parameters()):
            assert torch.equal(p_.grad, q.grad), (step, n)
        opt.step()
        opt_ref.step()
        for (n, p_), q in zip(m.named_parameters(), ref.parameters()):
            assert torch.allclose(p_, q, rtol=0, atol=1e-7), (step, n, float((p_ - q).abs().max()))
    with torch.no_grad():
        lp1, lp_ref = m.log_prob(x), ref.log_prob(x)
    assert float((lp1 - lp0).abs().max()) > 1e-3, "the inference path must see the stepped weights"
    assert torch.allclose(lp1, lp_ref, rtol=1e-5, atol=1e-4)
    flat.release()


def test_pair_backward_tail_on_the_side_stream(nfa):
    """Round 6 (late; opt-in, config.set_train_reduce_async): the pair backward's last two launches (reduction of the partial tiles, LU
    factor gradients: parameter gradients only) on a side stream under the next pair's kernels (nf_pair_train_bwd_head / _tail,
    _sidestream.py).  Taken when nothing can
    read the gradients before the join at the end of the backward pass -- registered gradient buffers whose .grad is unset, no tensor
    hooks --, otherwise the seven launches stay on one stream.  Either way the gradients are the same bits; what follows backward() on
    the current stream sees them complete (the join is an autograd end-of-pass callback); the step also records into one hipGraph
    (fork and join inside the capture)."""
    from bench import build_c2_model
    from normflows_amd import _sidestream, ops
    m = build_c2_model(num_layers=4, sigma=0.05).to(DEV)
    x = torch.randn(4096, 64, device=DEV)
    flat = nfa.dp.FlatParameters(m)
    sides = []
    orig = ops.pair_train_bwd

    def spy(*a, **k):
        sides.append(k.get("side"))
        return orig(*a, **k)
    ops.pair_train_bwd = spy
    try:
        def step():
            flat.zero_grad()
            loss = m.forward_kld(x)
            loss.backward()
            assert not _sidestream._dirty, "backward() returns with the side stream joined"
            assert flat.sync() == 0
            return float(loss.detach()), flat.grad.clone()       # (a read on the current stream right behind backward())
        nfa.config.set_train_reduce_async(False)
        l0, g0 = step()
        assert len(sides) == 4 and all(s_ is None for s_ in sides)
        nfa.config.set_train_reduce_async(True)
        del sides[:]
        for _ in range(3):
            l1, g1 = step()
            assert l1 == l0 and torch.equal(g1, g0)
        assert len(sides) == 12 and all(isinstance(s_, torch.cuda.Stream) for s_ in sides)
        # an existing .grad (no zero_grad): autograd would ACCUMULATE on the current stream -> one stream
        del sides[:]
        m.forward_kld(x).backward()
        assert len(sides) == 4 and all(s_ is None for s_ in sides)
        # a tensor hook on one parameter of the second pair: that pair stays on one stream, the others do not
        del sides[:]
        flat.zero_grad()
        seen = []
        h = m.flows[2].prqct.transform_net.final_layer.weight.register_hook(lambda g_: seen.append(float(g_.abs().sum())))
        m.forward_kld(x).backward()
        h.remove()
        assert len(seen) == 1 and sum(s_ is None for s_ in sides) == 1 and len(sides) == 4
        assert flat.sync() == 0 and torch.equal(flat.grad, g0)
        # the whole step in one hipGraph: the fork and the end-of-pass join are captured
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                step()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        flat.zero_grad()
        del sides[:]
        with torch.cuda.graph(g):
            m.forward_kld(x).backward()
        assert len(sides) == 4 and all(isinstance(s_, torch.cuda.Stream) for s_ in sides)
        flat.grad.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(flat.grad, g0)
    finally:
        ops.pair_train_bwd = orig
        nfa.config.set_train_reduce_async(False)          # (the default: measured, no gain -- config.py)
        flat.release()


@pytest.fixture(params=["wg128", "wg256"])
def train_workgroups(request, nfa):
    """The whole-layer training forward on both workgroup sizes (round 6, late): batches of <= 32 768 rows take the 128-row build
    (csrc/rqs_fused_nw4.hip) by default; "wg256" keeps them on the 256-row workgroups the benchmark batch runs on."""
    old = nfa.config.set_fused_small_batch(request.param == "wg128")
    yield request.param
    nfa.config.set_fused_small_batch(old)


def test_training_forward_workgroup_sizes_give_the_same_bits(nfa):
    """nf_rqs_fused_train_pair_fwd / _full_fwd on 128-row and on 256-row workgroups: same source, same arithmetic per row -- the loss
    and EVERY gradient of a training step agree bit for bit (the backward kernels read what the forward left row by row)."""
    from bench import build_c2_model
    m = build_c2_model(num_layers=3, sigma=0.05).to(DEV)
    res = {}
    try:
        for B in (1024, 4160):
            x = torch.randn(B, 64, device=DEV)
            for mode in (True, False):
                nfa.config.set_fused_small_batch(mode)
                m.zero_grad(set_to_none=True)
                loss = m.forward_kld(x)
                loss.backward()
                res[mode] = (float(loss.detach()), [p_.grad.clone() for p_ in m.parameters()])
            assert res[True][0] == res[False][0]
            for (n, _), a_, b_ in zip(m.named_parameters(), res[True][1], res[False][1]):
                assert torch.equal(a_, b_), (B, n)
    finally:
        nfa.config.set_fused_small_batch(True)


@pytest.mark.parametrize("shape", [(256, 12, 16, 16), (7, 5, 3, 3), (64, 48, 4, 4), (1, 1, 1, 1), (33, 3, 5, 7)])
def test_channel_sum_vs_torch(nfa, shape):
    """nf_channel_sum (the conditioner's last bias gradient, round 6): g.sum((0, 2, 3)) in one launch with a fixed order -- equal to the
    float64 sum to float32 rounding, the same bits on every call, errno codes for bad arguments."""
    import ctypes as C
    from normflows_amd import _lib as L, ops
    g = torch.randn(*shape, device=DEV)
    s1, s2 = ops.channel_sum(g), ops.channel_sum(g)
    ref = g.double().sum((0, 2, 3))
    assert torch.equal(s1, s2)
    assert float((s1.double() - ref).abs().max()) <= 1e-5 * max(float(g.abs().sum((0, 2, 3)).max()), 1.0)
    lib, st = L.lib(), L.stream()
    one = C.c_void_p(16)
    assert lib.nf_channel_sum(C.c_void_p(0), one, C.c_int64(4), C.c_int(3), C.c_int64(16), st) == -14       # NULL input
    assert lib.nf_channel_sum(one, one, C.c_int64(4), C.c_int(0), C.c_int64(16), st) == -22                 # C < 1
    assert lib.nf_channel_sum(one, one, C.c_int64(-1), C.c_int(3), C.c_int64(16), st) == -22                # B < 0


def test_ragged_training_batch_runs_padded_on_the_tile_kernels(nfa):
    """A differentiable density pass on a batch that is not a multiple of 64 rows (round 6, late; config.train_pad_batch): the model pads
    the batch with zero rows to whole 64-row tiles, runs the pair kernels and slices the result back -- the padding rows get a zero
    cotangent from the slice's backward, so every gradient equals the general kernels' on the ragged batch (to the tolerance two float32
    paths agree to), log_prob has the caller's shape, and the input gradient too."""
    from bench import build_c2_model
    from normflows_amd import ops
    m = build_c2_model(num_layers=4, sigma=0.05).to(DEV)
    B = 1024 + 77
    x = torch.randn(B, 64, device=DEV, requires_grad=True)
    calls = []
    orig = ops.pair_train_bwd

    def spy(*a, **k):
        calls.append(a[0].shape[0])
        return orig(*a, **k)
    ops.pair_train_bwd = spy
    try:
        def run():
            m.zero_grad(set_to_none=True)
            x.grad = None
            lp = m.log_prob(x)
            assert lp.shape == (B,)
            (-lp.mean()).backward()
            return lp.detach().clone(), x.grad.clone(), {n: p_.grad.clone() for n, p_ in m.named_parameters()}
        lp1, gx1, g1 = run()
        assert calls == [1152] * 4, calls                     # the pair kernels, on the padded batch
        nfa.config.set_train_pad_batch(False)
        del calls[:]
        lp0, gx0, g0 = run()
        assert calls == []                                    # the general kernels
    finally:
        ops.pair_train_bwd = orig
        nfa.config.set_train_pad_batch(True)
    assert_close(N(lp1), N(lp0), what="log_prob", rtol=1e-5, atol=2e-4)
    assert gx1.shape == (B, 64)
    bad = ((gx1 - gx0).abs() > 1e-4 * float(gx0.abs().max())).any(dim=1)
    assert int(bad.sum()) <= 2, int(bad.sum())                # (kink rows: a row within rounding of a knot)
    for n in g0:
        err = float((g1[n] - g0[n]).abs().max()) / max(float(g0[n].abs().max()), 1e-12)
        assert err < 2e-3, (n, err)


def test_backward_after_reforward_with_other_weights_raises(nfa):
    """The training Functions read layer-owned weight images (packed blob, transposed final weight, LU factors) in backward;
    a second forward of the same layer with OTHER weights overwrites them.  autograd's saved-tensor check catches in-place
    updates; a re-assigned Parameter it cannot see: the stamp check (autograd._stamp) raises instead of differentiating the
    first graph with the second call's weights.  Two forwards on the same weights (micro-batches) stay legal."""
    from bench import build_c2_model
    m = build_c2_model(num_layers=2, sigma=0.05).to(DEV)
    x = torch.randn(2048, 64, device=DEV)
    loss1 = m.forward_kld(x)
    lin = m.flows[0].prqct.transform_net.final_layer
    lin.weight = torch.nn.Parameter(lin.weight.detach() * 0.5)
    m.forward_kld(x)
    with pytest.raises(RuntimeError, match="ran forward again"):
        loss1.backward()
    m.zero_grad(set_to_none=True)
    l1, l2 = m.forward_kld(x[:1024]), m.forward_kld(x[1024:])
    (0.5 * (l1 + l2)).backward()
    g_two = [p_.grad.clone() for p_ in m.parameters()]
    m.zero_grad(set_to_none=True)
    m.forward_kld(x).backward()
    for (nm, p_), a in zip(m.named_parameters(), g_two):
        assert float((a - p_.grad).abs().max()) <= 1e-4 * max(float(p_.grad.abs().max()), 1e-6), nm


@pytest.mark.parametrize("pshape", [(1, 4, 5), (3, 1, 5), (3, 4, 1)])
def test_affine_const_flow_inner_broadcast_shapes(nfa, pshape):
    """AffineConstFlow / ActNorm with parameters that broadcast over inner dimensions other than trailing ones (coupling.py:30-54,
    normalization.py:21-39 take any broadcastable shape): values, log-det, data-dependent initialisation and parameter gradients
    against the formulas written out in torch."""
    torch.manual_seed(11)
    z = (torch.randn(6, 3, 4, 5, device=DEV) * 1.7 + 0.3)
    f = nfa.flows.AffineConstFlow(pshape).to(DEV)
    with torch.no_grad():
        f.s.normal_(0, 0.4)
        f.t.normal_(0, 0.4)
    reps = z[0].numel() // f.s[0].numel()
    for inverse in (False, True):
        with torch.no_grad():
            y0, l0 = (f.inverse if inverse else f.forward)(z)
        y1, l1 = (f.inverse if inverse else f.forward)(z)
        ref = (z - f.t) * torch.exp(-f.s) if inverse else z * torch.exp(f.s) + f.t
        lref = (-1.0 if inverse else 1.0) * reps * f.s.sum()
        assert_close(N(y0), N(ref), what="inference values", rtol=1e-5, atol=1e-5)
        assert_close(N(l0), N(lref), what="inference log-det", rtol=1e-5, atol=1e-5)
        assert_close(N(y1), N(ref), what="training values", rtol=1e-5, atol=1e-5)
        gs, gt = torch.autograd.grad((y1 * y1).sum() + l1.sum(), [f.s, f.t])
        rs, rt = torch.autograd.grad((ref * ref).sum() + lref * (l1.numel() if l1.dim() else 1), [f.s, f.t])   # per-sample log-dets under autograd
        assert_close(N(gt), N(rt), what="grad t", rtol=1e-4, atol=1e-4)
        assert_close(N(gs), N(rs), what="grad s", rtol=1e-4, atol=1e-3)
    an = nfa.flows.ActNorm(pshape).to(DEV)
    with torch.no_grad():
        y, _ = an.forward(z)
    dims = [0] + [i + 1 for i, d in enumerate(pshape) if d == 1]
    assert float(y.mean(dim=dims).abs().max()) < 1e-4 and float((y.std(dim=dims) - 1).abs().max()) < 1e-3


@pytest.mark.parametrize("K", [8, 10])
def test_utils_splines_entry_points_vs_reference(nfa, K):
    """utils.splines.{unconstrained_,}rational_quadratic_spline (utils/splines.py:16-219 under the reference's own names): the
    inference kernel (no_grad) and the forward + backward kernel pair (gradients asked for) against the reference's outputs
    (tests/golden/spline_K*_f32.npz); gradients reach inputs and all three parameter tensors."""
    g = load_golden("spline_K%d_f32" % K)
    sp = nfa.utils.splines
    w, h = T(g["w"]), T(g["h"])
    for tails, dkey, bound, ykey, lkey in (("linear", "d_lin", 3.0, "yl", "ladl"), ("circular", "d_cir", 2.5, "yc", "ladc")):
        with torch.no_grad():
            y0, l0 = sp.unconstrained_rational_quadratic_spline(T(g["xl"]), w, h, T(g[dkey]), tails=tails, tail_bound=bound)
        assert_close(N(y0), g[ykey], what=ykey, rtol=1e-5, atol=1e-5)
        assert_close(N(l0), g[lkey], what=lkey, rtol=5e-5, atol=5e-5)
        x = T(g["xl"]).clone().requires_grad_(True)
        ps = [t_.clone().requires_grad_(True) for t_ in (w, h, T(g[dkey]))]
        y1, l1 = sp.unconstrained_rational_quadratic_spline(x, *ps, tails=tails, tail_bound=bound)
        assert y1.shape == x.shape and l1.shape == x.shape
        assert_close(N(y1), g[ykey], what=ykey + " (training path)", rtol=1e-5, atol=1e-5)
        assert_close(N(l1), g[lkey], what=lkey + " (training path)", rtol=5e-5, atol=5e-5)
        (y1.sum() + l1.sum()).backward()
        assert all(t_.grad is not None and torch.isfinite(t_.grad).all() for t_ in [x] + ps)
        assert float(x.grad.abs().sum()) > 0 and float(ps[0].grad.abs().sum()) > 0
    with torch.no_grad():
        y, l = sp.rational_quadratic_spline(T(g["x01"]), w, h, T(g["d_none"]))
        yi, li = sp.rational_quadratic_spline(T(g["y01"]), w, h, T(g["d_none"]), inverse=True)
    assert_close(N(y), g["y01"], what="y01", rtol=1e-5, atol=1e-5)
    assert_close(N(l), g["lad01"], what="lad01", rtol=5e-5, atol=5e-5)
    assert_close(N(yi), g["x01_inv"], what="x01_inv", rtol=1e-5, atol=1e-5)
    # the reference's DEFAULT box [0, 1] x [0, 1] under autograd (the backward kernel's `tails=None` box; other square boxes scale to it): values as above,
    # gradients against central differences of the inference kernel in float64
    x = T(g["x01"]).clone().requires_grad_(True)
    ps = [t_.clone().requires_grad_(True) for t_ in (w, h, T(g["d_none"]))]
    y2, l2 = sp.rational_quadratic_spline(x, *ps)
    assert_close(N(y2), g["y01"], what="y01 (training path)", rtol=1e-5, atol=1e-5)
    assert_close(N(l2), g["lad01"], what="lad01 (training path)", rtol=5e-5, atol=5e-5)
    (y2.sum() + l2.sum()).backward()
    assert all(t_.grad is not None and torch.isfinite(t_.grad).all() for t_ in [x] + ps)
    x64 = T(g["x01"]).double()
    p64 = [t_.detach().double() for t_ in ps]
    eps_ = 1e-6
    with torch.no_grad():
        up = sp.rational_quadratic_spline(x64 + eps_, *p64)
        dn = sp.rational_quadratic_spline(x64 - eps_, *p64)
    fd = (up[0] + up[1] - dn[0] - dn[1]) / (2 * eps_)
    inner = (x64 > 0.01) & (x64 < 0.99)
    assert float(((x.grad.double() - fd).abs() / (1.0 + fd.abs()))[inner].max()) < 5e-3
    yi2, _ = sp.rational_quadratic_spline(y2.detach().clone().requires_grad_(True), *ps, inverse=True)
    assert_close(N(yi2), g["x01"], what="inverse (training path)", rtol=1e-4, atol=1e-4)
    # a square box that is not the unit one: [-1, 3] x [2, 6] against the inference kernel on the same box
    xb = (4.0 * T(g["x01"]) - 1.0).clone().requires_grad_(True)
    y3, l3 = sp.rational_quadratic_spline(xb, *ps, left=-1.0, right=3.0, bottom=2.0, top=6.0)
    with torch.no_grad():
        y3n, l3n = sp.rational_quadratic_spline(xb.detach(), *[t_.detach() for t_ in ps], left=-1.0, right=3.0, bottom=2.0, top=6.0)
    assert_close(N(y3), N(y3n), what="square box y", rtol=1e-5, atol=1e-5)
    assert_close(N(l3), N(l3n), what="square box lad", rtol=5e-5, atol=5e-5)
    with pytest.raises(NotImplementedError):        # right - left != top - bottom: no reduction to the unit box
        sp.rational_quadratic_spline(x, *ps, left=0.0, right=2.0, bottom=0.0, top=1.0)
    with pytest.raises(ValueError):
        sp.rational_quadratic_spline(T(g["x01"]), w, h, T(g["d_none"]), min_bin_width=0.2)
    knots = torch.tensor([0.0, 1.0, 2.0], device=DEV)
    assert sp.searchsorted(knots, torch.tensor([0.5, 2.0, 1.0], device=DEV)).tolist() == [0, 1, 1] and float(knots[-1]) == 2.0


# ---- MADE under autograd on the hand-written kernels (csrc/made_fwd.hip EPI 3, csrc/made_bwd.hip) --------------------------------------
def _spy_made(monkeypatch):
    from normflows_amd import ops
    calls = {"fwd": 0, "bwd": 0, "wgrad": 0}
    for key, name in (("fwd", "made_forward_train"), ("bwd", "made_backward"), ("wgrad", "made_wgrad")):
        real = getattr(ops, name)
        monkeypatch.setattr(ops, name, (lambda real, key: lambda *a, **k: (calls.__setitem__(key, calls[key] + 1), real(*a, **k))[1])(
            real, key))
    return calls


def _made_grads(made, x, gp):
    made.zero_grad(set_to_none=True)
    x = x.clone().requires_grad_(True)
    out = made(x)
    out.backward(gp.to(out.dtype))
    return out.detach(), x.grad, [p.grad.clone() for p in made.parameters()]


@pytest.mark.parametrize("D,H,NB,mult,B", [(20, 40, 2, 2, 130), (6, 16, 2, 23, 70), (33, 300, 1, 3, 65), (5, 7, 3, 2, 1),
                                           (128, 512, 2, 2, 300), (128, 512, 2, 23, 64), (64, 256, 2, 2, 1000), (96, 400, 3, 5, 257),
                                           (2, 3, 1, 1, 64), (128, 257, 1, 2, 129)])
def test_made_training_kernels_vs_autograd(nfa, monkeypatch, D, H, NB, mult, B):
    """MADE.forward under autograd (nets/made.py:296-304 inside core.py:87-102): nf_made_forward_train + nf_made_backward +
    nf_made_wgrad against torch autograd through library GEMMs on the same module in float64 (2e-5 of each tensor's scale; measured
    ~1e-6, the float32 library path's own error) -- output, input gradient, every weight / bias gradient (masked entries exactly zero);
    ragged batches, padded hidden widths, 1..3 blocks, mult D beyond one LDS chunk; the three kernels ran (spy); deterministic."""
    import copy
    torch.manual_seed(D * 1000 + H)
    made = nfa.nets.MADE(D, H, num_blocks=NB, output_multiplier=mult)
    with torch.no_grad():
        for p in made.parameters():
            p.add_(0.2 * torch.randn_like(p))
    made = made.to(DEV)
    x = torch.randn(B, D, device=DEV)
    gp = torch.randn(B, mult * D, device=DEV)
    calls = _spy_made(monkeypatch)
    o1, gx1, g1 = _made_grads(made, x, gp)
    assert calls == {"fwd": 1, "bwd": 1, "wgrad": 1}, calls
    o1b, gx1b, g1b = _made_grads(made, x, gp)
    assert torch.equal(o1, o1b) and torch.equal(gx1, gx1b) and all(torch.equal(a, b) for a, b in zip(g1, g1b))
    o2, gx2, g2 = _made_grads(copy.deepcopy(made).double(), x.double(), gp.double())
    assert calls["fwd"] == 2                                   # (the float64 module took torch's path)

    def rel(a, b):
        return float((a.double() - b).abs().max() / (b.abs().max() + 1e-30))
    assert rel(o1, o2) < 2e-5 and rel(gx1, gx2) < 2e-5, (rel(o1, o2), rel(gx1, gx2))
    for (name, p), a, b in zip(made.named_parameters(), g1, g2):
        assert rel(a, b) < 2e-5, (name, rel(a, b))
    for lin in made._linears():                                # the reference's weight.grad is zero under the mask (:80-81)
        assert float((lin.weight.grad * (1 - lin.mask)).abs().max()) == 0.0


@pytest.mark.parametrize("tag,direction", [("grad_maf_d128_h512", "forward"), ("grad_arnsf_d32_h64", "inverse")])
def test_autoregressive_layers_training_vs_reference_autograd(nfa, monkeypatch, tag, direction):
    """The single-pass direction of the autoregressive layers under autograd against the REFERENCE's autograd at kernel-sized widths
    (tests/golden/grad_maf_d128_h512.npz: MaskedAffineAutoregressive(128, 512), BASELINE configs[4]'s layer, affine/autoregressive.py:
    24-27; grad_arnsf_d32_h64.npz: AutoregressiveRationalQuadraticSpline(32, 2, 64).inverse = the density direction, wrapper.py:241-245):
    weights rebuilt from the seed; outputs, input gradient and a strided sample + sum of every parameter gradient: 1e-3 of scale vs the
    float32 leg and no further from the float64 leg than 4 x the reference's own float32 leg (q90); the MADE kernels ran (spy)."""
    g = load_golden(tag)
    if tag.startswith("grad_maf"):
        torch.manual_seed(1128)
        layer = nfa.flows.MaskedAffineAutoregressive(128, 512, num_blocks=2)
        sigma = 0.05
    else:
        torch.manual_seed(2032)
        layer = nfa.flows.AutoregressiveRationalQuadraticSpline(32, 2, 64, num_bins=8, tail_bound=3, init_identity=False)
        sigma = 0.2
    gen = torch.Generator().manual_seed(8)
    with torch.no_grad():
        for p in layer.parameters():
            p.add_(sigma * torch.randn(p.shape, generator=gen, dtype=p.dtype))
    layer = layer.to(DEV)
    calls = _spy_made(monkeypatch)
    x = T(g["x"]).requires_grad_(True)
    z, ld = getattr(layer, direction)(x)
    ((z * T(g["cz"])).sum() + (ld * T(g["cl"])).sum()).backward()
    assert calls == {"fwd": 1, "bwd": 1, "wgrad": 1}, calls
    stride = int(g["stride"])

    def err(a, ref):
        return np.abs(a.astype(np.float64) - ref) / max(1.0, float(np.abs(ref).max()))
    ours = {"z": N(z), "ld": N(ld), "gx": N(x.grad)}
    own, got = [], []
    for k, a in ours.items():
        assert err(a, g[k + "_f32"]).max() < 1e-3, (k, err(a, g[k + "_f32"]).max())
        own.append(err(g[k + "_f32"], g[k + "_f64"]).max())
        got.append(err(a, g[k + "_f64"]).max())
    for k, p in layer.named_parameters():
        key = k.replace(".", "__")
        flat = N(p.grad).reshape(-1)
        ref32, ref64 = g["g_f32__" + key], g["g_f64__" + key]
        assert err(flat[::stride], ref32).max() < 1e-3, (k, err(flat[::stride], ref32).max())
        chk = g["chk_f64__" + key]
        assert abs(float(flat.astype(np.float64).sum()) - chk[0]) < 1e-4 * max(1.0, chk[1]), k
        own.append(err(ref32, ref64).max())
        got.append(err(flat[::stride], ref64).max())
    assert np.quantile(got, 0.9) <= 4 * max(np.quantile(own, 0.9), 1e-7), (np.quantile(got, 0.9), np.quantile(own, 0.9))


def test_maf_density_direction_in_place_backward_vs_reference_autograd(nfa):
    """The round-6 density-direction backward of BASELINE configs[4]'s layer against the REFERENCE's autograd through its D = 128
    recorded MADE passes (tests/golden/grad_maf_inv_d128_h512.npz: MaskedAffineAutoregressive(128, 512).inverse, affine/
    autoregressive.py:29-38, 64 rows -- a batch that takes nf_maf_inverse_h_train, nf_maf_solve_t_tri on 15 regular-8 tiles and
    nf_made_wgrad_pos on the 512 scratch positions: spies): outputs, input gradient and a strided sample + sum of every parameter
    gradient to 1e-3 of scale vs the float32 leg and no further from the float64 leg than 4 x the reference's own float32 leg (q90)."""
    from normflows_amd import ops
    g = load_golden("grad_maf_inv_d128_h512")
    torch.manual_seed(1129)
    layer = nfa.flows.MaskedAffineAutoregressive(128, 512, num_blocks=2)
    gen = torch.Generator().manual_seed(8)
    with torch.no_grad():
        for p in layer.parameters():
            p.add_(0.05 * torch.randn(p.shape, generator=gen, dtype=p.dtype))
    layer = layer.to(DEV)
    calls = {}
    real = {k: getattr(ops, k) for k in ("made_wgrad_pos", "maf_solve_t", "maf_inverse_bits", "maf_scratch_rows", "made_wgrad")}
    try:
        for k, f in real.items():
            def spy(*a, _f=f, _k=k, **kw):
                calls[_k] = calls.get(_k, 0) + 1
                if _k == "maf_solve_t":
                    assert kw.get("table_host") is not None
                if _k == "maf_inverse_bits":
                    assert kw.get("want_params")
                return _f(*a, **kw)
            setattr(ops, k, spy)
        x = T(g["x"]).requires_grad_(True)
        z, ld = layer.inverse(x)
        ((z * T(g["cz"])).sum() + (ld * T(g["cl"])).sum()).backward()
    finally:
        for k, f in real.items():
            setattr(ops, k, f)
    assert calls == {"maf_inverse_bits": 1, "maf_solve_t": 1, "made_wgrad_pos": 1}, calls
    stride = int(g["stride"])

    def err(a, ref):
        return np.abs(a.astype(np.float64) - ref) / max(1.0, float(np.abs(ref).max()))
    ours = {"z": N(z), "ld": N(ld), "gx": N(x.grad)}
    own, got = [], []
    for k, a in ours.items():
        assert err(a, g[k + "_f32"]).max() < 1e-3, (k, err(a, g[k + "_f32"]).max())
        own.append(err(g[k + "_f32"], g[k + "_f64"]).max())
        got.append(err(a, g[k + "_f64"]).max())
    for k, p in layer.named_parameters():
        key = k.replace(".", "__")
        flat = N(p.grad).reshape(-1)
        ref32, ref64 = g["g_f32__" + key], g["g_f64__" + key]
        assert err(flat[::stride], ref32).max() < 1e-3, (k, err(flat[::stride], ref32).max())
        chk = g["chk_f64__" + key]
        assert abs(float(flat.astype(np.float64).sum()) - chk[0]) < 1e-4 * max(1.0, chk[1]), k
        own.append(err(ref32, ref64).max())
        got.append(err(flat[::stride], ref64).max())
    assert np.quantile(got, 0.9) <= 4 * max(np.quantile(own, 0.9), 1e-7), (np.quantile(got, 0.9), np.quantile(own, 0.9))


def test_256_slot_training_kernels_on_128_row_tiles_give_the_same_bits(nfa):
    """mlp_tile.hpp mf_tr128 (round 6, last session): nf_made_forward_train / nf_made_backward of a 256-slot network on <= 64 features run
    batches of >= 32 768 rows (multiples of 128) on 128-row tiles -- a work item spans two sample blocks, every weight fragment feeds eight
    MFMAs.  Each row's arithmetic is unchanged: outputs, input gradient and every parameter gradient bit for bit against the 64-row
    tiles, for GlowBlock's conv conditioner at config 4's 16x16 level and for a dense ResidualNet (wrapper.py:20-35's conditioner)."""
    torch.manual_seed(11)
    conv = nfa.nets.ConvNet2d([6, 256, 256, 12], [3, 1, 3], 0.0, init_zeros=False).to(DEV)
    xc, cc = torch.randn(256, 6, 16, 16, device=DEV), torch.randn(256, 12, 16, 16, device=DEV)     # (512 tiles: every workgroup takes a second one)
    res = nfa.nets.ResidualNet(24, 40, 200, num_blocks=2).to(DEV)
    xr, cr = torch.randn(24576, 24, device=DEV), torch.randn(24576, 40, device=DEV)                 # (192 tiles in one round against 384 in two)
    # ... and a MADE (triangular masks: row-blocks with different k ranges, k-group offsets in the backward's suffix items)
    made = nfa.nets.MADE(40, 150, num_blocks=2, output_multiplier=2, use_residual_blocks=True, random_mask=False,
                         activation=torch.nn.functional.relu).to(DEV)
    with torch.no_grad():
        for p in made.parameters():
            p.add_(0.05 * torch.randn_like(p))
    xm, cm = torch.randn(32768, 40, device=DEV), torch.randn(32768, 80, device=DEV)
    out = []
    from normflows_amd import ops
    seen = []
    real_f = ops.made_forward_train
    ops.made_forward_train = lambda *a, **k: (seen.append(1), real_f(*a, **k))[1]
    prev = nfa.config.set_made_tr128(True)
    try:
        for mode in (True, False):
            nfa.config.set_made_tr128(mode)
            r = []
            for net, x, c in ((conv, xc, cc), (res, xr, cr), (made, xm, cm)):
                net.zero_grad(set_to_none=True)
                xx = x.clone().requires_grad_(True)
                y = net(xx)
                (y * c).sum().backward()
                r += [y.detach().clone(), xx.grad.clone()] + [p.grad.clone() for p in net.parameters()]
            out.append(r)
    finally:
        nfa.config.set_made_tr128(prev)
        ops.made_forward_train = real_f
    assert len(seen) == 6, len(seen)            # (all three networks took the one-launch training forward, in both modes)
    assert len(out[0]) == len(out[1]) and all(torch.isfinite(a).all() for a in out[0])
    assert all(torch.equal(a, b) for a, b in zip(out[0], out[1]))


def test_made_training_full_batch_vs_library_path(nfa):
    """BASELINE configs[4]'s layer at B = 65 536 (a multiple of the 64-row tiles and the weight-gradient chunks): hand-written path vs
    torch autograd through library GEMMs (float32 both: two different summation orders over 65 536 rows), outputs to 1e-4, every
    gradient to 1e-3 of its scale."""
    torch.manual_seed(5)
    layer = nfa.flows.MaskedAffineAutoregressive(128, 512, num_blocks=2)
    gen = torch.Generator().manual_seed(9)
    with torch.no_grad():
        for p in layer.parameters():
            p.add_(0.02 * torch.randn(p.shape, generator=gen))
    layer = layer.to(DEV)
    x0 = torch.randn(65536, 128, generator=torch.Generator().manual_seed(1)).to(DEV)
    res = []
    for mode in (True, False):
        nfa.config.set_made_train(mode)
        try:
            layer.zero_grad(set_to_none=True)
            x = x0.clone().requires_grad_(True)
            z, ld = layer.forward(x)
            (z.square().mean() - ld.mean()).backward()
            res.append((z.detach(), x.grad, [p.grad.clone() for p in layer.parameters()]))
        finally:
            nfa.config.set_made_train(True)

    def rel(a, b):
        return float((a - b).abs().max() / (b.abs().max() + 1e-30))
    assert rel(res[0][0], res[1][0]) < 1e-4 and rel(res[0][1], res[1][1]) < 1e-3, (rel(res[0][0], res[1][0]), rel(res[0][1], res[1][1]))
    for (name, _), a, b in zip(layer.named_parameters(), res[0][2], res[1][2]):
        assert rel(a, b) < 1e-3, (name, rel(a, b))


def test_made_backward_uses_the_weights_of_its_forward(nfa):
    """An in-place parameter update between forward and backward (an optimizer step on retained graphs): the backward still
    differentiates the forward that ran -- the packs held by the graph are the forward-time ones, new ones are built for later calls."""
    torch.manual_seed(3)
    made = nfa.nets.MADE(12, 40, num_blocks=2, output_multiplier=2).to(DEV)
    x = torch.randn(70, 12, device=DEV)
    gp = torch.randn(70, 24, device=DEV)
    ref = _made_grads(made, x, gp)
    made.zero_grad(set_to_none=True)
    xg = x.clone().requires_grad_(True)
    out = made(xg)
    with torch.no_grad():
        for p in made.parameters():
            p.add_(0.5)
    out.backward(gp)
    assert torch.equal(xg.grad, ref[1]) and all(torch.equal(p.grad, r) for p, r in zip(made.parameters(), ref[2]))
    later = _made_grads(made, x, gp)                           # the next call sees the updated weights
    assert not torch.equal(later[0], ref[0])


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
@pytest.mark.parametrize("direction", [0, 1])
def test_maf_affine_backward_kernel_vs_autograd(nfa, dt, direction):
    """nf_maf_affine_bwd against torch autograd through the formula of affine/autoregressive.py:98-128, both directions, with either
    cotangent absent (the D-pass inverse uses the outputs of intermediate passes without their log-dets)."""
    from normflows_amd import autograd
    g = torch.Generator().manual_seed(7)
    B, D = 67, 13
    x0 = torch.randn(B, D, generator=g).to(dt).to(DEV)
    p0 = (1.5 * torch.randn(B, 2 * D, generator=g)).to(dt).to(DEV)
    cz = torch.randn(B, D, generator=g).to(dt).to(DEV)
    cl = torch.randn(B, generator=g).to(dt).to(DEV)

    def formula(x_, p_):
        pr = p_.view(B, D, 2)
        scale = torch.sigmoid(pr[..., 0] + 2.0) + 1e-3
        if direction == 0:
            return scale * x_ + pr[..., 1], torch.log(scale).sum(1)
        return (x_ - pr[..., 1]) / scale, -torch.log(scale).sum(1)
    for use_z, use_l in ((True, True), (True, False), (False, True)):
        res = []
        for fn in (lambda a, b: autograd.MafAffineFn.apply(a, b, direction), formula):
            x = x0.clone().requires_grad_(True)
            p = p0.clone().requires_grad_(True)
            z, ld = fn(x, p)
            loss = (z * cz).sum() if use_z else 0.0
            loss = loss + ((ld * cl).sum() if use_l else 0.0)
            loss.backward()
            res.append((z.detach(), ld.detach(), x.grad if x.grad is not None else torch.zeros_like(x), p.grad))
        tol = 2e-5 if dt == torch.float32 else 1e-12
        for a, b in zip(res[0], res[1]):
            assert float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max()))


@pytest.mark.parametrize("D,H", [(64, 256), (128, 160)])
def test_wide_nsf_training_step_vs_reference_autograd(nfa, monkeypatch, D, H):
    """forward_kld + backward (core.py:87-102) of NSF models whose conditioner has more than 128 hidden units against the REFERENCE's
    autograd (tests/golden/grad_model_nsf_wide_*.npz: 2 x [CoupledRationalQuadraticSpline(D, 2, hidden) + LULinearPermute(D)], B = 200,
    weights rebuilt from the seed): the ResidualNet runs forward, input-gradient chain and weight gradients on the MADE training kernels
    (no mask; spy), the spline on nf_rqs_coupling(_bwd).  Loss 1e-4; gradients 1e-3 of scale vs the float32 leg and within 4 x the
    reference's own float32-vs-float64 error (q90) vs the float64 leg."""
    from bench import build_c2_model
    g = load_golden("grad_model_nsf_wide_d%d_h%d" % (D, H))
    m = build_c2_model(num_layers=2, dim=D, hidden=H, seed=41 + D, sigma=0.05).to(DEV)
    calls = _spy_made(monkeypatch)
    x = T(g["x"]).requires_grad_(True)
    loss = m.forward_kld(x)
    loss.backward()
    assert calls == {"fwd": 2, "bwd": 2, "wgrad": 2}, calls
    assert abs(float(loss.detach()) - float(g["loss_f32"])) < 1e-4 * abs(float(g["loss_f32"]))
    stride = int(g["stride"])

    def err(a, ref):
        return np.abs(np.asarray(a, dtype=np.float64) - ref) / max(1e-6, float(np.abs(ref).max()))
    own, got = [err(g["gx_f32"], g["gx_f64"]).max()], [err(N(x.grad), g["gx_f64"]).max()]
    assert err(N(x.grad), g["gx_f32"]).max() < 1e-3
    for k, p in m.named_parameters():
        key = k.replace(".", "__")
        flat = (np.zeros(p.numel(), dtype=np.float32) if p.grad is None else N(p.grad).reshape(-1))
        ref32, ref64 = g["g_f32__" + key], g["g_f64__" + key]
        if float(np.abs(ref64).max()) == 0.0:
            assert float(np.abs(flat).max()) == 0.0, k
            continue
        assert err(flat[::stride], ref32).max() < 1e-3, (k, err(flat[::stride], ref32).max())
        chk = g["chk_f64__" + key]
        assert abs(float(flat.astype(np.float64).sum()) - chk[0]) < 1e-4 * max(1e-6, chk[1]), k
        own.append(err(ref32, ref64).max())
        got.append(err(flat[::stride], ref64).max())
    assert np.quantile(got, 0.9) <= 4 * max(np.quantile(own, 0.9), 1e-7), (np.quantile(got, 0.9), np.quantile(own, 0.9))
    # differential: torch autograd through library GEMMs on the same model
    nfa.config.set_made_train(False)
    try:
        m.zero_grad(set_to_none=True)
        x2 = T(g["x"]).requires_grad_(True)
        m.forward_kld(x2).backward()
    finally:
        nfa.config.set_made_train(True)
    assert calls["fwd"] == 2 and float((x2.grad - x.grad).abs().max()) < 1e-4 * float(x.grad.abs().max())


@pytest.mark.parametrize("D,B", [(128, 2048), (96, 1500), (70, 130)])
def test_lu_linear_permute_training_wide_vs_float64(nfa, D, B):
    """LULinearPermute (mixing.py:535-563) under autograd for 64 < D <= 128: the density direction on the row mat-vec kernels
    (nf_lu_factors, nf_rows_matvec[_affine], split-K weight-gradient reductions, nf_lu_param_grads) against the reference's formula
    (mixing.py:402-473: y = L (U x[:, perm]) + b, log|det| = sum log(softplus(d) + eps)) in plain float64 torch under torch autograd:
    outputs and every gradient to 2e-5 of scale."""
    torch.manual_seed(D)
    layer = nfa.flows.LULinearPermute(D, identity_init=False)
    with torch.no_grad():
        for p in layer.parameters():
            p.add_(0.05 * torch.randn_like(p))
    layer = layer.to(DEV)
    x0 = torch.randn(B, D, device=DEV)
    cz, cl = torch.randn(B, D, device=DEV), torch.randn(B, device=DEV)
    x = x0.clone().requires_grad_(True)
    z, ld = layer.inverse(x)
    ((z * cz).sum() + (ld * cl).sum()).backward()
    lin = layer.linear
    ours = [z.detach(), ld.detach(), x.grad, lin.lower_entries.grad, lin.upper_entries.grad, lin.unconstrained_upper_diag.grad,
            lin.bias.grad]
    le, ue, ud, b = (t.detach().double().requires_grad_(True) for t in (lin.lower_entries, lin.upper_entries,
                                                                      lin.unconstrained_upper_diag, lin.bias))
    xd = x0.double().requires_grad_(True)
    li, ui = np.tril_indices(D, -1), np.triu_indices(D, 1)
    Lm = torch.eye(D, device=DEV, dtype=torch.float64).index_put((T(li[0]), T(li[1])), le)
    diag = torch.nn.functional.softplus(ud) + lin.eps
    Um = torch.diag(diag).index_put((T(ui[0]), T(ui[1])), ue)
    zr = (xd[:, layer.permutation._permutation] @ Um.t()) @ Lm.t() + b
    ldr = torch.log(diag).sum() * torch.ones(B, device=DEV, dtype=torch.float64)
    ((zr * cz.double()).sum() + (ldr * cl.double()).sum()).backward()
    ref = [zr.detach(), ldr.detach(), xd.grad, le.grad, ue.grad, ud.grad, b.grad]
    for a, r in zip(ours, ref):
        assert float((a.double() - r).abs().max()) <= 2e-5 * max(1.0, float(r.abs().max())), float((a.double() - r).abs().max())


# ---- GlowBlock's conv conditioner under autograd without the convolution library (csrc/conv_rows.hip + MADE kernels, plain-MLP mode) ----
@pytest.mark.parametrize("Cin,hid,Cout,B,H,W", [(6, 256, 12, 8, 16, 16), (12, 256, 24, 5, 8, 8), (24, 256, 48, 7, 4, 4), (3, 16, 5, 2, 2, 2),
                                                (14, 300, 4, 3, 5, 7), (24, 64, 48, 256, 4, 4), (28, 256, 6, 2, 3, 3)])
def test_convnet_training_kernels_vs_autograd(nfa, monkeypatch, Cin, hid, Cout, B, H, W):
    """ConvNet2d([Cin, hid, hid, Cout], (3, 1, 3), LeakyReLU(0)) (nets/cnn.py:5-63) under autograd: nf_conv3x3_gather ->
    nf_made_forward_train (plain MLP 9 Cin -> hid -> hid -> 9 Cout over pixel rows) -> nf_conv3x3_gather_sum, and the mirrored backward
    (nf_made_backward, nf_made_wgrad) against torch autograd through the convolution library in float64: output, input gradient and
    all six parameter gradients to 2e-5 of scale (measured ~5e-7); image borders, ragged pixel counts, 9 Cin up to 252, padded hidden
    widths; the hand-written path ran (spy), twice bit-identical."""
    import copy
    from normflows_amd import ops
    torch.manual_seed(Cin + hid)
    net = nfa.nets.ConvNet2d([Cin, hid, hid, Cout], [3, 1, 3], init_zeros=False).to(DEV)
    x = torch.randn(B, Cin, H, W, device=DEV)
    go = torch.randn(B, Cout, H, W, device=DEV)
    calls = _spy_made(monkeypatch)
    res = []
    for n_, dt in ((net, torch.float32), (net, torch.float32), (copy.deepcopy(net).double(), torch.float64)):
        n_.zero_grad(set_to_none=True)
        xx = x.detach().clone().to(dt).requires_grad_(True)
        out = n_(xx)
        out.backward(go.to(dt))
        res.append([out.detach(), xx.grad] + [p.grad.clone() for p in n_.parameters()])
    assert calls == {"fwd": 2, "bwd": 2, "wgrad": 2}, calls
    assert all(torch.equal(a, b) for a, b in zip(res[0], res[1]))
    for a, b in zip(res[0], res[2]):
        assert float((a.double() - b).abs().max()) <= 2e-5 * float(b.abs().max()), float((a.double() - b).abs().max() / b.abs().max())


def test_training_follows_fused_optimizer_steps(nfa):
    """torch.optim.Adam(fused=True) updates parameters WITHOUT bumping Tensor._version: the training packs are gathered from the current
    parameters in every forward (nf_pack_gather) and every cached inference image carries the optimizer-step epoch (_keys.py) -- three
    fused-Adam steps on the hand-written path end at the same parameters and the same log-density as three steps through torch's own
    autograd, and the inference path right after a step sees the updated weights."""
    import copy
    torch.manual_seed(11)
    base = nfa.NormalizingFlow(nfa.distributions.DiagGaussian(12, trainable=False),
                               [nfa.flows.MaskedAffineAutoregressive(12, 40, num_blocks=2) for _ in range(2)])
    with torch.no_grad():
        for p in base.parameters():
            p.add_(0.1 * torch.randn_like(p))
    eps = torch.randn(300, 12, generator=torch.Generator().manual_seed(1)).to(DEV)
    xs = torch.randn(200, 12, generator=torch.Generator().manual_seed(2)).to(DEV)
    outs = []
    for mode in (True, False):
        m = copy.deepcopy(base).to(DEV)
        opt = torch.optim.Adam(m.parameters(), lr=5e-2, fused=True)
        nfa.config.set_made_train(mode)
        try:
            lps = []
            for _ in range(3):
                opt.zero_grad(set_to_none=True)
                z, logq = eps, torch.zeros(300, device=DEV)
                for f in m.flows:
                    z, ld = f(z)
                    logq = logq - ld
                (logq + 0.5 * (z ** 2).sum(1)).mean().backward()
                opt.step()
                with torch.no_grad():
                    lps.append(m.log_prob(xs).clone())          # one-pass inverse kernel on a cached pack: must follow the step
        finally:
            nfa.config.set_made_train(True)
        outs.append((lps, [p.detach().clone() for p in m.parameters()]))
    assert float((outs[0][0][0] - outs[0][0][2]).abs().max()) > 1e-3                      # the steps changed the model ...
    for a, b in zip(outs[0][0], outs[1][0]):
        assert float((a - b).abs().max()) < 2e-3 * max(1.0, float(b.abs().max()))        # ... the same way on both paths
    for a, b in zip(outs[0][1], outs[1][1]):
        assert float((a - b).abs().max()) < 2e-3


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
@pytest.mark.parametrize("C", [3, 12, 48, 60])
def test_inv1x1_lu_parametrisation_vjp_vs_autograd(nfa, dt, C):
    """nf_inv1x1_assemble + nf_inv1x1_lu_grads (autograd.Inv1x1WeightFn) against torch autograd through the reference's assembly
    (mixing.py:88-104: W = P (tril(L, -1) + I) (triu(U, 1) + diag(sign_S exp(log_S))), log|det| = sum log_S)."""
    from normflows_amd.autograd import Inv1x1WeightFn
    torch.manual_seed(C)
    conv = nfa.flows.Invertible1x1Conv(C, use_lu=True).to(dt).to(DEV)
    with torch.no_grad():
        conv.L.add_(0.3 * torch.randn_like(conv.L))
        conv.U.add_(0.3 * torch.randn_like(conv.U))
        conv.log_S.add_(0.3 * torch.randn_like(conv.log_S))
    cW, cl = torch.randn(C, C, device=DEV, dtype=dt), torch.randn((), device=DEV, dtype=dt)
    res = []
    for hand in (True, False):
        conv.zero_grad(set_to_none=True)
        if hand:
            W, l = Inv1x1WeightFn.apply(conv.P, conv.L, conv.U, conv.sign_S, conv.log_S)
        else:
            Lm = torch.tril(conv.L, diagonal=-1) + conv.eye
            Um = torch.triu(conv.U, diagonal=1) + torch.diag(conv.sign_S * torch.exp(conv.log_S))
            W, l = conv.P @ Lm @ Um, torch.sum(conv.log_S)
        ((W * cW).sum() + l * cl).backward()
        res.append([W.detach(), l.detach(), conv.L.grad.clone(), conv.U.grad.clone(), conv.log_S.grad.clone()])
    tol = 2e-5 if dt == torch.float32 else 1e-12
    for a, b in zip(res[0], res[1]):
        assert float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max()))
    # a P that is not a permutation matrix (nothing in mixing.py:88-104 needs it to be one): the assembly takes the product, not the gather
    with torch.no_grad():
        conv.P.copy_(torch.randn(C, C, device=DEV, dtype=dt) / C ** 0.5)
        W, _ = Inv1x1WeightFn.apply(conv.P, conv.L, conv.U, conv.sign_S, conv.log_S)
        Lm = torch.tril(conv.L, diagonal=-1) + conv.eye
        Um = torch.triu(conv.U, diagonal=1) + torch.diag(conv.sign_S * torch.exp(conv.log_S))
        Wr = conv.P @ Lm @ Um
    assert float((W - Wr).abs().max()) <= 10 * tol * max(1.0, float(Wr.abs().max()))


def test_training_kernels_random_shapes(nfa):
    """Seeded fuzz of the MADE / ResidualNet / conv-conditioner training paths over their whole envelope (features 2..128 resp. 9 Cin up
    to 252, hidden 1..512, 1..3 blocks, output multipliers 1..23, batches 1..400 incl. non-multiples of the 64-row tile and of the
    weight-gradient chunks) against float64 autograd through the library: every output and gradient to 3e-5 of its scale."""
    import copy
    rng = np.random.RandomState(4)

    def compare(net, x, go):
        res = []
        for n_, dt in ((net, torch.float32), (copy.deepcopy(net).double(), torch.float64)):
            n_.zero_grad(set_to_none=True)
            xx = x.detach().clone().to(dt).requires_grad_(True)
            out = n_(xx)
            out.backward(go.to(dt))
            res.append([out.detach(), xx.grad] + [p.grad for p in n_.parameters()])
        for a, b in zip(res[0], res[1]):
            assert float((a.double() - b).abs().max()) <= 3e-5 * max(float(b.abs().max()), 1e-6), (type(net).__name__, tuple(a.shape))
    for k in range(14):
        D, H, NB = int(rng.randint(2, 129)), int(rng.randint(1, 513)), int(rng.randint(1, 4))
        mult, B = int(rng.choice([1, 2, 3, 5, 23])), int(rng.randint(1, 401))
        torch.manual_seed(k)
        if k % 2 == 0:
            net = nfa.nets.MADE(D, H, num_blocks=NB, output_multiplier=mult)
            out_f = mult * D
        else:
            H = max(H, 129)                                    # (the ResidualNet route starts beyond 128 hidden units)
            out_f = int(rng.randint(1, 700))
            net = nfa.nets.ResidualNet(D, out_f, H, num_blocks=NB)
        with torch.no_grad():
            for p in net.parameters():
                p.add_(0.1 * torch.randn_like(p))
        net = net.to(DEV)
        compare(net, torch.randn(B, D, device=DEV), torch.randn(B, out_f, device=DEV))
    for k in range(6):
        Cin, hid, Cout = int(rng.randint(1, 29)), int(rng.randint(1, 513)), int(rng.randint(1, 49))
        if hid > 256:
            Cin = min(Cin, 14)                                 # (9 Cin <= 128 next to 512 hidden slots)
        B, H, W = int(rng.randint(1, 9)), int(rng.randint(1, 9)), int(rng.randint(1, 9))
        torch.manual_seed(100 + k)
        net = nfa.nets.ConvNet2d([Cin, hid, hid, Cout], [3, 1, 3], init_zeros=False).to(DEV)
        compare(net, torch.randn(B, Cin, H, W, device=DEV), torch.randn(B, Cout, H, W, device=DEV))



@pytest.mark.parametrize("onepass", [True, False])
@pytest.mark.parametrize("D,H,NB,B", [(20, 40, 2, 130), (64, 256, 2, 300), (128, 512, 2, 200), (7, 24, 3, 5), (17, 40, 1, 65), (33, 70, 2, 1000),
                                      (64, 252, 2, 77)])
def test_maf_density_direction_implicit_vs_d_pass_autograd(nfa, D, H, NB, B, onepass):
    """MaskedAffineAutoregressive.inverse under autograd (the reference's density direction: D sequential MADE passes,
    autoregressive.py:29-38): implicit differentiation (autograd.MafInverseFn) -- round 5: the linear system solved in ONE pass of
    nf_maf_solve_t on the transposed pack with the forward inverse's own ReLU masks (onepass); round 4: MADE chain sweeps until v stops
    changing -- against torch autograd through the D-pass loop itself (config.set_maf_implicit(False); each pass on the MADE training
    kernels): outputs 1e-4, every gradient 2e-4 of its scale; the sweep count stays at or below D (one pass: reported as 1)."""
    from normflows_amd.autograd import MafInverseFn
    nfa.config.set_maf_onepass(onepass)
    torch.manual_seed(D + H)
    layer = nfa.flows.MaskedAffineAutoregressive(D, H, num_blocks=NB)
    gen = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for p in layer.parameters():
            p.add_(0.05 * torch.randn(p.shape, generator=gen))
    layer = layer.to(DEV)
    z0 = torch.randn(B, D, generator=torch.Generator().manual_seed(1)).to(DEV)
    cx, cl = torch.randn(B, D, device=DEV), torch.randn(B, device=DEV)
    res = []
    for mode in (True, False):
        nfa.config.set_maf_implicit(mode)
        try:
            layer.zero_grad(set_to_none=True)
            z = z0.clone().requires_grad_(True)
            x, ld = layer.inverse(z)
            ((x * cx).sum() + (ld * cl).sum()).backward()
            res.append([x.detach(), ld.detach(), z.grad] + [p.grad.clone() for p in layer.parameters()])
        finally:
            nfa.config.set_maf_implicit(True)
    nfa.config.set_maf_onepass(True)
    assert (MafInverseFn.last_sweeps == 1) if onepass else (1 <= MafInverseFn.last_sweeps <= D)
    for k, (a, b) in enumerate(zip(res[0], res[1])):
        tol = 1e-4 if k < 2 else 2e-4
        assert float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max())), (k, float((a - b).abs().max()), float(b.abs().max()))


@pytest.mark.parametrize("kind,D,H,B,dtype", [("arnsf", 12, 40, 300, torch.float32), ("arnsf", 33, 64, 130, torch.float32),
                                              ("arnsf", 9, 24, 50, torch.float64), ("circular", 6, 32, 200, torch.float32),
                                              ("circular", 5, 16, 40, torch.float64), ("maf_ctx_free_f64", 10, 24, 60, torch.float64)])
def test_autoregressive_inverse_implicit_vs_d_pass_autograd(nfa, kind, D, H, B, dtype):
    """`Autoregressive.inverse` under autograd beyond MAF's one-pass kernels -- the autoregressive spline layer's sampling direction
    (neural_spline/autoregressive.py:94-134, wrapper.py:140-155), the circular variant (wrapper.py:158-235), a float64 MAF: implicit
    differentiation on the layer's own density-direction graph (autograd.ArInverseImplicitFn: one graph-free inverse, <= D + 1 backward
    sweeps of the net, one weight-gradient pass) against torch autograd through the reference's D recorded passes
    (config.set_ar_implicit(False)): outputs and every gradient to 2e-4 of scale in float32, 1e-9 in float64."""
    from normflows_amd.autograd import ArInverseImplicitFn
    torch.manual_seed(D * 7 + H)
    if kind == "arnsf":
        layer = nfa.flows.AutoregressiveRationalQuadraticSpline(D, 2, H)
        z0 = 1.5 * torch.randn(B, D)
    elif kind == "circular":
        layer = nfa.flows.CircularAutoregressiveRationalQuadraticSpline(D, 1, H, [1, 3],
                                                                        tail_bound=torch.tensor(([5.0, 3.14159, 4.0, 3.14159] + [5.0] * D)[:D]))
        z0 = torch.randn(B, D).clamp(-3.0, 3.0)
    else:
        layer = nfa.flows.MaskedAffineAutoregressive(D, H, num_blocks=2)
        z0 = torch.randn(B, D)
    gen = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for p in layer.parameters():
            p.add_(0.05 * torch.randn(p.shape, generator=gen))
    layer = layer.to(DEV).to(dtype)
    z0 = z0.to(DEV).to(dtype)
    cx, cl = torch.randn(B, D, device=DEV, dtype=dtype), torch.randn(B, device=DEV, dtype=dtype)
    # AutoregressiveRationalQuadraticSpline.forward = the sampling direction = the inner transform's inverse (D passes in the reference)
    run = (lambda zz: layer.inverse(zz)) if kind.startswith("maf") else (lambda zz: layer.forward(zz))
    res = []
    ArInverseImplicitFn.last_sweeps = 0
    for mode in (True, False):
        nfa.config.set_ar_implicit(mode)
        try:
            layer.zero_grad(set_to_none=True)
            z = z0.clone().requires_grad_(True)
            x, ld = run(z)
            ((x * cx).sum() + (ld * cl).sum()).backward()
            res.append([x.detach(), ld.detach(), z.grad] + [p.grad.clone() for p in layer.parameters()])
        finally:
            nfa.config.set_ar_implicit(True)
    assert 1 <= ArInverseImplicitFn.last_sweeps <= D + 1
    tol = 2e-4 if dtype == torch.float32 else 1e-9
    for k, (a, b) in enumerate(zip(res[0], res[1])):
        assert float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max())), (k, float((a - b).abs().max()), float(b.abs().max()))
    # one cotangent absent (set_materialize_grads(False): None reaches the backward)
    for which in ("x", "ld"):
        out = []
        for mode in (True, False):
            nfa.config.set_ar_implicit(mode)
            try:
                layer.zero_grad(set_to_none=True)
                z = z0.clone().requires_grad_(True)
                x, ld = run(z)
                ((x * cx).sum() if which == "x" else (ld * cl).sum()).backward()
                out.append([z.grad] + [p.grad.clone() for p in layer.parameters()])
            finally:
                nfa.config.set_ar_implicit(True)
        for k, (a, b) in enumerate(zip(out[0], out[1])):
            assert float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max())), (which, k, float((a - b).abs().max()))


def test_training_step_captures_into_one_graph(nfa):
    """forward + backward of the hand-written training paths (MADE Function, conv conditioner, 1x1-conv LU parametrisation, couplings,
    ActNorm) records into ONE hipGraph with PyTorch's whole-network capture recipe -- no host synchronisation, no host-side packing in
    the step -- and a replay reproduces the eager step's gradients bit for bit (Glow config 4: 71 -> 57 ms, tools/train_graph_probe.py)."""
    torch.manual_seed(0)
    fl = [[nfa.flows.GlowBlock(12, 64, split_mode="channel", scale=True) for _ in range(2)] + [nfa.flows.Squeeze()]]
    glow = nfa.MultiscaleFlow([nfa.distributions.DiagGaussian((12, 4, 4))], fl, [], class_cond=False).to(DEV)
    ximg = torch.rand(16, 3, 8, 8, device=DEV)
    with torch.no_grad():
        glow.log_prob(ximg)                       # ActNorm's data-dependent init
    maf = nfa.NormalizingFlow(nfa.distributions.DiagGaussian(12, trainable=False),
                              [nfa.flows.MaskedAffineAutoregressive(12, 40, num_blocks=2) for _ in range(2)]).to(DEV)
    eps = torch.randn(200, 12, device=DEV)

    def maf_loss(mm, e):
        z, logq = e, torch.zeros(e.shape[0], device=DEV)
        for f in mm.flows:
            z, ld = f(z)
            logq = logq - ld
        return (logq + 0.5 * (z ** 2).sum(1)).mean()
    # round 5: the DENSITY direction of the MAF too (forward_kld = flow.inverse under autograd): the implicit backward is one
    # nf_maf_solve_t launch per layer -- round 4's sweeps read a flag back every other sweep and could not be captured
    maf_d = nfa.NormalizingFlow(nfa.distributions.DiagGaussian(12, trainable=False),
                                [nfa.flows.MaskedAffineAutoregressive(12, 40, num_blocks=2) for _ in range(2)]).to(DEV)
    xd = torch.randn(200, 12, device=DEV)
    # round 6: ... and at a shape / batch that takes the in-place weight gradients (nf_made_wgrad_pos: 128 positions, 256 rows), MADE's
    # output from the inverse pass (nf_maf_inverse_h_train) and the host-table launchers (nf_maf_solve_t_tri)
    maf_p = nfa.NormalizingFlow(nfa.distributions.DiagGaussian(40, trainable=False),
                                [nfa.flows.MaskedAffineAutoregressive(40, 100, num_blocks=2) for _ in range(2)]).to(DEV)
    xp = torch.randn(256, 40, device=DEV)
    from normflows_amd import ops as _ops
    seen = []
    real_pos = _ops.made_wgrad_pos
    _ops.made_wgrad_pos = lambda *a, **k: (seen.append(1), real_pos(*a, **k))[1]
    for m, x, lossfn in ((glow, ximg, lambda mm, xx: mm.forward_kld(xx)), (maf, eps, maf_loss),
                         (maf_d, xd, lambda mm, xx: mm.forward_kld(xx)), (maf_p, xp, lambda mm, xx: mm.forward_kld(xx))):
        def step():
            m.zero_grad(set_to_none=True)
            lossfn(m, x).backward()
        step()
        eager = [p.grad.clone() for p in m.parameters()]
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                step()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        m.zero_grad(set_to_none=True)
        with torch.cuda.graph(g):
            lossfn(m, x).backward()
        g.replay()
        g.replay()
        torch.cuda.synchronize()
        assert all(torch.equal(a, p.grad) for a, p in zip(eager, m.parameters()))
    _ops.made_wgrad_pos = real_pos
    assert len(seen) >= 2 * 4           # (maf_p's two layers, in the eager, warm-up and captured steps)


def test_glow_parameter_gradient_launches_on_the_side_stream(nfa):
    """config.train_leaf_async (round 6): the launches of a GlowBlock's backward that only produce parameter gradients run on a side
    stream forked inside the backward functions and joined when the pass ends.  Same bits as the one-stream step -- eager, repeated
    (a race would show up as a difference between repeats), with gradient accumulation (an existing .grad: the Functions must stay
    on the current stream) and recorded into one hipGraph."""
    torch.manual_seed(3)
    fl = [[nfa.flows.GlowBlock(12, 256, split_mode="channel", scale=True) for _ in range(4)] + [nfa.flows.Squeeze()]]
    m = nfa.MultiscaleFlow([nfa.distributions.DiagGaussian((12, 16, 16))], fl, [], class_cond=False).to(DEV)
    x = torch.rand(64, 3, 32, 32, device=DEV)
    with torch.no_grad():
        m.log_prob(x)
        for p in m.parameters():
            p.add_(0.02 * torch.randn_like(p))

    def step(zero=True):
        if zero:
            m.zero_grad(set_to_none=True)
        m.forward_kld(x).backward()
    from importlib import import_module
    ss = import_module("normflows_amd._sidestream")
    orig = ss.fork
    try:
        nfa.config.set_train_leaf_async(False)
        step()
        ref = [p.grad.clone() for p in m.parameters()]
        step(zero=False)
        ref2 = [p.grad.clone() for p in m.parameters()]
        nfa.config.set_train_leaf_async(True)
        forks = []
        ss.fork = lambda *a, **k: (forks.append(1), orig(*a, **k))[1]
        for _ in range(3):
            step()
            torch.cuda.synchronize()
            assert all(torch.equal(a, p.grad) for a, p in zip(ref, m.parameters()))
        # conditioner + 1x1 weight per block and step (the LU factors' gradients of the level are one launch, config.glow_weights_batched:
        # it joins first and runs on the current stream)
        assert len(forks) == 3 * 4 * 2, len(forks)
        n0 = len(forks)
        step(zero=False)                                  # accumulation into an existing .grad: nothing may leave the current stream
        torch.cuda.synchronize()
        assert len(forks) == n0 + 4       # (the 1x1 convolution's weight gradient still forks: its consumer, the LU factors' backward, joins first)
        ss.fork = orig
        assert all(torch.equal(a, p.grad) for a, p in zip(ref2, m.parameters()))
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            step()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        m.zero_grad(set_to_none=True)
        with torch.cuda.graph(g):
            m.forward_kld(x).backward()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        assert all(torch.equal(a, p.grad) for a, p in zip(ref, m.parameters()))
    finally:
        ss.fork = orig
        nfa.config.set_train_leaf_async(False)


def test_glow_level_assembles_its_1x1_matrices_in_one_launch(nfa):
    """config.glow_weights_batched (round 6, late): in the density direction under autograd a level's Invertible1x1Convs get their
    matrices from ONE nf_inv1x1_assemble_multi launch before the first block runs and their LU factors' gradients from ONE
    nf_inv1x1_lu_grads_multi launch once every block's gW exists (autograd.Inv1x1WeightsFn), and its conditioners their packed weight
    streams from ONE nf_pack_gather_batch launch (nets.prefetch_train_packs) -- the same kernel bodies per layer as the
    per-block launches: loss and every gradient bit for bit; nothing is left waiting in the modules afterwards; a second backward
    through the same graph is refused like any once-differentiable Function's."""
    from normflows_amd import ops
    torch.manual_seed(5)
    L_, K_ = 2, 3
    q0, merges, flows = [], [], []
    for i in range(L_):
        fl = [nfa.flows.GlowBlock(3 * 2 ** (L_ + 1 - i), 64, split_mode="channel", scale=True) for _ in range(K_)] + [nfa.flows.Squeeze()]
        flows += [fl]
        if i > 0:
            merges += [nfa.flows.Merge()]
            latent = (3 * 2 ** (L_ - i), 16 // 2 ** (L_ - i), 16 // 2 ** (L_ - i))
        else:
            latent = (3 * 2 ** (L_ + 1), 16 // 2 ** L_, 16 // 2 ** L_)
        q0 += [nfa.distributions.DiagGaussian(latent)]
    m = nfa.MultiscaleFlow(q0, flows, merges, class_cond=False).to(DEV)
    x = torch.rand(32, 3, 16, 16, device=DEV)
    with torch.no_grad():
        m.log_prob(x)
        for p in m.parameters():
            p.add_(0.02 * torch.randn_like(p))
    res, counts = [], []
    names = ("inv1x1_assemble", "inv1x1_lu_grads", "inv1x1_assemble_multi", "inv1x1_lu_grads_multi", "pack_gather", "pack_gather_batch")
    real = tuple(getattr(ops, k) for k in names)
    try:
        for mode in (False, True):
            nfa.config.set_glow_weights_batched(mode)
            n = [0] * len(names)

            def spy(k):
                def f(*a, **kw):
                    n[k] += 1
                    return real[k](*a, **kw)
                return f
            for k, name in enumerate(names):
                setattr(ops, name, spy(k))
            m.zero_grad(set_to_none=True)
            loss = m.forward_kld(x)
            loss.backward()
            counts.append(tuple(n))
            res.append([loss.detach().clone()] + [p.grad.clone() for p in m.parameters()])
    finally:
        for name, f in zip(names, real):
            setattr(ops, name, f)
        nfa.config.set_glow_weights_batched(True)
    # (the conditioners' packed weight streams of the step likewise: one gather launch per level instead of one per block)
    assert counts == [(L_ * K_, L_ * K_, 0, 0, L_ * K_, 0), (0, 0, L_, L_, 0, L_)], counts
    assert all(torch.equal(a, b) for a, b in zip(res[0], res[1]))
    assert all("_w_prefetch" not in mod.__dict__ for mod in m.modules())


def test_glow_level_folds_its_log_dets_in_one_launch(nfa):
    """config.lazy_logdet (round 6, late): under autograd the `log_q += log_det` statements of a level's layers are collected
    (flows/affine.lazy_ld) and applied by ONE nf_ld_fold_multi launch in the same order -- loss and gradients bit for bit, with a
    mixed-sign list too."""
    from normflows_amd import ops
    torch.manual_seed(6)
    ld0 = torch.randn(300, device=DEV)
    terms = [torch.randn(300, device=DEV) for _ in range(130)]          # (> 120: two launches)
    neg = [bool(i % 3 == 1) for i in range(130)]
    want = ld0.clone()
    for t, s_ in zip(terms, neg):
        want = want - t if s_ else want + t
    got = ops.ld_fold_multi(ld0.clone(), terms, neg)
    assert torch.equal(got, want)
    fl = [[nfa.flows.GlowBlock(12, 64, split_mode="channel", scale=True) for _ in range(3)] + [nfa.flows.Squeeze()]]
    m = nfa.MultiscaleFlow([nfa.distributions.DiagGaussian((12, 8, 8))], fl, [], class_cond=False).to(DEV)
    x = torch.rand(32, 3, 16, 16, device=DEV)
    with torch.no_grad():
        m.log_prob(x)
        for p in m.parameters():
            p.add_(0.02 * torch.randn_like(p))
    res, calls = [], []
    real = ops.ld_fold_multi
    try:
        for mode in (False, True):
            nfa.config.set_lazy_logdet(mode)
            n = [0]

            def spy(*a, **k):
                n[0] += 1
                return real(*a, **k)
            ops.ld_fold_multi = spy
            m.zero_grad(set_to_none=True)
            xx = x.clone().requires_grad_(True)
            loss = m.forward_kld(xx)
            loss.backward()
            calls.append(n[0])
            res.append([loss.detach().clone(), xx.grad.clone()] + [p.grad.clone() for p in m.parameters()])
    finally:
        ops.ld_fold_multi = real
        nfa.config.set_lazy_logdet(True)
    assert calls == [0, 1], calls
    assert all(torch.equal(a, b) for a, b in zip(res[0], res[1]))


def test_glow_training_run_is_bit_identical_with_the_round6_switches_off(nfa):
    """Three Adam steps of a two-level Glow (16x16 level at 32 768 pixel rows: the 128-row tiles apply) with every switch of the round's
    last session ON (batched 1x1 matrices / LU-factor gradients / conditioner packs, lazy log-dets, 128-row tiles) and with all of them
    OFF: every one of them re-orders launches, none re-orders arithmetic -- the loss sequence and the final parameters are bit for bit
    the same."""
    def run(on):
        nfa.config.set_glow_weights_batched(on)
        nfa.config.set_lazy_logdet(on)
        nfa.config.set_made_tr128(on)
        torch.manual_seed(13)
        L_, K_ = 2, 2
        q0, merges, flows = [], [], []
        for i in range(L_):
            fl = [nfa.flows.GlowBlock(3 * 2 ** (L_ + 1 - i), 256, split_mode="channel", scale=True) for _ in range(K_)] + [nfa.flows.Squeeze()]
            flows += [fl]
            if i > 0:
                merges += [nfa.flows.Merge()]
                latent = (3 * 2 ** (L_ - i), 32 // 2 ** (L_ - i), 32 // 2 ** (L_ - i))
            else:
                latent = (3 * 2 ** (L_ + 1), 32 // 2 ** L_, 32 // 2 ** L_)
            q0 += [nfa.distributions.DiagGaussian(latent)]
        m = nfa.MultiscaleFlow(q0, flows, merges, class_cond=False).to(DEV)
        x = torch.rand(128, 3, 32, 32, generator=torch.Generator().manual_seed(1)).to(DEV)
        with torch.no_grad():
            m.log_prob(x)
        opt = torch.optim.Adam(m.parameters(), lr=1e-3)
        losses = []
        for _ in range(3):
            opt.zero_grad(set_to_none=True)
            loss = m.forward_kld(x)
            loss.backward()
            opt.step()
            losses.append(loss.detach().clone())
        return losses, [p.detach().clone() for p in m.parameters()]
    prev = nfa.config.set_made_tr128(True)
    try:
        a = run(True)
        b = run(False)
    finally:
        nfa.config.set_glow_weights_batched(True)
        nfa.config.set_lazy_logdet(True)
        nfa.config.set_made_tr128(prev)
    assert all(torch.isfinite(l) for l in a[0]) and not torch.equal(a[0][0], a[0][2])          # (the parameters moved)
    assert all(torch.equal(x_, y_) for x_, y_ in zip(a[0], b[0]))
    assert all(torch.equal(x_, y_) for x_, y_ in zip(a[1], b[1]))


def test_maf_one_pass_backward_on_format0_and_format1_packs(nfa):
    """The one-pass implicit backward with the forward on the format-1 pack (default: fast inverse kernel, masks in its positions,
    nf_maf_inverse_h_tri_bits) and on the format-0 pack (config.set_maf_tri(False): nf_maf_inverse_h_bits): the two position
    conventions give the same gradients (summation order differs: 2e-5), at config 5's layer shape (tile 0 = kind 2) and at a shape
    whose tiles are all generic."""
    for D, H in ((128, 512), (40, 100)):
        torch.manual_seed(D)
        layer = nfa.flows.MaskedAffineAutoregressive(D, H, num_blocks=2)
        with torch.no_grad():
            for p in layer.parameters():
                p.add_(0.03 * torch.randn_like(p))
        layer = layer.to(DEV)
        z0 = torch.randn(300, D, device=DEV)
        cx, cl = torch.randn(300, D, device=DEV), torch.randn(300, device=DEV)
        res = []
        try:
            for tri in (True, False):
                nfa.config.set_maf_tri(tri)
                layer.zero_grad(set_to_none=True)
                z = z0.clone().requires_grad_(True)
                x, ld = layer.inverse(z)
                ((x * cx).sum() + (ld * cl).sum()).backward()
                res.append([x.detach(), ld.detach(), z.grad] + [p.grad.clone() for p in layer.parameters()])
        finally:
            nfa.config.set_maf_tri(True)
        for a, b in zip(res[0], res[1]):
            assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max())), (D, float((a - b).abs().max()))


@pytest.mark.parametrize("D,H,NB,B", [(128, 512, 2, 300), (40, 100, 2, 64), (20, 40, 2, 130), (64, 252, 2, 77), (17, 40, 1, 65), (33, 70, 3, 31),
                                      (5, 12, 2, 1), (20, 40, 2, 33)])
def test_maf_weight_gradients_from_the_solve_scratch(nfa, D, H, NB, B):
    """The one-pass implicit backward hands the weight-gradient launch MADE's hidden gradients straight from the SOLVE's activation scratch
    (nf_maf_scratch_rows through maf_pack.solve_t_gradient_columns: the solve finalises every unit of the transposed network once from
    final values = the input-gradient chain at the solution) instead of running nf_made_backward once more.

    (1) G from the scratch against nf_made_backward's G for the same cotangent, ROW by row (a row of G depends on that sample alone): equal
    to 2e-5 of scale on every row whose ReLU masks agree between the two float32 passes that produced them -- the inverse kernel's own
    pass (the solve's masks) and nf_made_forward_train's (the chain's): a pre-activation within rounding of zero may fall on either side
    (about one unit in a million) and then that ROW differs by a finite amount, so at most 1 % of the rows (+ 1) may; rows
    beyond B of the padded tensors are zero.  (2) End to end: every gradient against the path with the extra chain pass
    (config.set_maf_solve_grads(False)): 2e-5 of scale where no mask differs, summed absolute error <= 1e-3 of the summed gradient
    otherwise; the chain launch is gone, and so is the MADE forward at the solution (`save` = the inverse pass's own scratch through
    the same rearrangement, the parameters from its last hidden tensor by one product with the masked final weight)."""
    from normflows_amd import ops
    torch.manual_seed(D + H)
    layer = nfa.flows.MaskedAffineAutoregressive(D, H, num_blocks=NB)
    gen = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for p in layer.parameters():
            p.add_(0.03 * torch.randn(p.shape, generator=gen))
    layer = layer.to(DEV)
    z0 = torch.randn(B, D, device=DEV)
    cx, cl = torch.randn(B, D, device=DEV), torch.randn(B, device=DEV)
    # (1) the two G tensors
    inv, fwd, bwd = layer._implicit_packs(DEV)
    assert isinstance(inv, dict) and inv["gcols"] is not None
    x, _, bits = ops.maf_inverse_bits(z0, inv["blob"], inv["table"], inv["hp"], inv["nb"], inv["tiles"], table_host=inv.get("table_host"))
    prm, save, tbits = ops.made_forward_train(x, fwd[0], fwd[1], fwd[2], 2 * D, bwd["NB"])
    v, scratch = ops.maf_solve_t(x, prm, cx, cl, bits, inv["tblob"], inv["ttable"], inv["hp"], inv["nb"], return_scratch=True)
    G1 = ops.maf_scratch_rows(scratch, inv["gcols"], B, inv["nb"], inv["hp"], sign=-1.0, reverse_layers=True)
    _, gp = ops.maf_affine_bwd(x, prm, -v, -cl, 0)
    _, G2 = ops.made_backward(gp, tbits, bwd["blob"], bwd["table"], D, bwd["Hp"], bwd["NB"])
    assert G1.shape == G2.shape and float(G1[:, B:].abs().max() if G1.shape[1] > B else 0.0) == 0.0
    scale = max(1.0, float(G2.abs().max()))
    row_err = (G1[:, :B] - G2[:, :B]).abs().amax(dim=(0, 2))
    flipped = int((row_err > 2e-5 * scale).sum())
    assert flipped <= B // 100 + 1, (flipped, B, float(row_err.max()), scale)
    # (2) end to end
    res, chains = [], []
    real, real_f = ops.made_backward, ops.made_forward_train
    try:
        for mode in (True, False):
            nfa.config.set_maf_solve_grads(mode)
            n = [0, 0]

            def spy(*a, **k):
                n[0] += 1
                return real(*a, **k)

            def spy_f(*a, **k):
                n[1] += 1
                return real_f(*a, **k)
            ops.made_backward, ops.made_forward_train = spy, spy_f
            layer.zero_grad(set_to_none=True)
            z = z0.clone().requires_grad_(True)
            xx, ld = layer.inverse(z)
            ((xx * cx).sum() + (ld * cl).sum()).backward()
            chains.append(tuple(n))
            res.append([z.grad] + [p.grad.clone() for p in layer.parameters()])
    finally:
        ops.made_backward, ops.made_forward_train = real, real_f
        nfa.config.set_maf_solve_grads(True)
    assert chains == [(0, 0), (1, 1)], chains          # (chain passes, MADE forward passes) inside the backward
    for k, (a, b) in enumerate(zip(res[0], res[1])):
        err = (a - b).abs()
        if flipped == 0:
            assert float(err.max()) <= 2e-5 * max(1.0, float(b.abs().max())), (k, float(err.max()), float(b.abs().max()))
        else:
            assert float(err.sum()) <= 1e-3 * float(b.abs().sum()), (k, flipped, float(err.sum()), float(b.abs().sum()))


@pytest.mark.parametrize("D,H,NB,B", [(128, 512, 2, 320), (128, 512, 2, 4096), (40, 100, 2, 64), (64, 252, 2, 128), (33, 120, 3, 192),
                                      (24, 100, 1, 64), (128, 512, 2, 300)])
def test_maf_weight_gradients_read_the_scratches_in_place(nfa, D, H, NB, B):
    """Round 6 (nf_made_wgrad_pos, maf_pack.position_wgrad_tables): the weight-gradient launch of the implicit backward contracts the
    solve's and the inverse pass's activation scratches where they are -- problems, mask-non-zero tiles and scatter maps over scratch
    positions, LDS-DMA requests of [two k-groups][half][16 samples][4] -- instead of over two row-major rearrangements
    (config.set_maf_wgrad_in_place(False): nf_maf_scratch_rows x 2 + nf_made_wgrad).  Same operands, same order of summation over the
    rows inside a chunk: the two paths agree to float32 rounding of the chunk sums (the chunk length follows the tile count, which may
    differ between slot and position space).  A batch that is not a multiple of 64 rows, or a position count that is not a multiple of
    128, keeps the rearrangement."""
    from normflows_amd import ops
    torch.manual_seed(D + H + B)
    layer = nfa.flows.MaskedAffineAutoregressive(D, H, num_blocks=NB)
    gen = torch.Generator().manual_seed(5)
    with torch.no_grad():
        for p in layer.parameters():
            p.add_(0.03 * torch.randn(p.shape, generator=gen))
    layer = layer.to(DEV)
    z0 = torch.randn(B, D, device=DEV)
    cx, cl = torch.randn(B, D, device=DEV), torch.randn(B, device=DEV)
    inv = layer._implicit_packs(DEV)[0]
    assert isinstance(inv, dict)
    direct = inv["pw"] is not None and B % 64 == 0 and inv["pw"]["positions"] == inv["hp"]
    if (D, H, B) in ((128, 512, 320), (128, 512, 4096), (40, 100, 64)):
        assert direct
    if B == 300:
        assert not direct
    res, calls = [], []
    real_r, real_p = ops.maf_scratch_rows, ops.made_wgrad_pos
    try:
        for mode in (True, False):
            nfa.config.set_maf_wgrad_in_place(mode)
            n = [0, 0]

            def spy_r(*a, **k):
                n[0] += 1
                return real_r(*a, **k)

            def spy_p(*a, **k):
                n[1] += 1
                return real_p(*a, **k)
            ops.maf_scratch_rows, ops.made_wgrad_pos = spy_r, spy_p
            layer.zero_grad(set_to_none=True)
            z = z0.clone().requires_grad_(True)
            xx, ld = layer.inverse(z)
            ((xx * cx).sum() + (ld * cl).sum()).backward()
            calls.append(tuple(n))
            res.append([z.grad] + [p.grad.clone() for p in layer.parameters()])
    finally:
        ops.maf_scratch_rows, ops.made_wgrad_pos = real_r, real_p
        nfa.config.set_maf_wgrad_in_place(True)
    assert calls == [((0, 1) if direct else (2, 0)), (2, 0)], calls
    for k, (a, b) in enumerate(zip(res[0], res[1])):
        assert a.shape == b.shape and torch.isfinite(a).all()
        assert float((a - b).abs().max()) <= 2e-6 * max(1.0, float(b.abs().max())), (k, float((a - b).abs().max()), float(b.abs().max()))
    names = [n_ for n_, _ in layer.named_parameters()]
    for n_, a in zip(names, res[0][1:]):       # the masked entries of weight.grad are exactly zero as in the reference (made.py:80-81)
        if n_.endswith("weight"):
            mod = layer.autoregressive_net
            for part in n_.split(".")[1:-1] if n_.startswith("autoregressive_net.") else n_.split(".")[:-1]:
                mod = getattr(mod, part) if not part.isdigit() else mod[int(part)]
            assert float(a[mod.mask.to(DEV) == 0].abs().max() if (mod.mask == 0).any() else 0.0) == 0.0, n_


@pytest.mark.parametrize("D,H,NB,B", [(128, 512, 2, 2048), (64, 252, 2, 640), (72, 284, 1, 1000), (24, 92, 3, 500), (40, 100, 2, 300)])
def test_maf_solve_regular8_tiles_are_bit_identical_to_the_generic_part(nfa, D, H, NB, B):
    """nf_maf_solve_t_tri (round 6, late): the tiles a format-1 transposed pack marks regular-8 run the statically unrolled sequential
    part (tf_step: compile-time targets / rows / mask bits, dot products over the register quads that can hold a final unit) -- the
    generic part's accumulators in the generic part's order with exact-zero terms left out: v and every value of the activation scratch
    (= MADE's hidden gradients for the weight-gradient launch) bit for bit; (40, 100) has no such tile and takes the generic launch."""
    from normflows_amd import ops
    torch.manual_seed(D + B)
    layer = nfa.flows.MaskedAffineAutoregressive(D, H, num_blocks=NB)
    gen = torch.Generator().manual_seed(9)
    with torch.no_grad():
        for p in layer.parameters():
            p.add_(0.03 * torch.randn(p.shape, generator=gen))
    layer = layer.to(DEV)
    inv = layer._implicit_packs(DEV)[0]
    th = inv["ttable_host"]
    nfast = int(sum(int(th[8 + 24 * t + 21]) for t in range(int(th[4]))))
    assert (nfast > 0) == ((D, H) != (40, 100))
    z = torch.randn(B, D, device=DEV)
    cx, cl = torch.randn(B, D, device=DEV), torch.randn(B, device=DEV)
    x, _, bits, _, prm = ops.maf_inverse_bits(z, inv["blob"], inv["table"], inv["hp"], inv["nb"], inv["tiles"],
                                              table_host=inv.get("table_host"), return_scratch=True, want_params=True)
    v0, s0 = ops.maf_solve_t(x, prm, cx, cl, bits, inv["tblob"], inv["ttable"], inv["hp"], inv["nb"], return_scratch=True)
    v1, s1 = ops.maf_solve_t(x, prm, cx, cl, bits, inv["tblob"], inv["ttable"], inv["hp"], inv["nb"], return_scratch=True, table_host=th)
    assert torch.isfinite(v1).all() and torch.equal(v0, v1)
    nS = B // 32 * 32 * (2 * NB + 1) * inv["hp"]
    assert torch.equal(s0[:nS], s1[:nS])
    try:       # the switch
        nfa.config.set_maf_solve_fast(False)
        v2 = ops.maf_solve_t(x, prm, cx, cl, bits, inv["tblob"], inv["ttable"], inv["hp"], inv["nb"], table_host=th)
    finally:
        nfa.config.set_maf_solve_fast(True)
    assert torch.equal(v0, v2)


@pytest.mark.parametrize("which", ["x_only", "ld_only"])
def test_maf_implicit_backward_with_one_cotangent_absent(nfa, which):
    """The implicit backward when the loss sees only the outputs or only the log-det (the other cotangent arrives as None), and with
    the optional tolerance stop: same gradients as autograd through the D-pass loop."""
    torch.manual_seed(4)
    layer = nfa.flows.MaskedAffineAutoregressive(24, 64, num_blocks=2)
    with torch.no_grad():
        for p in layer.parameters():
            p.add_(0.05 * torch.randn_like(p))
    layer = layer.to(DEV)
    z0 = torch.randn(150, 24, device=DEV)
    c = torch.randn(150, 24, device=DEV)
    res = []
    for mode, rtol in ((True, 0.0), (True, 1e-6), (False, 0.0)):
        nfa.config.set_maf_implicit(mode, rtol=rtol)
        try:
            layer.zero_grad(set_to_none=True)
            z = z0.clone().requires_grad_(True)
            x, ld = layer.inverse(z)
            ((x * c).sum() if which == "x_only" else (ld * c[:, 0]).sum()).backward()
            res.append([z.grad] + [p.grad.clone() for p in layer.parameters()])
        finally:
            nfa.config.set_maf_implicit(True, rtol=0.0)
    for r in res[:2]:
        for a, b in zip(r, res[2]):
            assert float((a - b).abs().max()) <= 2e-4 * max(1.0, float(b.abs().max()))


def test_higher_order_gradients_context_through_maf_inverse(nfa):
    """the implicit Functions behind Autoregressive.inverse are first-order only; config.higher_order_gradients() routes the
    density direction of a MAF layer through the D recorded passes on torch modules, so a gradient penalty (create_graph=True, then a
    second backward) works -- and outside the context the same request fails loudly instead of returning a wrong value."""
    torch.manual_seed(0)
    layer = nfa.flows.MaskedAffineAutoregressive(6, 16, num_blocks=1).to(DEV)
    z = torch.randn(64, 6, device=DEV, requires_grad=True)
    with nfa.config.higher_order_gradients():
        x, ld = layer.inverse(z)
        (gz,) = torch.autograd.grad((x.pow(2).sum() + ld.sum()), z, create_graph=True)
        gz.pow(2).sum().backward()
    assert all(p_.grad is not None and torch.isfinite(p_.grad).all() for p_ in layer.parameters())
    assert nfa.config.ar_implicit and nfa.config.maf_implicit and nfa.config.made_train
    layer.zero_grad(set_to_none=True)
    x, ld = layer.inverse(z)
    with pytest.raises(RuntimeError):
        (gz,) = torch.autograd.grad((x.pow(2).sum() + ld.sum()), z, create_graph=True)
        gz.pow(2).sum().backward()
