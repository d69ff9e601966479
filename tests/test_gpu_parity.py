"""GPU parity tests (run with -m gpu on an MI355X): every kernel through the C ABI (normflows_amd.ops / layer
classes) against (a) the golden vectors produced by the real reference and (b) the CPU oracle on seeded inputs,
plus size-independent properties at the benchmark's full size (round trip, sample/log_prob consistency)."""
import os

import numpy as np
import pytest
import torch

from conftest import TOL, assert_close, golden_state, ld_tol, load_golden

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _inference_mode():
    """Parity tests exercise the inference kernels (fused paths included); the training path has its own module."""
    with torch.no_grad():
        yield


@pytest.fixture(scope="module")
def nfa():
    import normflows_amd
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    # fail loudly if the native library is not the one in-tree
    assert normflows_amd.native_library_path().endswith("normalizing-flows_amd/lib/libnf_mi355x.so")
    normflows_amd._lib.lib()
    return normflows_amd


def T(a, dev=DEV):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def N(t):
    return t.detach().cpu().numpy()


def load_layer(layer, state, dtype):
    layer.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in state.items()}, strict=True)
    return layer.to(dtype).to(DEV)


# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("K", [2, 8, 10, 16])
def test_spline_kernel_vs_reference(nfa, K, tag):
    g = load_golden("spline_K%d_%s" % (K, tag))
    tol = TOL[g["w"].dtype]
    t10 = dict(rtol=tol["rtol"] * 10, atol=tol["atol"] * 10)
    tld = (ld_tol(g["w"].dtype), ld_tol(g["w"].dtype, True))     # closed form | root finding (conftest.ld_tol)
    y, lad = nfa.ops.rqs_spline(T(g["x01"]), T(g["w"]), T(g["h"]), T(g["d_none"]), inverse=False, tails=None)
    assert_close(N(y), g["y01"], what="y01", **t10)
    assert_close(N(lad), g["lad01"], what="lad01", **tld[0])
    y, lad = nfa.ops.rqs_spline(T(g["y01"]), T(g["w"]), T(g["h"]), T(g["d_none"]), inverse=True, tails=None)
    assert_close(N(y), g["x01_inv"], what="x01_inv", **t10)
    assert_close(N(lad), g["lad01_inv"], what="lad01_inv", **tld[1])
    for tails, dkey, bound, keys in (("linear", "d_lin", 3.0, ("yl", "ladl", "yl_inv", "ladl_inv")),
                                     ("circular", "d_cir", 2.5, ("yc", "ladc", "yc_inv", "ladc_inv"))):
        for inv in (False, True):
            y, lad = nfa.ops.rqs_spline(T(g["xl"]), T(g["w"]), T(g["h"]), T(g[dkey]), inverse=inv, tails=tails,
                                        tail_bound=bound)
            assert_close(N(y), g[keys[2 * inv]], what=keys[2 * inv], **t10)
            assert_close(N(lad), g[keys[2 * inv + 1]], what=keys[2 * inv + 1], **tld[int(inv)])


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_spline_on_knots_and_edges(nfa, oracle, tag):
    g = load_golden("spline_knots_" + tag)
    tol = TOL[g["w"].dtype]
    y, lad = nfa.ops.rqs_spline(T(g["x"]), T(g["w"]), T(g["h"]), T(g["d"]), tails="linear", tail_bound=3.0)
    assert_close(N(y), g["y"], what="y", **tol)
    assert_close(N(lad), g["lad"], what="lad", rtol=tol["rtol"] * 10, atol=tol["atol"] * 10)
    # edge semantics: +-3 inside, nextafter(3) outside, NaN / +-inf pass through with logabsdet exactly 0
    K = 8
    rng = np.random.default_rng(0)
    dt = g["w"].dtype
    w, h, d = (rng.standard_normal((7, n)).astype(dt) for n in (K, K, K - 1))
    x = np.array([3.0, -3.0, np.nextafter(np.float32(3), np.float32(4)), np.nan, np.inf, -np.inf, 0.0], dt)
    y, lad = nfa.ops.rqs_spline(T(x), T(w), T(h), T(d), tails="linear", tail_bound=3.0)
    y, lad = N(y), N(lad)
    yo, lo = oracle.rqs_spline(x, w, h, d, tails="linear", tail_bound=3.0)
    assert_close(y, yo, what="edge y", rtol=1e-5, atol=1e-5)
    assert y[2] == x[2] and lad[2] == 0.0
    assert np.isnan(y[3]) and lad[3] == 0.0 and y[4] == np.inf and lad[4] == 0.0 and y[5] == -np.inf and lad[5] == 0.0


def test_spline_empty_and_ragged(nfa):
    K = 8
    e = torch.empty(0, device=DEV)
    y, lad = nfa.ops.rqs_spline(e, torch.empty(0, K, device=DEV), torch.empty(0, K, device=DEV),
                                torch.empty(0, K - 1, device=DEV), tails="linear", tail_bound=3.0)
    assert y.numel() == 0 and lad.numel() == 0
    # min bin width too large for K -> ValueError like utils/splines.py:121-124
    with pytest.raises(ValueError):
        nfa.ops.rqs_spline(torch.zeros(3, device=DEV), torch.zeros(3, K, device=DEV), torch.zeros(3, K, device=DEV),
                           torch.zeros(3, K - 1, device=DEV), tails="linear", tail_bound=3.0, min_bin_width=0.2)


@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("d", [2, 5, 7, 64])
def test_coupled_rqs_layer_vs_reference(nfa, d, tag):
    g = load_golden("crqs_d%d_%s" % (d, tag))
    dt = torch.float32 if tag == "f32" else torch.float64
    tol = TOL[g["x"].dtype]
    K, hidden = int(g["K"]), int(g["hidden"])
    layer = nfa.flows.CoupledRationalQuadraticSpline(d, 2, hidden, num_bins=K, init_identity=False,
                                                     reverse_mask=(d == 5))
    layer = load_layer(layer, golden_state(g), dt)
    x = T(g["x"])
    # kernel alone, fed with the REFERENCE conditioner output (isolates nf_rqs_coupling)
    p = layer.prqct
    uw, uh, ud = p._uncond()
    kw = p._kernel_kwargs()
    y, ld = nfa.ops.rqs_coupling(x, T(g["cond_density"]), uw, uh, ud, p.identity_features, p.transform_features, K, 0,
                                 **kw)
    assert_close(N(y), g["z_inv"], what="kernel z_inv", rtol=tol["rtol"] * 5, atol=tol["atol"] * 5)
    assert_close(N(ld), g["ld_inv"], what="kernel ld_inv", **ld_tol(g["x"].dtype))
    # layer end to end, both directions (conditioner included)
    z, ld = layer.inverse(x)
    assert z.dtype == x.dtype and z.shape == x.shape and ld.shape == (x.shape[0],)
    assert_close(N(z), g["z_inv"], what="z_inv", rtol=tol["rtol"] * 10, atol=tol["atol"] * 10)
    assert_close(N(ld), g["ld_inv"], what="ld_inv", **ld_tol(g["x"].dtype))
    z, ld = layer.forward(x)
    assert_close(N(z), g["z_fwd"], what="z_fwd", rtol=tol["rtol"] * 20, atol=tol["atol"] * 20)
    assert_close(N(ld), g["ld_fwd"], what="ld_fwd", **ld_tol(g["x"].dtype, True))
    # round trip (flows/flow_test.py:40-48)
    xr, ldr = layer.inverse(z)
    inside = np.abs(g["x"]) < 2.9
    # random (non-identity-init) splines have bins with small slopes: the fp32 round trip amplifies rounding
    assert_close(N(xr)[inside], g["x"][inside], what="roundtrip", rtol=1e-3 if tag == "f32" else 1e-9,
                 atol=1e-3 if tag == "f32" else 1e-9)
    assert_close(N(ld + ldr), np.zeros(x.shape[0], g["x"].dtype), what="ld cancel", rtol=0,
                 atol=5e-3 if tag == "f32" else 1e-8)


@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("d", [3, 4, 64])
def test_lu_linear_permute_vs_reference(nfa, d, tag):
    g = load_golden("lulinear_d%d_%s" % (d, tag))
    dt = torch.float32 if tag == "f32" else torch.float64
    tol = TOL[g["x"].dtype]
    layer = load_layer(nfa.flows.LULinearPermute(d, identity_init=False), golden_state(g), dt)
    x = T(g["x"])
    z, ld = layer.inverse(x)
    assert_close(N(z), g["z_inv"], what="z_inv", rtol=tol["rtol"] * 5, atol=tol["atol"] * 5)
    assert_close(N(ld), g["ld_inv"], what="ld_inv", **ld_tol(g["x"].dtype))
    z, ld = layer.forward(x)
    assert_close(N(z), g["z_fwd"], what="z_fwd", rtol=tol["rtol"] * 200, atol=tol["atol"] * 200)
    assert_close(N(ld), g["ld_fwd"], what="ld_fwd", **ld_tol(g["x"].dtype, True))
    # permutation is bit exact: a layer with L = U = I and zero bias reproduces index_select exactly
    with torch.no_grad():
        layer.linear.lower_entries.zero_()
        layer.linear.upper_entries.zero_()
        layer.linear.bias.zero_()
        layer.linear.unconstrained_upper_diag.fill_(50.0)   # softplus(50)+eps = 50.001 -> scale only
    z, _ = layer.inverse(x)
    perm = golden_state(g)["permutation._permutation"]
    scale = N(z)[:, 0] / g["x"][:, perm[0]]
    assert np.allclose(N(z), g["x"][:, perm] * scale[:, None], rtol=1e-6)


def test_lu_accumulate_modes(nfa):
    torch.manual_seed(0)
    layer = nfa.flows.LULinearPermute(8, identity_init=False).to(DEV)
    x = torch.randn(300, 8, device=DEV)
    z, ld = layer.inverse(x)
    acc = torch.full((300,), 2.0, device=DEV)
    z2 = layer._run(x, True, acc, +1)
    assert torch.equal(z, z2) and torch.allclose(acc, 2.0 + ld)
    z3 = layer._run(x, True, acc, -1)
    assert torch.allclose(acc, torch.full_like(acc, 2.0), atol=1e-6)


@pytest.mark.parametrize("name", ["masked_affine_d2", "masked_affine_d7", "masked_affine_nonfinite"])
def test_masked_affine_vs_reference(nfa, name):
    g = load_golden(name)
    tol = TOL[g["z"].dtype]
    for direction, zk, lk in ((0, "z_fwd", "ld_fwd"), (1, "z_inv", "ld_inv")):
        y, ld = nfa.ops.masked_affine(T(g["z"]), T(g["b"]), T(g["s"]), T(g["t"]), direction)
        assert_close(N(y), g[zk], what=zk, **tol)
        assert_close(N(ld), g[lk], what=lk, **tol)
    if "sd__b" in g:  # full layer incl. MLP maps
        d = g["z"].shape[1]
        b = torch.tensor([1.0 if i % 2 == 0 else 0.0 for i in range(d)])
        layer = nfa.flows.MaskedAffineFlow(b, nfa.nets.MLP([d, 2 * d, d]), nfa.nets.MLP([d, 2 * d, d]))
        layer = load_layer(layer, golden_state(g), torch.float32)
        z, ld = layer.forward(T(g["z"]))
        assert_close(N(z), g["z_fwd"], what="layer fwd", rtol=1e-4, atol=1e-5)
        z, ld = layer.inverse(T(g["z"]))
        assert_close(N(ld), g["ld_inv"], what="layer ld_inv", rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("name,C,split,smap,scale", [
    ("affine_block_C4_channel_exp", 4, "channel", "exp", True),
    ("affine_block_C5_channel_sigmoid", 5, "channel", "sigmoid", True),
    ("affine_block_C5_channel_inv_sigmoid_inv", 5, "channel_inv", "sigmoid_inv", True),
    ("affine_block_C3_channel_inv_noscale", 3, "channel_inv", "exp", False),
    ("affine_block_C4_checkerboard_sigmoid", 4, "checkerboard", "sigmoid", True),
    ("affine_block_C6_checkerboard_inv_exp", 6, "checkerboard_inv", "exp", True),
])
def test_affine_coupling_block_vs_reference(nfa, name, C, split, smap, scale):
    g = load_golden(name)
    if "checkerboard" in split:
        ch = (C, 8, 8, (2 if scale else 1) * C)
    elif split == "channel":
        ch = ((C + 1) // 2, 8, 8, (2 if scale else 1) * (C // 2))
    else:
        ch = (C // 2, 8, 8, (2 if scale else 1) * ((C + 1) // 2))
    net = nfa.nets.ConvNet2d(ch, (3, 1, 3), 0.0, init_zeros=False)
    layer = load_layer(nfa.flows.AffineCouplingBlock(net, scale, smap, split), golden_state(g), torch.float32)
    for fn, zk, lk in ((layer.forward, "z_fwd", "ld_fwd"), (layer.inverse, "z_inv", "ld_inv")):
        z, ld = fn(T(g["z"]))
        assert_close(N(z), g[zk], what=zk, rtol=1e-4, atol=1e-4)   # conv conditioner goes through MIOpen
        assert_close(N(ld), g[lk], what=lk, rtol=1e-4, atol=1e-4)
    if "checkerboard" not in split:  # kernel alone on the reference's param tensor
        c1 = (C + 1) // 2 if split == "channel" else C // 2
        for direction, zk, lk in ((0, "z_fwd", "ld_fwd"), (1, "z_inv", "ld_inv")):
            y, ld = nfa.ops.affine_coupling(T(g["z"]), T(g["param"]), c1, split == "channel_inv", smap if scale else None,
                                            direction)
            assert_close(N(y), g[zk], what="kernel " + zk, rtol=2e-5, atol=2e-5)
            assert_close(N(ld), g[lk], what="kernel " + lk, rtol=1e-4, atol=1e-4)


def test_affine_block_2d(nfa):
    g = load_golden("affine_block_2d")
    layer = load_layer(nfa.flows.AffineCouplingBlock(nfa.nets.MLP([1, 16, 16, 2], init_zeros=False)), golden_state(g),
                       torch.float32)
    z, ld = layer.forward(T(g["z"]))
    assert_close(N(z), g["z_fwd"], what="fwd", rtol=1e-5, atol=1e-5)
    assert_close(N(ld), g["ld_fwd"], what="ld_fwd", rtol=1e-5, atol=1e-5)
    z, ld = layer.inverse(T(g["z"]))
    assert_close(N(z), g["z_inv"], what="inv", rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name,shape", [("actnorm_4d", (6, 1, 1)), ("actnorm_2d", (3,))])
def test_actnorm_vs_reference(nfa, name, shape):
    g = load_golden(name)
    a = nfa.flows.ActNorm(shape).to(DEV)
    z, ld = a.forward(T(g["z"]))                 # forward-first data-dependent init
    assert ld.dim() == 0                          # 0-dim log_det like the reference
    assert_close(N(a.s), g["s_fwd"], what="s_fwd", rtol=1e-5, atol=1e-5)
    assert_close(N(a.t), g["t_fwd"], what="t_fwd", rtol=1e-5, atol=1e-5)
    assert_close(N(z), g["z_fwd"], what="z_fwd", rtol=1e-5, atol=2e-5)
    assert_close(N(ld), g["ld_fwd"], what="ld_fwd", rtol=1e-4, atol=1e-4)
    assert float(a.data_dep_init_done) == 1.0
    z, ld = a.forward(T(g["z2"]))                # no re-init
    assert_close(N(z), g["z2_fwd"], what="z2_fwd", rtol=1e-5, atol=2e-5)
    z, ld = a.inverse(T(g["z2"]))
    assert_close(N(z), g["z2_inv"], what="z2_inv", rtol=1e-5, atol=2e-5)
    assert_close(N(ld), g["ld2_inv"], what="ld2_inv", rtol=1e-4, atol=1e-4)
    b = nfa.flows.ActNorm(shape).to(DEV)
    z, ld = b.inverse(T(g["z"]))                 # inverse-first init
    assert_close(N(b.s), g["s_inv"], what="s_inv", rtol=1e-5, atol=1e-5)
    assert_close(N(b.t), g["t_inv"], what="t_inv", rtol=1e-5, atol=1e-5)
    assert_close(N(z), g["z_inv"], what="z_inv", rtol=1e-5, atol=2e-5)
    assert_close(N(ld), g["ld_inv"], what="ld_inv", rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("C,use_lu", [(3, True), (4, True), (12, True), (48, True), (4, False)])
def test_inv1x1_vs_reference(nfa, C, use_lu):
    g = load_golden("inv1x1_C%d_%s" % (C, "lu" if use_lu else "plain"))
    layer = load_layer(nfa.flows.Invertible1x1Conv(C, use_lu), golden_state(g), torch.float32)
    if use_lu:
        W, _ = layer._weight(True)
        assert_close(N(W), g["W_inv_dir"], what="W", rtol=1e-5, atol=1e-5)
        W, _ = layer._weight(False)
        assert_close(N(W), g["W_fwd_dir"], what="W^-1", rtol=1e-4, atol=1e-4)
    z, ld = layer.inverse(T(g["z"]))
    assert ld.dim() == 0
    assert_close(N(z), g["z_inv"], what="z_inv", rtol=1e-4, atol=1e-4)
    assert_close(N(ld), g["ld_inv"], what="ld_inv", rtol=1e-5, atol=1e-4)
    z, ld = layer.forward(T(g["z"]))
    assert_close(N(z), g["z_fwd"], what="z_fwd", rtol=1e-3, atol=1e-3)
    assert_close(N(ld), g["ld_fwd"], what="ld_fwd", rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("C,split", [(4, "channel"), (5, "channel_inv"), (3, "checkerboard")])
def test_glowblock_vs_reference(nfa, C, split):
    g = load_golden("glowblock_C%d_%s" % (C, split))
    layer = nfa.flows.GlowBlock(C, 8, split_mode=split, use_lu=True, init_zeros=False)
    layer = load_layer(layer, golden_state(g, "sd0__"), torch.float32)   # un-initialised ActNorm state
    z, ld = layer.inverse(T(g["z"]))
    assert_close(N(z), g["z_inv"], what="z_inv", rtol=2e-4, atol=2e-4)
    assert_close(N(ld), g["ld_inv"], what="ld_inv", rtol=2e-4, atol=2e-3)
    post = golden_state(g)
    assert_close(N(layer.flows[-1].s), post["flows.%d.s" % (len(layer.flows) - 1)], what="actnorm s", rtol=1e-4, atol=1e-4)
    zf, ldf = layer.forward(z)
    assert_close(N(zf), g["z_fwd"], what="z_fwd", rtol=1e-3, atol=1e-3)
    assert_close(N(ldf), g["ld_fwd"], what="ld_fwd", rtol=2e-4, atol=2e-3)


@pytest.mark.parametrize("name,seed,cin,cout,leaky,B,H,W", [
    ("convnet_6_12_16x16", 31, 6, 12, 0.0, 3, 16, 16), ("convnet_12_24_8x8", 32, 12, 24, 0.1, 5, 8, 8),
    ("convnet_24_48_4x4", 33, 24, 48, 0.0, 17, 4, 4), ("convnet_3_5_4x8", 34, 3, 5, 0.2, 3, 4, 8)])
def test_glow_convnet_kernel_vs_reference(nfa, name, seed, cin, cout, leaky, B, H, W):
    """nf_glow_convnet (conv3x3 -> LeakyReLU -> conv1x1 -> LeakyReLU -> conv3x3 in one launch) against the reference's
    ConvNet2d output; weights = the seeded default initialisation on both sides (checksum in the fixture)."""
    g = load_golden(name)
    torch.manual_seed(seed)
    net = nfa.nets.ConvNet2d([cin, 256, 256, cout], [3, 1, 3], leaky, init_zeros=False)
    chk = np.array([float(p_.double().abs().sum()) for p_ in net.parameters()])
    np.testing.assert_allclose(chk, g["weight_checksum"], rtol=1e-12)
    net = net.to(DEV)
    c1, _, c2, _, c3 = net.net
    prm = [p_.detach() for p_ in (c1.weight, c1.bias, c2.weight, c2.bias, c3.weight, c3.bias)]
    x = T(g["x"])
    wide = torch.randn(B, cin + 3, H, W, device=DEV)
    wide[:, 2:2 + cin] = x
    outs = []
    for layout in (nfa.ops.GLOW_CONV_WIDE, nfa.ops.GLOW_CONV_SMALL, nfa.ops.GLOW_CONV_TINY):   # 256 / 64 / 16-pixel workgroups
        if (layout == nfa.ops.GLOW_CONV_SMALL and 64 % (H * W) != 0) or (layout == nfa.ops.GLOW_CONV_TINY and 16 % (H * W) != 0):
            continue
        blob = nfa.ops.glow_convnet_pack(*prm, layout=layout)
        out = nfa.ops.glow_convnet(x, blob, cout, leaky, layout)
        assert_close(N(out), g["out"], what="out (layout %d)" % layout, rtol=1e-4, atol=1e-4)
        # a channel slice of a wider NCHW tensor is read in place (image stride != Cin H W)
        assert torch.equal(nfa.ops.glow_convnet(wide[:, 2:2 + cin], blob, cout, leaky, layout), out)
        assert torch.equal(nfa.ops.glow_convnet(x, blob, cout, leaky, layout), out)          # run-to-run identical
        outs.append(out)
    # the module takes this path on its own for chip-filling batches, and agrees with its library path
    with torch.no_grad():
        lib = net._forward_inference(x)
        old, type(net).FUSED_MIN_PIXELS = type(net).FUSED_MIN_PIXELS, 0
        try:
            fused = net._fused_pack(x)
            if H * W <= 64:
                want = nfa.ops.GLOW_CONV_TINY if 16 % (H * W) == 0 else nfa.ops.GLOW_CONV_SMALL
                assert fused is not None and fused[1] == want and net.forward_split(x) is None
                assert torch.equal(net(x), outs[-1])
            else:
                assert fused is None      # 16x16 images: the wide kernel waits for >= 128 workgroups
        finally:
            type(net).FUSED_MIN_PIXELS = old
    assert_close(N(outs[0]), N(lib), what="fused vs library", rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("name,seed,C,smap,leaky,B,H,W", [
    ("glowblock256_C12_16x16", 41, 12, "sigmoid", 0.0, 3, 16, 16), ("glowblock256_C24_8x8", 42, 24, "exp", 0.1, 5, 8, 8),
    ("glowblock256_C48_4x4", 43, 48, "sigmoid_inv", 0.0, 19, 4, 4), ("glowblock256_C5_4x4", 44, 5, "sigmoid", 0.0, 6, 4, 4)])
def test_glow_block_one_launch_vs_reference(nfa, name, seed, C, smap, leaky, B, H, W):
    """nf_glow_block (coupling + conditioner + [1x1 conv, ActNorm] in one launch, both directions) against the
    reference's GlowBlock outputs and against the layer-by-layer path; seeded default weights on both sides."""
    g = load_golden(name)
    torch.manual_seed(seed)
    blk = nfa.flows.GlowBlock(C, 256, scale_map=smap, leaky=leaky, init_zeros=False)
    with torch.no_grad():
        blk.flows[0].flows[1].param_map.net[-1].weight.mul_(0.2)
    blk = blk.to(DEV)
    x = T(g["x"])
    cls = nfa.nets.ConvNet2d
    saved = cls.FUSED_MIN_PIXELS, cls.FUSED_WIDE_MIN_PIXELS
    with torch.no_grad():
        try:
            cls.FUSED_MIN_PIXELS = 1 << 40                       # layer by layer (library convolutions)
            blk.inverse(x)                                       # ActNorm's data-dependent initialisation
            chk = np.array([float(p_.double().abs().sum()) for p_ in blk.parameters()])
            np.testing.assert_allclose(chk, g["checksum"], rtol=1e-5)
            zi0, ldi0 = blk.inverse(x)
            zf0, ldf0 = blk.forward(x)
            cls.FUSED_MIN_PIXELS, cls.FUSED_WIDE_MIN_PIXELS = 0, 0
            assert blk._whole_block(x) is not None
            zi, ldi = blk.inverse(x)
            zf, ldf = blk.forward(x)
            acc = torch.ones(B, device=DEV)
            blk._run(x, True, acc, nfa.ops.L.LD_SUB)
            # every kernel that takes this image size, called directly
            net = blk.flows[0].flows[1].param_map
            prm = [p_.detach() for p_ in (net.net[0].weight, net.net[0].bias, net.net[2].weight, net.net[2].bias,
                                          net.net[4].weight, net.net[4].bias)]
            for layout, pxw in ((nfa.ops.GLOW_CONV_WIDE, 256), (nfa.ops.GLOW_CONV_SMALL, 64), (nfa.ops.GLOW_CONV_TINY, 16)):
                if pxw % (H * W) != 0:
                    continue
                blob = nfa.ops.glow_convnet_pack(*prm, layout=layout)
                for inverse, zr_, ldr_ in ((True, zi0, ldi0), (False, zf0, ldf0)):
                    Wp, bp, ldp = blk._fused_mix(inverse)
                    try:
                        y_, ld_ = nfa.ops.glow_block(x, blob, layout, Wp, bp, ldp, leaky, smap, 1 if inverse else 0)
                    except NotImplementedError:      # many channels at 256 pixels per workgroup: beyond the LDS
                        assert layout == nfa.ops.GLOW_CONV_WIDE and C >= 24
                        continue
                    assert_close(N(y_), N(zr_), what="layout %d dir %d z" % (layout, inverse), rtol=1e-4, atol=1e-4)
                    assert_close(N(ld_), N(ldr_), what="layout %d dir %d ld" % (layout, inverse), rtol=1e-4, atol=1e-3)
        finally:
            cls.FUSED_MIN_PIXELS, cls.FUSED_WIDE_MIN_PIXELS = saved
    for got, ref, what in ((zi, g["z_inv"], "z_inv"), (ldi, g["ld_inv"], "ld_inv"), (zf, g["z_fwd"], "z_fwd"),
                           (ldf, g["ld_fwd"], "ld_fwd")):
        assert_close(N(got), ref, what=what, rtol=2e-4, atol=2e-3 if what.startswith("ld") else 2e-4)
    assert_close(N(zi), N(zi0), what="z_inv vs layers", rtol=1e-4, atol=1e-4)
    assert_close(N(ldi), N(ldi0), what="ld_inv vs layers", rtol=1e-4, atol=1e-3)
    assert_close(N(zf), N(zf0), what="z_fwd vs layers", rtol=1e-4, atol=1e-4)
    assert_close(N(ldf), N(ldf0), what="ld_fwd vs layers", rtol=1e-4, atol=1e-3)
    assert_close(N(acc), 1.0 - N(ldi), what="acc", rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("cin,cout,H,W,B,leaky", [(6, 12, 16, 16, 2, 0.3), (5, 7, 8, 8, 9, 0.0), (9, 13, 4, 4, 33, 0.05),
                                                  (1, 2, 2, 2, 70, 1.0), (2, 3, 1, 4, 5, 0.0)])
def test_glow_convnet_kernels_vs_oracle(nfa, oracle, cin, cout, H, W, B, leaky):
    """Every GlowBlock-conditioner kernel that takes the image size against the CPU oracle (oracle.convnet2d) on random
    normal weights: odd channel counts, non-square and 1-pixel-high images, batches off the workgroup size, slopes 0..1."""
    g = torch.Generator().manual_seed(cin * 100 + cout)
    w1 = 0.3 * torch.randn(256, cin, 3, 3, generator=g); b1 = torch.randn(256, generator=g)
    w2 = 0.08 * torch.randn(256, 256, 1, 1, generator=g); b2 = torch.randn(256, generator=g)
    w3 = 0.05 * torch.randn(cout, 256, 3, 3, generator=g); b3 = torch.randn(cout, generator=g)
    x = torch.randn(B, cin, H, W, generator=g)
    ref = oracle.convnet2d(x.numpy().astype(np.float64), [t.numpy().astype(np.float64) for t in (w1, w2, w3)],
                           [t.numpy().astype(np.float64) for t in (b1, b2, b3)], leaky)
    prm = [t.to(DEV) for t in (w1, b1, w2, b2, w3, b3)]
    ran = 0
    for layout, pxw in ((nfa.ops.GLOW_CONV_WIDE, 256), (nfa.ops.GLOW_CONV_SMALL, 64), (nfa.ops.GLOW_CONV_TINY, 16)):
        if pxw % (H * W) != 0:
            continue
        blob = nfa.ops.glow_convnet_pack(*prm, layout=layout)
        out = nfa.ops.glow_convnet(x.to(DEV), blob, cout, leaky, layout)
        assert_close(N(out).astype(np.float64), ref, what="layout %d" % layout, rtol=1e-4, atol=1e-4)
        ran += 1
    assert ran >= 1


@pytest.mark.parametrize("C,H,W,B,smap,leaky", [(12, 16, 16, 2, "exp", 0.2), (7, 8, 8, 5, "sigmoid", 0.0), (10, 4, 4, 21, "sigmoid_inv", 0.05)])
def test_glow_block_one_launch_vs_oracle(nfa, oracle, C, H, W, B, smap, leaky):
    """nf_glow_block against the CPU oracle's GlowBlock composite on perturbed (non-default) parameters, both directions."""
    torch.manual_seed(C + H)
    blk = nfa.flows.GlowBlock(C, 256, scale_map=smap, leaky=leaky, init_zeros=False)
    with torch.no_grad():
        for p_ in blk.parameters():
            p_.add_(0.02 * torch.randn_like(p_))
        blk.flows[0].flows[1].param_map.net[-1].weight.mul_(0.2)
    blk = blk.to(DEV)
    x = torch.randn(B, C, H, W, generator=torch.Generator().manual_seed(3)).to(DEV)
    cls = nfa.nets.ConvNet2d
    saved = cls.FUSED_MIN_PIXELS, cls.FUSED_WIDE_MIN_PIXELS
    with torch.no_grad():
        try:
            cls.FUSED_MIN_PIXELS = 1 << 40
            blk.inverse(x)                                   # ActNorm initialisation, layer by layer
            cls.FUSED_MIN_PIXELS, cls.FUSED_WIDE_MIN_PIXELS = 0, 0
            assert blk._whole_block(x) is not None
            zi, ldi = blk.inverse(x)
            zf, ldf = blk.forward(x)
        finally:
            cls.FUSED_MIN_PIXELS, cls.FUSED_WIDE_MIN_PIXELS = saved
    st = {k: N(v).astype(np.float64) for k, v in blk.state_dict().items()}
    x64 = N(x).astype(np.float64)
    ozi, oldi, _ = oracle.glow_block(st, x64, True, leaky, smap)
    ozf, oldf, _ = oracle.glow_block(st, x64, False, leaky, smap)
    assert_close(N(zi).astype(np.float64), ozi, what="z_inv", rtol=2e-4, atol=2e-4)
    assert_close(N(ldi).astype(np.float64), oldi, what="ld_inv", rtol=2e-4, atol=2e-3)
    assert_close(N(zf).astype(np.float64), ozf, what="z_fwd", rtol=2e-4, atol=2e-4)
    assert_close(N(ldf).astype(np.float64), oldf, what="ld_fwd", rtol=2e-4, atol=2e-3)


def test_glow_model_hidden256_vs_reference(nfa):
    """The config-4 architecture at its real width against the reference's log_prob (fixture: seeded construction on both
    sides), with every GlowBlock forced through the one-launch kernels (12 images: thresholds lowered for the test)."""
    import importlib.util
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_golden.py")
    src = open(path).read()
    ns = {"torch": torch}
    exec(src[src.index("def build_glow_c4("):src.index("def gen_glow_model256(")], ns)   # the shared model builder only
    g = load_golden("model_glow_c4_hidden256")
    m = ns["build_glow_c4"](nfa, 3, 2, 256, seed=61)
    x = T(g["x"])
    cls = nfa.nets.ConvNet2d
    saved = cls.FUSED_MIN_PIXELS, cls.FUSED_WIDE_MIN_PIXELS
    m = m.to(DEV)
    with torch.no_grad():
        lp0 = m.log_prob(x)          # ActNorm initialisation (layer by layer: the mix needs initialised ActNorms)
        chk = float(sum(p_.double().abs().sum() for p_ in m.parameters()))
        assert abs(chk - float(g["checksum"])) < 1e-5 * abs(chk)
        try:
            cls.FUSED_MIN_PIXELS, cls.FUSED_WIDE_MIN_PIXELS = 0, 0
            assert all(b._whole_block(torch.empty(12, *shape, device=DEV)) is not None
                       for fl, shape in zip(m.flows, ((48, 4, 4), (24, 8, 8), (12, 16, 16))) for b in fl[:-1])
            lp = m.log_prob(x)
        finally:
            cls.FUSED_MIN_PIXELS, cls.FUSED_WIDE_MIN_PIXELS = saved
    assert _rel(N(lp0), g["log_prob_first"]) < 1e-4, _rel(N(lp0), g["log_prob_first"])   # north_star: fp32 log_prob <= 1e-4 rel
    assert _rel(N(lp), g["log_prob"]) < 1e-4, _rel(N(lp), g["log_prob"])


def _glow_c4_builder():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_golden.py")
    src = open(path).read()
    ns = {"torch": torch}
    exec(src[src.index("def build_glow_c4("):src.index("def gen_glow_model256(")], ns)   # the shared model builder only
    return ns["build_glow_c4"]


@pytest.mark.parametrize("force_block_kernels", [False, True])
def test_glow_config4_full_depth_vs_reference(nfa, force_block_kernels):
    """BASELINE configs[3] at FULL depth -- MultiscaleFlow L = 3, K = 32 blocks per level, hidden 256, 32x32x3 -- against the
    reference (core.py:588-616, :553-586; fixture model_glow_c4_full: seeded construction on both sides, 8 images):
    log_prob on the first call (data-dependent ActNorm initialisation of all 96 blocks), log_prob on the second call, and
    the sampling direction on fixed per-level base noise.  96 blocks of fp32 accumulation at the north-star bar of 1e-4
    relative; once with the library's own dispatch, once with every GlowBlock forced through the one-launch kernels."""
    g = load_golden("model_glow_c4_full")
    m = _glow_c4_builder()(nfa, 3, 32, 256, seed=63).to(DEV)
    x = T(g["x"])
    cls = nfa.nets.ConvNet2d
    saved = cls.FUSED_MIN_PIXELS, cls.FUSED_WIDE_MIN_PIXELS
    try:
        if force_block_kernels:
            cls.FUSED_MIN_PIXELS, cls.FUSED_WIDE_MIN_PIXELS = 0, 0
        lp0 = m.log_prob(x)          # ActNorm initialisation
        chk = float(sum(p_.double().abs().sum() for p_ in m.parameters()))
        if force_block_kernels:
            assert all(b._whole_block(torch.empty(8, *shape, device=DEV)) is not None
                       for fl, shape in zip(m.flows, ((48, 4, 4), (24, 8, 8), (12, 16, 16))) for b in fl[:-1])
        lp = m.log_prob(x)
        xs, lq = m.sample_from_noise([T(g["eps0"]), T(g["eps1"]), T(g["eps2"])])
        lps = m.log_prob(xs)
    finally:
        cls.FUSED_MIN_PIXELS, cls.FUSED_WIDE_MIN_PIXELS = saved
    assert _rel(N(lp0), g["log_prob_first"]) < 1e-4, _rel(N(lp0), g["log_prob_first"])
    # the initialisation wrote the same ActNorm parameters as the reference's (checksum of every parameter after it)
    assert abs(chk - float(g["checksum"])) < 1e-5 * abs(chk), (chk, float(g["checksum"]))
    assert _rel(N(lp), g["log_prob"]) < 1e-4, _rel(N(lp), g["log_prob"])
    assert_close(N(xs), g["sample"], what="sample", rtol=1e-4, atol=1e-4)
    assert _rel(N(lq), g["sample_logq"]) < 1e-4, _rel(N(lq), g["sample_logq"])
    assert _rel(N(lps), N(lq)) < 1e-4, _rel(N(lps), N(lq))          # core_test.py:144-196


def test_maf_config5_full_model_vs_reference(nfa):
    """BASELINE configs[4] as a MODEL: 10 x MaskedAffineAutoregressive(128, 512) under a DiagGaussian base against the
    reference (core.py:182-197 over affine/autoregressive.py:29-38, 98-128; fixture model_maf_c5_full, 64 rows):
    inverse direction (the reference's 128-pass loop per layer; here one nf_maf_inverse launch per layer), log_prob, the
    forward direction on fixed noise and its log_q."""
    g = load_golden("model_maf_c5_full")
    torch.manual_seed(2000)
    flows = [nfa.flows.MaskedAffineAutoregressive(128, 512, num_blocks=2) for _ in range(10)]
    m = nfa.NormalizingFlow(nfa.distributions.DiagGaussian(128, trainable=False), flows)
    _perturb(m, 0.02, 9)
    chk = float(sum(p_.double().abs().sum() for p_ in m.parameters()))
    assert abs(chk - float(g["checksum"])) < 1e-6 * abs(chk)
    m = m.to(DEV)
    z, ld = m.inverse_and_log_det(T(g["x"]))
    lp = m.log_prob(T(g["x"]))
    # north-star bar (1e-4 relative) against the reference's fp32 outputs AND against the reference evaluated in double
    # precision on the same weights (`*_f64`); the kernel's error against the double-precision values may also be no more
    # than a small multiple of what the reference's own fp32 arithmetic loses there (1.8e-5 on z, 1.2e-5 on log_prob)
    assert_close(N(z), g["z_inv"], what="z_inv", rtol=1e-4, atol=1e-4)
    assert_close(N(ld), g["ld_inv"], what="ld_inv", rtol=1e-4, atol=1e-4)
    assert _rel(N(lp), g["log_prob"]) < 1e-4, _rel(N(lp), g["log_prob"])
    for got, name in ((N(z), "z_inv"), (N(ld), "ld_inv"), (N(lp), "log_prob")):
        r64 = g[name + "_f64"]
        e_gpu = np.abs(got.astype(np.float64) - r64) / np.maximum(1.0, np.abs(r64))
        e_ref = np.abs(g[name].astype(np.float64) - r64) / np.maximum(1.0, np.abs(r64))
        assert e_gpu.max() < 1e-4, (name, e_gpu.max())
        for q in (0.9, 0.99, 1.0):
            assert np.quantile(e_gpu, q) <= 4 * np.quantile(e_ref, q) + 2e-6, (name, q, np.quantile(e_gpu, q), np.quantile(e_ref, q))
    zf, ldf = m.forward_and_log_det(T(g["eps"]))
    assert_close(N(zf), g["z_fwd"], what="z_fwd", rtol=1e-4, atol=1e-4)
    assert_close(N(ldf), g["ld_fwd"], what="ld_fwd", rtol=1e-4, atol=1e-4)
    xs, lq = m.sample_from_noise(T(g["eps"]))
    assert _rel(N(lq), g["sample_logq"]) < 1e-4


def test_multiscale_graphs_follow_parameter_reloads_and_do_not_alias(nfa):
    """MultiscaleFlow.use_graphs: results of successive replays are separate tensors, and load_state_dict / train() /
    refresh_graphs() drop recorded graphs so that a replay never evaluates stale packed weights."""
    m = _glow_c4_builder()(nfa, 3, 2, 256, seed=5).to(DEV)
    x1 = torch.rand(16, 3, 32, 32, device=DEV)
    x2 = torch.rand(16, 3, 32, 32, device=DEV)
    m.log_prob(x1)                                  # ActNorm initialisation
    e1, e2 = N(m.log_prob(x1)), N(m.log_prob(x2))
    m.use_graphs(True)
    outs = [m.log_prob(x1), m.log_prob(x2)]         # second replay must not overwrite the first result
    # (16 images leave some levels on the library convolutions, whose summation order may differ run to run: 1e-6, not bits)
    assert _rel(N(outs[0]), e1) < 1e-6 and _rel(N(outs[1]), e2) < 1e-6 and outs[0].data_ptr() != outs[1].data_ptr()
    sd2 = {k: (v * 1.1 if v.is_floating_point() and v.dim() > 0 and "flows" in k and k.endswith(".t") else v)
           for k, v in m.state_dict().items()}
    m.load_state_dict(sd2)                          # drops the graphs
    g3 = N(m.log_prob(x1))
    m.use_graphs(False)
    assert _rel(g3, N(m.log_prob(x1))) < 1e-6
    assert _rel(g3, e1) > 1e-5
    # in-place update + refresh_graphs()
    m.use_graphs(True)
    m.log_prob(x1)
    with torch.no_grad():
        for p_ in m.parameters():
            p_.mul_(0.999)
    m.refresh_graphs()
    g4 = N(m.log_prob(x1))
    m.use_graphs(False)
    assert _rel(g4, N(m.log_prob(x1))) < 1e-6


def test_glow_config4_shapes_through_the_block_kernels(nfa):
    """BASELINE configs[4] geometry (L = 3, hidden 256, 32x32x3, batch 256; 3 blocks per level instead of 32): the three
    levels run through the three GlowBlock kernels (256- / 64- / 16-pixel workgroups).  Size-independent properties at the
    full batch: log_prob through the block kernels == log_prob layer by layer, and sample()'s log_q == log_prob(sample)
    (core_test.py:144-196)."""
    torch.manual_seed(5)
    L_, K_, hidden, channels = 3, 3, 256, 3
    input_shape = (3, 32, 32)
    q0, merges, flows = [], [], []
    for i in range(L_):
        fl = [nfa.flows.GlowBlock(channels * 2 ** (L_ + 1 - i), hidden, split_mode="channel", scale=True, init_zeros=False)
              for _ in range(K_)]
        for b in fl:
            with torch.no_grad():
                b.flows[0].flows[1].param_map.net[-1].weight.mul_(0.1)
        fl += [nfa.flows.Squeeze()]
        flows += [fl]
        if i > 0:
            merges += [nfa.flows.Merge()]
            latent = (input_shape[0] * 2 ** (L_ - i), input_shape[1] // 2 ** (L_ - i), input_shape[2] // 2 ** (L_ - i))
        else:
            latent = (input_shape[0] * 2 ** (L_ + 1), input_shape[1] // 2 ** L_, input_shape[2] // 2 ** L_)
        q0 += [nfa.distributions.DiagGaussian(latent)]
    m = nfa.MultiscaleFlow(q0, flows, merges, class_cond=False).to(DEV)
    x = torch.rand(256, 3, 32, 32, device=DEV)
    cls = nfa.nets.ConvNet2d
    saved = cls.FUSED_MIN_PIXELS
    with torch.no_grad():
        m.log_prob(x)                                   # ActNorm initialisation
        layouts = set()
        for lvl, shape in enumerate(((256, 48, 4, 4), (256, 24, 8, 8), (256, 12, 16, 16))):
            w = flows[lvl][0]._whole_block(torch.empty(shape, device=DEV))
            assert w is not None
            layouts.add(w[1])
        assert layouts == {nfa.ops.GLOW_CONV_WIDE, nfa.ops.GLOW_CONV_SMALL, nfa.ops.GLOW_CONV_TINY}
        lp = m.log_prob(x)
        torch.manual_seed(1)
        xs, lq = m.sample(256)
        lps = m.log_prob(xs)
        try:
            cls.FUSED_MIN_PIXELS = 1 << 40
            lp_layers = m.log_prob(x)
        finally:
            cls.FUSED_MIN_PIXELS = saved
    assert _rel(N(lp), N(lp_layers)) < 1e-5, _rel(N(lp), N(lp_layers))
    assert _rel(N(lps), N(lq)) < 1e-4, _rel(N(lps), N(lq))


def test_diag_gaussian_and_squeeze(nfa):
    g = load_golden("diag_gaussian")
    q = nfa.distributions.DiagGaussian((3, 2, 2)).to(DEV)
    with torch.no_grad():
        q.loc.copy_(T(g["loc"]))
        q.log_scale.copy_(T(g["log_scale"]))
    assert_close(N(q.log_prob(T(g["z"]))), g["log_prob"], what="log_prob", rtol=1e-5, atol=1e-5)
    q.temperature = 0.7
    assert_close(N(q.log_prob(T(g["z"]))), g["log_prob_t07"], what="log_prob_t", rtol=1e-5, atol=1e-5)
    g = load_golden("squeeze")
    s = nfa.flows.Squeeze()
    assert np.array_equal(N(s.forward(T(g["z"]))[0]), g["fwd"])
    assert np.array_equal(N(s.inverse(T(g["z"]))[0]), g["inv"])


# ---- whole models ---------------------------------------------------------------------------------------------
def _rel(a, b):
    return np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))


@pytest.fixture(params=["wg128", "wg256"])
def chain_workgroups(request, nfa):
    """The fused chain on both workgroup sizes (round 6, late): batches of <= 32 768 rows run on 128-row workgroups by default
    (csrc/rqs_fused_nw4.hip); "wg256" keeps them on the 256-row workgroups of the headline configuration."""
    old = nfa.config.set_fused_small_batch(request.param == "wg128")
    yield request.param
    nfa.config.set_fused_small_batch(old)


@pytest.mark.parametrize("dim,hidden,bins,blocks,lu", [(64, 128, 8, 2, True), (64, 128, 8, 2, False), (64, 128, 4, 1, True),
                                                       (64, 128, 16, 2, True), (32, 64, 8, 2, True), (16, 32, 16, 1, False)])
def test_small_batch_workgroups_give_the_same_bits(nfa, dim, hidden, bins, blocks, lu):
    """nf_rqs_fused_chain on 128-row workgroups (the default at <= 32 768 rows) and on 256-row workgroups (nf_rqs_fused_small_batch(0)):
    the same kernel source built for two workgroup sizes -- a row's arithmetic does not depend on which wave of which workgroup owns it,
    so log_prob and sample agree BIT FOR BIT, for ragged and tiny batches, every bin count, narrower layers, with and without the LU.
    (The golden-vector tests of the chain therefore pin both builds; the whole-model ones also run on both: chain_workgroups.)"""
    torch.manual_seed(dim + bins)
    flows = []
    for i in range(3):
        flows += [nfa.flows.CoupledRationalQuadraticSpline(dim, blocks, hidden, num_bins=bins, init_identity=False, reverse_mask=bool(i & 1))]
        if lu:
            flows += [nfa.flows.LULinearPermute(dim)]
    m = nfa.NormalizingFlow(nfa.distributions.DiagGaussian(dim, trainable=False), flows)
    with torch.no_grad():
        for p_ in m.parameters():
            p_.add_(0.04 * torch.randn_like(p_))
    m = m.to(DEV)
    try:
        with torch.no_grad():
            for B in (1, 31, 257, 4099, 32768):
                x = 1.3 * torch.randn(B, dim, device=DEV)
                eps = torch.randn(B, dim, device=DEV)
                nfa.config.set_fused_small_batch(True)
                lp1 = m.log_prob(x)
                xs1, lq1 = m.sample_from_noise(eps)
                nfa.config.set_fused_small_batch(False)
                lp0 = m.log_prob(x)
                xs0, lq0 = m.sample_from_noise(eps)
                assert torch.equal(lp1, lp0) and torch.equal(xs1, xs0) and torch.equal(lq1, lq0), B
                assert torch.isfinite(lp1).all()
    finally:
        nfa.config.set_fused_small_batch(True)


def test_model_c2mini_vs_reference(nfa, chain_workgroups):
    from bench import build_c2_model
    g = load_golden("model_c2mini")
    m = build_c2_model(num_layers=4, dim=16, hidden=32, seed=0, sigma=0.05)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in golden_state(g).items()}, strict=True)
    m = m.to(DEV)
    lp = N(m.log_prob(T(g["x"])))
    assert _rel(lp, g["log_prob"]) < 1e-4          # north-star: fp32 log_prob within 1e-4 rel
    xs, lq = m.sample_from_noise(T(g["eps"]))
    assert_close(N(xs), g["sample"], what="sample", rtol=1e-4, atol=1e-4)
    assert _rel(N(lq), g["sample_logq"]) < 1e-4
    # hipGraph replay gives bit-identical results to eager launches
    m.use_graphs(True)
    lp_g = N(m.log_prob(T(g["x"])))
    lp_g2 = N(m.log_prob(T(g["x"])))
    assert np.array_equal(lp_g, lp) and np.array_equal(lp_g2, lp)


def test_model_c2_full_width_head_vs_reference(nfa, chain_workgroups):
    """The benchmark model itself (32 layers, d=64, hidden 128): seeded construction reproduces the reference's
    weights, so the reference's log_prob on the first 128 benchmark rows is a golden vector for it."""
    from bench import build_c2_model
    g = load_golden("model_c2_head")
    m = build_c2_model().to(DEV)
    lp = N(m.log_prob(T(g["x"])))
    assert _rel(lp, g["log_prob"]) < 1e-4, _rel(lp, g["log_prob"])


def test_model_c2_full_size_properties(nfa, oracle):
    """B = 65 536 (BASELINE config 2): oracle on a slice, determinism, sample/log_prob consistency
    (core_test.py:187), in-bound fraction."""
    from bench import build_c2_model, c2_inputs, state_to_numpy
    m = build_c2_model().to(DEV)
    x = c2_inputs().to(DEV)
    lp = m.log_prob(x)
    lp2 = m.log_prob(x)
    assert torch.equal(lp, lp2)                                        # deterministic
    assert torch.isfinite(lp).all()
    ora = oracle.OracleNSF(state_to_numpy(m), num_layers=len(m.flows))
    # every benchmark row against the CPU oracle on a many-core host (the GPU box: 256 threads, ~3 s through the OpenMP entry
    # point), 8 192 rows elsewhere (round 4 checked 256)
    rows = 65536 if (os.cpu_count() or 1) >= 64 else 8192
    ref = ora.log_prob_whole(N(x[:rows]))
    assert ref.shape == (rows,) and _rel(N(lp[:rows]), ref) < 1e-4, _rel(N(lp[:rows]), ref)
    nll = float(-lp.mean() / 64)
    assert 1.3 < nll < 1.8, nll                                        # reference: 1.5415 nats/dim
    g = torch.Generator().manual_seed(5)
    eps = torch.randn(65536, 64, generator=g).to(DEV)
    xs, lq = m.sample_from_noise(eps)
    lp_s = m.log_prob(xs)
    assert _rel(N(lp_s), N(lq)) < 1e-4                                 # log_prob(sample) == returned log_q


def test_model_c1_realnvp_vs_reference(nfa):
    g = load_golden("model_c1_realnvp")
    b = torch.tensor([1.0, 0.0])
    fl = []
    for i in range(4):
        s = nfa.nets.MLP([2, 4, 2], init_zeros=True)
        t = nfa.nets.MLP([2, 4, 2], init_zeros=True)
        fl += [nfa.flows.MaskedAffineFlow(b if i % 2 == 0 else 1 - b, t, s), nfa.flows.ActNorm(2)]
    m = nfa.NormalizingFlow(nfa.distributions.DiagGaussian(2), fl)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in golden_state(g, "sd0__").items()}, strict=True)
    m = m.to(DEV)
    lp = N(m.log_prob(T(g["x"])))                  # triggers the ActNorm data-dependent init (inverse-first)
    assert _rel(lp, g["log_prob"]) < 1e-4
    assert _rel(N(m.log_prob(T(g["x"]))), g["log_prob_second"]) < 1e-4
    xs, lq = m.sample_from_noise(T(g["eps"]))
    assert_close(N(xs), g["sample"], what="sample", rtol=1e-4, atol=1e-4)
    assert _rel(N(lq), g["sample_logq"]) < 1e-4


@pytest.mark.parametrize("force_block_kernels", [False, True])
def test_model_c4mini_glow_vs_reference(nfa, force_block_kernels):
    """Mini Glow (hidden 16) against the reference; with the thresholds lowered every GlowBlock runs on the one-launch
    kernels, whose hidden width is 256: narrower conditioners ride them zero-padded."""
    g = load_golden("model_c4mini_glow")
    L_, K_, hidden, channels = 2, 2, 16, 3
    input_shape = (3, 8, 8)
    q0, merges, flows = [], [], []
    for i in range(L_):
        fl = [nfa.flows.GlowBlock(channels * 2 ** (L_ + 1 - i), hidden, split_mode="channel", scale=True,
                                  init_zeros=False) for _ in range(K_)]
        fl += [nfa.flows.Squeeze()]
        flows += [fl]
        if i > 0:
            merges += [nfa.flows.Merge()]
            latent = (input_shape[0] * 2 ** (L_ - i), input_shape[1] // 2 ** (L_ - i), input_shape[2] // 2 ** (L_ - i))
        else:
            latent = (input_shape[0] * 2 ** (L_ + 1), input_shape[1] // 2 ** L_, input_shape[2] // 2 ** L_)
        q0 += [nfa.distributions.DiagGaussian(latent)]
    m = nfa.MultiscaleFlow(q0, flows, merges, class_cond=False)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in golden_state(g, "sd0__").items()}, strict=True)
    m = m.to(DEV)
    cls = nfa.nets.ConvNet2d
    saved = cls.FUSED_MIN_PIXELS, cls.FUSED_WIDE_MIN_PIXELS
    if force_block_kernels:
        cls.FUSED_MIN_PIXELS, cls.FUSED_WIDE_MIN_PIXELS = 0, 0
    try:
        _c4mini_checks(nfa, m, g, force_block_kernels)
    finally:
        cls.FUSED_MIN_PIXELS, cls.FUSED_WIDE_MIN_PIXELS = saved


def _c4mini_checks(nfa, m, g, forced):
    lp = N(m.log_prob(T(g["x"])))
    if forced:
        assert all(b._whole_block(torch.empty(6, *shape, device=DEV)) is not None
                   for fl, shape in zip(m.flows, ((24, 2, 2), (12, 4, 4))) for b in fl[:-1])
    assert _rel(lp, g["log_prob"]) < 1e-4, _rel(lp, g["log_prob"])
    assert _rel(N(m.log_prob(T(g["x"]))), g["log_prob_second"]) < 1e-4
    # sample / log_prob consistency (core_test.py:144-196)
    torch.manual_seed(0)
    xs, lq = m.sample(8)
    assert _rel(N(m.log_prob(xs)), N(lq)) < 1e-4, _rel(N(m.log_prob(xs)), N(lq))


def test_drop_in_under_reference_style_container(nfa):
    """Duck-typed use: a container that only calls flow(z) / flow.inverse(z) and `log_q += log_det`
    (the loops of normflows/core.py:36-38, 193-195) works with our layers, including 0-dim log-dets."""
    torch.manual_seed(0)
    flows = [nfa.flows.ActNorm(6), nfa.flows.LULinearPermute(6), nfa.flows.CoupledRationalQuadraticSpline(6, 1, 16)]
    flows = [f.to(DEV) for f in flows]
    x = torch.randn(33, 6, device=DEV)
    log_q = torch.zeros(33, device=DEV)
    z = x
    for f in reversed(flows):
        z, ld = f.inverse(z)
        log_q += ld
    m = nfa.NormalizingFlow(nfa.distributions.DiagGaussian(6, trainable=False), flows).to(DEV)
    z2, ld2 = m.inverse_and_log_det(x)
    assert torch.allclose(z, z2) and torch.allclose(log_q, ld2, atol=1e-5)


def test_drop_in_under_the_reference_own_containers(nfa):
    """The reference's REAL containers around our layers (runs wherever /root/reference and a GPU exist together; skipped on a
    box without the reference): `nf.NormalizingFlow(q0, [ours...])` log_prob / sample / forward_kld (core.py:9-60, 87-102,
    167-197) and `nf.MultiscaleFlow` (core.py:455-616) with our GlowBlock / Squeeze / Merge layers give bit-for-bit what our own
    containers give on the same layers, and our layers load the reference's state_dict unchanged."""
    import sys
    # the build container has /root/reference; on the GPU box the package arrives staged (tools/stage_reference.py -> .refstage/,
    # git-ignored, NF_REFERENCE_PATH) -- round 5 ran it there: profiles/r05_reference_containers_gpubox.log
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cands = [os.environ.get("NF_REFERENCE_PATH"), os.path.join(here, ".refstage"), "/root/reference"]
    ref = next((os.path.abspath(c) for c in cands if c and os.path.isdir(os.path.join(c, "normflows"))), None)
    if ref is None:
        pytest.skip("the reference is not on this box (stage it with tools/stage_reference.py)")
    sys.path.insert(0, ref)
    nf = pytest.importorskip("normflows")
    assert os.path.abspath(nf.__file__).startswith(ref), nf.__file__
    torch.manual_seed(0)
    flows = []
    for _ in range(3):
        flows += [nfa.flows.CoupledRationalQuadraticSpline(8, 1, 32), nfa.flows.LULinearPermute(8)]
    flows += [nfa.flows.ActNorm(8)]
    for f in flows:
        f.to(DEV)
    with torch.no_grad():
        for f in flows:
            for p_ in f.parameters():
                p_.add_(0.05 * torch.randn_like(p_))
    x = torch.randn(130, 8, device=DEV)
    ours = nfa.NormalizingFlow(nfa.distributions.DiagGaussian(8, trainable=False), flows).to(DEV)
    with torch.no_grad():
        ours.log_prob(x)                                   # ActNorm's data-dependent init happens once, here
    theirs = nf.NormalizingFlow(nf.distributions.DiagGaussian(8, trainable=False), flows).to(DEV)
    with torch.no_grad():
        # (1) the reference's loops (core.py:36-38, 193-195) call our layers one by one: BIT-identical to the same calls written out
        # here.  (First run on the GPU box in round 5: the original version of this test expected bit equality against our own
        # container, whose fused chain sums in another order, and failed by one unit in the fifth digit.)
        q0 = theirs.q0                                      # (the reference's own base distribution: torch ops)
        lp_t = theirs.log_prob(x)
        z_, lq_ = x, torch.zeros(len(x), device=DEV)
        for f in reversed(flows):
            z_, ld_ = f.inverse(z_)
            lq_ = lq_ + ld_
        assert torch.equal(lp_t, lq_ + q0.log_prob(z_))
        z_t, ld_t = theirs.inverse_and_log_det(x)
        assert torch.equal(z_t, z_)
        torch.manual_seed(5)
        xs_t, lq_t = theirs.sample(64)
        torch.manual_seed(5)
        zs, lqs = q0(64)
        for f in flows:
            zs, ld_ = f(zs)
            lqs = lqs - ld_
        assert torch.equal(xs_t, zs) and torch.equal(lq_t, lqs)
        # (2) against our default (layer pairs fused into one persistent launch): another summation order, 1e-5
        lp_f = ours.log_prob(x)
        assert _rel(N(lp_t), N(lp_f)) < 1e-5, _rel(N(lp_t), N(lp_f))
        torch.manual_seed(5)
        xs_f, lq_f = ours.sample(64)
        assert torch.allclose(xs_t, xs_f, atol=2e-5) and _rel(N(lq_t), N(lq_f)) < 1e-5
    lt = theirs.forward_kld(x)
    lo = ours.forward_kld(x)
    assert abs(float(lt) - float(lo)) < 2e-6 * abs(float(lo))
    # MultiscaleFlow (Glow, examples/glow.ipynb cell 2, reduced): the reference's container over our blocks
    torch.manual_seed(1)
    L_, K_, C, hidden = 2, 2, 3, 16
    q0, merges, gflows = [], [], []
    for i in range(L_):
        gflows += [[nfa.flows.GlowBlock(C * 2 ** (L_ + 1 - i), hidden, split_mode="channel", scale=True) for _ in range(K_)]
                   + [nfa.flows.Squeeze()]]
        if i > 0:
            merges += [nfa.flows.Merge()]
            shape = (C * 2 ** (L_ - i), 8 // 2 ** (L_ - i), 8 // 2 ** (L_ - i))
        else:
            shape = (C * 2 ** (L_ + 1), 8 // 2 ** L_, 8 // 2 ** L_)
        q0 += [nfa.distributions.DiagGaussian(shape, trainable=False)]
    img = torch.rand(16, C, 8, 8, device=DEV)
    ours_ms = nfa.MultiscaleFlow(q0, gflows, merges).to(DEV)
    with torch.no_grad():
        lp_o = ours_ms.log_prob(img)                       # initialises every ActNorm
    theirs_ms = nf.MultiscaleFlow(q0, gflows, merges).to(DEV)
    with torch.no_grad():
        lp_t = theirs_ms.log_prob(img)
        assert _rel(N(lp_t), N(lp_o)) < 1e-5               # (ours runs whole levels as one launch: another summation order)
        torch.manual_seed(9)
        xs_t, lq_t = theirs_ms.sample(8)
        torch.manual_seed(9)
        xs_o, lq_o = ours_ms.sample(8)
        assert torch.allclose(xs_t, xs_o, atol=1e-4) and _rel(N(lq_t), N(lq_o)) < 1e-5
    # and the other way round: our layers take the reference layers' state_dict unchanged
    torch.manual_seed(2)
    ref_layer = nf.flows.CoupledRationalQuadraticSpline(8, 1, 32)
    mine = nfa.flows.CoupledRationalQuadraticSpline(8, 1, 32)
    mine.load_state_dict(ref_layer.state_dict(), strict=True)
    with torch.no_grad():
        a, la = ref_layer.inverse(x.cpu())
        b, lb = mine.to(DEV).inverse(x)
    assert torch.allclose(a, b.cpu(), atol=1e-5) and torch.allclose(la, lb.cpu(), atol=1e-5)


def test_cpu_tensor_is_rejected(nfa):
    layer = nfa.flows.LULinearPermute(4)
    with pytest.raises(RuntimeError, match="no CPU path"):
        layer.inverse(torch.randn(3, 4))


# ---- fused MFMA coupling layer (csrc/rqs_fused.hip) -------------------------------------------------------------
@pytest.mark.parametrize("reverse_mask", [False, True])
@pytest.mark.parametrize("B", [1, 33, 128, 1000])
def test_fused_layer_vs_unfused_and_oracle(nfa, oracle, reverse_mask, B):
    """The fused kernel against (a) the unfused path (library GEMM + nf_rqs_coupling) and (b) the CPU oracle, on a
    strongly non-identity layer, ragged batch sizes, edge inputs, both directions."""
    torch.manual_seed(17)
    layer = nfa.flows.CoupledRationalQuadraticSpline(64, 2, 128, num_bins=8, init_identity=False,
                                                     reverse_mask=reverse_mask)
    with torch.no_grad():
        for p_ in layer.parameters():
            p_.add_(0.04 * torch.randn_like(p_))   # larger steps blow the hidden activations up (128-wide sums)
        u = layer.prqct.unconditional_transform
        u.unnormalized_widths.normal_()
        u.unnormalized_heights.normal_()
        u.unnormalized_derivatives.normal_()
    layer = layer.to(DEV)
    g = torch.Generator().manual_seed(B)
    x = 2.0 * torch.randn(B, 64, generator=g)
    x.view(-1)[:6] = torch.tensor([3.0, -3.0, 3.0000002, float("nan"), float("inf"), 0.0])[: min(6, x.numel())]
    xd = x.to(DEV)
    assert layer.prqct._fused_eligible(xd, None)
    st = {"flows.0." + k: v.detach().cpu().numpy() for k, v in layer.state_dict().items()}
    ora = oracle.OracleNSF(st, num_layers=1)
    for inverse in (True, False):
        layer.prqct.use_fused = True
        zf, ldf = (layer.inverse if inverse else layer.forward)(xd)
        layer.prqct.use_fused = False
        zu, ldu = (layer.inverse if inverse else layer.forward)(xd)
        # strongly non-identity splines: a few elements sit in bins with tiny slopes (ill-conditioned in fp32)
        assert_close(N(zf), N(zu), what="fused vs unfused z", rtol=2e-3, atol=2e-3)
        assert_close(N(ldf), N(ldu), what="fused vs unfused ld", rtol=2e-4, atol=2e-3)
        logq = np.zeros(B, np.float32)
        zo = ora.coupling(0, x.numpy(), 0 if inverse else 1, logq, +1)
        assert_close(N(zf), zo, what="fused vs oracle z", rtol=2e-3, atol=2e-3)
        assert np.mean(np.abs(N(zf) - zo)[np.isfinite(zo)] < 2e-5) > 0.98   # the bulk agrees to fp32 rounding
        assert_close(N(ldf), logq, what="fused vs oracle ld", rtol=2e-4, atol=2e-3)
        # condition-aware bound: against the oracle evaluated in DOUBLE precision the kernel may be no worse than a small
        # multiple of what the reference's own fp32 arithmetic (the fp32 oracle) loses on the same elements -- at the
        # tail quantiles and at the maximum, so that a 1e-3 regression in a few per cent of the elements cannot hide
        # behind the ill-conditioned ones
        st64 = {k: v.astype(np.float64) if v.dtype == np.float32 else v for k, v in st.items()}
        ora64 = oracle.OracleNSF(st64, num_layers=1)
        logq64 = np.zeros(B, np.float64)
        z64 = ora64.coupling(0, x.numpy().astype(np.float64), 0 if inverse else 1, logq64, +1)
        fin = np.isfinite(z64)
        e_gpu = (np.abs(N(zf).astype(np.float64) - z64) / (1 + np.abs(z64)))[fin]
        e_o32 = (np.abs(zo.astype(np.float64) - z64) / (1 + np.abs(z64)))[fin]
        for q in ((0.9, 0.99, 0.999, 1.0) if fin.any() else ()):
            assert np.quantile(e_gpu, q) <= 4 * np.quantile(e_o32, q) + 4e-6, (q, np.quantile(e_gpu, q), np.quantile(e_o32, q))
        finl = np.isfinite(logq64)
        l_gpu = (np.abs(N(ldf).astype(np.float64) - logq64) / np.maximum(1, np.abs(logq64)))[finl]
        l_o32 = (np.abs(logq.astype(np.float64) - logq64) / np.maximum(1, np.abs(logq64)))[finl]
        if finl.any():
            assert l_gpu.max() <= 4 * l_o32.max() + 2e-5, (l_gpu.max(), l_o32.max())
    # accumulate modes and repacking after a parameter update
    layer.prqct.use_fused = True
    acc = torch.full((B,), 1.5, device=DEV)
    z2 = layer._run(xd, True, acc, -1)
    zf, ldf = layer.inverse(xd)
    fin = torch.isfinite(ldf)
    assert torch.allclose(acc[fin], 1.5 - ldf[fin], atol=1e-5)
    with torch.no_grad():
        layer.prqct.transform_net.final_layer.bias.add_(0.05)
    z3, _ = layer.inverse(xd)
    layer.prqct.use_fused = False
    z4, _ = layer.inverse(xd)
    assert_close(N(z3), N(z4), what="repacked", rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("B", [5, 300])
def test_fused_pair_with_lu_vs_layerwise_and_oracle(nfa, oracle, B):
    """[CoupledRQS, LULinearPermute] issued as ONE kernel (run_chain) vs the two layers run separately vs oracle."""
    from normflows_amd.core import run_chain
    torch.manual_seed(23)
    crqs = nfa.flows.CoupledRationalQuadraticSpline(64, 2, 128, num_bins=8, init_identity=False)
    lu = nfa.flows.LULinearPermute(64, identity_init=False)
    with torch.no_grad():
        for p_ in list(crqs.parameters()) + list(lu.parameters()):
            p_.add_(0.1 * torch.randn_like(p_))
    crqs, lu = crqs.to(DEV), lu.to(DEV)
    flows = [crqs, lu]
    st = {}
    for i, f in enumerate(flows):
        st.update({"flows.%d.%s" % (i, k): v.detach().cpu().numpy() for k, v in f.state_dict().items()})
    ora = oracle.OracleNSF(st, num_layers=2)
    g = torch.Generator().manual_seed(B)
    x = 1.5 * torch.randn(B, 64, generator=g)
    xd = x.to(DEV)
    for inverse in (True, False):
        ld_f = torch.zeros(B, device=DEV)
        zf = run_chain(flows, xd, inverse, ld_f, +1)                      # fused pair
        crqs.prqct.use_fused = False
        ld_u = torch.zeros(B, device=DEV)
        zu = run_chain(flows, xd, inverse, ld_u, +1)                      # layer by layer, unfused kernels
        crqs.prqct.use_fused = True
        logq = np.zeros(B, np.float32)
        z = x.numpy()
        for i in ((1, 0) if inverse else (0, 1)):
            z = (ora.coupling if i == 0 else ora.lu)(i, z, 0 if inverse else 1, logq, +1)
        assert_close(N(zf), N(zu), what="pair vs layerwise z", rtol=1e-3, atol=1e-3)
        assert_close(N(ld_f), N(ld_u), what="pair vs layerwise ld", rtol=1e-3, atol=2e-3)
        assert_close(N(zf), z, what="pair vs oracle z", rtol=1e-3, atol=1e-3)
        assert_close(N(ld_f), logq, what="pair vs oracle ld", rtol=1e-3, atol=2e-3)
        assert np.mean(np.abs(N(zf) - z) < 5e-5) > 0.98


# ---- split-bf16 ("bf16x3") variant of the fused kernel: same fixtures, same tolerances --------------------------
@pytest.fixture
def bf16x3(nfa):
    nfa.config.set_fused_gemm("bf16x3")
    yield
    nfa.config.set_fused_gemm("f32")


@pytest.mark.parametrize("B", [33, 1000])
def test_x3_fused_layer_vs_unfused_and_oracle(nfa, oracle, bf16x3, B):
    test_fused_layer_vs_unfused_and_oracle(nfa, oracle, False, B)
    test_fused_layer_vs_unfused_and_oracle(nfa, oracle, True, B)


def test_x3_fused_pair_with_lu(nfa, oracle, bf16x3):
    test_fused_pair_with_lu_vs_layerwise_and_oracle(nfa, oracle, 300)


def test_x3_model_c2_head_vs_reference_and_exact_kernel(nfa, bf16x3):
    from bench import build_c2_model
    g = load_golden("model_c2_head")
    m = build_c2_model().to(DEV)
    lp = N(m.log_prob(T(g["x"])))
    assert _rel(lp, g["log_prob"]) < 1e-4, _rel(lp, g["log_prob"])
    nfa.config.set_fused_gemm("f32")
    lp32 = N(m.log_prob(T(g["x"])))
    # split-bf16 agrees with the exact-fp32 MFMA kernel at the fp32 rounding level
    assert _rel(lp, lp32) < 2e-5, _rel(lp, lp32)


def test_x3_model_c2_full_size_properties(nfa, oracle, bf16x3):
    test_model_c2_full_size_properties(nfa, oracle)


@pytest.mark.parametrize("inverse", [True, False])
def test_x3_chain_equals_layer_by_layer(nfa, bf16x3, inverse):
    """nf_rqs_fused_x3_chain (one persistent launch, rows resident in LDS between the layers) against one nf_rqs_fused_x3
    launch per layer pair: the same arithmetic in the same order."""
    from bench import build_c2_model
    m = build_c2_model(num_layers=5).to(DEV)
    g = torch.Generator().manual_seed(17)
    z = (torch.randn(777, 64, generator=g) * 1.5).to(DEV)
    run = m.inverse_and_log_det if inverse else m.forward_and_log_det
    try:
        with torch.no_grad():
            y1, ld1 = run(z)
            nfa.config.set_fused_chain(False)
            y2, ld2 = run(z)
    finally:
        nfa.config.set_fused_chain(True)
    assert torch.isfinite(y1).all() and torch.isfinite(ld1).all()
    assert float((y1 - y2).abs().max()) <= 1e-6 * float(y2.abs().max())
    assert float((ld1 - ld2).abs().max()) <= 1e-6 * float(ld2.abs().max()) + 1e-6


# ---- MAF (BASELINE configs[4]: inverse pass = D sequential MADE passes) ------------------------------------------
def _perturb(module, sigma, seed):
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p_ in module.parameters():
            p_.add_(sigma * torch.randn(p_.shape, generator=g, dtype=p_.dtype))


def test_maf_layer_vs_reference(nfa, oracle):
    g = load_golden("maf_d20")
    layer = load_layer(nfa.flows.MaskedAffineAutoregressive(20, 40, num_blocks=2), golden_state(g), torch.float32)
    y, ld = nfa.ops.maf_affine(T(g["x"]), T(g["params"]), 0)              # kernel alone on the reference's MADE output
    assert_close(N(y), g["z_fwd"], what="kernel z_fwd", rtol=1e-5, atol=1e-5)
    assert_close(N(ld), g["ld_fwd"], what="kernel ld_fwd", rtol=1e-5, atol=1e-5)
    z, ld = layer.forward(T(g["x"]))
    assert_close(N(z), g["z_fwd"], what="z_fwd", rtol=1e-4, atol=1e-5)
    assert_close(N(ld), g["ld_fwd"], what="ld_fwd", rtol=1e-4, atol=1e-5)
    z, ld = layer.inverse(T(g["x"]))
    assert_close(N(z), g["z_inv"], what="z_inv", rtol=1e-4, atol=1e-4)
    assert_close(N(ld), g["ld_inv"], what="ld_inv", rtol=1e-4, atol=1e-4)
    xr, ldr = layer.forward(z)                                            # round trip (autoregressive_test.py:9-28)
    assert_close(N(xr), g["x"], what="roundtrip", rtol=1e-4, atol=1e-4)


def test_maf_inverse_full_batch_round_trip_and_determinism(nfa):
    """BASELINE configs[4] geometry at the full batch (65 536 rows = 1024 waves, every CU busy): the one-pass inverse is
    bit-identical run to run and inverts the forward pass.  Size-dependent by design: the kernel's counted `vmcnt` waits
    (LDS-DMA ring, tile pairing) only misbehave under full memory load -- a too-permissive count passed every small-batch
    parity test and failed here."""
    torch.manual_seed(5)
    layer = nfa.flows.MaskedAffineAutoregressive(128, 512, num_blocks=2)
    _perturb(layer, 0.05, 9)
    layer = layer.to(DEV)
    x = torch.randn(65536, 128, device=DEV)
    with torch.no_grad():
        z1, ld1 = layer.inverse(x)
        z2, ld2 = layer.inverse(x)
        z3, ld3 = layer.inverse(x)
        xr, ldf = layer.forward(z1)
    assert torch.equal(z1, z2) and torch.equal(z1, z3) and torch.equal(ld1, ld2) and torch.equal(ld1, ld3)
    assert float((xr - x).abs().max()) < 2e-4
    assert float((ldf + ld1).abs().max()) < 2e-3


def test_maf_config5_width_vs_reference(nfa):
    """d = 128, hidden 512 (the BASELINE config-5 layer): seeded construction + the fixture's perturbation reproduce the
    reference's weights; forward and the 128-pass inverse match its outputs."""
    g = load_golden("maf_d128")
    torch.manual_seed(1000 + 128)
    layer = nfa.flows.MaskedAffineAutoregressive(128, 512, num_blocks=2)
    _perturb(layer, 0.05, 8)
    layer = layer.to(DEV)
    z, ld = layer.forward(T(g["x"]))
    assert_close(N(z), g["z_fwd"], what="z_fwd", rtol=2e-4, atol=2e-4)
    assert_close(N(ld), g["ld_fwd"], what="ld_fwd", rtol=2e-4, atol=2e-4)
    z, ld = layer.inverse(T(g["x"]))
    assert_close(N(z), g["z_inv"], what="z_inv", rtol=2e-4, atol=2e-4)
    assert_close(N(ld), g["ld_inv"], what="ld_inv", rtol=2e-4, atol=2e-4)
    # MADE mask structure (nets/made_test.py:77-105): the product of all masks is strictly lower triangular
    net = layer.autoregressive_net
    total = net.initial_layer.mask
    for blk in net.blocks:
        for lin in blk.linear_layers:
            total = lin.mask @ total
    total = net.final_layer.mask @ total
    conn = (total[0::2] > 0).float()
    assert torch.equal(conn, torch.tril(torch.ones_like(conn), -1))


@pytest.mark.parametrize("D,H,B", [(128, 512, 300), (17, 40, 64), (3, 2, 5), (40, 39, 129), (6, 150, 1),
                                   # tiles of 10-16 degrees: the shapes on which a dynamically indexed register write of the
                                   # sequential part went out of bounds in one build
                                   (32, 64, 64), (64, 128, 64), (96, 256, 64), (33, 39, 129)])
def test_maf_incremental_inverse_vs_d_pass(nfa, D, H, B):
    """nf_maf_inverse (one pass, every hidden unit finalised once) against the reference's D-pass structure
    (autoregressive.py:29-38) run through the same MADE; ragged batches; degrees with 1..32 units."""
    from normflows_amd.flows.autoregressive import Autoregressive
    torch.manual_seed(D * 1000 + H)
    layer = nfa.flows.MaskedAffineAutoregressive(D, H, num_blocks=2)
    _perturb(layer, 0.05 if D < 100 else 0.02, 3)
    layer = layer.to(DEV)
    assert layer._packed(DEV) is not None
    z = torch.randn(B, D, generator=torch.Generator().manual_seed(5)).to(DEV)
    x1, ld1 = layer.inverse(z)                       # incremental kernel
    x0, ld0 = Autoregressive.inverse(layer, z)       # D MADE passes
    assert_close(N(x1), N(x0), what="x", rtol=2e-4, atol=2e-4)
    assert_close(N(ld1), N(ld0), what="ld", rtol=2e-4, atol=2e-4)
    xr, ldr = layer.forward(x1)
    assert_close(N(xr), N(z), what="roundtrip", rtol=1e-3, atol=1e-3)
    assert_close(N(ldr), -N(ld1), what="roundtrip ld", rtol=1e-3, atol=1e-3)
    # parameter update invalidates the pack
    with torch.no_grad():
        layer.autoregressive_net.final_layer.bias.add_(0.1)
    x2, _ = layer.inverse(z)
    x3, _ = Autoregressive.inverse(layer, z)
    assert_close(N(x2), N(x3), what="x after update", rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("D,H,NB,rev,B,K", [(64, 256, 2, False, 300, 8), (128, 128, 2, False, 65, 8), (128, 256, 1, True, 129, 8),
                                            (96, 192, 2, False, 64, 8), (7, 300, 2, True, 33, 8), (66, 512, 3, False, 257, 8),
                                            (128, 512, 2, False, 4099, 8), (65, 129, 2, True, 1, 8),
                                            (64, 256, 2, False, 300, 4), (128, 128, 2, False, 65, 4), (66, 512, 3, True, 257, 4),
                                            (7, 300, 2, True, 33, 4), (128, 256, 1, True, 129, 16), (128, 128, 2, False, 193, 16),
                                            (128, 512, 2, False, 1030, 16), (30, 140, 2, False, 64, 16), (65, 129, 2, True, 1, 16)])
def test_nsf_wide_one_launch_vs_layerwise_and_oracle(nfa, oracle, D, H, NB, rev, B, K):
    """nf_nsf_wide (csrc/nsf_wide.hip: the coupling layer beyond the benchmark kernel's shapes as one launch -- ResidualNet on fp32
    MFMA with the activations on chip, the spline on the accumulator registers, the batch-shared spline on the identity half)
    against (a) the layer-wise path (library GEMMs + nf_rqs_coupling, itself pinned to the reference's fixtures) and (b) the CPU
    oracle in double precision, both directions, on a strongly non-identity layer with inputs beyond the tails.  Round 5: 4 and 16
    bins on the same schedule (nf_nsf_wide_k: 8 / 2 transform features per final-layer group instead of 4)."""
    torch.manual_seed(D * 7 + H)
    layer = nfa.flows.CoupledRationalQuadraticSpline(D, NB, H, num_bins=K, init_identity=False, reverse_mask=rev)
    with torch.no_grad():
        for p_ in layer.parameters():
            p_.add_(0.04 * torch.randn_like(p_))
        u = layer.prqct.unconditional_transform
        u.unnormalized_widths.normal_()
        u.unnormalized_heights.normal_()
        u.unnormalized_derivatives.normal_()
    layer = layer.to(DEV)
    assert layer.prqct._wide_pack(torch.zeros(1, D, device=DEV), None) is not None
    x = (1.7 * torch.randn(B, D, generator=torch.Generator().manual_seed(3))).to(DEV)
    x[0, 0] = 3.5
    x[0, D - 1] = -4.0
    if B >= 33:
        # non-finite inputs (utils/splines.py:28, :40-41: the element passes through with log-det 0): in a TRANSFORM column only that
        # element is affected -- the conditioner sees the identity features alone --, in an IDENTITY column the conditioner's
        # output, i.e. every transform column of the row, becomes NaN as in the reference; the layer-wise path is the yardstick
        x[5, 0] = float("nan")
        x[6, 1] = float("nan")
        x[7, 2] = float("inf")
        x[8, 3] = -float("inf")
    outs = {}
    for name, fn in (("inv", layer.inverse), ("fwd", layer.forward)):
        z1, ld1 = fn(x)
        nfa.config.set_nsf_wide(False)
        try:
            assert layer.prqct._wide_pack(x, None) is None
            z0, ld0 = fn(x)
        finally:
            nfa.config.set_nsf_wide(True)
        assert_close(N(z1), N(z0), what=name + " z", rtol=2e-5, atol=2e-5)
        # log-det: 1e-4 on every row but at most one in 250 -- a row with an element in a bin whose slope is several hundred
        # carries the float32 rounding of its knots amplified into the log-derivative: [66-512-3-False-257-8], sampling
        # direction, row 52 (tools/wide_row_diag.py, profiles/r06_wide_row_diag.txt): float64 evaluation of the same layer
        # -36.351265; layer-wise float32 -36.354618 (3.4e-3 off); the kernel with round 5's knot construction -36.352871
        # (1.6e-3 off), with round 6's (rqs_regs_h: the second half of the softmax summed on top of the first half's sum)
        # -36.347443 (3.8e-3 off) -- every float32 path is a few 1e-3 from the truth there, on different sides.  Such rows stay
        # within 2e-2 of each other.
        l1, l0 = N(ld1), N(ld0)
        off = ~np.isclose(l1, l0, rtol=1e-4, atol=1e-4, equal_nan=True)
        assert int(off.sum()) <= max(1, B // 250), (name, int(off.sum()), np.nonzero(off)[0][:8])
        assert_close(l1[~off], l0[~off], what=name + " ld", rtol=1e-4, atol=1e-4)
        assert_close(l1[off], l0[off], what=name + " ld (amplified rows)", rtol=2e-2, atol=2e-2)
        z2, ld2 = fn(x)
        assert torch.equal(torch.nan_to_num(z1), torch.nan_to_num(z2)) and torch.equal(torch.nan_to_num(ld1), torch.nan_to_num(ld2))   # deterministic
        acc = torch.full((B,), -1.5, device=DEV)
        fnr = layer.prqct._density if name == "inv" else layer.prqct._sample
        _, acc2 = fnr(x, None, acc.clone(), -1)
        assert torch.allclose(acc2, acc - ld1, atol=1e-6, equal_nan=True)
        outs[name] = (z1, ld1)
    if B >= 33 and not rev:          # column 1 is a transform column: the NaN stays in its own element, the row's log-det is finite
        zi_, ldi_ = outs["inv"]
        assert bool(torch.isnan(zi_[6, 1])) and int(torch.isnan(zi_[6]).sum()) == 1 and bool(torch.isfinite(ldi_[6]))
    keep = torch.isfinite(outs["inv"][0]).all(dim=1) & torch.isfinite(x).all(dim=1)
    zi, ldi = outs["inv"]
    x_all, x = x, x[keep]
    zi, ldi = zi[keep], ldi[keep]
    xr, ldr = layer.forward(zi)                                        # round trip (flow_test.py:40-48)
    # (random N(0, 1) spline logits: single bins with slopes of several hundred amplify the forward pass's 1e-6 -- the per-direction
    # comparisons above and the oracle's below are the tight ones)
    assert_close(N(xr), N(x), what="round trip", rtol=2e-3, atol=2e-3)
    assert_close(N(ldr), -N(ldi), what="round trip ld", rtol=5e-3, atol=5e-3)
    # the CPU oracle in double precision on the same weights (both directions)
    st = {"flows.0." + k: (v.detach().cpu().double().numpy() if v.is_floating_point() else v.cpu().numpy())
          for k, v in layer.state_dict().items()}
    ora = oracle.OracleNSF(st, num_layers=1, K=K, tail_bound=3.0)
    x64 = N(x_all).astype(np.float64)
    for name, direction in (("inv", 0), ("fwd", 1)):
        lq = np.zeros(B)
        zo = ora.coupling(0, x64, direction, lq, +1)
        # float32 kernel against float64 oracle: 99.9 % of the elements within 5e-5 (1 + |z|), all within 1e-3 -- with N(0, 1) spline
        # logits a handful of elements per 100 000 sit in bins whose slope amplifies float32 rounding of the conditioner output
        # (the float32 layer-wise path above agrees with the kernel to 2e-5 on every element)
        got_z, got_l = N(outs[name][0]).astype(np.float64), N(outs[name][1]).astype(np.float64)
        assert np.array_equal(np.isfinite(got_z), np.isfinite(zo)) and np.array_equal(np.isfinite(got_l), np.isfinite(lq))
        fz, fl = np.isfinite(zo), np.isfinite(lq)
        ez = (np.abs(got_z - zo) / (1.0 + np.abs(zo)))[fz]
        assert np.quantile(ez, 0.999) < 5e-5 and ez.max() < 1e-3, (name, float(np.quantile(ez, 0.999)), float(ez.max()))
        el = (np.abs(got_l - lq) / (1.0 + np.abs(lq)))[fl]
        assert np.quantile(el, 0.99) < 2e-4 and el.max() < 5e-3, (name, float(np.quantile(el, 0.99)), float(el.max()))


def test_round4_entry_points_reject_bad_arguments_and_replay_in_graphs(nfa):
    """nf_made_forward[_affine] / nf_nsf_wide[_tables] through the raw C ABI: the errno codes of include/nf_mi355x.h (-22 EINVAL,
    -14 EFAULT, -95 ENOTSUP) for bad arguments, success without touching memory for an empty batch; and both kernels inside a
    recorded hipGraph (use_graphs): replay = eager, bit for bit."""
    import ctypes as C
    from normflows_amd import _lib as L
    lib = L.lib()
    nul, one = C.c_void_p(0), C.c_void_p(16)
    st = L.stream()
    i64, i32, f64 = C.c_int64, C.c_int, C.c_double
    assert lib.nf_made_forward_affine(nul, nul, nul, nul, nul, i64(0), i32(8), i32(256), i32(0), st) == 0          # B = 0
    assert lib.nf_made_forward_affine(nul, nul, nul, nul, nul, i64(4), i32(8), i32(256), i32(0), st) == -14        # NULL buffers
    assert lib.nf_made_forward_affine(one, one, one, one, one, i64(4), i32(200), i32(256), i32(0), st) == -22      # D > 128
    assert lib.nf_made_forward_affine(one, one, one, one, one, i64(4), i32(8), i32(300), i32(0), st) == -95        # hidden_padded
    assert lib.nf_made_forward_affine(one, one, one, one, one, i64(4), i32(8), i32(256), i32(7), st) == -22        # acc
    assert lib.nf_made_forward(one, one, one, one, i64(4), i32(8), i32(512), i32(0), st) == -22                    # mult < 1
    assert lib.nf_nsf_wide(nul, nul, nul, nul, nul, nul, nul, i64(0), i32(64), i32(256), i32(0), i32(0), f64(3.0), f64(1e-3),
                           f64(1e-3), f64(1e-3), st) == 0
    assert lib.nf_nsf_wide(nul, nul, nul, nul, nul, nul, nul, i64(5), i32(64), i32(256), i32(0), i32(0), f64(3.0), f64(1e-3),
                           f64(1e-3), f64(1e-3), st) == -14
    assert lib.nf_nsf_wide(one, one, one, one, one, one, nul, i64(5), i32(64), i32(256), i32(2), i32(0), f64(3.0), f64(1e-3),
                           f64(1e-3), f64(1e-3), st) == -22                                                        # direction
    assert lib.nf_nsf_wide(one, one, one, one, one, one, nul, i64(5), i32(64), i32(192), i32(0), i32(0), f64(3.0), f64(1e-3),
                           f64(1e-3), f64(1e-3), st) == -95                                                        # hidden_padded
    assert lib.nf_nsf_wide(one, one, one, one, one, one, nul, i64(5), i32(64), i32(256), i32(0), i32(0), f64(3.0), f64(0.2),
                           f64(1e-3), f64(1e-3), st) == -22                                  # min_bin_width * 8 > 1 (utils/splines.py:121-124)
    assert lib.nf_nsf_wide_tables(one, one, one, one, i32(32), i32(10), f64(3.0), f64(1e-3), f64(1e-3), f64(1e-3), st) == -95   # bins
    assert lib.nf_nsf_wide_k(one, one, one, one, one, one, nul, i64(5), i32(64), i32(256), i32(10), i32(0), i32(0), f64(3.0), f64(1e-3),
                             f64(1e-3), f64(1e-3), st) == -95                                                      # bins: 4 | 8 | 16
    assert lib.nf_nsf_wide_k(one, one, one, one, one, one, nul, i64(5), i32(64), i32(256), i32(16), i32(0), i32(0), f64(3.0), f64(0.07),
                             f64(1e-3), f64(1e-3), st) == -22                                                      # min_bin_width * 16 > 1
    assert lib.nf_nsf_wide_k(nul, nul, nul, nul, nul, nul, nul, i64(0), i32(64), i32(256), i32(4), i32(0), i32(0), f64(3.0), f64(1e-3),
                             f64(1e-3), f64(1e-3), st) == 0
    assert lib.nf_nsf_wide_tables(one, one, one, one, i32(80), i32(8), f64(3.0), f64(1e-3), f64(1e-3), f64(1e-3), st) == -22    # > 64 features
    # hipGraph replay of a MAF model's forward direction and of a wide NSF model's log_prob
    torch.manual_seed(3)
    maf = nfa.NormalizingFlow(nfa.distributions.DiagGaussian(24, trainable=False),
                              [nfa.flows.MaskedAffineAutoregressive(24, 96, num_blocks=2) for _ in range(3)])
    _perturb(maf, 0.05, 2)
    maf = maf.to(DEV)
    flows = []
    for _ in range(2):
        flows += [nfa.flows.CoupledRationalQuadraticSpline(80, 2, 200, num_bins=8), nfa.flows.LULinearPermute(80)]
    wide = nfa.NormalizingFlow(nfa.distributions.DiagGaussian(80, trainable=False), flows)
    _perturb(wide, 0.03, 4)
    wide = wide.to(DEV)
    eps = torch.randn(500, 24, device=DEV)
    x = torch.randn(700, 80, device=DEV)
    e1 = maf.sample_from_noise(eps)
    e2 = wide.log_prob(x)
    maf.use_graphs(True)
    wide.use_graphs(True)
    try:
        for _ in range(2):
            g1 = maf.sample_from_noise(eps)
            g2 = wide.log_prob(x)
        assert torch.equal(g1[0], e1[0]) and torch.equal(g1[1], e1[1]) and torch.equal(g2, e2)
    finally:
        maf.use_graphs(False)
        wide.use_graphs(False)


def test_nsf_wide_full_batch_properties(nfa):
    """nf_nsf_wide at the BASELINE batch (65 536 rows = 1024 / 512 tiles, every CU busy, persistent workgroups walking several tiles
    with the weight ring wrapping between them): size-independent properties -- run-to-run bit equality, log_prob(sample) = the
    sampler's log_q (core_test.py:144-196), the inverse pass undoes the forward pass -- on a 4-pair model of each tile geometry."""
    for D, H in ((64, 256), (128, 128), (128, 512)):
        torch.manual_seed(D + H)
        flows = []
        for _ in range(4):
            flows += [nfa.flows.CoupledRationalQuadraticSpline(D, 2, H, num_bins=8), nfa.flows.LULinearPermute(D, identity_init=False)]
        m = nfa.NormalizingFlow(nfa.distributions.DiagGaussian(D, trainable=False), flows)
        _perturb(m, 0.01, 7)
        m = m.to(DEV)
        assert flows[0]._pair_eligible(torch.zeros(2, D, device=DEV), flows[1]) and not flows[0].prqct._fused_eligible(torch.zeros(2, D, device=DEV), None)
        eps = torch.randn(65536, D, generator=torch.Generator().manual_seed(1)).to(DEV)
        xs, lq = m.sample_from_noise(eps)
        xs2, lq2 = m.sample_from_noise(eps)
        assert torch.equal(xs, xs2) and torch.equal(lq, lq2)
        lp = m.log_prob(xs)
        assert torch.equal(lp, m.log_prob(xs))
        assert _rel(N(lp), N(lq)) < 1e-4, (D, H, _rel(N(lp), N(lq)))
        z, _ = m.inverse_and_log_det(xs)
        z0 = m.q0.loc.reshape(1, -1) + torch.exp(m.q0.log_scale.reshape(1, -1)) * eps
        assert float((z - z0).abs().max()) < 2e-3, (D, H, float((z - z0).abs().max()))


def test_one_launch_engines_on_random_shapes(nfa):
    """Seeded sweep over shapes for the three kernels of the 64-row-tile engine (nf_made_forward_affine, nf_made_forward_spline,
    nf_nsf_wide with and without the fused LU): feature counts 2..128 (odd ones, off every granule), hidden widths 1..512 (all
    three padded widths), 1..4 residual blocks, batches off the tile height -- each against the layer-wise path on the same weights."""
    rng = np.random.RandomState(20240924)
    for trial in range(36):
        kind = trial % 3
        D = int(rng.choice([2, 3, 5, 8, 17, 31, 32, 33, 64, 65, 100, 127, 128]))
        H = int(rng.choice([1, 7, 32, 100, 128, 129, 200, 256, 257, 400, 512]))
        NB = int(rng.randint(1, 5))
        B = int(rng.choice([1, 63, 64, 65, 127, 129, 200, 1000]))
        torch.manual_seed(1000 + trial)
        x = (1.5 * torch.randn(B, D, generator=torch.Generator().manual_seed(trial))).to(DEV)
        what = (kind, D, H, NB, B)
        if kind == 0:
            layer = nfa.flows.MaskedAffineAutoregressive(D, H, num_blocks=NB)
            _perturb(layer, 0.05, trial)
            layer = layer.to(DEV)
            assert layer.autoregressive_net.packed_forward(DEV) is not None, what
            fn, off = layer.forward, nfa.config.set_made_fused
        elif kind == 1:
            layer = nfa.flows.AutoregressiveRationalQuadraticSpline(D, NB, H, num_bins=8, init_identity=False)
            _perturb(layer, 0.05, trial)
            layer = layer.to(DEV)
            assert layer.mprqat.autoregressive_net.packed_forward(DEV, spline=True) is not None, what
            fn, off = layer.inverse, nfa.config.set_made_fused
        else:
            c = nfa.flows.CoupledRationalQuadraticSpline(D, NB, H, num_bins=8, init_identity=False, reverse_mask=bool(trial & 1))
            lu = nfa.flows.LULinearPermute(D, identity_init=False)
            m = nfa.NormalizingFlow(nfa.distributions.DiagGaussian(D, trainable=False), [c, lu])
            _perturb(m, 0.03, trial)
            m = m.to(DEV)
            if c.prqct._fused_eligible(x, None):          # (the benchmark kernel's range: not this test's subject)
                continue
            assert c.prqct._wide_pack(x, None) is not None, what
            fn = (lambda t, m=m: m.inverse_and_log_det(t)) if trial & 2 else (lambda t, m=m: m.forward_and_log_det(t))
            off = nfa.config.set_nsf_wide
        z1, l1 = fn(x)
        off(False)
        try:
            z0, l0 = fn(x)
        finally:
            off(True)
        ez = (np.abs(N(z1).astype(np.float64) - N(z0)) / (1.0 + np.abs(N(z0)))).ravel()
        el = (np.abs(N(l1).astype(np.float64) - N(l0)) / (1.0 + np.abs(N(l0)))).ravel()
        assert np.isfinite(ez).all() and np.isfinite(el).all(), what
        assert np.quantile(ez, 0.999) < 5e-5 and ez.max() < 2e-3 and np.quantile(el, 0.99) < 2e-4 and el.max() < 5e-3, \
            (what, float(ez.max()), float(el.max()))


@pytest.mark.parametrize("D,H", [(64, 256), (128, 128), (128, 256), (96, 192)])
def test_nsf_wide_models_vs_reference(nfa, monkeypatch, D, H):
    """nf_nsf_wide against the REFERENCE itself at the shapes it exists for (tests/golden/model_nsf_wide_*.npz: 3 x
    [CoupledRationalQuadraticSpline(D, 2, H) + LULinearPermute(D)], sigma 0.05, 96 rows; wrapper.py:20-35, nets/resnet.py:53-104,
    mixing.py:535-563 under core.py:167-197): weights rebuilt from the seed, log_prob and the sampling pass to 1e-4 of the float32 leg
    and no further from the float64 leg than 4 x the reference's own float32 leg; the one-launch kernel is what ran (spy)."""
    from bench import build_c2_model
    from normflows_amd import ops
    g = load_golden("model_nsf_wide_d%d_h%d" % (D, H))
    m = build_c2_model(num_layers=3, dim=D, hidden=H, seed=40 + D, sigma=0.05).to(DEV)
    calls = []
    real = ops.nsf_wide
    monkeypatch.setattr(ops, "nsf_wide", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    lp = N(m.log_prob(T(g["x"])))
    assert len(calls) == 3, calls
    xs, lq = m.sample_from_noise(T(g["eps"]))
    assert len(calls) == 6
    assert _rel(lp, g["log_prob_f32"]) < 1e-4, _rel(lp, g["log_prob_f32"])
    assert_close(N(xs), g["sample_f32"], what="sample", rtol=1e-4, atol=1e-4)
    assert _rel(N(lq), g["sample_logq_f32"]) < 1e-4
    for ours, key in ((lp, "log_prob"), (N(lq), "sample_logq")):
        own = _rel(g[key + "_f32"], g[key + "_f64"])
        assert _rel(ours, g[key + "_f64"]) <= max(4 * own, 2e-6), (key, _rel(ours, g[key + "_f64"]), own)


@pytest.mark.parametrize("D,H", [(64, 256), (128, 128), (96, 192), (33, 300)])
def test_nsf_wide_pairs_with_fused_lu_vs_layerwise(nfa, D, H):
    """[CoupledRationalQuadraticSpline, LULinearPermute] pairs beyond the benchmark kernel's shapes: nf_nsf_wide with the LU layer's
    dense matrix in the same launch (density: LU first, core.py:193-195; sampling: LU last, core.py:177-179) against the layer-wise
    path on a 3-pair model: log_prob, sample, log_prob(sample) = log_q (core_test.py:144-196)."""
    torch.manual_seed(D + H)
    flows = []
    for _ in range(3):
        flows += [nfa.flows.CoupledRationalQuadraticSpline(D, 2, H, num_bins=8), nfa.flows.LULinearPermute(D, identity_init=False)]
    m = nfa.NormalizingFlow(nfa.distributions.DiagGaussian(D, trainable=False), flows)
    _perturb(m, 0.03, 5)
    m = m.to(DEV)
    x = torch.randn(1000, D, generator=torch.Generator().manual_seed(2)).to(DEV)
    eps = torch.randn(777, D, generator=torch.Generator().manual_seed(3)).to(DEV)
    lp1 = m.log_prob(x)
    xs1, lq1 = m.sample_from_noise(eps)
    nfa.config.set_nsf_wide(False)
    try:
        lp0 = m.log_prob(x)
        xs0, lq0 = m.sample_from_noise(eps)
    finally:
        nfa.config.set_nsf_wide(True)
    assert _rel(N(lp1), N(lp0)) < 1e-4, _rel(N(lp1), N(lp0))
    assert_close(N(xs1), N(xs0), what="sample", rtol=1e-4, atol=1e-4)
    assert _rel(N(lq1), N(lq0)) < 1e-4
    assert _rel(N(m.log_prob(xs1)), N(lq1)) < 1e-4
    # a parameter update of the LU layer invalidates the pair's pack
    with torch.no_grad():
        flows[1].linear.bias.add_(0.05)
    lp2 = m.log_prob(x)
    nfa.config.set_nsf_wide(False)
    try:
        lp3 = m.log_prob(x)
    finally:
        nfa.config.set_nsf_wide(True)
    assert _rel(N(lp2), N(lp3)) < 1e-4 and _rel(N(lp2), N(lp1)) > 1e-6


@pytest.mark.parametrize("D,H,NB,B", [(128, 512, 2, 300), (128, 512, 2, 64), (20, 40, 2, 130), (6, 300, 1, 7), (33, 256, 3, 65),
                                      (3, 2, 2, 1), (64, 257, 2, 129), (127, 512, 1, 4100)])
def test_made_forward_one_launch_vs_layerwise(nfa, D, H, NB, B):
    """nf_made_forward_affine / nf_made_forward (csrc/made_fwd.hip: the whole MADE pass on fp32 MFMA over the non-zero blocks of
    the masks, units sorted by degree, + the affine epilogue) against the layer-by-layer path (library GEMMs on weight * mask,
    nets/made.py:80-81, 296-304, then nf_maf_affine): both hidden widths of the kernel (Hp = 256 / 512), 1-3 residual blocks,
    feature counts off the 4- and 8-column granules, batches off the 64-row tile."""
    torch.manual_seed(D * 1000 + H + NB)
    layer = nfa.flows.MaskedAffineAutoregressive(D, H, num_blocks=NB)
    _perturb(layer, 0.05 if D < 100 else 0.02, 3)
    layer = layer.to(DEV)
    made = layer.autoregressive_net
    assert made.packed_forward(DEV) is not None
    x = torch.randn(B, D, generator=torch.Generator().manual_seed(5)).to(DEV)
    z1, ld1 = layer.forward(x)                       # one launch
    p1 = made(x)                                     # nf_made_forward: raw parameters
    nfa.config.set_made_fused(False)
    try:
        assert made.packed_forward(DEV) is None
        z0, ld0 = layer.forward(x)
        p0 = made(x)
    finally:
        nfa.config.set_made_fused(True)
    assert_close(N(p1), N(p0), what="params", rtol=1e-4, atol=1e-4)
    assert_close(N(z1), N(z0), what="z", rtol=1e-4, atol=1e-4)
    assert_close(N(ld1), N(ld0), what="ld", rtol=1e-4, atol=1e-4)
    # accumulate into a caller's log-density, and bit-reproducible
    acc = torch.full((B,), 2.0, device=DEV)
    _, acc2 = nfa.ops.made_forward_affine(x, *made.packed_forward(DEV)[:3], logdet=acc.clone(), acc=-1)
    assert torch.equal(acc2, acc - ld1)
    z2, ld2 = layer.forward(x)
    assert torch.equal(z1, z2) and torch.equal(ld1, ld2)
    # a parameter update invalidates the pack
    with torch.no_grad():
        made.final_layer.bias.add_(0.1)
    z3, _ = layer.forward(x)
    nfa.config.set_made_fused(False)
    try:
        z4, _ = layer.forward(x)
    finally:
        nfa.config.set_made_fused(True)
    assert_close(N(z3), N(z4), what="z after update", rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("D,H,NB,B", [(64, 256, 2, 300), (6, 16, 2, 11), (33, 300, 1, 65), (128, 512, 2, 130), (5, 40, 3, 1)])
def test_made_forward_spline_one_launch_vs_layerwise(nfa, D, H, NB, B):
    """nf_made_forward_spline (the autoregressive spline layer's density direction as one launch: MADE on fp32 MFMA over the masks'
    non-zero blocks, final layer in groups of four features, the spline on the accumulator registers) against the layer-wise
    path (one-launch MADE or library GEMMs + nf_rqs_coupling): both hidden widths, 1-3 blocks, odd feature counts, ragged batches,
    inputs beyond the tails and non-finite inputs."""
    torch.manual_seed(D * 11 + H)
    layer = nfa.flows.AutoregressiveRationalQuadraticSpline(D, NB, H, num_bins=8, tail_bound=3, init_identity=False)
    _perturb(layer, 0.2 if D < 40 else 0.05, 9)
    layer = layer.to(DEV)
    t = layer.mprqat
    assert t.autoregressive_net.packed_forward(DEV, spline=True) is not None
    x = (1.7 * torch.randn(B, D, generator=torch.Generator().manual_seed(D))).to(DEV)
    x[0, 0] = 3.5
    if B > 8:
        x[1, D - 1] = -4.0
        x[2, 1] = float("nan")
        x[3, 0] = float("inf")
    z1, ld1 = layer.inverse(x)
    nfa.config.set_made_fused(False)
    try:
        z0, ld0 = layer.inverse(x)
    finally:
        nfa.config.set_made_fused(True)
    # strongly non-identity splines: single elements sit in bins whose slope amplifies the float32 rounding of the conditioner output
    # (two summation orders): 99.9 % of the elements within 5e-5 (1 + |z|), every element within 2e-3; non-finite patterns identical
    a, b_ = N(z1).astype(np.float64), N(z0).astype(np.float64)
    assert np.array_equal(np.isfinite(a), np.isfinite(b_))
    ez = (np.abs(a - b_) / (1.0 + np.abs(b_)))[np.isfinite(b_)]
    assert np.quantile(ez, 0.999) < 5e-5 and ez.max() < 2e-3, (float(np.quantile(ez, 0.999)), float(ez.max()))
    la, lb = N(ld1).astype(np.float64), N(ld0).astype(np.float64)
    assert np.array_equal(np.isfinite(la), np.isfinite(lb))
    el = (np.abs(la - lb) / (1.0 + np.abs(lb)))[np.isfinite(lb)]
    assert np.quantile(el, 0.99) < 2e-4 and el.max() < 5e-3, (float(np.quantile(el, 0.99)), float(el.max()))
    z2, ld2 = layer.inverse(x)
    assert torch.equal(torch.nan_to_num(z1), torch.nan_to_num(z2)) and torch.equal(torch.nan_to_num(ld1), torch.nan_to_num(ld2))
    acc = torch.full((B,), 0.5, device=DEV)
    _, acc2 = nfa.ops.made_forward_spline(x, *t.autoregressive_net.packed_forward(DEV, spline=True)[:3], 3.0, logdet=acc.clone(), acc=-1)
    assert torch.allclose(acc2, acc - ld1, atol=1e-6, equal_nan=True)
    if D <= 8:      # (long autoregressive chains amplify float32 rounding feature by feature: the round trip is only tight for few features)
        fin = torch.isfinite(x).all(dim=1)
        xr, ldr = layer.forward(z1[fin])                # the one-pass sampling kernel undoes it (autoregressive_test.py round trip)
        assert_close(N(xr), N(x[fin]), what="round trip", rtol=2e-3, atol=2e-3)


def test_made_forward_raw_parameters_for_the_spline_layer(nfa):
    """nf_made_forward with 23 outputs per feature (the autoregressive spline layer's MADE, neural_spline/autoregressive.py:57-73):
    several rounds of final row-blocks, output rows in the reference's order; AR-NSF's density direction rides on it."""
    torch.manual_seed(11)
    made = nfa.nets.MADE(features=16, hidden_features=96, num_blocks=2, output_multiplier=23)
    _perturb(made, 0.05, 4)
    made = made.to(DEV)
    x = torch.randn(77, 16, device=DEV)
    assert made.packed_forward(DEV) is not None
    p1 = made(x)
    nfa.config.set_made_fused(False)
    try:
        p0 = made(x)
    finally:
        nfa.config.set_made_fused(True)
    assert p1.shape == (77, 16 * 23)
    assert_close(N(p1), N(p0), what="params", rtol=1e-4, atol=1e-4)


def test_maf_incremental_unsupported_falls_back_to_d_pass(nfa):
    """hidden < D-1 leaves degrees without units: outside the kernel's structure -> the D-pass loop is used."""
    torch.manual_seed(0)
    layer = nfa.flows.MaskedAffineAutoregressive(12, 4, num_blocks=2).to(DEV)
    assert layer._packed(DEV) is None
    z = torch.randn(7, 12, device=DEV)
    x, ld = layer.inverse(z)
    xr, _ = layer.forward(x)
    assert_close(N(xr), N(z), what="roundtrip", rtol=1e-4, atol=1e-4)


# ---- AR-NSF (neural_spline/autoregressive.py, wrapper.py:188-245) ------------------------------------------------
@pytest.mark.parametrize("name,d,hidden,K,ident", [("arnsf_d6", 6, 16, 8, False), ("arnsf_d5_ident", 5, 12, 4, True)])
def test_arnsf_wrapper_vs_reference(nfa, name, d, hidden, K, ident):
    g = load_golden(name)
    layer = load_layer(nfa.flows.AutoregressiveRationalQuadraticSpline(d, 2, hidden, num_bins=K, tail_bound=3,
                                                                       init_identity=ident), golden_state(g), torch.float32)
    z, ld = layer.inverse(T(g["x"]))                 # density direction: one MADE pass + spline kernel
    assert_close(N(z), g["z_inv"], what="z_inv", rtol=1e-4, atol=1e-5)
    assert_close(N(ld), g["ld_inv"], what="ld_inv", rtol=1e-4, atol=1e-5)
    z, ld = layer.forward(T(g["x"]))                 # generative direction: D MADE passes
    assert_close(N(z), g["z_fwd"], what="z_fwd", rtol=1e-4, atol=1e-4)
    assert_close(N(ld), g["ld_fwd"], what="ld_fwd", rtol=1e-4, atol=1e-4)
    # round trip: the two directions are different kernels (incremental one-pass inverse vs one MADE pass + spline), so
    # z's 1e-5-level error is amplified by the spline's slope on the way back
    xr, ldr = layer.inverse(z)
    assert_close(N(xr), g["x"], what="roundtrip", rtol=1e-3, atol=1e-3)
    assert_close(N(ldr), -N(ld), what="roundtrip ld", rtol=1e-3, atol=2e-3)


def test_arnsf_transform_without_tails_vs_reference(nfa):
    g = load_golden("arnsf_notails")
    t = load_layer(nfa.flows.MaskedPiecewiseRationalQuadraticAutoregressive(4, 10, num_bins=5, tails=None, num_blocks=2,
                                                                            init_identity=False),
                   golden_state(g), torch.float32)
    z, ld = t.forward(T(g["x"]))
    assert_close(N(z), g["z_fwd"], what="z_fwd", rtol=1e-4, atol=1e-5)
    assert_close(N(ld), g["ld_fwd"], what="ld_fwd", rtol=1e-4, atol=1e-5)
    z, ld = t.inverse(T(g["x"]))
    assert_close(N(z), g["z_inv"], what="z_inv", rtol=1e-4, atol=1e-4)
    assert_close(N(ld), g["ld_inv"], what="ld_inv", rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("D,H,K,tails,B", [(64, 256, 8, "linear", 1000), (33, 96, 10, "linear", 130),
                                           (7, 20, 4, None, 65), (5, 12, 10, "circular", 64), (3, 2, 1, "linear", 1),
                                           (40, 300, 11, "linear", 200)])
def test_arnsf_incremental_inverse_matches_d_pass(nfa, D, H, K, tails, B):
    """nf_arnsf_inverse (one pass, every hidden unit finalised once) against the D-pass loop of
    affine/autoregressive.py:29-38 run through the unfused kernels, and the density direction as its inverse."""
    from normflows_amd.flows.autoregressive import Autoregressive
    torch.manual_seed(D + K)
    t = nfa.flows.MaskedPiecewiseRationalQuadraticAutoregressive(D, H, num_bins=K, tails=tails, tail_bound=2.5,
                                                                 num_blocks=2, init_identity=False).to(DEV)
    with torch.no_grad():
        for p in t.parameters():
            p.mul_(1.5)
    z = torch.rand(B, D, device=DEV) if tails is None else 2.0 * torch.randn(B, D, device=DEV)
    assert t._packed(DEV) is not None
    x, ld = t.inverse(z)
    xr, ldr = Autoregressive.inverse(t, z)
    assert_close(N(x), N(xr), what="x", rtol=1e-4, atol=2e-4)
    assert_close(N(ld), N(ldr), what="ld", rtol=1e-4, atol=1e-3)
    zb, ldb = t.forward(x)
    inb = N((z.abs() <= 2.5).all(1)) if tails is not None else np.ones(B, dtype=bool)
    assert_close(N(zb)[inb], N(z)[inb], what="roundtrip", rtol=1e-3, atol=1e-3)
    assert_close(N(ldb)[inb], -N(ld)[inb], what="roundtrip ld", rtol=1e-3, atol=2e-3)
    # accumulate protocol and cache invalidation on a parameter update
    acc = torch.ones(B, device=DEV)
    pk = t._packed(DEV)
    nfa.ops.arnsf_inverse(z, pk[0], pk[1], pk[2], K, tails, 2.5, logdet=acc, acc=nfa.ops.L.LD_SUB)
    assert_close(N(acc), 1.0 - N(ld), what="acc", rtol=1e-5, atol=1e-5)
    with torch.no_grad():
        t.autoregressive_net.final_layer.bias.add_(0.1)
    x2, _ = t.inverse(z)
    x2r, _ = Autoregressive.inverse(t, z)
    assert_close(N(x2), N(x2r), what="x after update", rtol=1e-4, atol=2e-4)
    assert K == 1 or not np.allclose(N(x2), N(x))   # one bin with linear tails is the identity whatever the parameters


def test_arnsf_in_normalizing_flow(nfa):
    """Stack of AR-NSF layers + LULinearPermute inside NormalizingFlow: sample() log_q == log_prob (core_test.py:187)."""
    torch.manual_seed(3)
    flows = []
    for _ in range(3):
        flows += [nfa.flows.AutoregressiveRationalQuadraticSpline(5, 2, 16, num_bins=6, init_identity=False),
                  nfa.flows.LULinearPermute(5)]
    m = nfa.NormalizingFlow(nfa.distributions.DiagGaussian(5, trainable=False), flows).to(DEV)
    x, lq = m.sample(64)
    assert_close(N(m.log_prob(x)), N(lq), what="log_prob(sample)", rtol=2e-4, atol=2e-4)


# ---- image-side glue: Logit transform, class-conditional base, both inside MultiscaleFlow ----------
def test_logit_transform_vs_reference(nfa):
    g = load_golden("logit_transform")
    t = nfa.transforms.Logit(alpha=0.05)
    x, ld = t.inverse(T(g["u"]))
    assert_close(N(x), g["x_inv"], what="x_inv", rtol=1e-5, atol=1e-5)
    assert_close(N(ld), g["ld_inv"], what="ld_inv", rtol=1e-5, atol=1e-4)
    x, ld = t.forward(T(g["v"]))
    assert_close(N(x), g["x_fwd"], what="x_fwd", rtol=1e-5, atol=1e-6)
    assert_close(N(ld), g["ld_fwd"], what="ld_fwd", rtol=1e-5, atol=1e-4)
    xr, ldr = t.forward(t.inverse(T(g["u"]))[0])
    assert_close(N(xr), g["u"], what="roundtrip", rtol=1e-5, atol=1e-6)
    x64, ld64 = t.inverse(T(g["u"]).double())
    assert x64.dtype == torch.float64
    assert_close(N(x64), g["x_inv"].astype(np.float64), what="fp64", rtol=1e-5, atol=1e-5)


def test_class_cond_diag_gaussian_vs_reference(nfa):
    g = load_golden("class_cond_gauss")
    q = nfa.distributions.ClassCondDiagGaussian((3, 2, 2), 4)
    q.load_state_dict({k: torch.from_numpy(v) for k, v in golden_state(g).items()}, strict=True)
    q = q.to(DEV)
    y = torch.from_numpy(g["y"]).to(DEV)
    with torch.no_grad():
        assert_close(N(q.log_prob(T(g["z"]), y)), g["log_prob"], what="labels", rtol=1e-5, atol=1e-5)
        assert_close(N(q.log_prob(T(g["z"]), T(g["ysoft"]))), g["log_prob_soft"], what="soft labels", rtol=1e-5, atol=1e-5)
        q.temperature = 0.7
        assert_close(N(q.log_prob(T(g["z"]), y)), g["log_prob_temp"], what="temperature", rtol=1e-5, atol=1e-5)
        q.temperature = None
        torch.manual_seed(1)
        z, lp = q(y=y)                                   # forward: sample + its log-density (base.py:296-324)
        assert z.shape == (7, 3, 2, 2)
        assert_close(N(q.log_prob(z, y)), N(lp), what="sample log_p", rtol=1e-4, atol=1e-4)
        bad = q.log_prob(T(g["z"]), torch.full((7,), 9, device=DEV))
        assert torch.isnan(bad).all()                    # out-of-range label: NaN, not a wild read


def test_model_glow_classcond_vs_reference(nfa):
    """examples/glow.ipynb structure (reduced): GlowBlocks + Squeeze + Merge, ClassCondDiagGaussian bases, Logit
    transform; log_prob(x, y) before / after the data-dependent ActNorm init; bits per dim helper runs."""
    g = load_golden("model_glow_classcond")
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
    m = nfa.MultiscaleFlow(q0, flows, merges, transform=nfa.transforms.Logit(0.05), class_cond=True)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in golden_state(g, "sd0__").items()}, strict=True)
    m = m.to(DEV)
    y = torch.from_numpy(g["y"]).to(DEV)
    lp = N(m.log_prob(T(g["x"]), y))
    assert _rel(lp, g["log_prob"]) < 2e-4, _rel(lp, g["log_prob"])
    assert _rel(N(m.log_prob(T(g["x"]), y)), g["log_prob_second"]) < 2e-4
    ref_sd = golden_state(g, "sd__")                     # ActNorm parameters after the init agree with the reference's
    for k, v in m.state_dict().items():
        assert_close(N(v).astype(np.float64), ref_sd[k].astype(np.float64), what=k, rtol=2e-3, atol=2e-3)
    torch.manual_seed(0)
    xs, lq = m.sample(y=y)
    assert _rel(N(m.log_prob(xs, y)), N(lq)) < 1e-3
    b = nfa.utils.bitsPerDim(m, T(g["x"]), y)
    assert b.shape == (6,) and torch.isfinite(b).all()


# ---- batch-size edge cases through every layer type ------------------------------------------------------------------
def _edge_layers(nfa):
    torch.manual_seed(11)
    mk = lambda f: f.to(DEV)
    b = torch.tensor([1.0, 0.0, 1.0, 0.0, 1.0, 0.0])
    return [
        ("crqs_unfused", mk(nfa.flows.CoupledRationalQuadraticSpline(6, 1, 16, num_bins=4)), (6,)),
        ("crqs_fused_shape", mk(nfa.flows.CoupledRationalQuadraticSpline(64, 2, 128, num_bins=8)), (64,)),
        ("lu", mk(nfa.flows.LULinearPermute(6)), (6,)),
        ("lu64", mk(nfa.flows.LULinearPermute(64)), (64,)),
        ("masked_affine", mk(nfa.flows.MaskedAffineFlow(b, nfa.nets.MLP([6, 8, 6], init_zeros=False),
                                                       nfa.nets.MLP([6, 8, 6], init_zeros=False))), (6,)),
        ("maf", mk(nfa.flows.MaskedAffineAutoregressive(6, 16, num_blocks=2)), (6,)),
        ("arnsf", mk(nfa.flows.AutoregressiveRationalQuadraticSpline(5, 2, 12, num_bins=4)), (5,)),
        ("glowblock", mk(nfa.flows.GlowBlock(4, 8, init_zeros=False)), (4, 4, 4)),
        ("squeeze", mk(nfa.flows.Squeeze()), (4, 4, 4)),
        ("logit", nfa.transforms.Logit(0.05), (3, 2, 2)),
    ]


@pytest.mark.parametrize("B", [0, 1, 63, 257])
def test_batch_size_edges_every_layer(nfa, B):
    """Empty, single-row and ragged batches (tile / wave / workgroup remainders) through every layer type: shapes,
    finite values, round trip, and row-wise agreement with a larger batch (no cross-row leakage)."""
    for name, layer, shape in _edge_layers(nfa):
        g = torch.Generator().manual_seed(5)
        big = torch.randn((300,) + shape, generator=g).to(DEV)
        if name == "logit":
            big = torch.sigmoid(big) * 0.98 + 0.01
        if name == "glowblock":
            layer.inverse(big)                      # ActNorm data-dependent init on a fixed batch first
        x = big[:B].contiguous()
        z, ld = layer.inverse(x)
        assert z.shape[0] == B, name
        per_row = torch.is_tensor(ld) and ld.dim() == 1 and ld.shape[0] == B
        assert per_row or not torch.is_tensor(ld) or ld.numel() == 1, name      # (B,) or the reference's 0-dim / int 0
        zb, ldb = layer.inverse(big)
        if B:
            assert torch.isfinite(z).all(), name
            assert_close(N(z), N(zb[:B]), what=name + " rows", rtol=1e-5, atol=1e-5)
            if per_row:
                assert_close(N(ld), N(ldb[:B]), what=name + " ld rows", rtol=1e-5, atol=1e-5)
            xr, ldr = layer.forward(z)
            assert_close(N(xr), N(x), what=name + " roundtrip", rtol=2e-3, atol=2e-3)


def test_model_with_empty_batch(nfa):
    m = nfa.NormalizingFlow(nfa.distributions.DiagGaussian(64, trainable=False),
                            [nfa.flows.CoupledRationalQuadraticSpline(64, 2, 128), nfa.flows.LULinearPermute(64)] * 2).to(DEV)
    lp = m.log_prob(torch.empty(0, 64, device=DEV))
    assert lp.shape == (0,)
    x, lq = m.sample(0)
    assert x.shape == (0, 64) and lq.shape == (0,)


# ---- remaining classes of the hot-path files -------------------------------------------------------------------------
def test_cc_affine_const_vs_reference(nfa):
    g = load_golden("cc_affine_const")
    cc = nfa.flows.CCAffineConst((3, 1, 1), 4)
    cc.load_state_dict({k: torch.from_numpy(v) for k, v in golden_state(g).items()}, strict=True)
    cc = cc.to(DEV)
    z, ld = cc.forward(T(g["z"]), T(g["y"]))
    assert_close(N(z), g["z_fwd"], what="z_fwd", rtol=1e-5, atol=1e-5)
    assert_close(N(ld), g["ld_fwd"], what="ld_fwd", rtol=1e-5, atol=1e-5)
    z, ld = cc.inverse(T(g["z"]), T(g["y"]))
    assert_close(N(z), g["z_inv"], what="z_inv", rtol=1e-5, atol=1e-5)
    assert_close(N(ld), g["ld_inv"], what="ld_inv", rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("use_lu", [True, False])
def test_invertible_affine_vs_reference(nfa, use_lu):
    g = load_golden("invertible_affine_lu%d" % int(use_lu))
    ia = nfa.flows.InvertibleAffine(7, use_lu=use_lu)
    ia.load_state_dict({k: torch.from_numpy(v) for k, v in golden_state(g).items()}, strict=True)
    ia = ia.to(DEV)
    z, ld = ia.forward(T(g["z"]))
    assert_close(N(z), g["z_fwd"], what="z_fwd", rtol=1e-4, atol=1e-5)
    assert_close(N(ld), g["ld_fwd"], what="ld_fwd", rtol=1e-5, atol=1e-5)
    z, ld = ia.inverse(T(g["z"]))
    assert_close(N(z), g["z_inv"], what="z_inv", rtol=1e-4, atol=1e-5)
    assert_close(N(ld), g["ld_inv"], what="ld_inv", rtol=1e-5, atol=1e-5)
    xr, _ = ia.forward(z)
    assert_close(N(xr), g["z"], what="roundtrip", rtol=1e-4, atol=1e-4)


def test_batchnorm_flow_vs_reference(nfa):
    g = load_golden("batchnorm_flow")
    z, ld = nfa.flows.BatchNorm().to(DEV).forward(T(g["z"]))
    assert_close(N(z), g["z_fwd"], what="z_fwd", rtol=1e-4, atol=1e-5)
    assert_close(N(ld), g["ld_fwd"], what="ld_fwd", rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("ncls", [0, 3])
def test_glow_base_vs_reference(nfa, ncls):
    g = load_golden("glow_base_cc%d" % ncls)
    gb = nfa.distributions.GlowBase((4, 2, 2), num_classes=ncls or None)
    gb.load_state_dict({k: torch.from_numpy(v) for k, v in golden_state(g).items()}, strict=True)
    gb = gb.to(DEV)
    y = torch.from_numpy(g["y"]).to(DEV)
    with torch.no_grad():
        lp = gb.log_prob(T(g["z"]), y) if ncls else gb.log_prob(T(g["z"]))
        assert_close(N(lp), g["log_prob"], what="log_prob", rtol=1e-5, atol=1e-5)
        gb.temperature = 0.8
        lp = gb.log_prob(T(g["z"]), y) if ncls else gb.log_prob(T(g["z"]))
        assert_close(N(lp), g["log_prob_temp"], what="temperature", rtol=1e-5, atol=1e-5)
        gb.temperature = None
        if ncls:
            soft = torch.nn.functional.one_hot(y, ncls).float()
            assert_close(N(gb.log_prob(T(g["z"]), soft)), g["log_prob"], what="one-hot", rtol=1e-5, atol=1e-5)
        torch.manual_seed(2)
        z, lq = gb(y=y) if ncls else gb(6)
        lp2 = gb.log_prob(z, y) if ncls else gb.log_prob(z)
        assert_close(N(lp2), N(lq), what="sample log_p", rtol=1e-4, atol=1e-4)


# ---- circular-coordinate spline layers: per-feature tails and bounds (utils/splines.py:48-66) --------------------------
@pytest.mark.parametrize("name,tb", [("circ_coupled_scalar", 3.0),
                                     ("circ_coupled_tensor", [3.0, np.pi, 2.0, np.pi, 3.5, 1.5])])
def test_circular_coupled_spline_vs_reference(nfa, name, tb):
    g = load_golden(name)
    tbv = torch.tensor(tb) if isinstance(tb, list) else tb
    layer = nfa.flows.CircularCoupledRationalQuadraticSpline(6, 2, 16, ind_circ=[1, 3, 4], num_bins=5, tail_bound=tbv,
                                                             init_identity=False)
    layer = load_layer(layer, golden_state(g), torch.float32)
    with torch.no_grad():
        z, ld = layer.inverse(T(g["x"]))
        assert_close(N(z), g["z_inv"], what="z_inv", rtol=1e-4, atol=1e-4)
        assert_close(N(ld), g["ld_inv"], what="ld_inv", rtol=1e-4, atol=1e-4)
        z, ld = layer.forward(T(g["x"]))
        assert_close(N(z), g["z_fwd"], what="z_fwd", rtol=1e-4, atol=1e-4)
        assert_close(N(ld), g["ld_fwd"], what="ld_fwd", rtol=1e-4, atol=1e-4)
        x_in = T(g["x"])[2:]                       # rows with every coordinate inside its interval: exact round trip
        xr, _ = layer.inverse(layer.forward(x_in)[0])
        assert_close(N(xr), N(x_in), what="roundtrip", rtol=1e-3, atol=1e-3)


def test_circular_autoregressive_spline_vs_reference(nfa):
    g = load_golden("circ_autoregressive")
    layer = nfa.flows.CircularAutoregressiveRationalQuadraticSpline(5, 2, 12, ind_circ=[0, 3], num_bins=4, tail_bound=2.5,
                                                                    permute_mask=False, init_identity=False)
    layer = load_layer(layer, golden_state(g), torch.float32)
    with torch.no_grad():
        z, ld = layer.inverse(T(g["x"]))
        assert_close(N(z), g["z_inv"], what="z_inv", rtol=1e-4, atol=1e-4)
        assert_close(N(ld), g["ld_inv"], what="ld_inv", rtol=1e-4, atol=1e-4)
        z, ld = layer.forward(T(g["x"]))
        assert_close(N(z), g["z_fwd"], what="z_fwd", rtol=1e-4, atol=1e-4)
        assert_close(N(ld), g["ld_fwd"], what="ld_fwd", rtol=1e-4, atol=1e-4)


def test_coupling_with_tensor_tail_bound_vs_reference(nfa):
    g = load_golden("coupling_tensor_bound")
    mask = nfa.utils.create_alternating_binary_mask(4, even=False)
    mk = lambda i, o: nfa.nets.ResidualNet(i, o, hidden_features=8, num_blocks=1)
    t = nfa.flows.PiecewiseRationalQuadraticCoupling(mask, mk, num_bins=4, tails="linear",
                                                     tail_bound=torch.tensor([2.0, 3.0, 1.5, 2.5]),
                                                     apply_unconditional_transform=True)
    t = load_layer(t, golden_state(g), torch.float32)
    with torch.no_grad():
        z, ld = t.forward(T(g["x"]))
        assert_close(N(z), g["z_fwd"], what="z_fwd", rtol=1e-4, atol=1e-5)
        assert_close(N(ld), g["ld_fwd"], what="ld_fwd", rtol=1e-4, atol=1e-5)
        z, ld = t.inverse(T(g["x"]))
        assert_close(N(z), g["z_inv"], what="z_inv", rtol=1e-4, atol=1e-4)
        assert_close(N(ld), g["ld_inv"], what="ld_inv", rtol=1e-4, atol=1e-4)


def test_periodic_wrap_and_shift(nfa):
    z = torch.tensor([[3.5, 0.2, -4.0], [0.1, 7.0, 2.0]], device=DEV)
    w, ld = nfa.flows.PeriodicWrap([0, 2], bound=3.0).inverse(z)
    assert_close(N(w), np.array([[-2.5, 0.2, 2.0], [0.1, 7.0, 2.0]], dtype=np.float32), what="wrap", rtol=1e-6, atol=1e-6)
    sh = nfa.flows.PeriodicShift([0], bound=3.0, shift=1.0)
    f, _ = sh.forward(z)
    b, _ = sh.inverse(f)
    assert_close(N(b)[:, 0], np.array([-2.5, 0.1], dtype=np.float32), what="shift roundtrip", rtol=1e-6, atol=1e-6)


def test_conditional_flow_vs_reference(nfa):
    """ConditionalNormalizingFlow: context through coupling conditioners (concat + GLU gate), MADE context layers and
    the conditional base distribution (core.py:216-366)."""
    g = load_golden("model_conditional_nsf")
    flows = []
    for _ in range(2):
        flows += [nfa.flows.CoupledRationalQuadraticSpline(4, 1, 8, num_context_channels=3, num_bins=4, init_identity=False),
                  nfa.flows.LULinearPermute(4)]
    flows += [nfa.flows.AutoregressiveRationalQuadraticSpline(4, 1, 8, num_context_channels=3, num_bins=4,
                                                              init_identity=False)]
    q0 = nfa.distributions.ConditionalDiagGaussian(4, torch.nn.Linear(3, 8))
    m = nfa.ConditionalNormalizingFlow(q0, flows)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in golden_state(g).items()}, strict=True)
    m = m.to(DEV)
    x, c = T(g["x"]), T(g["context"])
    with torch.no_grad():
        assert_close(N(m.log_prob(x, c)), g["log_prob"], what="log_prob", rtol=1e-4, atol=1e-4)
        z, ld = m.inverse_and_log_det(x, c)
        assert_close(N(z), g["z_inv"], what="z_inv", rtol=1e-4, atol=1e-4)
        assert_close(N(ld), g["ld_inv"], what="ld_inv", rtol=1e-4, atol=1e-4)
        xf, ldf = m.forward_and_log_det(x, c)
        assert_close(N(xf), g["z_fwd"], what="z_fwd", rtol=1e-4, atol=1e-4)
        assert_close(N(ldf), g["ld_fwd"], what="ld_fwd", rtol=1e-4, atol=1e-4)
        torch.manual_seed(4)
        xs, lq = m.sample(9, context=c)
        assert_close(N(m.log_prob(xs, c)), N(lq), what="log_prob(sample)", rtol=1e-3, atol=1e-3)


def test_image_spline_coupling_vs_reference(nfa):
    """NCHW inputs through PiecewiseRationalQuadraticCoupling (nsf/coupling.py:150-160): conv conditioner, channel mask,
    per-pixel unconditional transform."""
    import warnings
    g = load_golden("coupling_image")
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
        t = load_layer(t, golden_state(g), torch.float32)
        with torch.no_grad():
            z, ld = t.forward(T(g["x"]))
            assert_close(N(z), g["z_fwd"], what="z_fwd", rtol=1e-4, atol=1e-5)
            assert_close(N(ld), g["ld_fwd"], what="ld_fwd", rtol=1e-4, atol=1e-4)
            z, ld = t.inverse(T(g["x"]))
            assert_close(N(z), g["z_inv"], what="z_inv", rtol=1e-4, atol=1e-4)
            assert_close(N(ld), g["ld_inv"], what="ld_inv", rtol=1e-4, atol=1e-4)
            xr, ldr = t.forward(z)
            assert_close(N(xr), g["x"], what="roundtrip", rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("seed", [0, 5, 9, 14, 19, 23])
def test_fused_chain_vs_unfused_randomized(nfa, seed, chain_workgroups):
    """Randomised models of the benchmark shape (2-4 layer pairs, both mask parities, identity / random inits, perturbed
    weights), ragged batches with NaN / +-inf / tail-boundary inputs: the persistent fused chain and the unfused path
    (library GEMMs + nf_rqs_coupling + nf_lu_linear_permute) agree, including which rows are non-finite."""
    torch.manual_seed(seed)
    nl = 2 + seed % 3
    flows = []
    for _ in range(nl):
        flows += [nfa.flows.CoupledRationalQuadraticSpline(64, 2, 128, num_bins=8, init_identity=(seed % 2 == 0),
                                                           reverse_mask=bool(seed & 4)),
                  nfa.flows.LULinearPermute(64, identity_init=(seed % 3 == 0))]
    m = nfa.NormalizingFlow(nfa.distributions.DiagGaussian(64, trainable=False), flows)
    g = torch.Generator().manual_seed(100 + seed)
    with torch.no_grad():
        for p_ in m.parameters():
            p_.add_(0.03 * torch.randn(p_.shape, generator=g))
    m = m.to(DEV)
    B = [1, 31, 257, 4099, 8192][seed % 5]
    x = (2.2 * torch.randn(B, 64, generator=g)).to(DEV)
    if B > 3:
        x[0, :4] = torch.tensor([3.0, -3.0, 3.0000002, float("nan")], device=DEV)
        x[1, :2] = torch.tensor([float("inf"), -float("inf")], device=DEV)
    eps = torch.randn(B, 64, generator=g).to(DEV)
    a = m.log_prob(x)
    xs, lq = m.sample_from_noise(eps)
    for f in m.flows:
        if hasattr(f, "prqct"):
            f.prqct.use_fused = False
    b = m.log_prob(x)
    xs2, lq2 = m.sample_from_noise(eps)
    assert torch.equal(torch.isfinite(a), torch.isfinite(b))
    fin = torch.isfinite(b)
    assert float(((a[fin] - b[fin]).abs() / b[fin].abs().clamp_min(1.0)).max()) < 2e-5
    # sampling runs the ill-conditioned quadratic-root branch through randomly initialised (steep) splines: the two paths
    # use different transcendental implementations, so their samples -- and log q at those samples -- drift apart more
    assert float(((lq - lq2).abs() / lq2.abs().clamp_min(1.0)).max()) < 1e-3
    assert_close(N(xs), N(xs2), what="samples", rtol=2e-3, atol=2e-3)


# ---- RealNVP stack as one launch (csrc/realnvp_chain.hip) ------------------------------------------------------------
def _realnvp_model(nfa, d, hidden, nlayers, seed, leaky=0.0):
    torch.manual_seed(seed)
    b = torch.tensor([1.0 if i % 2 == 0 else 0.0 for i in range(d)])
    fl = []
    for i in range(nlayers):
        s_ = nfa.nets.MLP([d] + hidden + [d], leaky=leaky, init_zeros=False)
        t_ = nfa.nets.MLP([d] + hidden + [d], leaky=leaky, init_zeros=False) if i % 3 != 2 else None
        fl += [nfa.flows.MaskedAffineFlow(b if i % 2 == 0 else 1 - b, t_, s_), nfa.flows.ActNorm(d)]
    return nfa.NormalizingFlow(nfa.distributions.DiagGaussian(d, trainable=False), fl).to(DEV)


@pytest.mark.parametrize("d,hidden,nl,B,leaky", [(2, [4], 4, 1024, 0.0), (5, [16, 8], 3, 333, 0.1), (16, [64], 2, 65, 0.0),
                                                 (3, [], 2, 7, 0.0)])
def test_realnvp_chain_kernel_vs_layerwise(nfa, d, hidden, nl, B, leaky):
    """The one-launch RealNVP chain (MaskedAffineFlow with MLP conditioners + ActNorm) against the layer-by-layer kernels,
    both directions, after the data-dependent ActNorm initialisation."""
    from normflows_amd import core
    m = _realnvp_model(nfa, d, hidden, nl, seed=d + B, leaky=leaky)
    x = torch.randn(B, d, generator=torch.Generator().manual_seed(B)).to(DEV)
    with torch.no_grad():
        m.log_prob(x)                                   # ActNorm init (layer-by-layer path)
        calls = []
        orig = core._run_realnvp
        core._run_realnvp = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
        try:
            lp = m.log_prob(x)
            z1, ld1 = m.inverse_and_log_det(x)
            x1, lf1 = m.forward_and_log_det(x)
        finally:
            core._run_realnvp = orig
        assert len(calls) == 3                          # every pass was one chain launch
        lp0 = torch.zeros(B, device=DEV)
        zz = x
        for f in reversed(list(m.flows)):
            zz, l_ = f.inverse(zz)
            lp0 = lp0 + l_
        ref_lp = lp0 + m.q0.log_prob(zz)
        assert_close(N(lp), N(ref_lp), what="log_prob", rtol=1e-5, atol=1e-5)
        assert_close(N(z1), N(zz), what="z", rtol=1e-5, atol=1e-5)
        assert_close(N(ld1), N(lp0), what="ld", rtol=1e-5, atol=1e-5)
        xf, lf = x, torch.zeros(B, device=DEV)
        for f in m.flows:
            xf, l_ = f.forward(xf)
            lf = lf + l_
        assert_close(N(x1), N(xf), what="forward", rtol=1e-5, atol=1e-5)
        assert_close(N(lf1), N(lf), what="forward ld", rtol=1e-5, atol=1e-5)
        # a parameter update re-packs the blob
        m.flows[1].t.add_(0.25)           # in place under no_grad, as an optimiser step: bumps the version counter
        lp2 = m.log_prob(x)
        assert float((lp2 - lp).abs().max()) > 1e-3


def test_models_run_in_double_precision(nfa, oracle):
    """`.double()` models (examples/real_nvp.ipynb trains in float64): every layer has an fp64 kernel path; the NSF model
    in fp64 matches the fp64 oracle to 1e-10 and its own fp32 result to fp32 accuracy."""
    from bench import build_c2_model, state_to_numpy
    m = build_c2_model(num_layers=3, dim=64, hidden=128, seed=2, sigma=0.02)
    x = torch.randn(200, 64, generator=torch.Generator().manual_seed(9))
    ora = oracle.OracleNSF(state_to_numpy(m), num_layers=6, K=8, tail_bound=3.0)
    ref64 = ora.log_prob(x.numpy().astype(np.float64))
    m32 = m.to(DEV)
    lp32 = N(m32.log_prob(x.to(DEV)))
    m64 = m32.double()
    lp64 = m64.log_prob(x.double().to(DEV))
    assert lp64.dtype == torch.float64
    assert float(np.max(np.abs(N(lp64) - ref64) / np.maximum(1.0, np.abs(ref64)))) < 1e-10
    assert float(np.max(np.abs(lp32 - ref64) / np.maximum(1.0, np.abs(ref64)))) < 1e-5
    xs, lq = m64.sample(50)
    assert xs.dtype == torch.float64
    assert_close(N(m64.log_prob(xs)), N(lq), what="fp64 log_prob(sample)", rtol=1e-8, atol=1e-8)
    # RealNVP in double: the fp32-only chain kernel steps aside for the fp64 layer kernels
    rn = _realnvp_model(nfa, 2, [4], 3, seed=1).double()
    xr = torch.randn(64, 2, dtype=torch.float64, device=DEV)
    a = rn.log_prob(xr)
    z, ld = rn.inverse_and_log_det(xr)
    assert a.dtype == torch.float64 and torch.allclose(a, ld + rn.q0.log_prob(z), atol=1e-12)
    xb, _ = rn.forward_and_log_det(z)
    assert_close(N(xb), N(xr), what="fp64 roundtrip", rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("D,hidden,blocks,bins", [(2, 128, 2, 8), (5, 32, 1, 8), (7, 64, 2, 8), (16, 16, 0, 8), (32, 128, 3, 8),
                                                  (63, 100, 2, 8), (64, 64, 2, 8),
                                                  # the 4- and 16-bin instantiations of the fused kernel (full shape and padded)
                                                  (64, 128, 2, 4), (64, 128, 2, 16), (20, 64, 1, 4), (33, 96, 2, 16)])
def test_fused_kernel_on_narrower_layers_vs_unfused_and_oracle(nfa, oracle, D, hidden, blocks, bins):
    """Layers narrower than the fused kernel's shape (64 features, 128 hidden units) run on it zero-padded -- padding columns
    parked in the splines' tails, zero hidden units, LU factors extended by an identity block: same results as the unfused
    path (library GEMMs + nf_rqs_coupling / nf_lu kernels) and as the CPU oracle, both directions, both mask parities, alone
    and as a chain of [CoupledRQS, LULinearPermute] pairs."""
    from normflows_amd.core import run_chain
    torch.manual_seed(100 * D + hidden + bins)
    flows = []
    for i in range(3):
        c = nfa.flows.CoupledRationalQuadraticSpline(D, blocks, hidden, num_bins=bins, init_identity=False, reverse_mask=bool(i % 2))
        lu = nfa.flows.LULinearPermute(D, identity_init=False)
        with torch.no_grad():
            for p_ in list(c.parameters()) + list(lu.parameters()):
                p_.add_(0.05 * torch.randn_like(p_))
        flows += [c.to(DEV), lu.to(DEV)]
    B = 333
    g = torch.Generator().manual_seed(D)
    x = 1.5 * torch.randn(B, D, generator=g)
    x.view(-1)[:4] = torch.tensor([3.0, -3.0, 3.0000002, 0.0])[: min(4, x.numel())]
    xd = x.to(DEV)
    assert flows[0].prqct._fused_eligible(xd, None) and flows[0].prqct._fused_padded() == (D != 64 or hidden != 128)
    st = {}
    for i, f in enumerate(flows):
        st.update({"flows.%d.%s" % (i, k): v.detach().cpu().numpy() for k, v in f.state_dict().items()})
    for inverse in (True, False):
        for f in flows[0::2]:
            f.prqct.use_fused = True
        ld_f = torch.zeros(B, device=DEV)
        z_f = run_chain(flows, xd, inverse, ld_f, +1)                 # one persistent launch on padded rows
        z1, l1 = (flows[0].inverse if inverse else flows[0].forward)(xd)   # a single layer through the fused kernel
        for f in flows[0::2]:
            f.prqct.use_fused = False
        for f in flows[1::2]:
            f.use_dense = False
        ld_u = torch.zeros(B, device=DEV)
        z_u = run_chain(flows, xd, inverse, ld_u, +1)                 # layer by layer, unfused kernels
        z1u, l1u = (flows[0].inverse if inverse else flows[0].forward)(xd)
        for f in flows[1::2]:
            f.use_dense = True
        assert z_f.shape == (B, D)
        assert_close(N(z1), N(z1u), what="single layer z", rtol=2e-4, atol=2e-4)
        assert_close(N(l1), N(l1u), what="single layer ld", rtol=2e-4, atol=2e-4)
        assert_close(N(z_f), N(z_u), what="chain z", rtol=1e-3, atol=1e-3)
        assert_close(N(ld_f), N(ld_u), what="chain ld", rtol=1e-3, atol=1e-3)
        logq = np.zeros(B, np.float64)
        zo = x.numpy().astype(np.float64)
        st64 = {k: (v.astype(np.float64) if v.dtype == np.float32 else v) for k, v in st.items()}
        ora64 = oracle.OracleNSF(st64, num_layers=6, K=bins, tail_bound=3.0)
        for i in (range(5, -1, -1) if inverse else range(6)):
            zo = (ora64.coupling if ora64._is_coupling(i) else ora64.lu)(i, zo, 0 if inverse else 1, logq, +1)
        assert_close(N(z_f).astype(np.float64), zo, what="chain vs oracle z", rtol=1e-3, atol=1e-3)
        assert_close(N(ld_f).astype(np.float64), logq, what="chain vs oracle ld", rtol=1e-3, atol=1e-3)
        # condition-aware bound (the layers are strongly non-identity, some bins have tiny slopes): against the
        # double-precision oracle the kernel may lose no more than a small multiple of what the reference's own fp32
        # arithmetic (the fp32 oracle on the same weights) loses, quantile by quantile and at the maximum
        ora32 = oracle.OracleNSF(st, num_layers=6, K=bins, tail_bound=3.0)
        logq32 = np.zeros(B, np.float32)
        z32 = x.numpy().copy()
        for i in (range(5, -1, -1) if inverse else range(6)):
            z32 = (ora32.coupling if ora32._is_coupling(i) else ora32.lu)(i, z32, 0 if inverse else 1, logq32, +1)
        fin = np.isfinite(zo)
        e_gpu = (np.abs(N(z_f).astype(np.float64) - zo) / (1 + np.abs(zo)))[fin]
        e_o32 = (np.abs(z32.astype(np.float64) - zo) / (1 + np.abs(zo)))[fin]
        l_gpu = np.abs(N(ld_f).astype(np.float64) - logq) / np.maximum(1, np.abs(logq))
        l_o32 = np.abs(logq32.astype(np.float64) - logq) / np.maximum(1, np.abs(logq))
        for q in (0.9, 0.99, 0.999, 1.0):
            assert np.quantile(e_gpu, q) <= 4 * np.quantile(e_o32, q) + 4e-6, ("z", q, np.quantile(e_gpu, q), np.quantile(e_o32, q))
            assert np.quantile(l_gpu, q) <= 4 * np.quantile(l_o32, q) + 1e-5, ("ld", q, np.quantile(l_gpu, q), np.quantile(l_o32, q))
        assert np.quantile(l_gpu, 0.99) < 1e-4, np.quantile(l_gpu, 0.99)      # the north-star bar on log-det, bulk of the rows


@pytest.mark.parametrize("B", [65536, 1000, 33, 1])
def test_maf_inverse_both_mappings_agree(nfa, B):
    """The three generations of the one-pass inverse on the configs[4] layer shape: nf_maf_inverse_h_tri (round 5: format-1 pack,
    regular tiles on the triangular statically unrolled sequential part, two launches: tile 0 generic + tiles 1..15 fast),
    nf_maf_inverse_h (round 3: format 0) and nf_maf_inverse (round 2: 64 samples per wave) -- different summation orders only
    (2e-5); batches off the 32-sample wave / the 256-sample workgroup; run-to-run bit equality; round trip."""
    torch.manual_seed(B)
    layer = nfa.flows.MaskedAffineAutoregressive(128, 512, num_blocks=2).to(DEV)
    with torch.no_grad():
        for p_ in layer.parameters():
            p_.add_(0.02 * torch.randn_like(p_))
    z = torch.randn(B, 128, device=DEV)
    res = []
    try:
        for halves, tri in ((True, True), (True, False), (False, False)):
            nfa.config.set_maf_halves(halves)
            nfa.config.set_maf_tri(tri)
            pk = layer._packed(DEV)
            assert (pk[4] is not None) == tri and int(N(pk[1])[7]) == int(tri)
            x, ld = layer.inverse(z)
            res.append((x, ld))
        nfa.config.set_maf_halves(True)
        nfa.config.set_maf_tri(True)
        x2, ld2 = layer.inverse(z)
    finally:
        nfa.config.set_maf_halves(True)
        nfa.config.set_maf_tri(True)
    assert torch.equal(res[0][0], x2) and torch.equal(res[0][1], ld2)
    for other in (1, 2):
        assert_close(N(res[0][0]), N(res[other][0]), what="x", rtol=2e-5, atol=2e-5)
        assert_close(N(res[0][1]), N(res[other][1]), what="logdet", rtol=2e-5, atol=2e-5)
    zz, _ = layer.forward(res[0][0])
    assert_close(N(zz), N(z), what="round trip", rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("D,H,NB,B", [(128, 512, 2, 4099), (64, 252, 2, 1000), (10, 36, 1, 77), (12, 40, 3, 257), (96, 256, 2, 513),
                                      (128, 512, 1, 64), (64, 252, 1, 31), (64, 256, 2, 300), (9, 34, 1, 65), (8, 30, 3, 129)])
def test_maf_inverse_regular_tiles_triangular_path(nfa, D, H, NB, B):
    """nf_maf_inverse_h_tri (round 5) on structures whose tiles are regular from the start (64 / 252: the fast kernel also does
    feature 0), regular with short last tiles (10 / 36: one degree; 12 / 40: 3 units per degree), generic then regular (128 / 512:
    degrees 1-4 own five units -> tile 0 is "regular with extras" (kind 2: the fifth units in slot 7), as are 64 / 256 (m = 4) and
    9 / 34, 8 / 30 (m = 2, the latter a single tile); 96 / 256: eight generic tiles, then one regular) -- i.e. one, two launches per
    layer in both orders: against the reference's D-pass structure on the same weights, the three log-det accumulation modes
    ACROSS the launches, and bit-equal repeats.  Packer + schedule are pinned on CPU (tests/test_host.py, same shapes)."""
    from normflows_amd.flows.autoregressive import Autoregressive
    from normflows_amd.flows import maf_pack
    torch.manual_seed(D * 1000 + H + NB)
    layer = nfa.flows.MaskedAffineAutoregressive(D, H, num_blocks=NB)
    _perturb(layer, 0.05 if D < 100 else 0.02, 3)
    layer = layer.to(DEV)
    pk = layer._packed(DEV)
    assert pk is not None and pk[3] == NB and pk[4] is not None and int(pk[4][7]) == 1
    T = int(pk[4][4])
    kinds = [int(pk[4][8 + 24 * t + 20]) for t in range(T)]
    assert any(kinds)
    z = torch.randn(B, D, generator=torch.Generator().manual_seed(5)).to(DEV)
    x1, ld1 = layer.inverse(z)
    x0, ld0 = Autoregressive.inverse(layer, z)       # D MADE passes
    assert_close(N(x1), N(x0), what="x", rtol=2e-4, atol=2e-4)
    assert_close(N(ld1), N(ld0), what="ld", rtol=2e-4, atol=2e-4)
    xr, _ = layer.forward(x1)
    assert_close(N(xr), N(z), what="roundtrip", rtol=1e-3, atol=1e-3)
    for mode, ref in ((nfa.ops.L.LD_ADD, 1.0 + N(ld1)), (nfa.ops.L.LD_SUB, 1.0 - N(ld1))):
        acc = torch.ones(B, device=DEV)
        x2, _ = nfa.ops.maf_inverse(z, pk[0], pk[1], pk[2], logdet=acc, acc=mode, num_blocks=NB, table_host=pk[4])
        assert torch.equal(x2, x1)
        assert_close(N(acc), ref, what="acc %d" % mode, rtol=1e-6, atol=1e-6)
    # the C ABI's argument checks for the new entry point (nothing is launched on bad arguments)
    import ctypes as C
    lib, L = nfa.ops.L.lib(), nfa.ops.L
    th = np.ascontiguousarray(pk[4], dtype=np.int32)
    args = lambda t_host, hp: (L.ptr(z), L.ptr(x1), L.ptr(ld1), L.ptr(pk[0]), L.ptr(pk[1]), t_host, L.ptr(pk[0]), L.i64(B), L.i32(D),
                               L.i32(hp), L.i32(NB), L.i32(0), L.stream())
    assert lib.nf_maf_inverse_h_tri(*args(C.c_void_p(0), pk[2])) == -14
    bad = th.copy(); bad[7] = 0
    assert lib.nf_maf_inverse_h_tri(*args(C.c_void_p(bad.ctypes.data), pk[2])) == -22
    assert lib.nf_maf_inverse_h_tri(*args(C.c_void_p(th.ctypes.data), pk[2] + 32)) == -22


@pytest.mark.parametrize("D,H,NB,B", [(128, 512, 1, 4096), (40, 100, 3, 777), (128, 512, 3, 2048), (17, 40, 1, 65), (40, 39, 1, 129), (40, 39, 3, 129)])
def test_maf_incremental_inverse_other_block_counts(nfa, D, H, NB, B):
    """nf_maf_inverse_h for MADE conditioners of 1 and 3 residual blocks (nets/made.py:140-214; round 2 took two blocks only and
    sent the others through the D-pass loop): against the reference's D-pass structure on the same weights; the packer and
    the schedule for these block counts are pinned on CPU by tests/test_host.py::test_maf_pack_schedule_matches_d_pass."""
    from normflows_amd.flows.autoregressive import Autoregressive
    torch.manual_seed(D * 1000 + H + NB)
    layer = nfa.flows.MaskedAffineAutoregressive(D, H, num_blocks=NB)
    _perturb(layer, 0.05 if D < 100 else 0.02, 3)
    layer = layer.to(DEV)
    packed = layer._packed(DEV)
    assert packed is not None and packed[3] == NB
    z = torch.randn(B, D, generator=torch.Generator().manual_seed(5)).to(DEV)
    x1, ld1 = layer.inverse(z)                       # incremental kernel
    x0, ld0 = Autoregressive.inverse(layer, z)       # D MADE passes
    assert_close(N(x1), N(x0), what="x", rtol=2e-4, atol=2e-4)
    assert_close(N(ld1), N(ld0), what="ld", rtol=2e-4, atol=2e-4)
    xr, ldr = layer.forward(x1)
    assert_close(N(xr), N(z), what="roundtrip", rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize("B,parity,NT", [(65536, 0, 32), (1024, 1, 32), (4098, 0, 32), (4097, 1, 64), (32768, 0, 64)])
def test_rqs_coupling_pipelined_kernel_vs_wave_kernel(nfa, B, parity, NT):
    """rqs_coupling_pipe_kernel (round 3: the default NSF layer shape with the conditioner output materialised -- D = 64, 8 bins,
    linear tails, float32, B >= 1024 and even: LDS-DMA double buffering, 114 -> 54 us at B = 65 536) against the wave kernel (the
    same rows in chunks below 1024 rows), both mask parities, the three log-det accumulation modes, density and
    sample-transform; rows on / outside the bounds, NaN / inf; run-to-run bit equality (the same register routine compiled into
    two kernels: contraction differences of a few ulp)."""
    torch.manual_seed(B + parity)
    dev = DEV
    D = 2 * NT                                   # 64, or 128 (one sample per pass and wave)
    x = 1.6 * torch.randn(B, D, device=dev)
    x.view(-1)[:6] = torch.tensor([3.0, -3.0, 3.0000002, float("nan"), float("inf"), 0.0], device=dev)
    cond = torch.randn(B, NT * 23, device=dev)
    uw, uh, ud = torch.randn(NT, 8, device=dev), torch.randn(NT, 8, device=dev), torch.randn(NT, 7, device=dev)
    ii = torch.arange(parity, D, 2, device=dev)
    ti = torch.arange(1 - parity, D, 2, device=dev)
    kw = dict(tail_bound=3.0, wh_div=float(np.sqrt(128.0)))
    L = nfa._lib
    for mode in (L.RQS_DENSITY, L.RQS_SAMPLE_TRANSFORM):
        for acc in (None, L.LD_ADD, L.LD_SUB):
            base = torch.randn(B, device=dev)
            ld_a = None if acc is None else base.clone()
            ya = x.clone() if mode != L.RQS_DENSITY else None        # sampling: the call owns the transform columns only
            y1, l1 = nfa.ops.rqs_coupling(x, cond, uw, uh, ud, ii, ti, 8, mode, y=ya, logdet=ld_a, acc=acc, **kw)
            yb = x.clone() if mode != L.RQS_DENSITY else torch.empty_like(x)
            lb = torch.empty(B, device=dev) if acc is None else base.clone()
            step = 1022
            for s0 in range(0, B, step):                                       # below 1024 rows: the wave kernel
                sl = slice(s0, min(B, s0 + step))
                yc = yb[sl].contiguous()
                lc = lb[sl].contiguous()
                y2, l2 = nfa.ops.rqs_coupling(x[sl].contiguous(), cond[sl].contiguous(), uw, uh, ud, ii, ti, 8, mode, y=yc,
                                              logdet=lc, acc=(L.LD_WRITE if acc is None else acc), **kw)
                yb[sl] = y2
                lb[sl] = l2
            fin = torch.isfinite(yb)
            assert (torch.isfinite(y1) == fin).all()
            dy = (y1[fin] - yb[fin]).abs()
            assert float(dy.max()) <= 5e-5 and float(torch.quantile(dy[:1000000], 0.999)) <= 2e-6     # (random rows: a few ill-conditioned bins)
            assert float((l1 - lb).abs().max()) <= 5e-5 * max(1.0, float(lb.abs().max()))      # 64 terms, butterfly vs sequential order
            if acc is None:
                y3, l3 = nfa.ops.rqs_coupling(x, cond, uw, uh, ud, ii, ti, 8, mode, y=(x.clone() if ya is not None else None), **kw)
                assert torch.equal(torch.nan_to_num(y1), torch.nan_to_num(y3)) and torch.equal(l1, l3)


@pytest.mark.parametrize("D,B", [(128, 65536), (100, 1000), (65, 37)])
def test_lu_linear_permute_wide_dense_path(nfa, oracle, D, B):
    """LULinearPermute with 64 < D <= 128 as ONE dense product on fp32 MFMA (nf_rows_matvec_affine, round 3; the matrices composed
    in float64 once per parameter version) against the LDS-column kernel nf_lu_linear_permute and the oracle, both directions,
    log-det accumulation, round trip."""
    torch.manual_seed(D)
    layer = nfa.flows.LULinearPermute(D).to(DEV)
    with torch.no_grad():
        layer.linear.lower_entries.normal_(0, 0.1)
        layer.linear.upper_entries.normal_(0, 0.1)
        layer.linear.bias.normal_(0, 0.5)
        layer.linear.unconstrained_upper_diag.normal_(0.5, 0.3)
    x = torch.randn(B, D, device=DEV)
    st = {"flows.0." + k: v.detach().cpu().numpy() for k, v in layer.state_dict().items()}
    for inverse in (True, False):
        layer.use_dense = True
        acc = torch.full((B,), 0.25, device=DEV)
        z1 = layer._run(x, inverse, acc, +1)
        layer.use_dense = False
        acc2 = torch.full((B,), 0.25, device=DEV)
        z2 = layer._run(x, inverse, acc2, +1)
        layer.use_dense = True
        assert_close(N(z1), N(z2), what="dense vs LDS-column kernel", rtol=2e-5, atol=2e-5)
        assert_close(N(acc), N(acc2), what="log-det", rtol=1e-5, atol=1e-5)
    z, ld = layer.inverse(x)
    xr, ldr = layer.forward(z)
    assert_close(N(xr), N(x), what="round trip", rtol=1e-4, atol=1e-4)
    assert_close(N(ldr), -N(ld), what="round trip log-det", rtol=1e-5, atol=1e-5)
    with torch.no_grad():            # the cache follows parameter updates
        layer.linear.bias.add_(1.0)
    z3, _ = layer.inverse(x)
    assert_close(N(z3), N(z) + 1.0, what="bias update", rtol=1e-5, atol=1e-5)


def test_spline_debug_mode_device_flags(nfa):
    """the reference's run-time failures of utils/splines.py as device-side flags read back ONLY in debug mode --
    tails=None inputs outside the domain (the reference's gathers raise on bin index -1 / K, :154-160; our kernels clamp) and
    `assert (discriminant >= 0).all()` (:181; a NaN parameter row makes the discriminant NaN, which fails the reference's assert
    and is the only way to fail it in exact arithmetic).  Off by default: the same calls return without a host synchronisation."""
    from normflows_amd.utils import splines
    torch.manual_seed(3)
    K, n = 8, 257
    w, h, d = (torch.randn(n, K, device=DEV), torch.randn(n, K, device=DEV), torch.randn(n, K + 1, device=DEV))
    x = torch.rand(n, device=DEV)
    y, lad = splines.rational_quadratic_spline(x, w, h, d)                    # clean call
    xo = x.clone()
    xo[5] = 1.5
    yo, _ = splines.rational_quadratic_spline(xo, w, h, d)                    # default mode: clamped, no exception
    assert torch.isfinite(yo).all()
    wn = w.clone()
    wn[7, 2] = float("nan")
    nfa.config.set_debug_checks(True)
    try:
        y2, lad2 = splines.rational_quadratic_spline(x, w, h, d)
        assert torch.equal(y2, y) and torch.equal(lad2, lad)                  # clean input: same results, no exception
        xi, _ = splines.rational_quadratic_spline(y, w, h, d, inverse=True)
        assert_close(N(xi), N(x), what="round trip", rtol=1e-4, atol=1e-5)
        with pytest.raises(RuntimeError, match="outside the domain"):
            splines.rational_quadratic_spline(xo, w, h, d)
        with pytest.raises(AssertionError, match="discriminant"):
            splines.rational_quadratic_spline(y, wn, h, d, inverse=True)
        # linear tails: outside inputs are the identity, not an error
        z, _ = splines.unconstrained_rational_quadratic_spline(4.0 * torch.randn(n, device=DEV), w, h, d[:, :K - 1], tail_bound=1.0)
        assert torch.isfinite(z).all()
    finally:
        nfa.config.set_debug_checks(False)
