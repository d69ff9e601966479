"""GPU tests of the auxiliary subsystems: out-of-bounds canaries around every output buffer the
ops allocate, the closed-form backward kernels of the affine family against autograd of the reference formulas, the
row mat-vec kernel, roctx ranges, and the RCCL code path of the NLL all-reduce on one rank."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
GUARD = 256           # elements of canary on each side of every allocation
CANARY = 12345.678


@pytest.fixture(scope="module")
def nfa():
    import normflows_amd
    assert torch.cuda.is_available()
    return normflows_amd


class _GuardedTorch:
    """Stand-in for the `torch` module inside normflows_amd.ops: empty / empty_like / zeros return the middle of a larger
    allocation whose borders hold a canary pattern."""

    def __init__(self):
        self.allocs = []

    def __getattr__(self, name):
        return getattr(torch, name)

    def _alloc(self, shape, dtype, device, fill=None):
        n = int(np.prod(shape)) if len(shape) else 1
        flat = torch.full((n + 2 * GUARD,), CANARY, dtype=dtype, device=device)
        mid = flat[GUARD:GUARD + n]
        if fill is not None:
            mid.fill_(fill)
        self.allocs.append((flat, n))
        return mid.view(shape)

    def empty(self, *shape, dtype=torch.float32, device=None, **kw):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list, torch.Size)):
            shape = tuple(shape[0])
        if dtype not in (torch.float32, torch.float64):
            return torch.empty(*shape, dtype=dtype, device=device, **kw)
        return self._alloc(tuple(shape), dtype, device)

    def empty_like(self, t, **kw):
        if t.dtype not in (torch.float32, torch.float64):
            return torch.empty_like(t, **kw)
        return self._alloc(tuple(t.shape), t.dtype, t.device)

    def check(self):
        assert self.allocs, "no guarded allocation was made"
        for flat, n in self.allocs:
            lo, hi = flat[:GUARD], flat[GUARD + n:]
            assert bool((lo == CANARY).all()) and bool((hi == CANARY).all()), "a kernel wrote outside its output buffer"
        k = len(self.allocs)
        self.allocs = []
        return k


@pytest.fixture()
def guarded(nfa, monkeypatch):
    g = _GuardedTorch()
    monkeypatch.setattr(nfa.ops, "torch", g)
    return g


def test_output_buffers_keep_their_canaries(nfa, guarded):
    """Guard words around every output buffer allocated by the ops layer survive the kernels: ragged batch sizes (off every
    tile size), empty tails, both directions -- spline coupling (unfused and fused chain), LULinearPermute, the affine
    family, Invertible1x1Conv, Squeeze, DiagGaussian, the Glow level chain."""
    with torch.no_grad():
        torch.manual_seed(0)
        for B in (1, 33, 257, 1000):
            flows = []
            for _ in range(2):
                flows += [nfa.flows.CoupledRationalQuadraticSpline(64, 2, 128, num_bins=8), nfa.flows.LULinearPermute(64)]
            m = nfa.NormalizingFlow(nfa.distributions.DiagGaussian(64, trainable=False), flows).to(DEV)
            x = torch.randn(B, 64, device=DEV)
            lp = m.log_prob(x)                       # fused chain + DiagGaussian
            xs, lq = m.sample_from_noise(x)
            for f in flows:                          # layer by layer (unfused kernels)
                if hasattr(f, "prqct"):
                    f.prqct.use_fused = False
                z, ld = f.inverse(x)
                z, ld = f.forward(x)
            assert torch.isfinite(lp).all() and torch.isfinite(lq).all()
            assert guarded.check() >= 8
        for B, C, H in ((3, 6, 4), (17, 12, 8), (5, 3, 5)):
            z = torch.randn(B, C, H, H, device=DEV)
            an = nfa.flows.ActNorm((C, 1, 1)).to(DEV)
            y, _ = an.inverse(z)
            y, _ = an.forward(z)
            if C > 1:
                cv = nfa.flows.Invertible1x1Conv(C, use_lu=True).to(DEV)
                y, _ = cv.inverse(z)
                y, _ = cv.forward(z)
            if C % 4 == 0 and H % 2 == 0:
                sq = nfa.flows.Squeeze()
                y, _ = sq.forward(z)
                y, _ = sq.inverse(z)
            blk = nfa.flows.AffineCouplingBlock(nfa.nets.ConvNet2d([(C + 1) // 2, 8, 8, 2 * (C // 2)], (3, 1, 3), init_zeros=False),
                                                scale=True, scale_map="sigmoid").to(DEV) if C > 1 else None
            if blk is not None:
                y, _ = blk.forward(z)
                y, _ = blk.inverse(z)
            b = (torch.arange(C * H * H, device=DEV).reshape(1, C, H, H) % 2).float()
            maf = nfa.flows.MaskedAffineFlow(b)
            y, _ = maf.forward(z)
            assert guarded.check() >= 3
        # Glow level chain: 3 blocks, 8x8, a batch off the workgroup's image count
        cls = nfa.nets.ConvNet2d
        saved = cls.FUSED_MIN_PIXELS, cls.FUSED_WIDE_MIN_PIXELS
        try:
            cls.FUSED_MIN_PIXELS, cls.FUSED_WIDE_MIN_PIXELS = 0, 0
            blocks = [nfa.flows.GlowBlock(24, 256, init_zeros=False).to(DEV) for _ in range(3)]
            done = 0
            for H in (8, 4, 16):
                B = 37
                zz = torch.randn(B, 24, H, H, device=DEV)
                for bk in blocks:
                    zz, _ = bk.inverse(zz)           # initialises the ActNorms (layer by layer), later calls: one launch each
                from normflows_amd.flows.glow import plan_level, run_level
                try:
                    for inverse in (True, False):
                        seq = blocks[::-1] if inverse else blocks
                        n, entries, layout, slope, smap = plan_level(seq, B, 24, H, H, inverse)
                        assert n == 3
                        ld = torch.zeros(B, device=DEV)
                        big = torch.randn(B, 6, 2 * H, 2 * H, device=DEV)
                        o0, o1 = run_level(seq, entries, layout, slope, smap, big, None, True, inverse, ld, +1, cout0=12)
                        o0, o1 = run_level(seq, entries, layout, slope, smap, o0, o1, False, inverse, ld, +1, out_squeezed=True)
                        assert torch.isfinite(o0).all()
                except NotImplementedError:          # 24 channels x 256 pixels: beyond one workgroup's LDS (nothing launched)
                    assert H == 16
                    guarded.allocs = []
                    continue
                done += 1
                assert guarded.check() >= 4
            assert done >= 2
        finally:
            cls.FUSED_MIN_PIXELS, cls.FUSED_WIDE_MIN_PIXELS = saved


def _vjp_check(out_fn, ref_fn, inputs, rtol, atol):
    """Gradients of sum(y * cy) + sum(ld * cl) through `out_fn` (our autograd Function) and `ref_fn` (plain torch formula)."""
    res = []
    for fn in (out_fn, ref_fn):
        leaves = [t.detach().clone().requires_grad_(True) for t in inputs]
        y, ld = fn(*leaves)
        g = torch.Generator(device="cpu").manual_seed(5)
        cy = torch.randn(y.shape, generator=g, dtype=torch.float64).to(y.dtype).to(y.device)
        cl = torch.randn(ld.shape, generator=g, dtype=torch.float64).to(y.dtype).to(y.device)
        ((y * cy).sum() + (ld * cl).sum()).backward()
        res.append([y.detach(), ld.detach()] + [t.grad for t in leaves])
    for i, (a, b) in enumerate(zip(*res)):
        assert_close(a.cpu().numpy(), b.cpu().numpy(), what="tensor %d" % i, rtol=rtol, atol=atol)


@pytest.mark.parametrize("B,C,H,W", [(256, 12, 16, 16), (64, 48, 4, 4), (7, 3, 5, 3), (9, 300, 2, 2), (33, 256, 1, 2), (3, 1, 4, 4)])
def test_actnorm_backward_reduction_over_channel_counts(nfa, B, C, H, W):
    """nf_actnorm_bwd's second stage (round 5: thread = (channel, slice of the batch splits), slices added through LDS in slice order)
    at Glow's channel counts, one channel, a count that leaves idle slices and counts >= the workgroup width: gradients of s and t
    (and z) against torch autograd of coupling.py:38-54, float32 and float64, both directions."""
    from normflows_amd import autograd as ag
    for dtype, rtol in ((torch.float32, 3e-5), (torch.float64, 1e-11)):
        torch.manual_seed(C + B)
        z = torch.randn(B, C, H, W, dtype=dtype, device=DEV)
        sv, tv = 0.2 * torch.randn(C, dtype=dtype, device=DEV), torch.randn(C, dtype=dtype, device=DEV)
        for d in (0, 1):
            def ref_an(z_, s_, t_):
                s4, t4 = s_.view(1, C, 1, 1), t_.view(1, C, 1, 1)
                ones = torch.ones(B, dtype=dtype, device=DEV)
                if d == 0:
                    return z_ * torch.exp(s4) + t4, H * W * s_.sum() * ones
                return (z_ - t4) * torch.exp(-s4), -H * W * s_.sum() * ones
            _vjp_check(lambda z_, s_, t_: ag.ActNormFn.apply(z_, s_, t_, d), ref_an, (z, sv, tv), rtol, 300 * rtol)


@pytest.mark.parametrize("dtype,rtol", [(torch.float32, 2e-5), (torch.float64, 1e-11)])
def test_affine_family_backward_kernels_vs_torch_autograd(nfa, dtype, rtol):
    """csrc/affine_bwd.hip against PyTorch autograd of the reference formulas (coupling.py:38-54, :117-171, :209-229,
    mixing.py:106-133): both directions, every scale map, odd shapes, with and without s / t."""
    from normflows_amd import autograd as ag
    torch.manual_seed(1)
    B, C, H, W = 5, 7, 3, 5
    z = torch.randn(B, C, H, W, dtype=dtype, device=DEV)
    b = (torch.arange(C * H * W, device=DEV).reshape(1, C, H, W) % 2).to(dtype)
    s, t = 0.3 * torch.randn_like(z), torch.randn_like(z)
    for d in (0, 1):
        def ref_ma(z_, s_, t_):
            if d == 0:
                return b * z_ + (1 - b) * (z_ * torch.exp(s_) + t_), ((1 - b) * s_).flatten(1).sum(1)
            return b * z_ + (1 - b) * (z_ - t_) * torch.exp(-s_), -((1 - b) * s_).flatten(1).sum(1)
        _vjp_check(lambda z_, s_, t_: ag.MaskedAffineFn.apply(z_, b, s_, t_, d), ref_ma, (z, s, t), rtol, 10 * rtol)
        c1 = (C + 1) // 2
        for smap in ("exp", "sigmoid", "sigmoid_inv", None):
            for flip in (False, True):
                P = (C - c1) * (1 if smap is None else 2)
                prm = 0.5 * torch.randn(B, P, H, W, dtype=dtype, device=DEV)
                _vjp_check(lambda z_, p_: ag.AffineCouplingFn.apply(z_, p_, c1, flip, smap, d),
                           lambda z_, p_: ag._coupling_formula(z_, p_, c1, flip, smap, d), (z, prm), rtol, 10 * rtol)
        sv, tv = 0.2 * torch.randn(C, dtype=dtype, device=DEV), torch.randn(C, dtype=dtype, device=DEV)

        def ref_an(z_, s_, t_):
            s4, t4 = s_.view(1, C, 1, 1), t_.view(1, C, 1, 1)
            ones = torch.ones(B, dtype=dtype, device=DEV)
            if d == 0:
                return z_ * torch.exp(s4) + t4, H * W * s_.sum() * ones
            return (z_ - t4) * torch.exp(-s4), -H * W * s_.sum() * ones
        _vjp_check(lambda z_, s_, t_: ag.ActNormFn.apply(z_, s_, t_, d), ref_an, (z, sv, tv), rtol, 100 * rtol)
    Wm = torch.randn(C, C, dtype=dtype, device=DEV)
    ldu = torch.tensor(0.37, dtype=dtype, device=DEV)
    _vjp_check(lambda z_, W_, l_: ag.Inv1x1Fn.apply(z_, W_, l_),
               lambda z_, W_, l_: (torch.einsum("oc,bchw->bohw", W_, z_), H * W * l_ * torch.ones(B, dtype=dtype, device=DEV)),
               (z, Wm, ldu), rtol, 100 * rtol)


@pytest.mark.parametrize("B,D", [(1, 64), (300, 64), (65536, 64), (77, 17), (129, 3)])
def test_rows_matvec_kernel(nfa, B, D):
    torch.manual_seed(B + D)
    x = torch.randn(B, D, device=DEV)
    Wm = torch.randn(D, D, device=DEV)
    y = nfa.ops.rows_matvec(x, Wm)
    ref = (x.double() @ Wm.double().t()).float()
    assert_close(y.cpu().numpy(), ref.cpu().numpy(), what="rows_matvec", rtol=1e-5, atol=1e-4)
    assert torch.equal(y, nfa.ops.rows_matvec(x, Wm))     # deterministic


@pytest.mark.parametrize("B,D", [(65536, 64), (1000, 64), (333, 20), (70, 5)])
def test_rows_matvec2_equals_two_products(nfa, B, D):
    """nf_rows_matvec2 (u = W1 x, y = W2 u + b in one launch, u through the permuted contraction order) against two
    nf_rows_matvec launches and fp64."""
    torch.manual_seed(B + D)
    x = torch.randn(B, D, device=DEV)
    W1, W2 = torch.randn(D, D, device=DEV), torch.randn(D, D, device=DEV)
    b = torch.randn(D, device=DEV)
    c = torch.tensor([0.75], device=DEV)
    ld0 = torch.randn(B, device=DEV)
    ld = ld0.clone()
    u, y, ld = nfa.ops.rows_matvec2(x, W1, W2, b, c, -1.0, logdet=ld, acc=nfa._lib.LD_ADD)
    u1 = nfa.ops.rows_matvec(x, W1)
    y1, _ = nfa.ops.rows_matvec_affine(u1, W2, b)
    assert torch.equal(u, u1)
    ref = ((x.double() @ W1.double().t()) @ W2.double().t() + b.double()).float()
    assert_close(y.cpu().numpy(), ref.cpu().numpy(), what="rows_matvec2", rtol=1e-5, atol=1e-3)
    assert float((y - y1).abs().max()) <= 1e-5 * float(y1.abs().max())
    assert torch.allclose(ld, ld0 - 0.75)
    u2, y2, _ = nfa.ops.rows_matvec2(x, W1, W2, b)
    assert torch.equal(y, y2) and torch.equal(u, u2)


def test_roctx_ranges_do_not_change_results(nfa):
    torch.manual_seed(0)
    flows = [nfa.flows.CoupledRationalQuadraticSpline(64, 2, 128, num_bins=8), nfa.flows.LULinearPermute(64), nfa.flows.ActNorm(64)]
    m = nfa.NormalizingFlow(nfa.distributions.DiagGaussian(64, trainable=False), flows).to(DEV)
    x = torch.randn(100, 64, device=DEV)
    with torch.no_grad():
        a = m.log_prob(x)
        a = m.log_prob(x)
        nfa.config.set_roctx(True)
        try:
            b = m.log_prob(x)
        finally:
            nfa.config.set_roctx(False)
    assert torch.equal(a, b)


_NCCL_WORKER = r"""
import os, sys
sys.path.insert(0, os.environ["NF_ROOT"])
import torch, torch.distributed as dist
import normflows_amd as nfa
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))     # RCCL, one rank
lq = torch.randn(1000, device="cuda:0")
nll = nfa.dp.global_nll(lq)
t = torch.ones(4, device="cuda:0", dtype=torch.float64)
dist.all_reduce(t)                                                      # world of one: the collective still runs on RCCL
torch.cuda.synchronize()
assert abs(float(nll) + float(lq.double().mean())) < 1e-9 and float(t.sum()) == 4.0
lin = torch.nn.Linear(4, 2).cuda()
for p in lin.parameters():
    p.grad = torch.ones_like(p)
nfa.dp.allreduce_gradients(lin.parameters())
dist.barrier()
print("NCCL_OK", dist.get_backend())
dist.destroy_process_group()
"""


def test_rccl_path_on_one_rank(nfa, tmp_path):
    """The `nccl` (= RCCL) process group of bench.py's N > 1 path, initialised and exercised on ONE rank: the 8-GPU run is
    the driver's, but the backend, device binding and the collective calls are at least executed on hardware here."""
    script = tmp_path / "nccl_worker.py"
    script.write_text(_NCCL_WORKER)
    env = dict(os.environ, NF_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29688", RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    assert "NCCL_OK nccl" in out.stdout


def test_mfma_clock_probe_reports_a_plausible_clock(nfa):
    """nf_mfma_clock_probe (bench.py's extra roofline field): between 1 and 2.6 GHz, and repeatable to a few per cent."""
    a = nfa.ops.mfma_clock_mhz(DEV)
    b = nfa.ops.mfma_clock_mhz(DEV)
    assert 1000.0 < a < 2600.0 and 1000.0 < b < 2600.0, (a, b)
    assert abs(a - b) < 0.1 * a, (a, b)


_WAITS_WORKER = r"""
import hashlib, json, os, sys
sys.path.insert(0, os.environ["NF_ROOT"])
import torch
import normflows_amd as nfa
from bench import build_c2_model
dev = "cuda:0"
out = {}

def digest(*ts):
    h = hashlib.sha256()
    for t in ts:
        h.update(t.detach().float().cpu().numpy().tobytes())
    return h.hexdigest()

# (1) the training step at the benchmark layer shape on FULL tiles: rqs_fused_kernel<TRAIN> (vmcnt(14) behind the parameter-row
#     stores), final_bwd_kernel (23 / 14 / 7), wgrad_ring_kernel, resblock_bwd / lu_bwd
m = build_c2_model(num_layers=2, dim=64, hidden=128, seed=0, sigma=0.05).to(dev)
x = torch.randn(4096, 64, generator=torch.Generator().manual_seed(1)).to(dev)
loss = m.forward_kld(x)
loss.backward()
out["train_step"] = digest(loss, *[p.grad for p in m.parameters()])
# (1b) MADE under autograd: made_wgrad_kernel's LDS-DMA ring (vmcnt(4) per step) over several steps and chunks
mafl = nfa.flows.MaskedAffineAutoregressive(128, 512, num_blocks=2).to(dev)
xm = torch.randn(8192, 128, generator=torch.Generator().manual_seed(6)).to(dev).requires_grad_(True)
zm, ldm_ = mafl.forward(xm)
(zm.square().sum() + ldm_.sum()).backward()
out["made_train"] = digest(zm, xm.grad, *[p.grad for p in mafl.parameters()])
torch.set_grad_enabled(False)
# (2) the software-pipelined spline kernels (rqs_spline.hip / rqs_bwd.hip: vmcnt(loads + stores) per pass)
layer = nfa.flows.CoupledRationalQuadraticSpline(64, 2, 128, init_identity=False).to(dev)
layer.prqct.use_fused = False
xs = (1.5 * torch.randn(65536, 64, generator=torch.Generator().manual_seed(2))).to(dev)
z, ld = layer.inverse(xs)
zf, ldf = layer.forward(xs)
out["spline_pipe"] = digest(z, ld, zf, ldf)
# (3) the split-bf16 chain (rqs_fused_x3.hip: 3-slot ring, vmcnt(3))
nfa.config.set_fused_gemm("bf16x3")
m4 = build_c2_model(num_layers=4, dim=64, hidden=128, seed=1, sigma=0.02).to(dev)
out["x3_chain"] = digest(m4.log_prob(xs[:8192]))
nfa.config.set_fused_gemm("f32")
# (4) Glow levels (glow_conv.hip: weight rings of all three kernels) and (5) the MAF inverse kernels
import importlib.util
spec = importlib.util.spec_from_file_location("cb", os.path.join(os.environ["NF_ROOT"], "tools", "config_bench.py"))
torch.manual_seed(0)
L_, K_, hidden, channels = 3, 4, 256, 3
q0, merges, flows = [], [], []
for i in range(L_):
    fl = [nfa.flows.GlowBlock(channels * 2 ** (L_ + 1 - i), hidden, split_mode="channel", scale=True) for _ in range(K_)] + [nfa.flows.Squeeze()]
    flows += [fl]
    if i > 0:
        merges += [nfa.flows.Merge()]
        latent = (channels * 2 ** (L_ - i), 32 // 2 ** (L_ - i), 32 // 2 ** (L_ - i))
    else:
        latent = (channels * 2 ** (L_ + 1), 32 // 2 ** L_, 32 // 2 ** L_)
    q0 += [nfa.distributions.DiagGaussian(latent)]
g = nfa.MultiscaleFlow(q0, flows, merges, class_cond=False).to(dev)
img = torch.rand(256, 3, 32, 32, generator=torch.Generator().manual_seed(3)).to(dev)
g.log_prob(img)
out["glow_levels"] = digest(g.log_prob(img))
maf = nfa.flows.MaskedAffineAutoregressive(128, 512, num_blocks=2).to(dev)
y, ldm = maf.inverse(xs.new_empty(8192, 128).normal_(generator=None) if False else torch.randn(8192, 128, generator=torch.Generator().manual_seed(4)).to(dev))
out["maf_inverse"] = digest(y, ldm)
ar = nfa.flows.AutoregressiveRationalQuadraticSpline(16, 2, 64).to(dev)
ya, lda = ar.forward(torch.randn(4096, 16, generator=torch.Generator().manual_seed(5)).to(dev))
out["arnsf_inverse"] = digest(ya, lda)
print("DIGESTS " + json.dumps(out))
"""


def test_counted_waits_equal_full_drains(nfa, tmp_path):
    """Every hand-counted `s_waitcnt vmcnt(N)` (NF_WAIT_VMCNT: ring acquires that leave N younger requests in flight -- correct only
    while N equals what the compiler actually emits between request and wait) against a build in which each of them is a full
    drain (lib/variants/safe_waits.so, -DNF_SAFE_WAITS, built by __graft_entry__.build()): the deterministic kernels that use them
    -- the one-launch training forward, nf_final_bwd, the ring weight gradients (wgrad.hip, made_bwd.hip), the pipelined spline
    kernels, the split-bf16 chain,
    the Glow level kernels, both MAF inverse kernels -- must give BIT-identical results at sizes with full tiles.  A count that
    has become too lax reads a stage that has not landed: silent corruption on full tiles only (round-3 ADVICE)."""
    import json
    from normflows_amd import _lib
    if not os.path.exists(_lib.SAFE_WAITS_LIB):
        pytest.skip("lib/variants/safe_waits.so was not built")
    script = tmp_path / "waits_worker.py"
    script.write_text(_WAITS_WORKER)
    res = []
    for variant in (None, _lib.SAFE_WAITS_LIB):
        env = {k: v for k, v in os.environ.items() if k != "NF_MI355X_LIB"}
        env["NF_ROOT"] = ROOT
        if variant:
            env["NF_MI355X_LIB"] = variant
        out = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-2500:]
        line = [l for l in out.stdout.splitlines() if l.startswith("DIGESTS ")][-1]
        res.append(json.loads(line[len("DIGESTS "):]))
    assert res[0].keys() == res[1].keys() and len(res[0]) == 7
    for k in res[0]:
        assert res[0][k] == res[1][k], "%s: the counted-wait build and the full-drain build differ" % k
