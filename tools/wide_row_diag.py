"""Round-6 diagnosis: one row of test_nsf_wide_one_launch_vs_layerwise_and_oracle[66-512-3-False-257-8] whose log-det differs between
the one-launch kernel and the layer-wise path by 7e-3 in the sampling direction: which one is closer to the float64 layer-wise value?"""
import os, sys, copy
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import normflows_amd as nfa
DEV = "cuda:0"
torch.set_grad_enabled(False)      # inference: with gradients enabled the layer runs its training kernels, not nf_nsf_wide
D, H, NB, rev, B, K = 66, 512, 3, False, 257, 8
torch.manual_seed(D * 7 + H)
layer = nfa.flows.CoupledRationalQuadraticSpline(D, NB, H, num_bins=K, init_identity=False, reverse_mask=rev)
with torch.no_grad():
    for p_ in layer.parameters():
        p_.add_(0.04 * torch.randn_like(p_))
    u = layer.prqct.unconditional_transform
    u.unnormalized_widths.normal_(); u.unnormalized_heights.normal_(); u.unnormalized_derivatives.normal_()
layer64 = copy.deepcopy(layer).double().to(DEV)
layer = layer.to(DEV)
x = (1.7 * torch.randn(B, D, generator=torch.Generator().manual_seed(3))).to(DEV)
x[0, 0] = 3.5; x[0, D - 1] = -4.0
x[5, 0] = float("nan"); x[6, 1] = float("nan"); x[7, 2] = float("inf"); x[8, 3] = -float("inf")
for name in ("inverse", "forward"):
    z1, ld1 = getattr(layer, name)(x)
    nfa.config.set_nsf_wide(False)
    assert layer.prqct._wide_pack(x, None) is None
    z0, ld0 = getattr(layer, name)(x)
    nfa.config.set_nsf_wide(True)
    print(name, "paths bit-identical:", bool(torch.equal(torch.nan_to_num(ld1), torch.nan_to_num(ld0))))
    z64, ld64 = getattr(layer64, name)(x.double())
    e1 = torch.nan_to_num((ld1.double() - ld64).abs()); e0 = torch.nan_to_num((ld0.double() - ld64).abs())
    zz1 = torch.nan_to_num((z1.double() - z64).abs()).amax(1); zz0 = torch.nan_to_num((z0.double() - z64).abs()).amax(1)
    worst = torch.argsort(torch.maximum(e1, e0), descending=True)[:4].tolist()
    print(name, "max |ld - ld64|: one-launch %.3e  layer-wise %.3e;  max |z - z64|: %.3e  %.3e" % (e1.max(), e0.max(), zz1.max(), zz0.max()))
    for r in worst:
        print("   row %d: ld64 %.6f  one-launch %.6f (err %.3e)  layer-wise %.6f (err %.3e)   z err %.2e / %.2e"
              % (r, ld64[r], ld1[r], e1[r], ld0[r], e0[r], zz1[r], zz0[r]))
