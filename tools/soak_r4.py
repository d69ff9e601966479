import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import normflows_amd as nfa
torch.set_grad_enabled(False)
dev="cuda:0"
torch.manual_seed(0)
def rel(a,b): return float(((a-b).abs()/(1+b.abs())).max())
# MAF forward, big odd batch
maf = nfa.flows.MaskedAffineAutoregressive(128, 512, num_blocks=2).to(dev)
for p in maf.parameters(): p.add_(0.02*torch.randn_like(p))
x = torch.randn(1000003, 128, device=dev)
z, ld = maf.forward(x)
nfa.config.set_made_fused(False); z0, ld0 = maf.forward(x[-70000:]); nfa.config.set_made_fused(True)
print("maf fwd B=1000003:", rel(z[-70000:], z0), rel(ld[-70000:], ld0), bool(torch.isfinite(z).all()))
# AR-NSF density
ar = nfa.flows.AutoregressiveRationalQuadraticSpline(100, 2, 400).to(dev)
for p in ar.parameters(): p.add_(0.02*torch.randn_like(p))
x = torch.randn(300001, 100, device=dev)
z, ld = ar.inverse(x)
nfa.config.set_made_fused(False); z0, ld0 = ar.inverse(x[-5000:]); nfa.config.set_made_fused(True)
print("arnsf density B=300001:", rel(z[-5000:], z0), rel(ld[-5000:], ld0))
# wide NSF pair, both directions
c = nfa.flows.CoupledRationalQuadraticSpline(128, 2, 512); lu = nfa.flows.LULinearPermute(128, identity_init=False)
m = nfa.NormalizingFlow(nfa.distributions.DiagGaussian(128, trainable=False), [c, lu]).to(dev)
for p in m.parameters(): p.add_(0.02*torch.randn_like(p))
x = torch.randn(300001, 128, device=dev)
for fn in (m.inverse_and_log_det, m.forward_and_log_det):
    z, ld = fn(x)
    nfa.config.set_nsf_wide(False); z0, ld0 = fn(x[-5000:]); nfa.config.set_nsf_wide(True)
    print("wide pair B=300001:", rel(z[-5000:], z0), rel(ld[-5000:], ld0))
torch.cuda.synchronize(); print("peak mem GB", torch.cuda.max_memory_allocated()/2**30)
