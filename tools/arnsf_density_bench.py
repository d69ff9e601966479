import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import normflows_amd as nfa
torch.set_grad_enabled(False)
dev="cuda:0"
torch.manual_seed(0)
flows=[nfa.flows.AutoregressiveRationalQuadraticSpline(64, 2, 256) for _ in range(4)]
m=nfa.NormalizingFlow(nfa.distributions.DiagGaussian(64, trainable=False), flows).to(dev)
for p in m.parameters(): p.add_(0.01*torch.randn_like(p))
x=torch.randn(65536,64,device=dev)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n
a=t(lambda: m.log_prob(x)); lp1=m.log_prob(x)
nfa.config.set_made_fused(False)
b=t(lambda: m.log_prob(x)); lp0=m.log_prob(x)
nfa.config.set_made_fused(True)
made=flows[0].mprqat.autoregressive_net
c=t(lambda: made(x))
print("AR-NSF density 4 layers d=64 h=256: one-launch MADE %.3f ms (%.0f us/layer; MADE alone %.0f us), layer-wise %.3f ms; max rel diff %.2e" % (a, a*250, c*1e3, b, float(((lp1-lp0).abs()/lp0.abs().clamp_min(1)).max())))
