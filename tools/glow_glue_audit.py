"""Round 6: where the copy / fill / cat launches of config 4's training step come from (call sites of Tensor.contiguous() that copy,
clone(), torch.cat, torch.zeros*, Tensor.add_/add on the host side), one step.  python tools/glow_glue_audit.py"""
import collections, os, sys, traceback
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", ROOT))
import normflows_amd as nfa
dev = "cuda:0"
torch.manual_seed(0)
L_, K_, hidden, channels = 3, 32, 256, 3
q0, merges, flows = [], [], []
for i in range(L_):
    fl = [nfa.flows.GlowBlock(channels * 2 ** (L_ + 1 - i), hidden, split_mode="channel", scale=True) for _ in range(K_)]
    fl += [nfa.flows.Squeeze()]
    flows += [fl]
    if i > 0:
        merges += [nfa.flows.Merge()]
        latent = (3 * 2 ** (L_ - i), 32 // 2 ** (L_ - i), 32 // 2 ** (L_ - i))
    else:
        latent = (3 * 2 ** (L_ + 1), 32 // 2 ** L_, 32 // 2 ** L_)
    q0 += [nfa.distributions.DiagGaussian(latent)]
m = nfa.MultiscaleFlow(q0, flows, merges, class_cond=False).to(dev)
x = torch.rand(256, 3, 32, 32, device=dev)
with torch.no_grad():
    m.log_prob(x)
for _ in range(2):
    m.zero_grad(set_to_none=True); m.forward_kld(x).backward()
sites = collections.Counter()
def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "normalizing-flows_amd" in fr.filename or "normflows_amd" in fr.filename:
            return "%s:%d" % (os.path.basename(fr.filename), fr.lineno)
    return "?"
orig_contig, orig_clone, orig_cat, orig_zeros, orig_zl = torch.Tensor.contiguous, torch.Tensor.clone, torch.cat, torch.zeros, torch.zeros_like
def contig(self, *a, **k):
    if not self.is_contiguous():
        sites[("contiguous-copy", site())] += 1
    return orig_contig(self, *a, **k)
def clone(self, *a, **k):
    sites[("clone", site())] += 1
    return orig_clone(self, *a, **k)
def cat(*a, **k):
    sites[("cat", site())] += 1
    return orig_cat(*a, **k)
def zeros(*a, **k):
    sites[("zeros", site())] += 1
    return orig_zeros(*a, **k)
def zl(*a, **k):
    sites[("zeros_like", site())] += 1
    return orig_zl(*a, **k)
torch.Tensor.contiguous, torch.Tensor.clone, torch.cat, torch.zeros, torch.zeros_like = contig, clone, cat, zeros, zl
m.zero_grad(set_to_none=True); m.forward_kld(x).backward()
torch.cuda.synchronize()
torch.Tensor.contiguous, torch.Tensor.clone, torch.cat, torch.zeros, torch.zeros_like = orig_contig, orig_clone, orig_cat, orig_zeros, orig_zl
for (op, s), n in sites.most_common(40):
    print("%5d  %-16s %s" % (n, op, s))
