"""Round 6: python stacks of the aten::copy_ / aten::clone / aten::add calls of one config-4 training step (torch profiler)."""
import os, sys, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", ROOT))
import normflows_amd as nfa
dev = "cuda:0"
torch.manual_seed(0)
fl = [[nfa.flows.GlowBlock(12, 256, split_mode="channel", scale=True) for _ in range(4)] + [nfa.flows.Squeeze()]]
m = nfa.MultiscaleFlow([nfa.distributions.DiagGaussian((12, 16, 16))], fl, [], class_cond=False).to(dev)
x = torch.rand(64, 3, 32, 32, device=dev)
with torch.no_grad():
    m.log_prob(x)
for _ in range(2):
    m.zero_grad(set_to_none=True); m.forward_kld(x).backward()
from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CPU], with_stack=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    m.zero_grad(set_to_none=True); m.forward_kld(x).backward()
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::add", "aten::add_", "aten::fill_", "aten::zero_", "aten::zeros", "aten::cat"):
        st = [fr for fr in (ev.stack or ()) if "normalizing-flows_amd" in fr or "normflows_amd" in fr or "autograd" in fr or "torch/" in fr]
        cnt[(ev.name, " <- ".join(s.split("/")[-1][:60] for s in st[:2]) or "(no python frame: inside the autograd engine)")] += 1
for (n, s), c in cnt.most_common(60):
    print("%4d %-12s %s" % (c, n, s))
