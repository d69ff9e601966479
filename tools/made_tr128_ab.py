"""Round 6: the 256-slot training kernels on 128-row tiles (mlp_tile.hpp mf_tr128) against 64-row tiles -- the environment switch
NF_MADE_TR128 is read once per process, so this script runs one mode, saves its results next to the other mode's and compares when both
exist.  Config 4's 16x16 conditioner (ConvNet2d 6 -> 256 -> 256 -> 12 on 256 images) forward + backward, and a ResidualNet-free MADE."""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import normflows_amd as nfa

mode = os.environ.get("NF_MADE_TR128", "1")
dev = "cuda:0"
torch.manual_seed(0)
net = nfa.nets.ConvNet2d([6, 256, 256, 12], [3, 1, 3], 0.0, init_zeros=False).to(dev)
x = torch.randn(256, 6, 16, 16, device=dev)
co = torch.randn(256, 12, 16, 16, device=dev)


def step():
    net.zero_grad(set_to_none=True)
    xx = x.clone().requires_grad_(True)
    out = net(xx)
    (out * co).sum().backward()
    return [out.detach(), xx.grad] + [p.grad for p in net.parameters()]


res = step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) * 50
path = "/tmp/made_tr128_%s.pt" % mode
torch.save([t.cpu() for t in res], path)
out = {"mode_tr128": mode, "conv_fwd_bwd_ms": round(ms, 4)}
other = "/tmp/made_tr128_%s.pt" % ("0" if mode != "0" else "1")
if os.path.exists(other):
    o = torch.load(other)
    out["bitwise_equal_to_other_mode"] = all(torch.equal(a.cpu(), b) for a, b in zip(res, o))
    out["max_abs_diff"] = max(float((a.cpu() - b).abs().max()) for a, b in zip(res, o))
print(json.dumps(out), flush=True)
