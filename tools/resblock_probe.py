"""nf_resblock_bwd against the separate kernels (nf_rows_block backward + nf_linear_wgrad_pair [+ the initial layer's addmm /
nf_linear_wgrad]): values and time."""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import normflows_amd as nfa
from normflows_amd import ops
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
torch.manual_seed(0)
gh = torch.randn(B, 128, device=dev); t = torch.randn(B, 128, device=dev); h = torch.randn(B, 128, device=dev)
W1 = torch.randn(128, 128, device=dev) * 0.1; W2 = torch.randn(128, 128, device=dev) * 0.1
x = torch.randn(B, 64, device=dev); wfull = torch.randn(128, 64, device=dev) * 0.1; wfull[:, 1::2] = 0; wfull_t = wfull.t().contiguous()
gx0 = torch.randn(B, 64, device=dev)

def old(init):
    gt, gh_in = ops.rows_block(gh, W2, None, W1, None, trans=True, mask1=t, mask2=h, relu=False)
    gw2, gb2, gw1, gb1 = ops.linear_wgrad_pair(gh, t, gt, h, relu_x=True)
    if not init:
        return gh_in, gw1, gb1, gw2, gb2
    gx = gx0.clone(); gx.addmm_(gh_in, wfull)
    gw0, gb0 = ops.linear_wgrad(gh_in, x, want_bias=True)
    return gx, gw1, gb1, gw2, gb2, gw0, gb0

def new(init):
    if not init:
        return ops.resblock_bwd(gh, t, h, W1, W2)
    gx = gx0.clone()
    r = ops.resblock_bwd(gh, t, h, W1, W2, x=x, wfull=wfull_t, gx=gx)
    return (gx,) + r[1:]

def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

for init in (False, True):
    try:
        a, b = old(init), new(init)
    except Exception as ex:
        print("init", init, "FAILED", ex); continue
    names = ["gh_in/gx", "dW1", "db1", "dW2", "db2", "dW0", "db0"]
    for nm, u, v in zip(names, a, b):
        sc = float(u.abs().max())
        print("init=%d %-8s max|diff| %.3e (scale %.3e) rel %.2e" % (init, nm, float((u - v).abs().max()), sc, float((u - v).abs().max()) / sc))
    # determinism
    c = new(init)
    print("init=%d deterministic:" % init, all(torch.equal(u, v) for u, v in zip(b, c)))
    print("init=%d old %.1f us   new %.1f us" % (init, timeit(lambda: old(init)), timeit(lambda: new(init))))
