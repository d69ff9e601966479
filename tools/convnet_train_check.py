"""ConvNet2d (3x3 -> 1x1 -> 3x3, ReLU) under autograd: the hand-written path (conv_rows.hip + the MADE training kernels in plain-MLP
mode) against torch autograd through the convolution library in float64."""
import os, sys, copy
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import normflows_amd as nfa

dev = "cuda:0"
bad = 0
for Cin, hid, Cout, B, H, W in [(6, 256, 12, 8, 16, 16), (12, 256, 24, 5, 8, 8), (24, 256, 48, 7, 4, 4), (3, 16, 5, 2, 2, 2), (14, 300, 4, 3, 5, 7),
                                (24, 64, 48, 256, 4, 4)]:
    torch.manual_seed(Cin + hid)
    net = nfa.nets.ConvNet2d([Cin, hid, hid, Cout], [3, 1, 3], init_zeros=False).to(dev)
    x = torch.randn(B, Cin, H, W, device=dev)
    go = torch.randn(B, Cout, H, W, device=dev)
    res = []
    for n_, dt in ((net, torch.float32), (copy.deepcopy(net).double(), torch.float64)):
        xx = x.detach().clone().to(dt).requires_grad_(True)
        out = n_(xx)
        out.backward(go.to(dt))
        res.append([out.detach(), xx.grad] + [p.grad for p in n_.parameters()])
    worst = max(float((a.double() - b).abs().max() / (b.abs().max() + 1e-30)) for a, b in zip(res[0], res[1]))
    print("Cin %d hid %d Cout %d B %d %dx%d: worst relative error vs float64 autograd %.2e" % (Cin, hid, Cout, B, H, W, worst), flush=True)
    bad += worst > 2e-5
print("FAILED" if bad else "ALL OK")
