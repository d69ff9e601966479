"""MADE under autograd: the hand-written path (nf_made_forward_train / nf_made_backward / nf_made_wgrad) against torch autograd
through library GEMMs on the same module (float32 and float64), shape by shape; optional timing at config 5's layer."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import normflows_amd as nfa

dev = "cuda:0"


def grads_of(made, x, gp):
    for p in made.parameters():
        p.grad = None
    x = x.clone().requires_grad_(True)
    out = made(x)
    out.backward(gp.to(out.dtype))
    return out.detach(), x.grad, [p.grad.clone() for p in made.parameters()]


def check(D, H, NB, mult, B, seed=0):
    torch.manual_seed(seed)
    made = nfa.nets.MADE(D, H, num_blocks=NB, output_multiplier=mult)
    with torch.no_grad():
        for p in made.parameters():
            p.add_(0.2 * torch.randn_like(p))
    made = made.to(dev)
    x = torch.randn(B, D, device=dev)
    gp = torch.randn(B, mult * D, device=dev)
    nfa.config.set_made_train(True)
    o1, gx1, g1 = grads_of(made, x, gp)
    nfa.config.set_made_train(False)
    o0, gx0, g0 = grads_of(made, x, gp)
    nfa.config.set_made_train(True)
    md = __import__("copy").deepcopy(made).double()
    o2, gx2, g2 = grads_of(md, x.double(), gp.double())

    def rel(a, b):
        return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))
    names = [n for n, _ in made.named_parameters()]
    worst = max(rel(a, b) for a, b in zip(g1, g2))
    worst0 = max(rel(a, b) for a, b in zip(g0, g2))
    wn = max(zip(g1, g2, names), key=lambda t: rel(t[0], t[1]))[2]
    print("D %d H %d NB %d mult %d B %d | out %.2e  gx %.2e (lib %.2e)  params %.2e (lib %.2e) worst %s" %
          (D, H, NB, mult, B, rel(o1, o2), rel(gx1, gx2), rel(gx0, gx2), worst, worst0, wn), flush=True)
    return max(rel(o1, o2), rel(gx1, gx2), worst)


def timing(B=65536, D=128, H=512, mult=2):
    torch.manual_seed(0)
    made = nfa.nets.MADE(D, H, num_blocks=2, output_multiplier=mult).to(dev)
    x = torch.randn(B, D, device=dev)
    gp = torch.randn(B, mult * D, device=dev)
    for mode in (True, False):
        nfa.config.set_made_train(mode)
        for _ in range(3):
            grads_of(made, x, gp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            grads_of(made, x, gp)
        torch.cuda.synchronize()
        print("config-5 layer fwd+bwd, made_train=%s: %.3f ms" % (mode, (time.perf_counter() - t0) * 100), flush=True)
    nfa.config.set_made_train(True)


if __name__ == "__main__":
    bad = 0
    for args in [(20, 40, 2, 2, 130), (6, 16, 2, 23, 70), (33, 300, 1, 3, 65), (5, 7, 3, 2, 1), (128, 512, 2, 2, 300),
                 (128, 512, 2, 23, 64), (64, 256, 2, 2, 1000), (96, 400, 3, 5, 257)]:
        e = check(*args)
        bad += e > 2e-4
    print("FAILED" if bad else "ALL OK", flush=True)
    if "--time" in sys.argv:
        timing()
