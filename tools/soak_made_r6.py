"""Round 6 (last session): the single-pass direction of config 5's layer under autograd (nf_made_forward_train / nf_made_backward /
nf_made_wgrad) at 1 000 000 rows -- tensors of more than 2^31 elements -- against torch autograd through library GEMMs."""
import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import normflows_amd as nfa
dev = "cuda:0"
torch.manual_seed(5)
layer = nfa.flows.MaskedAffineAutoregressive(128, 512, num_blocks=2)
gen = torch.Generator().manual_seed(9)
with torch.no_grad():
    for p in layer.parameters():
        p.add_(0.02 * torch.randn(p.shape, generator=gen))
layer = layer.to(dev)
out = {}
for B in (65536, 262144, 1000000):
    x0 = torch.randn(B, 128, device=dev)
    res = []
    for mode in (True, False):
        nfa.config.set_made_train(mode)
        try:
            layer.zero_grad(set_to_none=True)
            x = x0.clone().requires_grad_(True)
            z, ld = layer.forward(x)
            (z.square().mean() - ld.mean()).backward()
            torch.cuda.synchronize()
            res.append([z.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in layer.parameters()])
        finally:
            nfa.config.set_made_train(True)
    rels = [float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)) for a, b in zip(res[0], res[1])]
    names = ["z", "gx"] + [n for n, _ in layer.named_parameters()]
    worst = max(range(len(rels)), key=lambda i: rels[i])
    out["B%d" % B] = {"max_rel_diff_vs_library_path": max(rels), "worst_tensor": names[worst], "z": rels[0], "gx": rels[1],
                     "finite": all(bool(torch.isfinite(t).all()) for t in res[0])}
    del res, x0
    torch.cuda.empty_cache()
print(json.dumps(out), flush=True)
