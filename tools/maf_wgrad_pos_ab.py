"""Round 6: config 5's layer (MaskedAffineAutoregressive(128, 512), B = 65 536), density direction forward + backward, with the
weight-gradient launch reading the one-pass kernels' scratches in place (config.maf_wgrad_in_place) and with the two rearrangements --
alternating on one box."""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import normflows_amd as nfa

dev = "cuda:0"
torch.manual_seed(0)
layer = nfa.flows.MaskedAffineAutoregressive(128, 512, num_blocks=2)
gen = torch.Generator().manual_seed(3)
with torch.no_grad():
    for p in layer.parameters():
        p.add_(0.02 * torch.randn(p.shape, generator=gen))
layer = layer.to(dev)
z = torch.randn(65536, 128, device=dev)


def step():
    layer.zero_grad(set_to_none=True)
    zz = z.clone().requires_grad_(True)
    x, ld = layer.inverse(zz)
    (0.5 * (x ** 2).sum(1) - ld).mean().backward()


out = {"True": [], "False": []}
grads = {}
for rnd in range(3):
    for mode in (False, True):
        nfa.config.set_maf_wgrad_in_place(mode)
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        out[str(mode)].append(round((time.perf_counter() - t0) * 200, 3))
        grads[mode] = [p.grad.clone() for p in layer.parameters()]
out["max_rel_diff"] = max(float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)) for a, b in zip(grads[True], grads[False]))
print(json.dumps({"layer_fwd_bwd_ms_in_place": out["True"], "layer_fwd_bwd_ms_rearranged": out["False"],
                  "max_rel_diff_of_gradients": out["max_rel_diff"]}), flush=True)
