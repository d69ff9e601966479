"""forward_kld-style training of a MAF layer in the reference's DENSITY direction (flow.inverse = D MADE passes): implicit
differentiation (autograd.MafInverseFn) at BASELINE configs[4]'s layer, B = 65 536; the D-pass autograd path at a batch that fits."""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import normflows_amd as nfa
from normflows_amd.autograd import MafInverseFn

dev = "cuda:0"
torch.manual_seed(0)
layer = nfa.flows.MaskedAffineAutoregressive(128, 512, num_blocks=2)
sigma = float(sys.argv[sys.argv.index("--sigma") + 1]) if "--sigma" in sys.argv else 0.02
gen = torch.Generator().manual_seed(3)
with torch.no_grad():
    for p in layer.parameters():
        p.add_(sigma * torch.randn(p.shape, generator=gen))
layer = layer.to(dev)
out = {"sigma": sigma}
# round 5: implicit differentiation = ONE pass of nf_maf_solve_t per layer (config.maf_onepass, default); "sweeps" = round 4's
# iteration of nf_made_backward until v stops changing
for B, modes in ((65536, ("onepass", "sweeps")), (2048, ("onepass", "sweeps", False))):
    z = torch.randn(B, 128, device=dev)

    def step():
        layer.zero_grad(set_to_none=True)
        zz = z.clone().requires_grad_(True)
        x, ld = layer.inverse(zz)
        (0.5 * (x ** 2).sum(1) - ld).mean().backward()
    for mode in modes:
        nfa.config.set_maf_implicit(bool(mode))
        nfa.config.set_maf_onepass(mode == "onepass")
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        out["B%d_%s_ms" % (B, ("implicit_" + mode) if mode else "d_pass_autograd")] = (time.perf_counter() - t0) * 500
        if mode == "sweeps":
            out["B%d_sweeps" % B] = MafInverseFn.last_sweeps
    nfa.config.set_maf_implicit(True)
    nfa.config.set_maf_onepass(True)
print(json.dumps(out), flush=True)
