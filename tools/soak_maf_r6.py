"""Round 6 (last session): MAF's density-direction forward + backward on the round's new path (nf_maf_inverse_h_train, nf_maf_solve_t_tri,
nf_made_wgrad_pos) against round 5's (rearranged scratches, generic solve) at large and odd batch sizes: 1 000 000 / 64 / 65 600 rows.
Found on the way: nf_maf_solve_t_scratch_floats was bound as a 32-bit int (_lib.py), so either path failed above ~720 000 rows."""
import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import normflows_amd as nfa
dev = "cuda:0"
out = {}
for D, H, B in ((128, 512, 1000000 // 64 * 64), (128, 512, 64), (40, 100, 65600 // 64 * 64), (128, 512, 65536 + 64)):
    torch.manual_seed(D + B % 1000)
    layer = nfa.flows.MaskedAffineAutoregressive(D, H, num_blocks=2)
    gen = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for p in layer.parameters():
            p.add_(0.02 * torch.randn(p.shape, generator=gen))
    layer = layer.to(dev)
    z0 = torch.randn(B, D, device=dev)
    res = []
    for mode in (True, False):
        nfa.config.set_maf_wgrad_in_place(mode)
        nfa.config.set_maf_solve_fast(mode)
        layer.zero_grad(set_to_none=True)
        z = z0.clone().requires_grad_(True)
        x, ld = layer.inverse(z)
        (0.5 * (x ** 2).sum(1) - ld).mean().backward()
        res.append([x.detach().clone(), z.grad.clone()] + [p.grad.clone() for p in layer.parameters()])
        torch.cuda.synchronize()
    nfa.config.set_maf_wgrad_in_place(True); nfa.config.set_maf_solve_fast(True)
    rel = max(float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)) for a, b in zip(res[0], res[1]))
    fin = all(bool(torch.isfinite(a).all()) for a in res[0])
    out["D%d_H%d_B%d" % (D, H, B)] = {"max_rel_diff_new_vs_round5_path": rel, "finite": fin}
    del res, z0
    torch.cuda.empty_cache()
print(json.dumps(out))
