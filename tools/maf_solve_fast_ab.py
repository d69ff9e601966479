"""Round 6: nf_maf_solve_t_tri (regular-8 tiles on the statically unrolled sequential part) against nf_maf_solve_t on the same format-1
transposed pack: bitwise comparison and timing at BASELINE configs[4]'s layer (B = 65 536) and at smaller shapes."""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import normflows_amd as nfa
from normflows_amd import ops

dev = "cuda:0"
out = {}
for D, H, NB, B in ((128, 512, 2, 65536), (64, 252, 2, 4096), (40, 100, 2, 300), (72, 284, 1, 1000), (24, 92, 3, 500)):
    torch.manual_seed(D)
    layer = nfa.flows.MaskedAffineAutoregressive(D, H, num_blocks=NB)
    gen = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for p in layer.parameters():
            p.add_(0.03 * torch.randn(p.shape, generator=gen))
    layer = layer.to(dev)
    inv, fwd, bwd = layer._implicit_packs(dev)
    z = torch.randn(B, D, device=dev)
    cx, cl = torch.randn(B, D, device=dev), torch.randn(B, device=dev)
    x, _, bits, scr, prm = ops.maf_inverse_bits(z, inv["blob"], inv["table"], inv["hp"], inv["nb"], inv["tiles"],
                                                table_host=inv.get("table_host"), return_scratch=True, want_params=True)
    th = inv.get("ttable_host")
    nfast = 0 if th is None else int(sum(int(th[8 + 24 * t + 21]) for t in range(int(th[4]))))
    res = {}
    for mode in ("generic", "fast"):
        kw = dict(table_host=th) if mode == "fast" else {}
        v, s = ops.maf_solve_t(x, prm, cx, cl, bits, inv["tblob"], inv["ttable"], inv["hp"], inv["nb"], return_scratch=True, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            ops.maf_solve_t(x, prm, cx, cl, bits, inv["tblob"], inv["ttable"], inv["hp"], inv["nb"], **kw)
        torch.cuda.synchronize()
        res[mode] = (v, s, (time.perf_counter() - t0) * 200)
    nS = B // 32 * 32 * (2 * NB + 1) * inv["hp"] if B % 32 == 0 else 0       # (whole wave tiles only: the rest holds padding rows)
    out["D%d_H%d_NB%d_B%d" % (D, H, NB, B)] = {
        "tiles": None if th is None else int(th[4]), "regular8_tiles": nfast,
        "v_bitwise_equal": bool(torch.equal(res["generic"][0], res["fast"][0])),
        "scratch_bitwise_equal": bool(torch.equal(res["generic"][1][:nS], res["fast"][1][:nS])) if nS else None,
        "generic_ms": round(res["generic"][2], 3), "fast_ms": round(res["fast"][2], 3)}
print(json.dumps(out), flush=True)
