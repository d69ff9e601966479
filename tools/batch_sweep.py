"""Round 6: the benchmark model's log_prob pass and training step over batch sizes (how far below 65 536 rows the row-tile kernels keep
the chip busy).  python tools/batch_sweep.py [--json out.json]"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", ROOT))
import normflows_amd as nfa
from bench import build_c2_model
dev = "cuda:0"
m = build_c2_model().to(dev)
flat = nfa.dp.FlatParameters(m)
opt = torch.optim.Adam(flat.parameters(), lr=1e-5, fused=True)
out = []
for B in (1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072):
    x = torch.randn(B, 64, device=dev)
    with torch.no_grad():
        for _ in range(3):
            m.log_prob(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            m.log_prob(x)
        torch.cuda.synchronize(); inf_ms = (time.perf_counter() - t0) * 1e3 / 20

    def step():
        flat.zero_grad(); m.forward_kld(x).backward(); flat.sync(); opt.step()
    for _ in range(3):
        step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8):
        step()
    torch.cuda.synchronize(); tr_ms = (time.perf_counter() - t0) * 1e3 / 8
    out.append({"rows": B, "log_prob_ms": round(inf_ms, 3), "log_prob_mrows_per_s": round(B / inf_ms / 1e3, 2),
                "train_step_ms": round(tr_ms, 3), "train_krows_per_s": round(B / tr_ms, 1)})
    print(out[-1], flush=True)
if "--json" in sys.argv:
    json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
