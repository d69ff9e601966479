#!/usr/bin/env python3
"""log_prob of the benchmark model shape (32 x [CoupledRQS d=64, hidden 128, 2 blocks + LULinearPermute], B = 65 536) with
4 / 8 / 16 spline bins: the fused persistent chain (nf_rqs_fused_chain, one instantiation per bin count) against the unfused
path (library GEMMs + nf_rqs_coupling + dense LU kernel) on the same weights."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import normflows_amd as nfa  # noqa: E402
from bench import build_c2_model, c2_inputs  # noqa: E402

dev = torch.device("cuda:0")
x = c2_inputs(65536).to(dev)


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


rows = []
for bins in (4, 8, 16):
    m = build_c2_model(bins=bins).to(dev)
    eps = torch.randn(65536, 64, device=dev)
    with torch.no_grad():
        lp = m.log_prob(x)
        zs, _ = m.sample_from_noise(eps) if hasattr(m, "sample_from_noise") else (None, None)
        t_f = timed(lambda: m.log_prob(x))
        t_s = timed(lambda: m.sample_from_noise(eps))
        for f in m.flows:
            if hasattr(f, "prqct"):
                f.prqct.use_fused = False
        lpu = m.log_prob(x)
        zu, _ = m.sample_from_noise(eps)
        t_u = timed(lambda: m.log_prob(x), reps=2)
    rel = float(((lp - lpu).abs() / lpu.abs().clamp_min(1.0)).max())
    rels = float(((zs - zu).abs() / (1.0 + zu.abs())).max())
    print("bins %2d: fused chain log_prob %.3f ms (%.2f M rows/s), sample %.3f ms; unfused log_prob %.1f ms (%.1fx); max rel diff "
          "log_prob %.1e, samples %.1e" % (bins, t_f, 65536 / t_f / 1e3, t_s, t_u, t_u / t_f, rel, rels))
    rows.append({"bins": bins, "log_prob_ms": t_f, "rows_per_s": 65536 / t_f * 1e3, "sample_ms": t_s, "unfused_log_prob_ms": t_u,
                 "max_rel_diff_log_prob_vs_unfused": rel, "max_rel_diff_samples_vs_unfused": rels})
if len(sys.argv) > 2 and sys.argv[1] == "--json":
    import json
    os.makedirs(os.path.dirname(os.path.abspath(sys.argv[2])), exist_ok=True)
    json.dump({"workload": "32 x [CoupledRQS(64, 2 blocks, 128 hidden, K bins) + LULinearPermute(64)], B = 65536, eager", "rows": rows},
              open(sys.argv[2], "w"), indent=1)
