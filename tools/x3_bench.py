#!/usr/bin/env python3
"""Times one log_prob pass of the benchmark model on the split-bf16 path (persistent chain, no graph).  NF_MI355X_LIB selects
a build variant (tools/build_variant.py; ablations give wrong results, timing only)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import normflows_amd as nfa  # noqa: E402
from bench import build_c2_model, c2_inputs  # noqa: E402

dev = torch.device("cuda:0")
m = build_c2_model().to(dev)
# rows per GPU: argv (default: the benchmark's 65 536); several values = the batch sweep behind DESIGN 7.2's hypothesis that the
# per-layer stream of split weights (2 MB per layer and CU) is only amortised by more rows per CU
batches = [int(v) for v in sys.argv[1:] if v.isdigit()] or [65536]
FLOP_ROW = 344064.0 * 32 * 6            # algorithmic fp32 FLOP per row x six bf16 products per fp32 product
out = []
for mode in (os.environ.get("NF_GEMM", "bf16x3"),) + (("f32",) if len(batches) > 1 else ()):
    nfa.config.set_fused_gemm(mode)
    for Bn in batches:
        x = c2_inputs(Bn).to(dev)
        with torch.no_grad():
            for _ in range(3):
                m.log_prob(x)
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(5):
                    m.log_prob(x)
                e.record()
                torch.cuda.synchronize()
                ts.append(s.elapsed_time(e) / 5)
        ts.sort()
        med = ts[len(ts) // 2]
        frac = (Bn * FLOP_ROW / (med * 1e-3)) / 2.5166e15 if mode == "bf16x3" else (Bn * FLOP_ROW / 6 / (med * 1e-3)) / 157.3e12
        out.append(dict(mode=mode, rows=Bn, ms_median=round(med, 4), ms_min=round(ts[0], 4), mrows_per_s=round(Bn / med / 1e3, 3),
                        frac_of_peak=round(frac, 4)))
        print("log_prob %s rows %d: median %.3f ms  min %.3f ms  (%.2f M rows/s, %.3f of the %s MFMA peak)  [%s]"
              % (mode, Bn, med, ts[0], Bn / med / 1e3, frac, "bf16" if mode == "bf16x3" else "fp32",
                 os.environ.get("NF_MI355X_LIB", "default")))
        del x
nfa.config.set_fused_gemm("f32")
if "--json" in sys.argv:
    import json
    json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
