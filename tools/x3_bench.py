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
x = c2_inputs(65536).to(dev)
nfa.config.set_fused_gemm(os.environ.get("NF_GEMM", "bf16x3"))
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
print("log_prob %s: median %.3f ms  min %.3f ms  (%.2f M rows/s)  [%s]" % (nfa.config.fused_gemm, ts[len(ts) // 2], ts[0],
      65536 / ts[len(ts) // 2] / 1e3, os.environ.get("NF_MI355X_LIB", "default")))
