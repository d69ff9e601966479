#!/usr/bin/env python3
"""log_prob of the benchmark model shape with 128 / 64 / 32 hidden units: narrower conditioners are packed into the fused kernel's
128-unit blob zero-padded, and the HB = 2 / 1 instantiations skip the padding's row-blocks and k-groups; against the unfused
path on the same weights."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import normflows_amd as nfa
from bench import build_c2_model, c2_inputs
dev = torch.device("cuda:0")
x = c2_inputs(65536).to(dev)
def timed(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
for hidden in (128, 64, 32):
    m = build_c2_model(hidden=hidden).to(dev)
    with torch.no_grad():
        lp = m.log_prob(x)
        t = timed(lambda: m.log_prob(x))
        for f in m.flows:
            if hasattr(f, "prqct"): f.prqct.use_fused = False
        lpu = m.log_prob(x)
        tu = timed(lambda: m.log_prob(x), reps=2)
    print("hidden %3d: fused chain %.3f ms (%.2f M rows/s), unfused %.1f ms; max rel diff %.1e" % (hidden, t, 65536 / t / 1e3, tu,
          float(((lp - lpu).abs() / lpu.abs().clamp_min(1.0)).max())))
