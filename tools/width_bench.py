#!/usr/bin/env python3
"""log_prob of 32 x [CoupledRQS(d, 2 blocks, 128 hidden, 8 bins) + LULinearPermute(d)] at B = 65 536 for d = 64 / 32 / 16 / 8:
narrower layers run on the fused kernel's 64 columns zero-padded, and the final-layer groups of all-padding 16-column chunks are
skipped; against the unfused path on the same weights."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import normflows_amd as nfa  # noqa: E402
from bench import build_c2_model  # noqa: E402

dev = torch.device("cuda:0")


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


for d in (64, 32, 16, 8):
    m = build_c2_model(dim=d).to(dev)
    x = torch.randn(65536, d, generator=torch.Generator().manual_seed(d)).to(dev)
    with torch.no_grad():
        lp = m.log_prob(x)
        t = timed(lambda: m.log_prob(x))
        for f in m.flows:
            if hasattr(f, "prqct"):
                f.prqct.use_fused = False
        lpu = m.log_prob(x)
        tu = timed(lambda: m.log_prob(x), reps=2)
    print("d %2d: fused chain %.3f ms (%.2f M rows/s), unfused %.1f ms; max rel diff %.1e" % (
        d, t, 65536 / t / 1e3, tu, float(((lp - lpu).abs() / lpu.abs().clamp_min(1.0)).max())))
