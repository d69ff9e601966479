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


for bins in (4, 8, 16):
    m = build_c2_model(bins=bins).to(dev)
    with torch.no_grad():
        lp = m.log_prob(x)
        t_f = timed(lambda: m.log_prob(x))
        for f in m.flows:
            if hasattr(f, "prqct"):
                f.prqct.use_fused = False
        lpu = m.log_prob(x)
        t_u = timed(lambda: m.log_prob(x), reps=2)
    rel = float(((lp - lpu).abs() / lpu.abs().clamp_min(1.0)).max())
    print("bins %2d: fused chain %.3f ms (%.2f M rows/s), unfused %.1f ms, speed-up %.1fx, max rel diff of log_prob %.1e" % (
        bins, t_f, 65536 / t_f / 1e3, t_u, t_u / t_f, rel))
