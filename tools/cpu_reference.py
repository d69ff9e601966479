#!/usr/bin/env python3
"""The REFERENCE leg of the CPU baseline (SURVEY.md section 8d last row): normflows 1.7.3 itself,
imported from /root/reference, PyTorch CPU, on the benchmark of record -- BASELINE configs[1], the same seeded model and the
same rows bench.py runs on the GPU (bench.build_c2_model(lib=normflows), bench.c2_inputs).

The reference tree exists only in the build container (the GPU box has no /root/reference), so this script runs THERE and
its JSON output is committed under profiles/; bench.py reads that file and reports it as the `kind: "reference"` entry of
`cpu_baseline` next to the `kind: "port"` entry it times live on the GPU box's own host cores.

    python tools/cpu_reference.py [--rows 65536] [--out profiles/r02_cpu_reference.json]

Threads = every core of the container, fp32, no_grad, 1 warm-up + best of 3 wall-clock passes for `log_prob`
(core.py:182-197 over utils/splines.py:16-219) and for `sample` (core.py:167-180).
"""
import argparse
import json
import os
import platform
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


def best_of(fn, n):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        out = fn()
        ts.append(time.perf_counter() - t0)
    return min(ts), ts, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=65536)
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_cpu_reference.json"))
    a = ap.parse_args()
    import normflows as nf
    from bench import DIM, build_c2_model, c2_inputs
    cores = len(os.sched_getaffinity(0))
    torch.set_num_threads(cores)
    model = build_c2_model(lib=nf)
    x = c2_inputs(a.rows, DIM)
    res = {"what": "normflows %s (the reference itself, /root/reference) on BASELINE configs[1]: 32 x [CoupledRQS(64, 2, 128, "
                   "K=8) + LULinearPermute(64)] + DiagGaussian, fp32, no_grad" % nf.__version__,
           "where": "build container", "cores": cores, "cpu_model": cpu_model(), "torch": torch.__version__,
           "torch_threads": torch.get_num_threads(), "rows": a.rows, "repeats": a.repeats}
    with torch.no_grad():
        # in-bound fraction per layer in the density direction (the reference's CPU time is data dependent: only
        # in-bound elements enter the spline, utils/splines.py:77-80)
        z, fr = x, []
        for f in reversed(model.flows):
            if isinstance(f, nf.flows.CoupledRationalQuadraticSpline):
                fr.append(float(((z >= -3.0) & (z <= 3.0)).float().mean()))
            z, _ = f.inverse(z)
        res["in_bound_fraction_min_over_layers"] = min(fr)
        model.log_prob(x[:4096])                                     # warm-up
        t, ts, lp = best_of(lambda: model.log_prob(x), a.repeats)
        res["log_prob"] = {"best_s": t, "all_s": ts, "samples_per_s": a.rows / t,
                           "nll_nats_per_dim": float(-lp.mean() / DIM)}
        torch.manual_seed(1)
        model.sample(4096)
        t, ts, (xs, lq) = best_of(lambda: model.sample(a.rows), a.repeats)
        res["sample"] = {"best_s": t, "all_s": ts, "samples_per_s": a.rows / t,
                         "in_bound_fraction_of_samples": float(((xs >= -3.0) & (xs <= 3.0)).float().mean())}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
