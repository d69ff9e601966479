#!/usr/bin/env python3
"""The REFERENCE leg of the CPU baseline (SURVEY.md section 8d last row): normflows 1.7.3 itself,
imported from /root/reference, PyTorch CPU, on the benchmark of record -- BASELINE configs[1], the same seeded model and the
same rows bench.py runs on the GPU (bench.build_c2_model(lib=normflows), bench.c2_inputs).

The reference tree exists only in the build container (the GPU box has no /root/reference).  Rounds 2-4 ran this script THERE
(8 cores); round 5 stages the package for one `gpurun` call (tools/stage_reference.py -> .refstage/, git-ignored) and runs it ON
THE GPU BOX's host cores with `--ref .refstage --where "gpu box"`.  The JSON output is committed under profiles/; bench.py reads
the newest one and reports it as the `kind: "reference"` entry of `cpu_baseline` (with where it was measured and the core count)
next to the `kind: "port"` entry it times live.

    python tools/cpu_reference.py [--rows 65536] [--ref DIR] [--where TEXT] [--out profiles/rNN_cpu_reference.json]

Threads = every core of the container (or the best of a `--threads` sweep), fp32, no_grad, 1 warm-up + best of `--repeats`
wall-clock passes for `log_prob` (core.py:182-197 over utils/splines.py:16-219) and for `sample` (core.py:167-180); the JSON is
rewritten after every stage, so a run cut short by a time limit still leaves what it measured.
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
    ap.add_argument("--ref", default=os.environ.get("NF_REFERENCE_PATH", "/root/reference"),
                    help="directory that holds the reference's `normflows/` package")
    ap.add_argument("--where", default="build container")
    ap.add_argument("--threads", default="",
                    help="comma-separated torch thread counts to try (default: every core); the best log_prob rate is reported. "
                         "On the 256-thread GPU box ALL cores is pathological for the reference's thousands of small ATen ops "
                         "(round 5: the all-core run did not finish 6 passes in 15 min), hence the sweep")
    ap.add_argument("--quick", action="store_true", help="skip the per-layer in-bound scan and the sample direction")
    ap.add_argument("--per-setting-timeout", type=float, default=0.0,
                    help="seconds; > 0: every thread count of the sweep runs in its OWN process under this limit, and a setting that "
                         "does not finish is recorded as a data point (`timeout_s`) instead of stalling the sweep (round 6: the all-core "
                         "setting of the 256-thread GPU box)")
    ap.add_argument("--worker", type=int, default=0, help=argparse.SUPPRESS)      # internal: one setting, print its entry, exit
    a = ap.parse_args()
    sys.path.insert(0, os.path.abspath(a.ref))
    import normflows as nf
    assert os.path.abspath(nf.__file__).startswith(os.path.abspath(a.ref)), (nf.__file__, a.ref)
    from bench import DIM, build_c2_model, c2_inputs
    cores = len(os.sched_getaffinity(0))
    tlist = [int(t) for t in a.threads.split(",") if t] or [cores]
    model = build_c2_model(lib=nf)
    x = c2_inputs(a.rows, DIM)
    res = {"what": "normflows %s (the reference itself) on BASELINE configs[1]: 32 x [CoupledRQS(64, 2, 128, "
                   "K=8) + LULinearPermute(64)] + DiagGaussian, fp32, no_grad" % nf.__version__,
           "where": a.where, "host_threads": cores, "cpu_model": cpu_model(), "torch": torch.__version__,
           "rows": a.rows, "repeats": a.repeats, "sweep": []}

    def dump():
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(res, open(a.out, "w"), indent=1)

    with torch.no_grad():
        if not a.quick:
            # in-bound fraction per layer in the density direction (the reference's CPU time is data dependent: only
            # in-bound elements enter the spline, utils/splines.py:77-80)
            torch.set_num_threads(min(cores, 32))
            z, fr = x, []
            for f in reversed(model.flows):
                if isinstance(f, nf.flows.CoupledRationalQuadraticSpline):
                    fr.append(float(((z >= -3.0) & (z <= 3.0)).float().mean()))
                z, _ = f.inverse(z)
            res["in_bound_fraction_min_over_layers"] = min(fr)
            dump()
        def one_setting(nt):
            torch.set_num_threads(nt)
            model.log_prob(x[:min(4096, a.rows)])                                     # warm-up
            t, ts, lp = best_of(lambda: model.log_prob(x), a.repeats)
            return {"torch_threads": nt, "best_s": t, "all_s": ts, "samples_per_s": a.rows / t,
                    "nll_nats_per_dim": float(-lp.mean() / DIM)}

        if a.worker:
            print("ENTRY " + json.dumps(one_setting(a.worker)), flush=True)
            return
        best = None
        for nt in tlist:
            if a.per_setting_timeout > 0:
                import subprocess
                cmd = [sys.executable, os.path.abspath(__file__), "--worker", str(nt), "--rows", str(a.rows), "--repeats", str(a.repeats),
                       "--ref", a.ref, "--quick"]
                t0 = time.perf_counter()
                try:
                    out = subprocess.run(cmd, capture_output=True, text=True, timeout=a.per_setting_timeout)
                    lines = [l for l in out.stdout.splitlines() if l.startswith("ENTRY ")]
                    ent = json.loads(lines[-1][6:]) if lines else {"torch_threads": nt, "error": (out.stderr or out.stdout)[-300:],
                                                                   "samples_per_s": None}
                except subprocess.TimeoutExpired:
                    ent = {"torch_threads": nt, "timeout_s": a.per_setting_timeout, "samples_per_s": None,
                           "note": "1 warm-up (4096 rows) + %d passes did not finish within the limit" % a.repeats,
                           "wall_s": time.perf_counter() - t0}
            else:
                ent = one_setting(nt)
            res["sweep"].append(ent)
            if ent.get("samples_per_s") is None:
                dump()
                print(json.dumps(ent), flush=True)
                continue
            if best is None or ent["samples_per_s"] > best["samples_per_s"]:
                best = ent
            res["log_prob"] = best
            res["cores"] = res["torch_threads"] = best["torch_threads"]
            dump()
            print(json.dumps(ent), flush=True)
        if not a.quick:
            torch.set_num_threads(best["torch_threads"])
            torch.manual_seed(1)
            model.sample(min(4096, a.rows))
            t, ts, (xs, lq) = best_of(lambda: model.sample(a.rows), a.repeats)
            res["sample"] = {"best_s": t, "all_s": ts, "samples_per_s": a.rows / t,
                             "in_bound_fraction_of_samples": float(((xs >= -3.0) & (xs <= 3.0)).float().mean())}
    dump()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
