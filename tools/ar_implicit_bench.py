#!/usr/bin/env python3
"""Autoregressive spline layer, sampling direction under autograd (neural_spline/autoregressive.py:94-134 through
wrapper.py:140-155: reverse-KLD training / differentiable sampling): implicit differentiation (autograd.ArInverseImplicitFn) against the
reference's D recorded passes (config.set_ar_implicit(False)), forward + backward of one layer.
python tools/ar_implicit_bench.py [--out gpurun_out/ar_implicit.json]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import normflows_amd as nfa  # noqa: E402
from normflows_amd.autograd import ArInverseImplicitFn  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--out", default="")
a = ap.parse_args()
dev = torch.device("cuda:0")
rows = []
for D, H, B in ((16, 128, 16384), (64, 256, 16384), (128, 512, 4096)):
    torch.manual_seed(D)
    layer = nfa.flows.AutoregressiveRationalQuadraticSpline(D, 2, H)
    with torch.no_grad():
        for p in layer.parameters():
            p.add_(0.03 * torch.randn_like(p))
    layer = layer.to(dev)
    z0 = torch.randn(B, D, device=dev)
    rec = {"D": D, "hidden": H, "rows": B}
    for mode in (True, False):
        nfa.config.set_ar_implicit(mode)
        ms = []
        torch.cuda.reset_peak_memory_stats()
        for it in range(4):
            layer.zero_grad(set_to_none=True)
            z = z0.clone().requires_grad_(True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            x, ld = layer(z)
            (x.square().sum() - ld.sum()).backward()
            torch.cuda.synchronize()
            ms.append((time.perf_counter() - t0) * 1e3)
        key = "implicit" if mode else "d_pass"
        rec[key + "_ms"] = round(min(ms[1:]), 2)
        rec[key + "_peak_mem_MB"] = round(torch.cuda.max_memory_allocated() / 2 ** 20, 1)
        if mode:
            rec["sweeps"] = ArInverseImplicitFn.last_sweeps
            g_imp = [z.grad.clone()] + [p.grad.clone() for p in layer.parameters()]
        else:
            g_ref = [z.grad.clone()] + [p.grad.clone() for p in layer.parameters()]
    # ground truth: the D recorded passes in float64
    import copy
    l64 = copy.deepcopy(layer).double()
    z = z0.double().requires_grad_(True)
    x, ld = l64(z)
    (x.square().sum() - ld.sum()).backward()
    g64 = [z.grad] + [p.grad for p in l64.parameters()]
    nfa.config.set_ar_implicit(True)
    err = lambda g: max(float((u.double() - w).abs().max()) / max(1.0, float(w.abs().max())) for u, w in zip(g, g64))
    rec["implicit_err_vs_f64_of_scale"], rec["d_pass_err_vs_f64_of_scale"] = err(g_imp), err(g_ref)
    rec["worst_grad_err_of_scale"] = max(float((u - w).abs().max()) / max(1.0, float(w.abs().max())) for u, w in zip(g_imp, g_ref))
    rec["speedup"] = round(rec["d_pass_ms"] / rec["implicit_ms"], 2)
    print(json.dumps(rec))
    rows.append(rec)
if a.out:
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump({"what": "AR-NSF layer forward (sampling direction) + backward, implicit vs D recorded passes", "cases": rows}, open(a.out, "w"), indent=1)
