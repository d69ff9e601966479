#!/usr/bin/env python3
"""Secondary measurement: one training step (forward_kld + backward + Adam) of the BASELINE configs[1] model on an
MI355X through the autograd path (HIP forward kernels, HIP spline backward, library GEMMs for the conditioner and
the LU parameter gradients).  python tools/train_bench.py [--batch 65536] [--steps 5]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import build_c2_model, c2_inputs  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=65536)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--no-wgrad-pair", action="store_true", help="ablation: the residual blocks' weight gradients as two launches")
ap.add_argument("--no-matvec2", action="store_true", help="ablation: LULinearPermute's chained products as two launches each")
ap.add_argument("--no-train-full", action="store_true", help="ablation: per-module Functions instead of the whole-layer forward launch")
ap.add_argument("--no-resblock-bwd", action="store_true", help="ablation: residual-block backward as separate kernels")
ap.add_argument("--no-lu-bwd", action="store_true", help="ablation: LULinearPermute's backward as separate kernels")
ap.add_argument("--no-final-bwd", action="store_true", help="ablation: stand-alone spline backward + library GEMM instead of nf_final_bwd")
ap.add_argument("--no-prepack", action="store_true", help="ablation: every layer packs its own weights / LU factors (2 x 32 launches)")
ap.add_argument("--fused-adam", action="store_true", help="torch.optim.Adam(fused=True) instead of the default foreach implementation")
ap.add_argument("--flat", action="store_true", help="round 6: dp.FlatParameters -- parameters as views of one flat tensor, gradients written into one "
                "flat buffer by the kernels, Adam(fused) on ONE tensor")
ap.add_argument("--no-pair", action="store_true", help="ablation: LULinearPermuteFn + CouplingTrainFn instead of the fused pair (round 6)")
ap.add_argument("--no-onecall", action="store_true", help="ablation: the layer's backward kernel by kernel with one reduction launch each (rounds 3-5)")
ap.add_argument("--async", dest="async_", action="store_true", help="the pair backward's reduction launches on a side stream (round 6, late; measured: no gain)")
a = ap.parse_args()
if a.async_:
    import normflows_amd
    normflows_amd.config.set_train_reduce_async(True)
if a.no_train_full:
    import normflows_amd
    normflows_amd.config.set_train_full(False)
if a.no_resblock_bwd:
    import normflows_amd
    normflows_amd.config.set_resblock_bwd(False)
if a.no_lu_bwd:
    import normflows_amd
    normflows_amd.config.set_lu_bwd_fused(False)
if a.no_final_bwd:
    import normflows_amd
    normflows_amd.config.set_final_bwd_fused(False)
if a.no_prepack:
    import normflows_amd
    normflows_amd.config.set_train_prepack(False)
if a.no_matvec2:
    import normflows_amd
    normflows_amd.config.set_lu_matvec2(False)
if a.no_wgrad_pair:
    import normflows_amd
    normflows_amd.config.set_wgrad_pair(False)
if a.no_onecall:
    import normflows_amd
    normflows_amd.config.set_train_bwd_onecall(False)
if a.no_pair:
    import normflows_amd
    normflows_amd.config.set_train_pair(False)
dev = torch.device("cuda:0")
m = build_c2_model().to(dev)
x = c2_inputs(a.batch).to(dev)
flat = None
if a.flat:
    import normflows_amd
    flat = normflows_amd.dp.FlatParameters(m)
    opt = torch.optim.Adam(flat.parameters(), lr=1e-4, fused=True)
else:
    opt = torch.optim.Adam(m.parameters(), lr=1e-4, fused=True) if a.fused_adam else torch.optim.Adam(m.parameters(), lr=1e-4)
for i in range(2 + a.steps):
    if i == 2:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    if flat is not None:
        flat.zero_grad()
    else:
        opt.zero_grad(set_to_none=True)
    loss = m.forward_kld(x)
    loss.backward()
    if flat is not None:
        flat.sync()
    opt.step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
print("train step: %.2f ms  -> %.0f samples/s  (loss %.4f, peak mem %.1f GB)%s" % (
    dt * 1e3, a.batch / dt, float(loss), torch.cuda.max_memory_allocated() / 2 ** 30,
    "  [flat parameters]" if flat is not None else ""))
