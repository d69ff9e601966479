#!/usr/bin/env python3
"""Where the small launches of one training step of the BASELINE configs[1] model come from: the aten fill / copy / zero / clone /
index operators of one step (torch.profiler, CPU side with Python stacks), counted per calling source line of this package.
python tools/train_launch_audit.py [--batch 65536] [--out gpurun_out/train_launch_audit.json]"""
import argparse
import collections
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import build_c2_model, c2_inputs  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=65536)
ap.add_argument("--out", default="")
ap.add_argument("--model", default="c2", choices=["c2", "glow"], help="c2: the benchmark model (+ Adam); glow: config 4, forward_kld + backward")
a = ap.parse_args()
dev = torch.device("cuda:0")
if a.model == "c2":
    m = build_c2_model().to(dev)
    x = c2_inputs(a.batch).to(dev)
    opt = torch.optim.Adam(m.parameters(), lr=1e-4)
else:
    import normflows_amd as nfa
    torch.manual_seed(0)
    q0, merges, flows = [], [], []
    for i in range(3):
        flows += [[nfa.flows.GlowBlock(3 * 2 ** (4 - i), 256, split_mode="channel", scale=True) for _ in range(32)] + [nfa.flows.Squeeze()]]
        if i > 0:
            merges += [nfa.flows.Merge()]
        q0 += [nfa.distributions.DiagGaussian((3 * 2 ** (3 - i), 32 // 2 ** (3 - i), 32 // 2 ** (3 - i)) if i > 0 else (48, 4, 4))]
    m = nfa.MultiscaleFlow(q0, flows, merges, class_cond=False).to(dev)
    x = torch.rand(256, 3, 32, 32, device=dev)
    with torch.no_grad():
        m.log_prob(x)
    opt = None


def step():
    m.zero_grad(set_to_none=True)
    loss = m.forward_kld(x)
    loss.backward()
    if opt is not None:
        opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
WATCH = ("aten::fill_", "aten::zero_", "aten::copy_", "aten::clone", "aten::index_select", "aten::contiguous", "aten::neg",
         "aten::sum", "aten::add_", "aten::mul", "aten::add", "aten::index_copy_", "aten::zeros", "aten::zeros_like", "aten::cat",
         "aten::sub", "aten::div", "aten::exp", "aten::mean", "aten::empty_like", "aten::view_as", "aten::mul_", "aten::sigmoid",
         "aten::log", "aten::index", "aten::index_put_", "aten::triu", "aten::tril", "aten::diag", "aten::matmul", "aten::mm")
by_site = collections.Counter()
kernels = collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CUDA:
        kernels[ev.name[:60]] += 1
        continue
    if ev.name not in WATCH:
        continue
    site = "?"
    for fr in ev.stack or ():
        if ("normalizing-flows_amd" in fr or "normflows_amd" in fr or "bench.py" in fr) and "_lib.py" not in fr:
            site = fr.split("/")[-1]
            break
    by_site[(ev.name, site)] += 1
rows = [{"op": k[0], "site": k[1], "calls": v} for k, v in by_site.most_common(60)]
for r in rows:
    print("%5d  %-20s %s" % (r["calls"], r["op"], r["site"]))
print("--- device launches in the step")
for k, v in kernels.most_common(25):
    print("%5d  %s" % (v, k))
if a.out:
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump({"batch": a.batch, "aten_calls_by_site": rows, "device_launches": dict(kernels.most_common(40))}, open(a.out, "w"), indent=1)
