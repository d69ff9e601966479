#!/usr/bin/env python3
"""Where the small library launches of a training step come from: one profiled step (torch.profiler, Python stacks) of the
configs[1] model, aten operators that launch device work grouped by the innermost frame of this package."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import build_c2_model, c2_inputs  # noqa: E402

dev = torch.device("cuda:0")
m = build_c2_model().to(dev)
x = c2_inputs(65536).to(dev)
opt = torch.optim.Adam(m.parameters(), lr=1e-4)


def step():
    opt.zero_grad(set_to_none=True)
    loss = m.forward_kld(x)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
by = collections.Counter()
tm = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.self_device_time_total <= 0:
        continue
    where = "?"
    for fr in ev.stack or []:
        if "normalizing-flows_amd" in fr or "normflows_amd" in fr or "bench.py" in fr or "optim" in fr:
            where = fr.split("/")[-1]
            if "ops.py" not in where and "_lib.py" not in where:
                break
    where = str(ev.input_shapes)[:70]
    by[(ev.name, where)] += 1
    tm[(ev.name, where)] += ev.self_device_time_total
tot = sum(tm.values())
print("leaf aten ops with device time: %d launches, %.2f ms" % (sum(by.values()), tot / 1e3))
for k, n in sorted(by.items(), key=lambda kv: -tm[kv[0]])[:45]:
    print("%5d  %8.1f us  %-28s %s" % (n, tm[k], k[0], k[1]))
