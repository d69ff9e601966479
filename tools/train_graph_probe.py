#!/usr/bin/env python3
"""Can a whole training step (forward_kld + backward + Adam) of the benchmark model be captured into a hipGraph?
Eager vs graphed step time, same losses.  python tools/train_graph_probe.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import build_c2_model, c2_inputs  # noqa: E402

dev = torch.device("cuda:0")
m = build_c2_model().to(dev)
x = c2_inputs(65536).to(dev)
opt = torch.optim.Adam(m.parameters(), lr=1e-4, capturable=True)


def step():
    opt.zero_grad(set_to_none=False)
    loss = m.forward_kld(x)
    loss.backward()
    opt.step()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    l_e = step()
torch.cuda.synchronize()
print("eager: %.1f ms/step, loss %.4f" % ((time.perf_counter() - t0) / 5 * 1e3, float(l_e.detach())))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2):
        step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        static_loss = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    print("graphed: %.1f ms/step, loss %.4f" % ((time.perf_counter() - t0) / 5 * 1e3, float(static_loss.detach())))
except Exception as exc:   # noqa: BLE001
    print("capture failed:", repr(exc)[:500])
