#!/usr/bin/env python3
"""BASELINE configs[4] (MAF, 10 layers, d = 128, hidden 512, B = 65 536), forward_kld + backward in the density direction only: the step
`rocprofv3 --kernel-trace --stats` is pointed at to see what the implicit backward is made of (no D-pass comparison legs).
python tools/maf_density_profile.py [--steps 5]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import normflows_amd as nfa  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--batch", type=int, default=65536)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
flows = [nfa.flows.MaskedAffineAutoregressive(128, 512, num_blocks=2) for _ in range(10)]
m = nfa.NormalizingFlow(nfa.distributions.DiagGaussian(128, trainable=False), flows)
g = torch.Generator().manual_seed(1234)
with torch.no_grad():
    for p in m.parameters():
        p.add_(0.01 * torch.randn(p.shape, generator=g))
m = m.to(dev)
x = torch.randn(a.batch, 128, device=dev)
for i in range(2 + a.steps):
    if i == 2:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    m.zero_grad(set_to_none=True)
    m.forward_kld(x).backward()
torch.cuda.synchronize()
print("forward_kld + backward (density direction): %.2f ms per step" % ((time.perf_counter() - t0) / a.steps * 1e3))
