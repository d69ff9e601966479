#!/usr/bin/env python3
"""Density estimation with a neural spline flow on the MI355X path -- the training loop of the reference's
examples (forward KL, Adam), with `normflows_amd` in place of `normflows`.

    python examples/nsf_density.py [--steps 300] [--batch 4096] [--dim 16]

Data: a synthetic mixture (two interleaved noisy arcs in the first two coordinates, Gaussian noise elsewhere).
Runs on cuda:0; the layers have no CPU path.
"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import normflows_amd as nf  # noqa: E402


def sample_data(n, dim, gen, device):
    half = n // 2
    t = torch.rand(n, generator=gen) * math.pi
    x = torch.empty(n, dim)
    x[:half, 0], x[:half, 1] = torch.cos(t[:half]), torch.sin(t[:half])
    x[half:, 0], x[half:, 1] = 1.0 - torch.cos(t[half:]), 0.5 - torch.sin(t[half:])
    x[:, :2] += 0.08 * torch.randn(n, 2, generator=gen)
    x[:, 2:] = 0.5 * torch.randn(n, dim - 2, generator=gen)
    return x.to(device)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--dim", type=int, default=16)
    ap.add_argument("--layers", type=int, default=6)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    flows = []
    for _ in range(a.layers):
        flows += [nf.flows.CoupledRationalQuadraticSpline(a.dim, 2, 64, num_bins=8), nf.flows.LULinearPermute(a.dim)]
    model = nf.NormalizingFlow(nf.distributions.DiagGaussian(a.dim, trainable=False), flows).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=2e-3)
    gen = torch.Generator().manual_seed(1)
    first = last = None
    for step in range(a.steps):
        x = sample_data(a.batch, a.dim, gen, dev)
        opt.zero_grad()
        loss = model.forward_kld(x)
        if torch.isfinite(loss):
            loss.backward()
            opt.step()
        if step % 50 == 0 or step == a.steps - 1:
            print("step %4d  forward KL %.4f nats  (%.4f nats/dim)" % (step, loss.item(), loss.item() / a.dim), flush=True)
        first = loss.item() if first is None else first
        last = loss.item()
    with torch.no_grad():
        x = sample_data(8192, a.dim, gen, dev)
        nll = -model.log_prob(x).mean()
        xs, lq = model.sample(8192)
        print("held-out NLL %.4f nats; log_prob(sample) - log_q max diff %.2e" % (
            float(nll), float((model.log_prob(xs) - lq).abs().max())))
    return first, last


if __name__ == "__main__":
    main()
