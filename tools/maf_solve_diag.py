"""Diagnostic: rows where the one-pass implicit backward (nf_maf_solve_t) and the sweeps disagree most, against the float64 truth
(autograd through the D-pass loop in float64 on the CPU) -- is the difference conditioning (both float32 paths off by the same order)
or a defect of one path?  python tools/maf_solve_diag.py [B]"""
import copy
import json
import os
import sys

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")))
import normflows_amd as nfa  # noqa: E402
from normflows_amd.flows.autoregressive import Autoregressive  # noqa: E402

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 250_001
torch.manual_seed(0)
layer = nfa.flows.MaskedAffineAutoregressive(128, 512, num_blocks=2)
with torch.no_grad():
    for p in layer.parameters():
        p.add_(0.02 * torch.randn_like(p))
layer = layer.to(dev)
z0 = torch.randn(B, 128, device=dev)
cx, cl = torch.randn(B, 128, device=dev), torch.randn(B, device=dev)
gz = []
for onepass in (True, False):
    nfa.config.set_maf_onepass(onepass)
    z = z0.clone().requires_grad_(True)
    x, ld = layer.inverse(z)
    ((x * cx).sum() + (ld * cl).sum()).backward()
    gz.append(z.grad.clone())
nfa.config.set_maf_onepass(True)
d = (gz[0] - gz[1]).abs().max(1).values
rows = torch.topk(d, 6).indices
# float64 truth on the same device with plain torch ops (library GEMMs): the reference's D-pass loop of autoregressive.py:29-38
import torch.nn.functional as F
net = layer.autoregressive_net
lins = [net.initial_layer] + [l for b in net.blocks for l in b.linear_layers] + [net.final_layer]
W = [(l.weight.detach() * l.mask).double() for l in lins]
bs = [l.bias.detach().double() for l in lins]


def made64(xx):
    h = F.linear(xx, W[0], bs[0])
    for b in range(len(net.blocks)):
        t = F.linear(torch.relu(h), W[1 + 2 * b], bs[1 + 2 * b])
        h = h + F.linear(torch.relu(t), W[2 + 2 * b], bs[2 + 2 * b])
    return F.linear(h, W[-1], bs[-1])


zc = z0[rows].double().requires_grad_(True)
xr = torch.zeros_like(zc)
for _ in range(128):
    prm = made64(xr).view(len(rows), 128, 2)
    scale = torch.sigmoid(prm[..., 0] + 2.0) + 1e-3
    xr = (zc - prm[..., 1]) / scale
ldr = -torch.log(scale).sum(1)
((xr * cx[rows].double()).sum() + (ldr * cl[rows].double()).sum()).backward()
truth = zc.grad.cpu()
out = {"B": B, "rows": rows.tolist(), "max_abs_gz": float(gz[1].abs().max()),
       "onepass_minus_sweeps_max_abs": [float(v) for v in d[rows]],
       "onepass_vs_f64_max_abs": [float(v) for v in (gz[0][rows].double().cpu() - truth).abs().max(1).values],
       "sweeps_vs_f64_max_abs": [float(v) for v in (gz[1][rows].double().cpu() - truth).abs().max(1).values],
       "truth_row_max_abs": [float(v) for v in truth.abs().max(1).values]}
print(json.dumps(out), flush=True)
