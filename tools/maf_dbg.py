#!/usr/bin/env python3
"""python tools/maf_dbg.py D H B [num_blocks]: nf_maf_inverse (incremental) against the D-pass loop on one random layer."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import normflows_amd as nfa  # noqa: E402
from normflows_amd.flows.autoregressive import Autoregressive  # noqa: E402

D, H, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
NB = int(sys.argv[4]) if len(sys.argv) > 4 else 2
dev = torch.device("cuda:0")
torch.manual_seed(D * 1000 + H)
layer = nfa.flows.MaskedAffineAutoregressive(D, H, num_blocks=NB).to(dev)
with torch.no_grad():
    for p in layer.parameters():
        p.add_(0.05 * torch.randn_like(p))
z = torch.randn(B, D, device=dev)
with torch.no_grad():
    x1, ld1 = layer.inverse(z)
    torch.cuda.synchronize()
    print("kernel ok; finite:", bool(torch.isfinite(x1).all()))
    x0, ld0 = Autoregressive.inverse(layer, z)
    torch.cuda.synchronize()
print("D=%d H=%d B=%d: max |dx| %.3e  max |dld| %.3e" % (D, H, B, float((x1 - x0).abs().max()), float((ld1 - ld0).abs().max())))
