#!/usr/bin/env python3
"""Times nf_rqs_fused_train_fwd (final Linear + coupling transform, training variant) alone on the benchmark layer shape.
NF_MI355X_LIB selects a build variant (tools/build_variant.py)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import normflows_amd as nfa  # noqa: E402
from normflows_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B = 65536
torch.manual_seed(0)
x = torch.randn(B, 64, device=dev)
h2 = torch.randn(B, 128, device=dev)
wf = torch.randn(736, 128, device=dev) * 0.05
bf = torch.randn(736, device=dev) * 0.1
uw, uh, ud = torch.randn(32, 8, device=dev), torch.randn(32, 8, device=dev), torch.randn(32, 7, device=dev)
blob = ops.rqs_fused_train_blob(2, dev)
ops.rqs_fused_pack_final(blob, wf, bf, uw, uh, ud, 2)
for _ in range(3):
    ops.rqs_fused_train_fwd(x, h2, blob, 0, 2)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        ops.rqs_fused_train_fwd(x, h2, blob, 0, 2)
    e.record()
    torch.cuda.synchronize()
    ts.append(s.elapsed_time(e) / 10 * 1e3)
ts.sort()
print("nf_rqs_fused_train_fwd B=%d: median %.1f us  min %.1f us  [%s]" % (B, ts[len(ts) // 2], ts[0], os.environ.get("NF_MI355X_LIB", "default")))
