#!/usr/bin/env python3
"""Times nf_rqs_fused_train_full_fwd (whole-layer training forward) alone on the benchmark layer shape; NF_MI355X_LIB selects a
build variant (tools/build_variant.py, e.g. -DNF_ABL_NOACT / -DNF_ABL_NOCOND: without the activation / parameter-row stores)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import normflows_amd as nfa  # noqa: E402,F401
from normflows_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
B, D = 65536, 64
torch.manual_seed(0)
x = torch.randn(B, D, device=dev)
uw, uh, ud = torch.randn(32, 8, device=dev), torch.randn(32, 8, device=dev), torch.randn(32, 7, device=dev)
iidx = torch.arange(0, D, 2, device=dev)
blob = ops.rqs_fused_train_blob(2, dev)
wf, bf = torch.randn(736, 128, device=dev) * 0.05, torch.randn(736, device=dev) * 0.1
w0, b0 = torch.randn(128, 32, device=dev) * 0.1, torch.randn(128, device=dev) * 0.1
wb = [torch.randn(128, 128, device=dev) * 0.05 for _ in range(4)]
bb = [torch.randn(128, device=dev) * 0.1 for _ in range(4)]
wfull_t, wpad = torch.zeros(64, 128, device=dev), torch.zeros(32, 24, 128, device=dev)
ops.rqs_fused_pack_all(blob, w0, b0, wb, bb, wf, bf, uw, uh, ud, wfull=wfull_t, wpad=wpad, identity_idx=iidx)
for _ in range(3):
    ops.rqs_fused_train_full_fwd(x, blob, 0, 2)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        ops.rqs_fused_train_full_fwd(x, blob, 0, 2)
    e.record()
    torch.cuda.synchronize()
    ts.append(s.elapsed_time(e) / 10 * 1e3)
ts.sort()
print("nf_rqs_fused_train_full_fwd B=%d: median %.1f us  min %.1f us  [%s]" % (B, ts[len(ts) // 2], ts[0], os.environ.get("NF_MI355X_LIB", "default")))
