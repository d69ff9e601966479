#!/usr/bin/env python3
"""Times nf_linear_wgrad against torch (rocBLAS) on the conditioner's weight-gradient shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import normflows_amd as nfa  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for B, M, N in [(65536, 128, 128), (65536, 128, 32), (65536, 736, 128), (65536, 768, 128), (65536, 64, 64)]:
    dy = torch.randn(B, M, device=dev)
    x = torch.randn(B, N, device=dev)
    t_ours = timeit(lambda: nfa.ops.linear_wgrad(dy, x))
    t_lib = timeit(lambda: (dy.t() @ x, dy.sum(0)))
    gf = 2.0 * B * M * N / 1e9
    print("B=%d M=%d N=%d: nf_linear_wgrad %.1f us (%.1f TFLOP/s)   torch mm+sum %.1f us" % (B, M, N, t_ours, gf / t_ours * 1e3, t_lib))
