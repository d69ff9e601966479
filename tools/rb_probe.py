"""Stand-alone time of nf_rows_block (residual block forward / backward in one launch).  With NF_MI355X_LIB pointing at a build
with -DNF_RB_ABL_COALESCED (tools/build_variant.py rbcoal "-DNF_RB_ABL_COALESCED" rows_linear.hip) every global access of a tile
is issued as consecutive 16-byte pieces per lane (wrong results): measured 62 / 67 us -> 60 / 62 us, i.e. the kernel is not bound by
its per-lane-row access pattern."""
import sys, torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import normflows_amd as nfa
from normflows_amd import ops
dev = torch.device('cuda:0')
B = 65536
x = torch.randn(B, 128, device=dev); t = torch.randn(B, 128, device=dev); g = torch.randn(B, 128, device=dev)
W1 = torch.randn(128, 128, device=dev) * 0.05; W2 = torch.randn(128, 128, device=dev) * 0.05; b = torch.zeros(128, device=dev)
def timeit(f, n=50):
    for _ in range(10): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
print("fwd  %.1f us" % timeit(lambda: ops.rows_block(x, W1, b, W2, b)))
print("bwd  %.1f us" % timeit(lambda: ops.rows_block(g, W2, None, W1, None, trans=True, mask1=t, mask2=x, relu=False)))
