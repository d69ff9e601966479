"""Time of nf_rqs_coupling_bwd_p24 (software-pipelined spline backward) at the benchmark shape; with NF_MI355X_LIB pointing at an
ablation build (tools/build_variant.py ... -DNF_BWD_ABL_NOMATH / -DNF_BWD_ABL_NOIDENT) the same launch without the arithmetic."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import normflows_amd as nfa
from normflows_amd import ops
from bench import build_c2_model
dev = torch.device("cuda:0")
B = 65536
m = build_c2_model(num_layers=1).to(dev)
c = m.flows[0].prqct
u = c.unconditional_transform
x = torch.randn(B, 64, device=dev)
gy, gld = torch.randn(B, 64, device=dev), torch.randn(B, device=dev)
cond24 = 0.5 * torch.randn(B, 32, 24, device=dev)
f = lambda: ops.rqs_coupling_bwd_p24(x, gy, gld, cond24, u.unnormalized_widths.detach(), u.unnormalized_heights.detach(),
                                     u.unnormalized_derivatives.detach(), c.identity_features, c.transform_features,
                                     tail_bound=3.0, wh_div=float(128 ** 0.5))
for _ in range(5): f()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(30): f()
e.record(); torch.cuda.synchronize()
print("%s: %.1f us per launch" % (os.environ.get("NF_MI355X_LIB", "product build").split("/")[-1], s.elapsed_time(e) / 30 * 1e3))
