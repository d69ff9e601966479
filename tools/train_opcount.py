"""ATen-level op counts of one training step (which host-side ops launch the small fill / copy kernels)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_c2_model
dev = torch.device("cuda:0")
m = build_c2_model().to(dev)
x = torch.randn(65536, 64, device=dev)
opt = torch.optim.Adam(m.parameters(), lr=1e-4)
def step():
    opt.zero_grad(set_to_none=True)
    loss = m.forward_kld(x); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step()
torch.cuda.synchronize()
ka = prof.key_averages(group_by_stack_n=6)
rows = [e for e in ka if e.key in ("aten::zeros", "aten::zeros_like", "aten::fill_", "aten::zero_", "aten::clone", "aten::copy_", "aten::contiguous", "aten::neg", "aten::index_select")]
rows.sort(key=lambda e: -e.count)
for e in rows[:40]:
    st = [s for s in e.stack if "normalizing-flows_amd" in s or "torch/optim" in s or "bench" in s][:3]
    print(e.count, e.key, " | ".join(s.split("normalizing-flows_amd/")[-1][:70] for s in st))
