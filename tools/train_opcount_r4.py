import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import normflows_amd as nfa
from bench import build_c2_model, c2_inputs
dev="cuda:0"
m = build_c2_model().to(dev); x = c2_inputs().to(dev)
opt = torch.optim.Adam(m.parameters(), lr=1e-4, fused=True)
def step():
    opt.zero_grad(set_to_none=True); l = m.forward_kld(x); l.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=False) as prof:
    step(); torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True)
rows = [(e.key, e.count, str(e.input_shapes)[:80], e.self_device_time_total) for e in ka if e.key in ("aten::fill_", "aten::zero_", "aten::zeros", "aten::copy_", "aten::clone", "aten::contiguous", "aten::zeros_like", "aten::empty", "aten::sum", "aten::add_", "aten::mul")]
rows.sort(key=lambda r: -r[1])
for r in rows[:40]: print(r)
