"""forward_kld + backward of NSF models whose conditioner is wider than 128 hidden units (4 x [CoupledRQS(D, 2, hidden) + LULinearPermute(D)],
B = 65 536): the ResidualNet on the MADE training kernels (csrc/made_bwd.hip, dense) vs torch autograd through library GEMMs."""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import normflows_amd as nfa
from bench import build_c2_model, c2_inputs

dev = "cuda:0"
out = {}
for D, H in ((64, 256), (128, 256)):
    m = build_c2_model(num_layers=4, dim=D, hidden=H, seed=1, sigma=0.01).to(dev)
    x = c2_inputs(65536, D).to(dev)

    def step():
        m.zero_grad(set_to_none=True)
        m.forward_kld(x).backward()
    r = {}
    for mode in (True, False):
        nfa.config.set_made_train(mode)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            step()
        torch.cuda.synchronize()
        r["hand_written_ms" if mode else "library_ms"] = (time.perf_counter() - t0) * 250
    nfa.config.set_made_train(True)
    out["d%d_h%d" % (D, H)] = r
print(json.dumps(out), flush=True)
