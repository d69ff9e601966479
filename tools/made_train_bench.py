"""Forward + backward of one MADE / MAF layer under autograd at BASELINE configs[4]'s layer (d = 128, hidden 512, B = 65 536):
hand-written path vs torch autograd through library GEMMs; run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import normflows_amd as nfa

dev = "cuda:0"
B = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 65536
mult = int(sys.argv[sys.argv.index("--mult") + 1]) if "--mult" in sys.argv else 2
modes = (True,) if "--only" in sys.argv else (True, False)
torch.manual_seed(0)
made = nfa.nets.MADE(128, 512, num_blocks=2, output_multiplier=mult).to(dev)
x = torch.randn(B, 128, device=dev)
gp = torch.randn(B, mult * 128, device=dev)


def step():
    made.zero_grad(set_to_none=True)
    xx = x.clone().requires_grad_(True)
    out = made(xx)
    out.backward(gp)


res = {}
for mode in modes:
    nfa.config.set_made_train(mode)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    res["hand_written_ms" if mode else "library_ms"] = (time.perf_counter() - t0) * 100
nfa.config.set_made_train(True)
flop = 3 * 2 * B * (128 * 512 + 4 * 512 * 512 + 512 * mult * 128)
res.update(batch=B, mult=mult, dense_flop_fwd_bwd=flop)
if "hand_written_ms" in res:
    res["dense_tflops"] = flop / res["hand_written_ms"] / 1e9
print(json.dumps(res), flush=True)
