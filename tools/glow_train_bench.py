"""forward_kld + backward of BASELINE configs[3]'s model (Glow L = 3, K = 32, hidden 256, 32x32x3, batch 256): where the training
step's time goes (run under rocprofv3 --kernel-trace --stats)."""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import normflows_amd as nfa

dev = "cuda:0"
torch.manual_seed(0)
L_, K_, hidden, channels = 3, 32, 256, 3
input_shape = (3, 32, 32)
q0, merges, flows = [], [], []
for i in range(L_):
    fl = [nfa.flows.GlowBlock(channels * 2 ** (L_ + 1 - i), hidden, split_mode="channel", scale=True) for _ in range(K_)]
    fl += [nfa.flows.Squeeze()]
    flows += [fl]
    if i > 0:
        merges += [nfa.flows.Merge()]
        latent = (input_shape[0] * 2 ** (L_ - i), input_shape[1] // 2 ** (L_ - i), input_shape[2] // 2 ** (L_ - i))
    else:
        latent = (input_shape[0] * 2 ** (L_ + 1), input_shape[1] // 2 ** L_, input_shape[2] // 2 ** L_)
    q0 += [nfa.distributions.DiagGaussian(latent)]
m = nfa.MultiscaleFlow(q0, flows, merges, class_cond=False).to(dev)
x = torch.rand(256, 3, 32, 32, device=dev)
with torch.no_grad():
    m.log_prob(x)                                 # ActNorm's data-dependent init


def step():
    m.zero_grad(set_to_none=True)
    m.forward_kld(x).backward()


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    step()
torch.cuda.synchronize()
print(json.dumps({"glow_c4_forward_kld_backward_ms": (time.perf_counter() - t0) * 1e3 / 3}), flush=True)
