"""One training step of BASELINE configs[4]'s model (MAF, 10 x [MaskedAffineAutoregressive(128, 512) + Permute], B = 65 536) in its
single-pass direction: loss = reverse-KLD-style objective through `forward` (sample + log_q), backward, Adam -- hand-written MADE
kernels vs torch autograd through library GEMMs."""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import normflows_amd as nfa

dev = "cuda:0"
B = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 65536
modes = (True,) if "--only" in sys.argv else (True, False)
torch.manual_seed(0)
flows = []
for _ in range(10):
    flows += [nfa.flows.MaskedAffineAutoregressive(128, 512, num_blocks=2), nfa.flows.Permute(128, mode="swap")]
model = nfa.NormalizingFlow(nfa.distributions.DiagGaussian(128), flows).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)
eps = torch.randn(B, 128, device=dev)


def step():
    opt.zero_grad(set_to_none=True)
    z = eps
    logq = torch.zeros(B, device=dev)
    for f in model.flows:                      # core.py:167-180 (sample): z, log_det = flow(z); log_q -= log_det
        z, ld = f(z)
        logq = logq - ld
    loss = (logq + 0.5 * (z ** 2).sum(1)).mean()
    loss.backward()
    opt.step()
    return loss


res = {"batch": B}
for mode in modes:
    nfa.config.set_made_train(mode)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    res["hand_written_ms" if mode else "library_ms"] = (time.perf_counter() - t0) * 200
nfa.config.set_made_train(True)
print(json.dumps(res), flush=True)
