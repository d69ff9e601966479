"""Round 6 (last session): config 4's model at 4 x its batch (1 024 images: 262 144 pixel rows at the 16x16 level) and at a ragged batch
(77 images), forward_kld + backward with the session's switches on and off: bit-identical loss and gradients, finite."""
import os, sys, json
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
with torch.no_grad():
    m.log_prob(torch.rand(256, 3, 32, 32, device=dev))
out = {}
for B in (1024, 77):
    x = torch.rand(B, 3, 32, 32, device=dev)
    res = []
    for on in (True, False):
        nfa.config.set_glow_weights_batched(on); nfa.config.set_lazy_logdet(on); nfa.config.set_made_tr128(on)
        m.zero_grad(set_to_none=True)
        loss = m.forward_kld(x)
        loss.backward()
        torch.cuda.synchronize()
        res.append([loss.detach().clone()] + [p.grad.clone() for p in m.parameters()])
    nfa.config.set_glow_weights_batched(True); nfa.config.set_lazy_logdet(True); nfa.config.set_made_tr128(True)
    out["B%d" % B] = {"loss": float(res[0][0]), "finite": all(bool(torch.isfinite(t).all()) for t in res[0]),
                     "bitwise_equal_on_off": all(torch.equal(a, b) for a, b in zip(res[0], res[1])),
                     "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}
    del res, x
    torch.cuda.empty_cache()
print(json.dumps(out), flush=True)
