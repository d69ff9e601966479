"""Round 6: config 4's training step (forward_kld + backward, eager and as one hipGraph) with one switch of config.py on and off --
alternating on one box; also checks that both give the same gradient bits.  NF_AB = leaf_async (the parameter-gradient launches on the
side stream, default) | weights_batched (a level's 1x1-convolution matrices and LU-factor gradients in one launch each) | lazy_logdet
(a level's log-det statements as one launch)."""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import normflows_amd as nfa

dev = "cuda:0"
torch.manual_seed(0)
L_, K_, hidden, channels = 3, int(os.environ.get("NF_GLOW_K", "32")), 256, 3
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


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n


def capture():
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    m.zero_grad(set_to_none=True)
    with torch.cuda.graph(g):
        m.forward_kld(x).backward()
    return g


res = {"eager_ms": {}, "graph_ms": {}}
grads = {}
for rnd in range(2):
    for mode in (False, True):
        getattr(nfa.config, {"weights_batched": "set_glow_weights_batched", "lazy_logdet": "set_lazy_logdet"}.get(os.environ.get("NF_AB"), "set_train_leaf_async"))(mode)
        step(); step()
        e = timed(step, 3)
        grads.setdefault(mode, [p.grad.clone() for p in m.parameters()])
        g = capture()
        g.replay()
        r = timed(g.replay, 5)
        gg = [p.grad.clone() for p in m.parameters()]
        same = all(torch.equal(a, b) for a, b in zip(grads[mode], gg))
        res["eager_ms"].setdefault(str(mode), []).append(round(e, 2))
        res["graph_ms"].setdefault(str(mode), []).append(round(r, 2))
        res.setdefault("graph_equals_eager", {}).setdefault(str(mode), []).append(same)
        del g
        m.zero_grad(set_to_none=True)
res["async_equals_sync_bits"] = all(torch.equal(a, b) for a, b in zip(grads[False], grads[True]))
res["peak_mem_gb"] = torch.cuda.max_memory_allocated() / 2 ** 30
print(json.dumps(res), flush=True)
