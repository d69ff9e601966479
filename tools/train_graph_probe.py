"""Does a whole training step (forward_kld + backward) of the hand-written paths capture into ONE hipGraph (PyTorch's whole-network
capture recipe)?  Glow config 4 and the MAF single-pass direction; prints eager vs replay time."""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import normflows_amd as nfa

dev = "cuda:0"


def glow():
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
        m.log_prob(x)
    return m, x, lambda mm, xx: mm.forward_kld(xx)


def maf():
    torch.manual_seed(0)
    m = nfa.NormalizingFlow(nfa.distributions.DiagGaussian(128, trainable=False),
                            [nfa.flows.MaskedAffineAutoregressive(128, 512, num_blocks=2) for _ in range(10)]).to(dev)
    eps = torch.randn(65536, 128, device=dev)

    def loss(mm, e):
        z, logq = e, torch.zeros(e.shape[0], device=dev)
        for f in mm.flows:
            z, ld = f(z)
            logq = logq - ld
        return (logq + 0.5 * (z ** 2).sum(1)).mean()
    return m, eps, loss


def c2():
    """the benchmark model (BASELINE configs[1]) and batch: forward_kld + backward"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from bench import build_c2_model, c2_inputs
    m = build_c2_model().to(dev)
    return m, c2_inputs().to(dev), lambda mm, xx: mm.forward_kld(xx)


def maf_density():
    """config 5 in the DENSITY direction (forward_kld = flow.inverse under autograd): one-pass implicit backward (round 5)"""
    torch.manual_seed(0)
    m = nfa.NormalizingFlow(nfa.distributions.DiagGaussian(128, trainable=False),
                            [nfa.flows.MaskedAffineAutoregressive(128, 512, num_blocks=2) for _ in range(10)]).to(dev)
    return m, torch.randn(65536, 128, device=dev), lambda mm, xx: mm.forward_kld(xx)


out = {}
which = sys.argv[1:] or ["glow_c4", "maf_c5_single_pass", "c2_benchmark_model", "maf_c5_density"]
for name, build in (("glow_c4", glow), ("maf_c5_single_pass", maf), ("c2_benchmark_model", c2), ("maf_c5_density", maf_density)):
    if name not in which:
        continue
    m, x, lossfn = build()

    def step():
        m.zero_grad(set_to_none=True)
        lossfn(m, x).backward()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    out[name + "_eager_ms"] = (time.perf_counter() - t0) * 1e3 / 3
    eager = [p.grad.clone() for p in m.parameters()]
    try:
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        m.zero_grad(set_to_none=True)
        with torch.cuda.graph(g):
            loss = lossfn(m, x)
            loss.backward()
        g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        out[name + "_graph_replay_ms"] = (time.perf_counter() - t0) * 1e3 / 5
        out[name + "_replay_equals_eager_step"] = all(torch.equal(a, p.grad) for a, p in zip(eager, m.parameters()))
    except Exception as e:                                      # noqa: BLE001
        out[name + "_graph_error"] = repr(e)[:300]
print(json.dumps(out), flush=True)
