"""log_prob time of NSF models next to the benchmark shape: [CoupledRationalQuadraticSpline(D, 2, hidden) + LULinearPermute(D)] x 4
at B = 65 536 for (D, hidden) in the fused kernel's range and beyond it (hidden 256, D 128: library GEMMs for the conditioner,
nf_rqs_coupling's pipelined kernel, the dense LU product).  python tools/wide_bench.py [--json out.json] [--only D hidden] [--bins K]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import normflows_amd as nfa                      # noqa: E402

dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
B, pairs, out = 65536, 4, []
shapes = ((64, 128), (64, 256), (128, 128), (128, 256), (96, 192))
if "--only" in sys.argv:
    i = sys.argv.index("--only")
    shapes = ((int(sys.argv[i + 1]), int(sys.argv[i + 2])),)
K = int(sys.argv[sys.argv.index("--bins") + 1]) if "--bins" in sys.argv else 8
for D, hidden in shapes:
    torch.manual_seed(0)
    flows = []
    for _ in range(pairs):
        flows += [nfa.flows.CoupledRationalQuadraticSpline(D, 2, hidden, num_bins=K), nfa.flows.LULinearPermute(D)]
    m = nfa.NormalizingFlow(nfa.distributions.DiagGaussian(D, trainable=False), flows).to(dev)
    for p in m.parameters():
        p.add_(0.01 * torch.randn_like(p))
    x = torch.randn(B, D, device=dev)

    def timed():
        for _ in range(3):
            m.log_prob(x)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            m.log_prob(x)
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / 10

    ms = timed()
    lp1 = m.log_prob(x)
    nfa.config.set_nsf_wide(False)          # round 3's path for the shapes beyond the benchmark kernel's: library GEMMs + nf_rqs_coupling
    ms_lw = timed()
    lp0 = m.log_prob(x)
    nfa.config.set_nsf_wide(True)
    flop = 2.0 * B * pairs * (D // 2 * hidden + 4 * hidden * hidden + hidden * (D // 2) * (3 * K - 1) + D * D)
    out.append(dict(D=D, hidden=hidden, bins=K, pairs=pairs, rows=B, ms=round(ms, 3), us_per_pair=round(ms * 1e3 / pairs, 1),
                    mrows_per_s=round(B / ms / 1e3, 2), tflops=round(flop / ms / 1e9, 1),
                    frac_of_fp32_mfma_peak=round(flop / ms / 1e9 / 157.3, 3), layerwise_ms=round(ms_lw, 3),
                    max_rel_diff_log_prob_vs_layerwise=float(((lp1 - lp0).abs() / lp0.abs().clamp_min(1.0)).max())))
    print("D = %3d hidden = %3d: %7.3f ms for %d pairs = %6.1f us per pair, %5.2f M rows/s, %5.1f TFLOP/s (fp32)"
          % (D, hidden, ms, pairs, ms * 1e3 / pairs, B / ms / 1e3, flop / ms / 1e9))
if "--json" in sys.argv:
    json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
