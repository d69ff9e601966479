"""Where do two training runs of one model part ways?  (round-3 investigation of test_model_level_prepack_gives_identical_steps)

Runs the 3-step Adam trajectory of the test for the mode pairs ON/OFF, OFF/OFF, ON/ON (config.train_prepack) from the same
deep-copied weights and prints, per step, the largest relative difference of the loss, of every parameter's gradient and of
every parameter -- so that a pack bug (step-1 gradients already differ between ON and OFF but not between OFF and OFF) can be
told from run-to-run noise amplified by the optimizer (OFF/OFF parts ways as well).

    python tools/prepack_diag.py [rows] [layers]
"""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import normflows_amd as nfa                      # noqa: E402
from bench import build_c2_model                 # noqa: E402


def run(m0, x, on, steps=3, lr=1e-3):
    nfa.config.set_train_prepack(on)
    m = copy.deepcopy(m0)
    opt = torch.optim.Adam(m.parameters(), lr=lr)
    out = []
    for _ in range(steps):
        opt.zero_grad(set_to_none=True)
        loss = m.forward_kld(x)
        loss.backward()
        g = [p.grad.clone() for p in m.parameters()]
        opt.step()
        out.append((float(loss.detach()), g, [p.detach().clone() for p in m.parameters()]))
    return out, [n for n, _ in m.named_parameters()]


def worst(names, A, B):
    w = (0.0, "", 0.0, 0.0)
    for n, a, b in zip(names, A, B):
        scale = max(float(b.abs().max()), 1e-30)
        d = float((a - b).abs().max())
        if d / scale > w[0]:
            w = (d / scale, n, d, scale)
    return w


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    layers = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m0 = build_c2_model(num_layers=layers, sigma=0.05).to(dev)
    x = torch.randn(rows, 64, device=dev)
    for pa, pb in ((True, False), (False, False), (True, True)):
        ra, names = run(m0, x, pa)
        rb, _ = run(m0, x, pb)
        print("== prepack %s vs %s" % (pa, pb))
        for s, (a, b) in enumerate(zip(ra, rb)):
            wg, wp = worst(names, a[1], b[1]), worst(names, a[2], b[2])
            print("  step %d: loss %.9g vs %.9g | grad worst rel %.3e (%s: |d| %.3e of %.3e) | param worst rel %.3e (%s: |d| %.3e)"
                  % (s + 1, a[0], b[0], wg[0], wg[1], wg[2], wg[3], wp[0], wp[1], wp[2]))
        # which gradient elements are small enough for Adam's g / (|g| + eps) to amplify their noise (step 1)
        tiny = 0
        tot = 0
        for n, g in zip(names, ra[0][1]):
            tiny += int((g.abs() < 1e-6).sum() - (g == 0).sum())
            tot += g.numel()
        print("  step-1 gradient elements with 0 < |g| < 1e-6: %d of %d" % (tiny, tot))
    nfa.config.set_train_prepack(True)


if __name__ == "__main__":
    main()
