"""Round 6 diagnosis: the one row of grad_model_c2_w64h128 whose input gradient differs between the pair path (LU composed) and
the separate layers sits on a kink (a ReLU pre-activation / knot within float32 rounding): perturbing the INPUT of that row by a few
ulps flips the gradient of EITHER path between the same two values."""
import os, sys, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import normflows_amd as nfa
import test_gpu_training as T
m, g = T._c2_train_model_and_fixture(nfa)
ref = g["gx_f32"]
scale = np.abs(ref).max()
row = 153
x0 = T.T(g["x"])
for on in (True, False):
    nfa.config.set_train_pair(on)
    out = []
    for k in range(-4, 5):
        x = x0.clone()
        x[row] = x[row] * (1.0 + k * 1.2e-7)
        x.requires_grad_(True)
        m.zero_grad(set_to_none=True)
        m.forward_kld(x).backward()
        e = float(np.abs(x.grad[row].cpu().numpy() - ref[row]).max() / scale)
        out.append("%.1e" % e)
    print("pair" if on else "separate", "row 153 error vs reference for input scaled by 1 + k 1.2e-7, k = -4..4:", " ".join(out))
