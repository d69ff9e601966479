#!/usr/bin/env python3
"""Phase breakdown of the GlowBlock kernels inside a level chain (debug build with -DNF_GL_TRACE, see
tools/build_variant.py): workgroup 0 stamps the 100 MHz wall clock at the phase boundaries of every block.
    python tools/build_variant.py trace "-DNF_GL_TRACE" glow_conv.hip
    NF_MI355X_LIB=normalizing-flows_amd/lib/variants/trace.so python tools/glow_trace.py"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import normflows_amd as nfa  # noqa: E402

dev = torch.device("cuda:0")
NAMES = ["prologue (load/mix/pad)", "GEMM 1", "GEMM 2+3", "col2im", "coupling/mix/log-det"]


def main():
    lib = nfa._lib.lib()
    torch.manual_seed(0)
    L_, K_, hidden, channels = 3, 32, 256, 3
    q0, merges, flows = [], [], []
    for i in range(L_):
        fl = [nfa.flows.GlowBlock(channels * 2 ** (L_ + 1 - i), hidden, split_mode="channel", scale=True) for _ in range(K_)]
        fl += [nfa.flows.Squeeze()]
        flows += [fl]
        if i > 0:
            merges += [nfa.flows.Merge()]
            latent = (3 * 2 ** (L_ - i), 32 // 2 ** (L_ - i), 32 // 2 ** (L_ - i))
        else:
            latent = (3 * 2 ** (L_ + 1), 32 // 2 ** L_, 32 // 2 ** L_)
        q0 += [nfa.distributions.DiagGaussian(latent)]
    m = nfa.MultiscaleFlow(q0, flows, merges, class_cond=False).to(dev)
    x = torch.rand(256, 3, 32, 32, device=dev)
    with torch.no_grad():
        m.log_prob(x)
        m.log_prob(x)
        for lvl, name in ((2, "16x16 wide"), (1, "8x8 small"), (0, "4x4 tiny")):
            buf = torch.zeros(8 * 64, dtype=torch.int64, device=dev)
            # trace only this level: run the passes level by level
            z = x
            log_q = torch.zeros(256, device=dev)
            for i in range(L_ - 1, -1, -1):
                lib.nf_glow_debug_trace(C.c_void_p(buf.data_ptr() if i == lvl else 0))
                z, z_ = m._level_pass(i, z, None, True, log_q, +1)
            torch.cuda.synchronize()
            t = buf.cpu().view(64, 8)[:K_].double() * 0.01   # us
            ph = [(t[:, k + 1] - t[:, k]).mean().item() for k in range(5)]
            gap = (t[1:, 0] - t[:-1, 5]).mean().item()
            print("%-12s per block %.1f us: " % (name, (t[-1, 5] - t[0, 0]).item() / K_) +
                  ", ".join("%s %.1f" % (n, v) for n, v in zip(NAMES, ph)) + ", between blocks %.2f" % gap)
    lib.nf_glow_debug_trace(C.c_void_p(0))


if __name__ == "__main__":
    main()
