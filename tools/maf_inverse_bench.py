#!/usr/bin/env python3
"""BASELINE configs[4], the direction it names: the inverse (sampling) pass of 10 x MaskedAffineAutoregressive(128, hidden 512, 2
blocks) at B = 65 536 -- nothing else in the process (no training legs, no library-GEMM comparison), so a rocprofv3 run of this
command holds only the inverse kernels.  Prints one JSON line per variant: format 1 (round 5: nf_maf_inverse_h_tri, regular tiles on
the triangular sequential part + 8-deep activation ring) and, with --ablate, format 0 (nf_maf_inverse_h, rounds 3-4).

    python tools/maf_inverse_bench.py [--ablate] [--layers 10] [--batch 65536] [--reps 5]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import normflows_amd as nfa  # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ablate", action="store_true")
    ap.add_argument("--layers", type=int, default=10)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    flows = [nfa.flows.MaskedAffineAutoregressive(128, 512, num_blocks=2) for _ in range(a.layers)]
    m = nfa.NormalizingFlow(nfa.distributions.DiagGaussian(128, trainable=False), flows).to(dev)
    x = torch.randn(a.batch, 128, device=dev)
    nnz = 0
    for f in flows:
        net = f.autoregressive_net
        for lin in [net.initial_layer, net.final_layer] + [l for b in net.blocks for l in b.linear_layers]:
            nnz += int(lin.mask.sum().item())
    flop = 2.0 * nnz * a.batch
    out = {}
    with torch.no_grad():
        for tri in ((True, False) if a.ablate else (True,)):
            nfa.config.set_maf_tri(tri)
            z, ld = m.inverse_and_log_det(x)
            dt = timed(lambda: m.inverse_and_log_det(x), a.reps)
            zf, _ = m.forward_and_log_det(z)
            res = {"variant": "format 1 (nf_maf_inverse_h_tri)" if tri else "format 0 (nf_maf_inverse_h)", "layers": a.layers,
                   "batch": a.batch, "inverse_pass_ms": dt * 1e3, "ms_per_layer": dt * 1e3 / a.layers,
                   "samples_per_s": a.batch / dt, "round_trip_max_abs_err": float((zf - x).abs().max()),
                   "roofline": {"bound": "mfma", "achieved": flop / dt / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                                "frac": flop / dt / 157.3e12, "flop_per_pass_masked": flop}}
            out[tri] = (z, ld)
            print(json.dumps(res), flush=True)
        nfa.config.set_maf_tri(True)
        if a.ablate:
            print(json.dumps({"max_abs_diff_x_between_formats": float((out[True][0] - out[False][0]).abs().max()),
                              "max_abs_diff_logdet_between_formats": float((out[True][1] - out[False][1]).abs().max())}))


if __name__ == "__main__":
    main()
