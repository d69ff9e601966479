#!/usr/bin/env python3
"""Single-pass direction of a MAF layer at BASELINE configs[4] (d = 128, hidden 512, 2 blocks, B = 65 536): the one-launch kernel
(nf_made_forward_affine, csrc/made_fwd.hip) against the layer-by-layer path (library GEMMs on weight * mask + nf_maf_affine),
HIP-event timing on the launch stream, fp32 MFMA roofline on the dense and on the masked (executed) work.

    python tools/made_bench.py [D H NB B]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import normflows_amd as nfa  # noqa: E402


def events_ms(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    D, H, NB, B = [int(v) for v in sys.argv[1:5]] if len(sys.argv) >= 5 else (128, 512, 2, 65536)
    from made_fwd_emulator import work_fraction
    torch.manual_seed(0)
    layer = nfa.flows.MaskedAffineAutoregressive(D, H, num_blocks=NB).to("cuda:0")
    x = torch.randn(B, D, device="cuda:0")
    made = layer.autoregressive_net
    with torch.no_grad():
        packed = made.packed_forward(x.device)
        frac = work_fraction(packed[1].cpu().numpy())
        fused = events_ms(lambda: layer.forward(x))
        z1, ld1 = layer.forward(x)
        nfa.config.set_made_fused(False)
        lib = events_ms(lambda: layer.forward(x))
        z0, ld0 = layer.forward(x)
        nfa.config.set_made_fused(True)
    dense = 2.0 * (D * H + 2 * NB * H * H + H * 2 * D) * B
    Hp, Dp = int(packed[2]), (D + 7) // 8 * 8
    executed = 2.0 * (Dp * Hp + 2 * NB * Hp * Hp + Hp * ((2 * D + 31) // 32 * 32)) * frac * B
    res = {"shape": {"D": D, "H": H, "num_blocks": NB, "B": B}, "one_launch_ms": fused, "layerwise_ms": lib,
           "dense_flop": dense, "masked_fraction_executed": frac,
           "roofline_dense": {"bound": "mfma", "achieved": dense / fused / 1e9, "peak": 157.3, "unit": "TFLOP/s",
                              "frac": dense / fused / 1e9 / 157.3},
           "roofline_executed": {"bound": "mfma", "achieved": executed / fused / 1e9, "peak": 157.3, "unit": "TFLOP/s",
                                 "frac": executed / fused / 1e9 / 157.3},
           "max_abs_diff_z": float((z1 - z0).abs().max()), "max_abs_diff_ld": float((ld1 - ld0).abs().max())}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
