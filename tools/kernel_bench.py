#!/usr/bin/env python3
"""Times the stand-alone (unfused) C-ABI kernels on an MI355X and prints achieved GB/s against algorithmic bytes.
    python tools/kernel_bench.py            (on the GPU box)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import normflows_amd as nfa  # noqa: E402
from normflows_amd import ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, reps=20):
    """Seconds per call.  The calls are recorded into ONE hipGraph and replayed, so the number is the kernels' own time: a Python
    loop over ctypes launches cannot issue faster than ~13 us per call (round 3's table carried that floor in every row below
    ~25 us).  Falls back to the eager loop when an op cannot be captured."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    graph = None
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                for _ in range(reps):
                    fn()
        torch.cuda.current_stream().wait_stream(side)
        g.replay()
        torch.cuda.synchronize()
        graph = g
    except Exception:   # noqa: BLE001
        torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    if graph is not None:
        graph.replay()
    else:
        for _ in range(reps):
            fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3


ROWS = []


def report(name, sec, nbytes):
    print("%-52s %9.1f us  %8.1f GB/s  (%.1f%% of 8 TB/s)" % (name, sec * 1e6, nbytes / sec / 1e9, nbytes / sec / 8e12 * 100))
    ROWS.append({"kernel": name, "us": sec * 1e6, "algorithmic_bytes": nbytes, "GBps": nbytes / sec / 1e9,
                 "frac_of_8TBps": nbytes / sec / 8e12})


def main():
    torch.manual_seed(0)
    B, D, K = 65536, 64, 8
    layer = nfa.flows.CoupledRationalQuadraticSpline(D, 2, 128, num_bins=K).to(dev)
    p = layer.prqct
    x = torch.randn(B, D, device=dev)
    cond = (0.3 * torch.randn(B, 32 * 23, device=dev)).contiguous()
    uw, uh, ud = p._uncond()
    kw = p._kernel_kwargs()
    ld = torch.zeros(B, device=dev)
    for mode, nm in ((0, "density"), (1, "sample/identity"), (2, "sample/transform")):
        y = torch.empty_like(x)
        t = timeit(lambda: ops.rqs_coupling(x, cond, uw, uh, ud, p.identity_features, p.transform_features, K, mode, y=y,
                                            logdet=ld, acc=1, **kw))
        nb = B * (2 * D * 4 + 8 + (32 * 23 * 4 if mode != 1 else 0))
        report("nf_rqs_coupling f32 " + nm, t, nb)
    xd, cd = x.double(), cond.double()
    ldd = torch.zeros(B, device=dev, dtype=torch.float64)
    t = timeit(lambda: ops.rqs_coupling(xd, cd, uw.double(), uh.double(), ud.double(), p.identity_features,
                                        p.transform_features, K, 0, logdet=ldd, acc=1, **kw), reps=5)
    report("nf_rqs_coupling f64 density", t, B * (2 * D * 8 + 16 + 32 * 23 * 8))
    lu = nfa.flows.LULinearPermute(D, identity_init=False).to(dev)
    for inverse, nm in ((True, "density"), (False, "sample")):
        with torch.no_grad():       # inference path: nf_lu_compose (cached) + nf_rows_matvec_affine
            t = timeit(lambda: lu._run(x, inverse, ld, +1))
        report("LULinearPermute f32 %s (dense MFMA mat-vec)" % nm, t, B * (2 * D * 4 + 8))
        lu.use_dense = False
        with torch.no_grad():
            t = timeit(lambda: lu._run(x, inverse, ld, +1))
        lu.use_dense = True
        report("nf_lu_linear_permute f32 %s (LDS tile)" % nm, t, B * (2 * D * 4 + 8))
    w = torch.randn(B, D, K, device=dev)
    h = torch.randn(B, D, K, device=dev)
    d = torch.randn(B, D, K - 1, device=dev)
    t = timeit(lambda: ops.rqs_spline(x, w, h, d, tails="linear", tail_bound=3.0), reps=5)
    report("nf_rqs_spline f32 (elementwise)", t, B * D * (23 * 4 + 12))
    q = nfa.distributions.DiagGaussian(D, trainable=False).to(dev)
    t = timeit(lambda: q._log_prob_acc(x, ld, +1))
    report("nf_diag_gaussian_log_prob f32", t, B * (D * 4 + 8))
    zr = torch.randn(B, D, device=dev)
    bm = (torch.arange(D, device=dev) % 2).float()
    sr, tr = 0.1 * torch.randn(B, D, device=dev), torch.randn(B, D, device=dev)
    t = timeit(lambda: ops.masked_affine(zr, bm, sr, tr, 0, logdet=ld, acc=1))
    report("nf_masked_affine (%d,%d)" % (B, D), t, B * D * 16 + B * 8)
    t = timeit(lambda: ops.masked_affine_bwd(zr, bm, sr, tr, tr, ld, 0))
    report("nf_masked_affine_bwd (%d,%d)" % (B, D), t, B * D * 28 + B * 4)
    Wm = torch.randn(D, D, device=dev)
    t = timeit(lambda: ops.rows_matvec(zr, Wm))
    report("nf_rows_matvec (%d,%d)" % (B, D), t, B * D * 8)
    # image-side kernels at the Glow shapes of BASELINE configs[3] (B = 256)
    for C, HW in ((12, 16), (24, 8), (48, 4)):
        z = torch.randn(256, C, HW, HW, device=dev)
        an = nfa.flows.ActNorm((C, 1, 1)).to(dev)
        an.inverse(z)
        ldz = torch.zeros(256, device=dev)
        t = timeit(lambda: an._run(z, True, ldz, +1))
        report("nf_actnorm (256,%d,%d,%d)" % (C, HW, HW), t, z.numel() * 8)
        conv = nfa.flows.Invertible1x1Conv(C, True).to(dev)
        t = timeit(lambda: conv._run(z, True, ldz, +1))
        report("nf_inv1x1 assemble+conv (256,%d,%d,%d)" % (C, HW, HW), t, z.numel() * 8)
        param = torch.randn(256, 2 * (C // 2), HW, HW, device=dev)
        t = timeit(lambda: ops.affine_coupling(z, param, (C + 1) // 2, False, "sigmoid", 1, logdet=ldz, acc=1))
        report("nf_affine_coupling (256,%d,%d,%d)" % (C, HW, HW), t, z.numel() * 8 + param.numel() * 4)
    zi = torch.randn(256, 3, 32, 32, device=dev)
    t = timeit(lambda: ops.squeeze(zi, 1))
    report("nf_squeeze (256,3,32,32)", t, zi.numel() * 8)
    # the same layer kernels at sizes that can saturate HBM (transform kernels as GB/s / 8 TB/s)
    Bi = 8192
    for C, HW in ((12, 16), (48, 4)):
        z = torch.randn(Bi, C, HW, HW, device=dev)
        gyz = torch.randn_like(z)
        ldz = torch.zeros(Bi, device=dev)
        an = nfa.flows.ActNorm((C, 1, 1)).to(dev)
        an.inverse(z)
        t = timeit(lambda: an._run(z, True, ldz, +1))
        report("nf_actnorm (%d,%d,%d,%d)" % (Bi, C, HW, HW), t, z.numel() * 8)
        t = timeit(lambda: ops.actnorm_bwd(z, an.s.detach().view(-1), an.t.detach().view(-1), gyz, ldz, 1))
        report("nf_actnorm_bwd (%d,%d,%d,%d)" % (Bi, C, HW, HW), t, z.numel() * 12)
        conv = nfa.flows.Invertible1x1Conv(C, True).to(dev)
        conv._run(z, True, ldz, +1)
        t = timeit(lambda: conv._run(z, True, ldz, +1))
        report("nf_inv1x1 conv (%d,%d,%d,%d)" % (Bi, C, HW, HW), t, z.numel() * 8)
        t = timeit(lambda: ops.inv1x1_wgrad(z, gyz, ldz))
        report("nf_inv1x1_wgrad (%d,%d,%d,%d)" % (Bi, C, HW, HW), t, z.numel() * 8)
        param = torch.randn(Bi, 2 * (C // 2), HW, HW, device=dev)
        t = timeit(lambda: ops.affine_coupling(z, param, (C + 1) // 2, False, "sigmoid", 1, logdet=ldz, acc=1))
        report("nf_affine_coupling (%d,%d,%d,%d)" % (Bi, C, HW, HW), t, z.numel() * 8 + param.numel() * 4)
        t = timeit(lambda: ops.affine_coupling_bwd(z, param, gyz, ldz, (C + 1) // 2, False, "sigmoid", 1))
        report("nf_affine_coupling_bwd (%d,%d,%d,%d)" % (Bi, C, HW, HW), t, z.numel() * 12 + param.numel() * 8)
    zi = torch.randn(Bi, 3, 32, 32, device=dev)
    t = timeit(lambda: ops.squeeze(zi, 1))
    report("nf_squeeze (%d,3,32,32)" % Bi, t, zi.numel() * 8)
    xa, W1, W2 = torch.randn(B, 128, device=dev), torch.randn(128, 128, device=dev), torch.randn(128, 128, device=dev)
    b1 = torch.randn(128, device=dev)
    t = timeit(lambda: ops.rows_block(xa, W1, b1, W2, b1))
    report("nf_rows_block forward (residual block, 128) [%.0f TF]" % (4.0 * B * 128 * 128 / t / 1e12), t, B * 128 * 12)
    tt = torch.randn(B, 128, device=dev)
    t = timeit(lambda: ops.rows_block(xa, W2, None, W1, None, trans=True, mask1=tt, mask2=xa, relu=False))
    report("nf_rows_block backward (residual block, 128) [%.0f TF]" % (4.0 * B * 128 * 128 / t / 1e12), t, B * 128 * 20)

    def lib_block():
        h = torch.nn.functional.linear(torch.relu(xa), W1, b1)
        return xa + torch.nn.functional.linear(torch.relu(h), W2, b1)
    t = timeit(lib_block)
    report("  library residual block forward (2 GEMM + 3 elementwise)", t, B * 128 * 12)
    # ---- round-2 training kernels of the benchmark layer ----
    cond24 = torch.randn(B, 32, 24, device=dev)
    cond24[:, :, 23] = 0.0
    gy, gld = torch.randn(B, D, device=dev), torch.randn(B, device=dev)
    iidx, tidx = p.identity_features, p.transform_features
    t = timeit(lambda: ops.rqs_coupling_bwd_p24(x, gy, gld, cond24, uw, uh, ud, iidx, tidx, tail_bound=3.0,
                                                wh_div=float(128 ** 0.5)))
    report("nf_rqs_coupling_bwd_p24 (software-pipelined spline backward)", t, B * (3 * D * 4 + 4 + 2 * 32 * 24 * 4))
    h2 = torch.randn(B, 128, device=dev)
    g2 = torch.randn(B, 768, device=dev)
    t = timeit(lambda: ops.linear_wgrad(g2, h2, skip_every=24))
    report("nf_linear_wgrad_skip 768x128 (LDS-DMA ring tile) [%.0f TF]" % (2.0 * B * 768 * 128 / t / 1e12), t, B * (768 + 128) * 4)
    ga, gb_ = torch.randn(B, 128, device=dev), torch.randn(B, 128, device=dev)
    t = timeit(lambda: ops.linear_wgrad_pair(ga, xa, gb_, tt, relu_x=True))
    report("nf_linear_wgrad_pair 2 x (128x128) [%.0f TF]" % (4.0 * B * 128 * 128 / t / 1e12), t, 4 * B * 128 * 4)
    t = timeit(lambda: ops.linear_wgrad(ga, xa, relu_x=True))
    report("  nf_linear_wgrad_act 128x128 alone [%.0f TF]" % (2.0 * B * 128 * 128 / t / 1e12), t, 2 * B * 128 * 4)
    Wm1, Wm2 = torch.randn(D, D, device=dev), torch.randn(D, D, device=dev)
    t = timeit(lambda: ops.rows_matvec2(x, Wm1, Wm2))
    report("nf_rows_matvec2 (two chained 64x64 products, u kept)", t, B * D * 4 * 3)
    blob = ops.rqs_fused_train_blob(2, dev)
    wf, bf = torch.randn(736, 128, device=dev) * 0.05, torch.randn(736, device=dev) * 0.1
    ops.rqs_fused_pack_final(blob, wf, bf, uw, uh, ud, 2)
    t = timeit(lambda: ops.rqs_fused_train_fwd(x, h2, blob, 0, 2))
    report("nf_rqs_fused_train_fwd (final Linear + coupling) [%.0f TF]" % (2.0 * B * 736 * 128 / t / 1e12), t,
           B * (2 * D * 4 + 128 * 4 + 32 * 24 * 4 + 4))
    # whole-layer training forward and the one-pass backward kernels (round 2, second half)
    w0, b0 = torch.randn(128, 32, device=dev) * 0.1, torch.randn(128, device=dev) * 0.1
    wb = [torch.randn(128, 128, device=dev) * 0.05 for _ in range(4)]
    bb = [torch.randn(128, device=dev) * 0.1 for _ in range(4)]
    wfull_t, wpad = torch.zeros(64, 128, device=dev), torch.zeros(32, 24, 128, device=dev)
    t = timeit(lambda: ops.rqs_fused_pack_all(blob, w0, b0, wb, bb, wf, bf, uw, uh, ud, wfull=wfull_t, wpad=wpad, identity_idx=iidx))
    report("nf_rqs_fused_pack_all (whole layer's blob + backward weight images) [latency]", t, blob.numel() * 4)
    t = timeit(lambda: ops.rqs_fused_train_full_fwd(x, blob, 0, 2))
    report("nf_rqs_fused_train_full_fwd (whole layer forward + saved tensors) [%.0f TF]" % (2.0 * B * 128 * (32 + 4 * 128 + 736) / t / 1e12),
           t, B * (2 * D * 4 + 5 * 128 * 4 + 32 * 24 * 4 + 4))
    t = timeit(lambda: ops.resblock_bwd(ga, tt, xa, W1, W2))
    report("nf_resblock_bwd (block backward: 2 dgrad + 2 wgrad, one pass) [%.0f TF]" % (8.0 * B * 128 * 128 / t / 1e12), t, 4 * B * 128 * 4)
    gxx = torch.randn(B, D, device=dev)
    t = timeit(lambda: ops.resblock_bwd(ga, tt, xa, W1, W2, x=x, wfull=wfull_t, gx=gxx))
    report("nf_resblock_bwd + initial layer [%.0f TF]" % ((8.0 * 128 + 4.0 * 64) * B * 128 / t / 1e12), t, (3 * 128 + 3 * D) * B * 4)
    uu = torch.randn(B, D, device=dev)
    t = timeit(lambda: ops.lu_bwd(gy, uu, x, Wm1, Wm2))
    report("nf_lu_bwd (LU backward, D = 64, one pass) [%.0f TF]" % (8.0 * B * D * D / t / 1e12), t, 4 * B * D * 4)
    z2 = torch.randn(1024, 2, device=dev)
    b = torch.tensor([1.0, 0.0], device=dev)
    s2, t2 = torch.randn(1024, 2, device=dev), torch.randn(1024, 2, device=dev)
    ld2 = torch.zeros(1024, device=dev)
    t = timeit(lambda: ops.masked_affine(z2, b, s2, t2, 0, logdet=ld2, acc=1))
    report("nf_masked_affine (1024,2) [latency]", t, 1024 * 2 * 16)


if __name__ == "__main__":
    with torch.no_grad():      # the inference kernels (layers route to their autograd Functions when gradients are on)
        main()
    if len(sys.argv) > 2 and sys.argv[1] == "--json":
        import json
        json.dump({"what": "stand-alone layer kernels: HIP-event time per launch, algorithmic bytes / time against the 8 TB/s HBM "
                           "peak (tools/kernel_bench.py, 1x MI355X)", "rows": ROWS}, open(sys.argv[2], "w"), indent=1)
