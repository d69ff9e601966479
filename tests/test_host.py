"""CPU-side tests (run here without a GPU): the C-ABI library loads and exports every declared symbol, argument
validation of the C ABI (no compute), host logic (masks, state_dict compatibility with the reference, pair
scheduling), the fail-loud behaviour without a HIP device, and the data-parallel path under gloo (world_size 2)."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, golden_state, load_golden

REF = "/root/reference"


@pytest.fixture(scope="module")
def nfa():
    import __graft_entry__
    import normflows_amd
    if not os.path.exists(normflows_amd.native_library_path()):
        __graft_entry__.build()
    return normflows_amd


def test_library_loads_and_exports_every_declared_symbol(nfa):
    lib = nfa._lib.lib()
    declared = nfa._lib.exported_symbols_declared()
    assert len(declared) >= 18
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert b"gfx950" in lib.nf_version()
    assert lib.nf_max_bins() >= 16
    assert lib.nf_strerror(-22) == b"invalid argument"


def test_c_abi_argument_validation_without_gpu(nfa):
    """Bad arguments are rejected before any launch (errno-style codes)."""
    lib = nfa._lib.lib()
    i32, i64, f64, vp = ctypes.c_int, ctypes.c_int64, ctypes.c_double, ctypes.c_void_p
    null = vp(0)
    one = vp(16)  # never dereferenced on the host
    # utils/splines.py:121-124: min_bin_width * K > 1 -> ValueError (EINVAL)
    rc = lib.nf_rqs_spline(one, one, i64(8), one, i64(8), one, i64(7), one, one, i64(4), i32(8), i32(1), f64(3.0), f64(0),
                           f64(1), f64(0), f64(1), f64(0.2), f64(1e-3), f64(1e-3), f64(1.0), i32(0), i32(0), null)
    assert rc == -22
    # K beyond the compiled range -> ERANGE
    rc = lib.nf_rqs_spline(one, one, i64(8), one, i64(8), one, i64(7), one, one, i64(4), i32(1000), i32(1), f64(3.0),
                           f64(0), f64(1), f64(0), f64(1), f64(1e-5), f64(1e-5), f64(1e-3), f64(1.0), i32(0), i32(0), null)
    assert rc == -34
    # null buffer -> EFAULT
    rc = lib.nf_lu_linear_permute(null, one, one, one, one, one, one, one, i64(4), i32(4), f64(1e-3), i32(0), i32(0),
                                  i32(0), null)
    assert rc == -14
    # unknown dtype -> ENOTSUP ; bad direction -> EINVAL ; empty batch -> OK without touching pointers
    assert lib.nf_masked_affine(one, one, null, null, one, one, i64(4), i64(2), i32(0), i32(0), i32(7), null) == -95
    assert lib.nf_masked_affine(one, one, null, null, one, one, i64(4), i64(2), i32(3), i32(0), i32(0), null) == -22
    assert lib.nf_masked_affine(null, null, null, null, null, null, i64(0), i64(2), i32(0), i32(0), i32(0), null) == 0
    # fused kernel only takes the benchmark shape
    lib.nf_rqs_fused_pack_size.restype = ctypes.c_int64
    assert lib.nf_rqs_fused_pack_size(i32(32), i32(32), i32(128), i32(2), i32(8)) > 600 * 1024
    assert lib.nf_rqs_fused_pack_size(i32(8), i32(8), i32(32), i32(2), i32(8)) == -95


def test_every_int64_function_of_the_header_is_bound_as_int64(nfa):
    """ctypes assumes `int` for a foreign function's return value: every function include/nf_mi355x.h declares `int64_t` (scratch and
    pack sizes) must be bound as c_int64 by _lib.lib() -- derived from the header, not from a hand-kept list (round 6: the solve's
    scratch size arrived truncated above 2^31 floats, ~720 000 rows of config 5's layer)."""
    lib = nfa._lib.lib()
    names = nfa._lib.int64_functions_declared()
    assert "nf_maf_solve_t_scratch_floats" in names and "nf_maf_inverse_h_scratch_floats" in names and len(names) >= 15
    for fn in names:
        assert getattr(lib, fn).restype is ctypes.c_int64, fn
    i32, i64 = ctypes.c_int, ctypes.c_int64
    assert lib.nf_maf_solve_t_scratch_floats(i64(999936), i32(128), i32(512), i32(2)) == 999936 * (5 * 512 + 256 + 5 * 32)     # > 2^31
    assert lib.nf_maf_inverse_h_scratch_floats(i64(999936), i32(128), i32(512), i32(2)) == 999936 * (5 * 512 + 128 + 5 * 32)


def test_c_abi_argument_validation_newer_entry_points(nfa):
    """Same discipline for the entry points added after the first bench: nothing is launched on bad arguments."""
    lib = nfa._lib.lib()
    i32, i64, f64, vp = ctypes.c_int, ctypes.c_int64, ctypes.c_double, ctypes.c_void_p
    null, one = vp(0), vp(16)
    lib.nf_maf_inverse_scratch_floats.restype = ctypes.c_int64
    lib.nf_linear_wgrad_scratch_floats.restype = ctypes.c_int64
    # MAF inverse: padded hidden size must be a positive multiple of 32; scratch = 64-row tiles x (5 Hp + Dp)
    assert lib.nf_maf_inverse(one, one, one, one, one, one, i64(8), i32(128), i32(500), i32(0), null) == -22
    assert lib.nf_maf_inverse(one, one, one, null, one, one, i64(8), i32(128), i32(512), i32(0), null) == -14
    assert lib.nf_maf_inverse(null, null, null, null, null, null, i64(0), i32(128), i32(512), i32(0), null) == 0
    lib.nf_maf_inverse_h_scratch_floats.restype = ctypes.c_int64
    assert lib.nf_maf_inverse_h(one, one, one, one, one, one, i64(8), i32(128), i32(500), i32(2), i32(0), null) == -22
    assert lib.nf_maf_inverse_h(one, one, one, one, one, one, i64(8), i32(128), i32(512), i32(4), i32(0), null) == -95   # 1..3 residual blocks
    assert lib.nf_maf_inverse_h(one, one, one, null, one, one, i64(8), i32(128), i32(512), i32(2), i32(0), null) == -14
    assert lib.nf_maf_inverse_h(null, null, null, null, null, null, i64(0), i32(128), i32(512), i32(1), i32(0), null) == 0
    assert lib.nf_maf_inverse_h_scratch_floats(i64(64), i32(128), i32(512), i32(2)) == 64 * (5 * 512 + 128 + 5 * 32)
    # format-1 entry point (round 5): the HOST copy of the table is mandatory and must describe the call
    th = np.zeros(8 + 24 * 16, dtype=np.int32)
    th[:8] = [128, 128, 512, 512, 16, 2, 2, 1]
    hp_ = vp(th.ctypes.data)
    assert lib.nf_maf_inverse_h_tri(one, one, one, one, one, null, one, i64(8), i32(128), i32(512), i32(2), i32(0), null) == -14
    assert lib.nf_maf_inverse_h_tri(one, one, one, one, one, hp_, one, i64(8), i32(128), i32(512), i32(1), i32(0), null) == -22   # num_blocks != table's
    assert lib.nf_maf_inverse_h_tri(one, one, one, one, one, hp_, one, i64(8), i32(64), i32(512), i32(2), i32(0), null) == -22     # D != table's
    assert lib.nf_maf_inverse_h_tri(one, one, one, null, one, hp_, one, i64(8), i32(128), i32(512), i32(2), i32(0), null) == -14   # blob
    assert lib.nf_maf_inverse_h_tri(null, null, null, null, null, hp_, null, i64(0), i32(128), i32(512), i32(2), i32(0), null) == 0
    th[7] = 0
    assert lib.nf_maf_inverse_h_tri(one, one, one, one, one, hp_, one, i64(8), i32(128), i32(512), i32(2), i32(0), null) == -22   # a format-0 table
    # the one-pass implicit backward (round 5): forward with ReLU masks + the transposed solve
    lib.nf_maf_solve_t_scratch_floats.restype = ctypes.c_int64
    assert lib.nf_maf_inverse_h_bits(one, one, one, one, one, one, null, i64(8), i32(128), i32(512), i32(2), i32(0), null) == -14   # bits
    assert lib.nf_maf_inverse_h_bits(one, one, one, one, one, one, one, i64(8), i32(128), i32(500), i32(2), i32(0), null) == -22
    assert lib.nf_maf_solve_t_scratch_floats(i64(64), i32(128), i32(512), i32(2)) == 64 * (5 * 512 + 256 + 5 * 32)
    assert lib.nf_maf_solve_t_scratch_floats(i64(64), i32(128), i32(512), i32(4)) == -22
    assert lib.nf_maf_solve_t(one, one, one, null, one, one, one, one, one, i64(8), i32(128), i32(500), i32(2), null) == -22
    assert lib.nf_maf_solve_t(one, one, one, null, one, one, one, one, one, i64(8), i32(128), i32(512), i32(4), null) == -95
    assert lib.nf_maf_solve_t(one, one, one, null, null, one, one, one, one, i64(8), i32(128), i32(512), i32(2), null) == -14      # bits
    assert lib.nf_maf_solve_t(null, null, null, null, null, null, null, null, null, i64(0), i32(128), i32(512), i32(2), null) == 0
    f64 = ctypes.c_double
    assert lib.nf_maf_scratch_rows(one, one, one, i64(8), i32(2), i32(500), i32(512), f64(-1.0), i32(1), null) == -22     # positions % 32
    assert lib.nf_maf_scratch_rows(one, one, one, i64(8), i32(4), i32(512), i32(512), f64(-1.0), i32(1), null) == -22     # blocks
    assert lib.nf_maf_scratch_rows(one, one, one, i64(8), i32(2), i32(512), i32(510), f64(-1.0), i32(1), null) == -22     # ldo % 4
    assert lib.nf_maf_scratch_rows(one, null, one, i64(8), i32(2), i32(512), i32(512), f64(-1.0), i32(1), null) == -14
    assert lib.nf_maf_scratch_rows(null, null, null, i64(0), i32(2), i32(512), i32(512), f64(-1.0), i32(1), null) == 0
    # round 6: one layer of a scratch; the weight gradients from the scratches in place; the training forward that stores MADE's output
    assert lib.nf_maf_scratch_layer(one, one, one, i64(8), i32(2), i32(512), i32(512), i32(5), null) == -22                 # layer >= 2 blocks + 1
    assert lib.nf_maf_scratch_layer(one, one, one, i64(8), i32(2), i32(500), i32(512), i32(4), null) == -22
    assert lib.nf_maf_scratch_layer(one, null, one, i64(8), i32(2), i32(512), i32(512), i32(4), null) == -14
    assert lib.nf_maf_scratch_layer(null, null, null, i64(0), i32(2), i32(512), i32(512), i32(4), null) == 0
    wg = lambda B, ntiles, nl, pos, a=one: lib.nf_made_wgrad_pos(a, one, one, one, one, one, one, one, one, i32(ntiles), i64(B), i32(nl), i32(pos), null)
    assert wg(64, 0, 5, 512) == -22 and wg(64, 4, 0, 512) == -22 and wg(-64, 4, 5, 512) == -22
    assert wg(100, 4, 5, 512) == -95             # rows: a multiple of 64 (the scratch holds no zero rows beyond the batch)
    assert wg(64, 4, 5, 160) == -95              # positions: a multiple of 128
    assert wg(64, 4, 5, 512, null) == -14 and wg(0, 4, 5, 512, null) == 0
    th[7] = 2        # (a transposed pack's table)
    so = lambda a, hp=512, D_=128, th_=hp_: lib.nf_maf_solve_t_tri(a, one, one, null, one, one, one, one, th_, one, i64(8), i32(D_), i32(hp), i32(2), null)
    assert so(one, th_=null) == -14 and so(one, 500) == -22 and so(one, D_=64) == -22 and so(null) == -14
    assert lib.nf_maf_solve_t_tri(null, null, null, null, null, null, null, null, hp_, null, i64(0), i32(128), i32(512), i32(2), null) == 0
    th[7] = 0
    tr = lambda bits, prm, hp=512, th_=null: lib.nf_maf_inverse_h_train(one, one, one, one, one, th_, one, bits, prm, i64(8), i32(128), i32(hp),
                                                                       i32(2), i32(0), null)
    assert tr(null, one) == -14 and tr(one, null) == -14 and tr(one, one, 500) == -22
    th[7] = 1
    assert lib.nf_maf_inverse_h_train(one, one, one, one, one, hp_, one, one, one, i64(8), i32(64), i32(512), i32(2), i32(0), null) == -22   # D != table's
    th[7] = 0
    # debug-mode spline check
    assert lib.nf_rqs_spline_check(one, one, i64(8), i32(0), f64(1.0), f64(0.0), f64(1.0), f64(0.0), f64(1.0), i32(1), i32(0), null, null) == -14
    assert lib.nf_rqs_spline_check(one, one, i64(8), i32(7), f64(1.0), f64(0.0), f64(1.0), f64(0.0), f64(1.0), i32(1), i32(0), one, null) == -22
    assert lib.nf_rqs_spline_check(null, null, i64(0), i32(0), f64(1.0), f64(0.0), f64(1.0), f64(0.0), f64(1.0), i32(1), i32(0), null, null) == 0
    assert lib.nf_maf_inverse_scratch_floats(i64(65), i32(128), i32(512)) == 2 * 64 * (5 * 512 + 128 + 5 * 32)   # + pair stash

    def arnsf(K, tails, hp=512, B=8, blob=one):
        return lib.nf_arnsf_inverse(one, one, one, blob, one, one, i64(B), i32(64), i32(hp), i32(K), i32(tails), f64(3.0),
                                    f64(1e-3), f64(1e-3), f64(1e-3), i32(0), null)
    assert arnsf(12, 1) == -95          # K = 11: the 3K-1 = 32 rows of linear tails fit one block, 3K / 3K+1 do not
    assert arnsf(11, 0) == -95 and arnsf(11, 2) == -95
    assert arnsf(8, 3) == -22 and arnsf(0, 1) == -22 and arnsf(8, 1, hp=500) == -22
    assert arnsf(8, 1, blob=null) == -14
    assert arnsf(8, 1, B=0) == 0
    # GlowBlock kernels: layout choice, shape limits, argument checks
    lib.nf_glow_convnet_pack_size.restype = ctypes.c_int64
    assert lib.nf_glow_convnet_layout(i64(256), i32(16), i32(16)) == 0      # 256 workgroups of 256 pixels
    assert lib.nf_glow_convnet_layout(i64(256), i32(8), i32(8)) == 1        # 64 such workgroups: the 64-pixel kernel
    assert lib.nf_glow_convnet_layout(i64(256), i32(4), i32(4)) == 2        # 4096 pixels: 16-pixel row-split workgroups
    assert lib.nf_glow_convnet_layout(i64(4), i32(16), i32(16)) == 0 and lib.nf_glow_convnet_layout(i64(4), i32(5), i32(5)) == -95
    # one size for every layout: header + biases + max(32 stages of 16 KB, 8 waves x (2 x 16 + 2 x 16 + 16) / 4 slots of 4 KB)
    assert lib.nf_glow_convnet_pack_size(i32(6), i32(12), i32(256)) == 4 * (64 + 576 + max((8 + 16 + 8) * 4096, 8 * 20 * 1024))
    assert lib.nf_glow_convnet_pack_size(i32(6), i32(12), i32(128)) == -95
    gc = lambda B, H, W, layout, slope=0.0, x=one: lib.nf_glow_convnet(x, i64(6 * H * W), one, one, i64(B), i32(6), i32(H),
                                                                     i32(W), i32(12), i32(256), f64(slope), i32(layout), null)
    assert gc(4, 16, 16, 1) == -95 and gc(4, 5, 5, 0) == -95                 # whole images must tile the workgroup
    assert gc(4, 16, 16, 3) == -22 and gc(4, 16, 16, 2) == -95 and gc(4, 16, 16, 0, slope=1.5) == -22
    assert gc(4, 16, 16, 0, x=null) == -14 and gc(0, 16, 16, 0) == 0
    # nf_glow_level(in0, in1, cin0, in_sq, out0, out1, cout0, out_sq, logdet, table, nblocks, B, C, H, W, hidden, slope, smap,
    #               direction, acc, layout, stream)
    def gl(C=12, smap=1, direction=1, nblocks=1, B=0, in_sq=0, out_sq=0, cin0=12, cout0=12, in1=null, out1=null, table=one):
        return lib.nf_glow_level(one, in1, i32(cin0), i32(in_sq), one, out1, i32(cout0), i32(out_sq), one, table, i32(nblocks),
                                 i64(B), i32(C), i32(8), i32(8), i32(256), f64(0.0), i32(smap), i32(direction), i32(0), i32(1), null)
    assert gl() == 0 and gl(C=1) == -22 and gl(smap=3) == -22 and gl(direction=2) == -22
    assert gl(nblocks=0) == -34 and gl(nblocks=65) == -34                  # 1..64 blocks per launch
    assert gl(C=10, cin0=10, cout0=10, in_sq=1) == -22                      # the squeeze views need C % 4 == 0
    assert gl(B=4, cin0=6) == -14 and gl(B=4, cout0=6) == -14               # the second tensor of a merge / split is missing
    assert gl(B=4, table=null) == -14 and gl(B=4, cin0=13) == -22
    # per-feature backward: type codes required with NF_TAILS_FEATURE, refused otherwise
    def bwd_ft(tails, tt, ti=null, uw=null):
        return lib.nf_rqs_coupling_bwd_ft(one, one, one, one, uw, uw, uw, one, i32(2), one, i32(2), i64(4), i32(4), i32(4),
                                          i32(tails), f64(3.0), f64(1e-3), f64(1e-3), f64(1e-3), f64(1.0), i32(0), one, one,
                                          uw, uw, uw, i32(0), tt, null, ti, null, null)
    assert bwd_ft(3, null) == -14 and bwd_ft(3, one, null, one) == -14
    assert bwd_ft(1, one) == -22 and bwd_ft(4, one) == -22
    assert lib.nf_rqs_coupling_bwd(one, one, one, one, null, null, null, one, i32(2), one, i32(2), i64(4), i32(4), i32(4),
                                   i32(3), f64(3.0), f64(1e-3), f64(1e-3), f64(1e-3), f64(1.0), i32(0), one, one, null,
                                   null, null, i32(0), null) == -22
    # weight gradient: N <= 128 columns per launch, accumulate is 0 / 1
    assert lib.nf_linear_wgrad(one, one, one, one, one, i64(64), i32(8), i32(200), i32(0), null) == -95
    assert lib.nf_linear_wgrad(one, one, one, one, one, i64(64), i32(8), i32(8), i32(2), null) == -22
    assert lib.nf_linear_wgrad(one, one, null, one, one, i64(64), i32(8), i32(8), i32(0), null) == -14
    assert lib.nf_linear_wgrad_scratch_floats(i64(65536), i32(128), i32(128)) > 0
    # logit: alpha in [0, 0.5)
    assert lib.nf_logit(one, one, one, i64(4), i64(12), f64(0.5), i32(0), i32(0), i32(0), null) == -22
    assert lib.nf_logit(one, one, one, i64(4), i64(12), f64(0.05), i32(2), i32(0), i32(0), null) == -22
    assert lib.nf_logit(null, null, null, i64(0), i64(12), f64(0.05), i32(0), i32(0), i32(0), null) == 0
    # row-wise Gaussian: without an index the table needs one row per sample
    assert lib.nf_diag_gaussian_log_prob_rows(one, one, one, null, i64(3), f64(0), one, i64(8), i64(4), i32(0), i32(0),
                                              null) == -22
    assert lib.nf_diag_gaussian_log_prob_rows(one, null, one, one, i64(3), f64(0), one, i64(8), i64(4), i32(0), i32(0),
                                              null) == -14
    # per-feature tails: type arrays only with NF_TAILS_FEATURE (3), and required with it
    args = lambda tails, tt: (one, one, one, one, null, null, null, one, i32(2), one, i32(2), i64(4), i32(4), i32(4),
                              i32(tails), f64(3.0), f64(1e-3), f64(1e-3), f64(1e-3), f64(1.0), i32(0), i32(0), i32(0), tt,
                              null, null, null, null)
    assert lib.nf_rqs_coupling_ft(*args(1, one)) == -22
    assert lib.nf_rqs_coupling_ft(*args(3, null)) == -14
    assert lib.nf_bias_leaky_relu(one, null, i64(2), i32(3), i64(4), f64(0.0), i32(0), null) == -14
    assert lib.nf_bias_leaky_relu(one, one, i64(2), i32(0), i64(4), f64(0.0), i32(0), null) == -22
    # RealNVP chain: at most 16 coordinates and 64-wide conditioners
    assert lib.nf_realnvp_chain(one, one, one, one, i64(8), i32(17), i32(8), i32(0), i32(0), null) == -95
    assert lib.nf_realnvp_chain(one, one, one, one, i64(8), i32(2), i32(65), i32(0), i32(0), null) == -95
    assert lib.nf_realnvp_chain(one, one, one, null, i64(8), i32(2), i32(4), i32(0), i32(0), null) == -14
    assert lib.nf_realnvp_chain(one, one, one, one, i64(8), i32(2), i32(4), i32(2), i32(0), null) == -22


def test_c_abi_argument_validation_training_entry_points(nfa):
    """The one-pass backward kernels and the whole-layer training forward refuse what they do not implement before any launch."""
    lib = nfa._lib.lib()
    i32, i64, f64, vp = ctypes.c_int, ctypes.c_int64, ctypes.c_double, ctypes.c_void_p
    null, one = vp(0), vp(16)
    lib.nf_resblock_bwd_scratch_floats.restype = ctypes.c_int64
    lib.nf_lu_bwd_scratch_floats.restype = ctypes.c_int64
    per = 128 * 128 + 128
    assert lib.nf_resblock_bwd_scratch_floats(i64(65536), i32(0)) == 2 * 256 * per
    assert lib.nf_resblock_bwd_scratch_floats(i64(65536), i32(1)) == 2 * 256 * per + 256 * (128 * 64 + 128)
    assert lib.nf_resblock_bwd_scratch_floats(i64(192), i32(0)) == 2 * 3 * per           # one workgroup per 64-row tile
    assert lib.nf_resblock_bwd_scratch_floats(i64(100), i32(0)) == -22
    assert lib.nf_lu_bwd_scratch_floats(i64(65536)) == 2 * 512 * (64 * 64 + 64) + 64 and lib.nf_lu_bwd_scratch_floats(i64(65)) == -22

    def rb(B=128, H=128, D=64, gh=one, gh_in=one, x=null, wfull=null, gx=null, dw0=null, db0=null, cmap=null, nc=0):
        return lib.nf_resblock_bwd(gh, one, one, one, one, gh_in, one, one, one, one, x, wfull, gx, dw0, db0, cmap, i32(nc), one,
                                   i64(B), i32(H), i32(D), null)
    assert rb(B=100) == -95 and rb(H=64) == -95 and rb(B=0) == -95        # 64-row tiles, hidden 128
    assert rb(gh=null) == -14 and rb(gh_in=null) == -14                    # without the initial layer gh_in is an output
    assert rb(x=one) == -14 and rb(x=one, wfull=one, gx=one, dw0=one, db0=one, D=32) == -95
    assert rb(x=one, wfull=one, gx=one, dw0=one, db0=one, cmap=one, nc=65) == -22
    assert rb(gh=vp(20)) == -22                                            # 16-byte aligned rows

    def lu(B=128, D=64, gy=one, gx=one):
        return lib.nf_lu_bwd(gy, one, one, one, one, gx, one, one, one, one, i64(B), i32(D), null)
    assert lu(B=100) == -95 and lu(D=32) == -95 and lu(gy=null) == -14 and lu(gx=vp(24)) == -22
    lf = lambda B=128, D=64, x=one, acc=1, ld=one, ldc=one: lib.nf_lu_fwd(x, one, one, one, one, one, ld, ldc, f64(1.0), i32(acc),
                                                                          i64(B), i32(D), null)
    assert lf(B=100) == -95 and lf(D=32) == -95 and lf(acc=5) == -22 and lf(x=null) == -14 and lf(ldc=null) == -14
    assert lf(x=vp(20)) == -22

    def full(B=128, D=64, H=128, nb=2, K=8, x=one, acc=1, parity=0):
        return lib.nf_rqs_fused_train_full_fwd(x, one, one, one, one, one, i32(parity), i64(B), i32(D), i32(H), i32(nb), i32(K),
                                               f64(3.0), f64(1e-3), f64(1e-3), f64(1e-3), i32(acc), null)
    assert full(D=32) == -95 and full(H=64) == -95 and full(K=4) == -95 and full(nb=17) == -95
    assert full(parity=2) == -22 and full(acc=7) == -22 and full(x=null) == -14 and full(B=0) == 0

    def pack(wfull=null, wpad=null, iidx=null, K=8, w0=one):
        return lib.nf_rqs_fused_pack_all(one, w0, one, null, null, one, one, one, one, one, i32(128), i32(0), i32(K), f64(3.0),
                                         f64(1e-3), f64(1e-3), f64(1e-3), wfull, wpad, iidx, null)
    assert pack(K=16) == -95 and pack(w0=null) == -14 and pack(wfull=one) == -14 and pack(wfull=one, wpad=one) == -14
    pm = lambda n=4, table=one, nb=2, K=8: lib.nf_rqs_fused_pack_all_multi(table, i32(n), i32(128), i32(nb), i32(K), f64(3.0),
                                                                          f64(1e-3), f64(1e-3), f64(1e-3), null)
    assert pm(K=4) == -95 and pm(nb=17) == -95 and pm(n=-1) == -22 and pm(table=null) == -14 and pm(n=0, table=null) == 0
    lm = lambda n=4, table=one, D=64: lib.nf_lu_factors_multi(table, i32(n), f64(1e-3), i32(D), null)
    assert lm(D=1) == -22 and lm(n=70000) == -22 and lm(table=null) == -14 and lm(n=0, table=null) == 0
    # round 6: the one-call layer / pair backward, the pair forward, the composed LU backward and the LU stage's pack
    for fn in ("nf_coupling_train_bwd_scratch_floats", "nf_pair_train_bwd_scratch_floats", "nf_lu_bwd_composed_scratch_floats"):
        getattr(lib, fn).restype = ctypes.c_int64
    per768 = 768 * 128 + 768
    chunks = lib.nf_linear_wgrad_chunks(i64(65536), i32(768), i32(128))
    assert chunks == 82 and lib.nf_resblock_bwd_grid(i64(65536)) == 256 and lib.nf_resblock_bwd_grid(i64(100)) == -22
    want = 65536 * 768 + 2 * 65536 * 128 + lib.nf_final_bwd_partials(i64(65536)) * 768 + chunks * per768 \
        + lib.nf_resblock_bwd_scratch_floats(i64(65536), i32(1)) + lib.nf_resblock_bwd_scratch_floats(i64(65536), i32(0))
    assert lib.nf_coupling_train_bwd_scratch_floats(i64(65536), i32(2)) == want
    assert lib.nf_pair_train_bwd_scratch_floats(i64(65536), i32(2)) == want + 65536 * 64 + 512 * (64 * 64 + 64) + 64 * 64 + 4
    assert lib.nf_coupling_train_bwd_scratch_floats(i64(100), i32(2)) == -95 and lib.nf_coupling_train_bwd_scratch_floats(i64(128), i32(0)) == -95
    assert lib.nf_coupling_train_bwd_scratch_floats(i64(128), i32(7)) == -95
    assert lib.nf_lu_bwd_composed_scratch_floats(i64(65536)) == 512 * (64 * 64 + 64) and lib.nf_lu_bwd_composed_scratch_floats(i64(65)) == -22
    two = (vp * 2)(one, one)
    four = (vp * 4)(one, one, one, one)

    def cb(B=128, D=64, H=128, nb=1, K=8, x=one, parity=0, cmap=null, nc=0, wb=two, gb=four, scratch=one):
        return lib.nf_coupling_train_bwd(x, one, one, one, one, one, one, one, wb, one, one, one, cmap, i32(nc), one, one, one, one, one,
                                         one, one, one, gb, scratch, i32(parity), i64(B), i32(D), i32(H), i32(nb), i32(K), f64(3.0),
                                         f64(1e-3), f64(1e-3), f64(1e-3), null)
    assert cb(D=32) == -95 and cb(H=64) == -95 and cb(K=4) == -95 and cb(nb=0) == -95 and cb(nb=7) == -95 and cb(B=100) == -95
    assert cb(parity=2) == -22 and cb(cmap=one, nc=65) == -22 and cb(x=null) == -14 and cb(wb=null) == -14 and cb(scratch=null) == -14
    assert cb(wb=(vp * 2)(one, null)) == -14 and cb(gb=(vp * 4)(one, one, null, one)) == -14
    assert cb(scratch=vp(20)) == -22                                       # 16-byte aligned scratch (the reduction's loads)

    def pb(B=128, x_in=one, Wd=one, gl=one, nb=1):
        return lib.nf_pair_train_bwd(x_in, one, one, one, one, one, one, one, one, two, one, one, one, null, i32(0), Wd, one, one, one,
                                     one, f64(1e-3), one, gl, one, one, one, one, one, one, one, one, one, one, four, one, i32(0),
                                     i64(B), i32(64), i32(128), i32(nb), i32(8), f64(3.0), f64(1e-3), f64(1e-3), f64(1e-3), null)
    assert pb(x_in=null) == -14 and pb(Wd=null) == -14 and pb(gl=null) == -14 and pb(B=100) == -95 and pb(nb=0) == -95

    def pf(B=128, D=64, H=128, nb=2, K=8, x=one, xlu=one, acc=1, parity=0):
        return lib.nf_rqs_fused_train_pair_fwd(x, xlu, one, one, one, one, one, i32(parity), i64(B), i32(D), i32(H), i32(nb), i32(K),
                                               f64(3.0), f64(1e-3), f64(1e-3), f64(1e-3), i32(acc), null)
    assert pf(D=32) == -95 and pf(H=64) == -95 and pf(K=16) == -95 and pf(parity=3) == -22 and pf(acc=9) == -22
    assert pf(x=null) == -14 and pf(xlu=null) == -14 and pf(B=0) == 0
    lc = lambda B=128, D=64, g=one, gx=one, dW=one: lib.nf_lu_bwd_composed(g, one, one, gx, dW, one, one, i64(B), i32(D), null)
    assert lc(B=100) == -95 and lc(D=32) == -95 and lc(g=null) == -14 and lc(dW=null) == -14 and lc(gx=vp(24)) == -22
    lp = lambda D=64, dW=one, perm=one: lib.nf_lu_param_grads_composed(dW, one, one, perm, null, i64(0), one, f64(1e-3), one, one, one,
                                                                       i32(D), null)
    assert lp(D=32) == -95 and lp(dW=null) == -14 and lp(perm=null) == -14
    lt = lambda n=4, table=one, nb=2, D=64: lib.nf_lu_pack_train_multi(table, i32(n), i32(nb), i32(D), f64(1e-3), null)
    assert lt(D=32) == -95 and lt(nb=17) == -95 and lt(n=-1) == -22 and lt(table=null) == -14 and lt(n=0, table=null) == 0
    wp = lambda B=128, M=768, N=128, dY=one, wb=1: lib.nf_linear_wgrad_partials(dY, one, one, i64(B), i32(M), i32(N), i32(0), i32(wb), null)
    assert wp(N=200) == -95 and wp(dY=null) == -14 and wp(wb=2) == -22 and wp(B=0) == -22


def test_masks_bit_exact(nfa):
    m = nfa.utils.create_alternating_binary_mask(7, even=False)
    assert m.dtype == torch.uint8 and m.tolist() == [0, 1, 0, 1, 0, 1, 0]
    assert nfa.utils.create_alternating_binary_mask(4, even=True).tolist() == [1, 0, 1, 0]
    assert nfa.utils.create_mid_split_binary_mask(5).tolist() == [1, 1, 1, 0, 0]
    a = nfa.utils.create_random_binary_mask(9, seed=3)
    b = nfa.utils.create_random_binary_mask(9, seed=3)
    assert torch.equal(a, b) and int(a.sum()) == 5


def test_layers_refuse_cpu_tensors(nfa):
    for layer, z in ((nfa.flows.LULinearPermute(4), torch.randn(3, 4)),
                     (nfa.flows.CoupledRationalQuadraticSpline(4, 1, 8), torch.randn(3, 4)),
                     (nfa.flows.ActNorm(4), torch.randn(3, 4)),
                     (nfa.flows.Invertible1x1Conv(4, True), torch.randn(2, 4, 3, 3)),
                     (nfa.flows.MaskedAffineFlow(torch.tensor([1.0, 0.0, 1.0, 0.0])), torch.randn(3, 4))):
        with pytest.raises(RuntimeError, match="no CPU path"):
            layer.inverse(z)
    with pytest.raises(RuntimeError, match="no CPU path"):
        nfa.distributions.DiagGaussian(4).log_prob(torch.randn(3, 4))


def test_missing_library_fails_loudly(nfa, tmp_path, monkeypatch):
    monkeypatch.setattr(nfa._lib, "_lib", None)
    monkeypatch.setattr(nfa._lib, "LIBPATH", str(tmp_path / "nope.so"))
    with pytest.raises(nfa._lib.NativeLibraryError, match="no CPU or eager fallback"):
        nfa._lib.lib()


def test_golden_state_dicts_load_strict(nfa):
    """Reference checkpoints (state_dict = the on-disk format, core.py:199-213) load with strict=True."""
    g = load_golden("model_c2mini")
    from bench import build_c2_model
    m = build_c2_model(num_layers=4, dim=16, hidden=32, seed=0, sigma=0.05)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in golden_state(g).items()}, strict=True)
    g = load_golden("glowblock_C4_channel")
    blk = nfa.flows.GlowBlock(4, 8, split_mode="channel", use_lu=True, init_zeros=False)
    blk.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in golden_state(g, "sd0__").items()}, strict=True)


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout only exists in the build container")
def test_seeded_construction_is_bit_identical_to_reference(nfa):
    """Same constructor calls under the same seed give bit-identical parameters, permutations and masks
    (the bench model is reproduced on the GPU box without shipping 21.8 MB of weights)."""
    sys.path.insert(0, REF)
    import normflows as nf
    from bench import build_c2_model
    a = build_c2_model(num_layers=3, lib=nf).state_dict()
    b = build_c2_model(num_layers=3, lib=nfa).state_dict()
    assert list(a.keys()) == list(b.keys())
    for k in a:
        assert a[k].dtype == b[k].dtype and torch.equal(a[k], b[k]), k
    torch.manual_seed(3)
    ga = nf.flows.GlowBlock(12, 16, use_lu=True)
    torch.manual_seed(3)
    gb = nfa.flows.GlowBlock(12, 16, use_lu=True)
    sa, sb = ga.state_dict(), gb.state_dict()
    assert list(sa.keys()) == list(sb.keys()) and all(torch.equal(sa[k], sb[k]) for k in sa)
    # both directions of strict loading
    gb.load_state_dict(sa, strict=True)
    ga.load_state_dict(sb, strict=True)


def test_pair_scheduler_orders(nfa, monkeypatch):
    """run_chain visits [CoupledRQS, LULinearPermute] pairs in the reference's order (core.py:177-179, 193-195)."""
    from normflows_amd import core
    calls = []

    class Fake(nfa.flows.Flow):
        def __init__(self, name):
            super().__init__()
            self.name = name

        def _run(self, z, inverse, ld, acc, **kw):
            calls.append((self.name, inverse))
            return z

    flows = [Fake("a"), Fake("b"), Fake("c")]
    ld = torch.zeros(2)
    core.run_chain(flows, torch.zeros(2, 3), True, ld, +1)
    assert calls == [("c", True), ("b", True), ("a", True)]
    calls.clear()
    core.run_chain(flows, torch.zeros(2, 3), False, ld, -1)
    assert calls == [("a", False), ("b", False), ("c", False)]


def test_dp_shard_bounds(nfa):
    from normflows_amd import dp
    for n, w in ((10, 3), (524288, 8), (5, 8), (0, 2)):
        spans = [dp.shard_bounds(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


_DP_WORKER = r"""
import os, sys
sys.path.insert(0, os.environ["NF_ROOT"]); sys.path.insert(0, os.path.join(os.environ["NF_ROOT"], "oracle"))
import numpy as np, torch, torch.distributed as dist
import normflows_amd as nfa, nf_oracle
from bench import build_c2_model, state_to_numpy
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
m = build_c2_model(num_layers=2, dim=8, hidden=16, seed=0, sigma=0.05)
ora = nf_oracle.OracleNSF(state_to_numpy(m), num_layers=4)
g = torch.Generator().manual_seed(99)
x = torch.randn(37, 8, generator=g)                       # same global batch on every rank
log_prob = lambda t: torch.from_numpy(ora.log_prob(t.numpy()))   # CPU stand-in for the HIP log_prob (tests only)
local = nfa.dp.shard_rows(x)
nll = nfa.dp.sharded_forward_kld(log_prob, local)
full = -log_prob(x).double().mean()
assert abs(float(nll) - float(full)) < 1e-6, (float(nll), float(full))
lo, hi = nfa.dp.shard_bounds(37, world, rank)
assert local.shape[0] == hi - lo
# uneven shards whose global size changes while one rank's local size does not (37 -> 38 rows over 2 ranks: 19/18 then
# 19/19): the count travels with the sum in the SAME collective, so the ranks cannot desynchronise or reuse a stale count
x2 = torch.randn(38, 8, generator=g)
nll2 = nfa.dp.sharded_forward_kld(log_prob, nfa.dp.shard_rows(x2))
full2 = -log_prob(x2).double().mean()
assert abs(float(nll2) - float(full2)) < 1e-6, (float(nll2), float(full2))
# gradient averaging: rank-dependent gradients -> identical averaged gradients on every rank, bucketed
lin = torch.nn.Linear(5, 3)
for p in lin.parameters():
    p.grad = torch.full_like(p, float(rank + 1))
nb = nfa.dp.allreduce_gradients(lin.parameters(), bucket_bytes=32)   # tiny buckets: forces several collectives
assert nb >= 2
for p in lin.parameters():
    assert torch.allclose(p.grad, torch.full_like(p, (1 + world) / 2.0))
# the same through ONE persistent flat gradient buffer (gradients are views of it; autograd accumulates in place; the
# collective runs on the buffer itself -- no cat, no copy back), against the bucketed result of a real backward
torch.manual_seed(5)
net = torch.nn.Sequential(torch.nn.Linear(6, 4), torch.nn.Tanh(), torch.nn.Linear(4, 2))
fg = nfa.dp.FlatGradients(net.parameters())
xg = torch.randn(9, 6, generator=torch.Generator().manual_seed(100 + rank))      # rank-dependent rows
for step in range(2):
    fg.zero()
    net(xg).pow(2).sum().backward()
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in fg.views)           # autograd accumulated in place
    local = [p.grad.clone() for p in net.parameters()]
    assert nfa.dp.allreduce_gradients(fg, bucket_bytes=40) >= 2
    ref = [g_.clone() for g_ in local]
    for g_ in ref:
        dist.all_reduce(g_)
        g_.div_(world)
    assert all(torch.allclose(p.grad, r_, atol=1e-6) for p, r_ in zip(net.parameters(), ref))
# ... and overlapped with backward: buckets in reverse parameter order, each all-reduce started from an autograd hook as soon as
# the bucket's last gradient exists; finish() waits, scales, scatters -- same averages
net2 = torch.nn.Sequential(torch.nn.Linear(6, 4), torch.nn.Tanh(), torch.nn.Linear(4, 2))
net2.load_state_dict(net.state_dict())
avg = nfa.dp.OverlappedGradientAverager(net2.parameters(), bucket_bytes=40)
assert len(avg.buckets) >= 2 and avg.buckets[0][0] is list(net2.parameters())[-1]
for step in range(2):
    net2.zero_grad(set_to_none=True)
    net2(xg).pow(2).sum().backward()
    assert len(avg._pending) == len(avg.buckets)                                  # every bucket was started DURING backward
    assert avg.finish() == len(avg.buckets)
    assert all(torch.allclose(p.grad, r_, atol=1e-6) for p, r_ in zip(net2.parameters(), ref))
# contract (ADVICE r04): a second backward before finish() raises instead of averaging stale buckets ...
net2.zero_grad(set_to_none=True)
net2(xg).pow(2).sum().backward()
try:
    net2(xg).pow(2).sum().backward()
    raise SystemExit("a second backward() before finish() must raise")
except RuntimeError as e:
    assert "finish()" in str(e), e
avg.finish()
# ... gradient accumulation goes through no_sync(): two micro-batches, the first without collectives
net2.zero_grad(set_to_none=True)
with avg.no_sync():
    net2(xg[:4]).pow(2).sum().backward()
    assert not avg._pending and avg._next == 0
net2(xg[4:]).pow(2).sum().backward()
assert avg.finish() == len(avg.buckets)
assert all(torch.allclose(p.grad, r_, atol=1e-5) for p, r_ in zip(net2.parameters(), ref)), "accumulated == full batch (sum loss)"
avg.remove()
# ... and a parameter that is unused on ONE rank only: same sequence of equally sized collectives on both ranks (no hang); the
# rank without a gradient gets nothing written, the other the average with zeros
heads = torch.nn.ModuleDict({"a": torch.nn.Linear(4, 3), "b": torch.nn.Linear(4, 3)})
for p in heads.parameters():
    torch.nn.init.constant_(p, 0.5)
avg3 = nfa.dp.OverlappedGradientAverager(heads.parameters(), bucket_bytes=16)
xh = torch.ones(2, 4)
loss = heads["a"](xh).sum() * (rank + 1)
if rank == 0:
    loss = loss + heads["b"](xh).sum()
loss.backward()
assert avg3.finish() == len(avg3.buckets)
assert torch.allclose(heads["a"].bias.grad, torch.full((3,), 2.0 * (1 + world) / 2.0))
if rank == 0:
    assert torch.allclose(heads["b"].bias.grad, torch.full((3,), 2.0 / world))
else:
    assert heads["b"].bias.grad is None
avg3.remove()
net.zero_grad(set_to_none=True)                                                   # a dropped view is re-attached
fg.zero()
assert all(p.grad is v for p, v in fg.views)
# ActNorm data-dependent init under DP: global per-channel mean / unbiased std from the ranks' local moments
xa = torch.randn(23, 5, 3, generator=g) * 2.0 + 0.7            # (rows, channels, pixels), same on every rank
lo, hi = nfa.dp.shard_bounds(23, world, rank)
loc = xa[lo:hi].transpose(0, 1).reshape(5, -1)                 # channel-major local elements
gm, gs = nfa.dp.combine_moments(loc.mean(1), loc.std(1), loc.shape[1])
ref = xa.transpose(0, 1).reshape(5, -1)
assert torch.allclose(gm, ref.mean(1), atol=1e-6) and torch.allclose(gs, ref.std(1), atol=1e-6), (gm, ref.mean(1))
if rank == 0:
    print("DP_OK", float(nll))
dist.destroy_process_group()
"""


_DP8_WORKER = r"""
import os, sys
sys.path.insert(0, os.environ["NF_ROOT"])
import torch, torch.distributed as dist
import normflows_amd as nfa
from bench import build_c2_model
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
assert world == 8
# BASELINE configs[2]'s row count + 3: uneven shards (three ranks hold one row more); the count travels with the sum
N = 524288 + 3
lo, hi = nfa.dp.shard_bounds(N, world, rank)
assert hi - lo == 65536 + (1 if rank < 3 else 0)
g = torch.Generator().manual_seed(7)
x = torch.randn(N, 4, generator=g)                               # the same global batch on every rank
log_prob = lambda t: -0.5 * (t.double() ** 2).sum(1) - 3.0       # stand-in for the HIP log_prob: the test is the sharding
calls = []
real = dist.all_reduce
def counted(t, *a, **k):
    calls.append(t.numel() * t.element_size())
    return real(t, *a, **k)
dist.all_reduce = counted
nll = nfa.dp.sharded_forward_kld(log_prob, nfa.dp.shard_rows(x))
dist.all_reduce = real
assert calls == [16], calls                                       # ONE 16-byte collective per evaluated batch
full = -log_prob(x).mean()
assert abs(float(nll) - float(full)) < 1e-9 * abs(float(full)), (float(nll), float(full))
# the REAL configs[1] parameter list (5 443 584 floats = 21.8 MB) through the overlapped averager with 8 MB buckets
m = build_c2_model()
params = list(m.parameters())
assert sum(p.numel() for p in params) == 5443584
avg = nfa.dp.OverlappedGradientAverager(params, bucket_bytes=8 << 20)
assert 3 <= len(avg.buckets) <= 4 and all(sum(q.numel() for q in b) * 4 <= (8 << 20) for b in avg.buckets)
loss = sum((p * float(rank + 1)).sum() for p in params)
loss.backward()
assert len(avg._pending) == len(avg.buckets)                      # every bucket's all-reduce started during backward
assert [i for i, _, _ in avg._pending] == list(range(len(avg.buckets)))   # ... in index order
assert avg.finish() == len(avg.buckets)
want = (1 + world) / 2.0
assert all(bool((p.grad == want).all()) for p in params)
avg.remove()
# round 6: the SAME buckets reduced in place on slices of one flat gradient buffer (dp.FlatParameters: parameters are views of one
# flat tensor, gradients are written / copied into one flat buffer) -- no torch.cat in front of the collective, no copy back; rank
# dependent NON-constant gradients, equal to the gather / scatter averager above to rounding (a ring all-reduce sums an element's
# eight contributions in an order that depends on its POSITION in the message, and a bucket's parameters sit in reverse order in
# the gathered message: the two paths may differ in the last bit, never more)
ref_avg = nfa.dp.OverlappedGradientAverager(params, bucket_bytes=8 << 20)
gen = torch.Generator().manual_seed(1000 + rank)
noise = [torch.randn(p.shape, generator=gen) for p in params]
for p in params:
    p.grad = None
sum((p * n_).sum() for p, n_ in zip(params, noise)).backward()
ref_avg.finish()
ref = [p.grad.clone() for p in params]
ref_avg.remove()
before = [p.detach().clone() for p in params]
flat = nfa.dp.FlatParameters(m)
assert flat.param.numel() == 5443584 and all(torch.equal(p, b_) for p, b_ in zip(params, before))      # values unchanged ...
assert all(p.data_ptr() == flat.param.data_ptr() + 4 * lo_ for p, (lo_, _) in zip(params, flat.offsets))   # ... now views of one tensor
avg2 = nfa.dp.OverlappedGradientAverager(params, bucket_bytes=8 << 20, flat=flat)
assert [sum(q.numel() for q in b) for b in avg2.buckets] == [hi_ - lo_ for _, lo_, hi_ in avg2._ranges]
cat_calls = []
real_cat = torch.cat
torch.cat = lambda *a, **k: (cat_calls.append(1), real_cat(*a, **k))[1]
flat.zero_grad()
sum((p * n_).sum() for p, n_ in zip(params, noise)).backward()
assert len(avg2._pending) == len(avg2.buckets) and avg2.finish() == len(avg2.buckets)
torch.cat = real_cat
assert not cat_calls, "the flat path must not gather buckets"
assert flat.sync() == 0 or True
off = 0
for p, r_ in zip(params, ref):
    assert torch.allclose(p.grad, r_, rtol=2e-6, atol=1e-6), "in-place flat averaging must equal the gather / scatter path"
    assert torch.equal(flat.grad[off:off + p.numel()].view_as(p), p.grad)
    off += p.numel()
avg2.remove()
flat.release()
if rank == 0:
    print("DP8_OK", float(nll), len(avg.buckets))
dist.destroy_process_group()
"""


def test_dp_world_size_8_uneven_shards_and_real_parameter_list(nfa, tmp_path):
    """Readiness for the 8-GPU run nobody can launch from here: world size 8 over gloo -- 524 288 + 3 rows in
    uneven shards give the unsharded NLL with ONE 16-byte all-reduce; the benchmark model's real parameter list (5.4 M floats)
    goes through OverlappedGradientAverager's 8 MB buckets, every collective started during backward, in index order."""
    script = tmp_path / "dp8_worker.py"
    script.write_text(_DP8_WORKER)
    env = dict(os.environ, NF_ROOT=ROOT, OMP_NUM_THREADS="1", MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8",
                          "--master-addr", "127.0.0.1", "--master-port", "29677", str(script)], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "DP8_OK" in out.stdout


def test_dp_nll_matches_unsharded_under_gloo(nfa, tmp_path):
    """world_size 2, gloo: the row-sharded NLL (one all-reduce of [sum log_q, n]) equals the unsharded NLL; bucketed
    gradient averaging; ActNorm's global initialisation statistics from per-rank moments."""
    script = tmp_path / "dp_worker.py"
    script.write_text(_DP_WORKER)
    env = dict(os.environ, NF_ROOT=ROOT, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29671", str(script)], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "DP_OK" in out.stdout


@pytest.mark.parametrize("D,H,NB", [(128, 512, 2), (17, 40, 2), (3, 2, 2), (40, 39, 2), (6, 150, 2), (17, 40, 1), (33, 70, 3), (128, 512, 1),
                                    (32, 64, 2), (96, 256, 2), (64, 252, 2), (10, 36, 1), (12, 40, 3), (64, 256, 2), (9, 34, 1), (8, 30, 3)])
def test_maf_pack_schedule_matches_d_pass(D, H, NB):
    """flows/maf_pack.py + the kernel's tile/step schedule (tests/maf_emulator.py restates it in numpy) reproduce the
    fixed point of the reference's D-pass inverse (autoregressive.py:29-38) computed with plain torch in fp64."""
    import normflows_amd as nfa
    from normflows_amd import nets
    from normflows_amd.flows import maf_pack
    from maf_emulator import emulate_inverse
    torch.manual_seed(D + H)
    made = nets.MADE(features=D, hidden_features=H, num_blocks=NB, output_multiplier=2)
    with torch.no_grad():
        for p in made.parameters():
            p.add_((0.1 if D < 100 else 0.01) * torch.randn_like(p))
    assert maf_pack.pack_made(made) is None or NB == 2          # round 2's kernel / the rows layout: two blocks only
    blob, table = maf_pack.pack_made(made, blocks=(1, 2, 3))
    assert table[6] == NB
    assert table[0] == D and table[3] % 32 == 0 and table[3] >= H
    blob1, table1 = maf_pack.pack_made(made, blocks=(1, 2, 3), tri=True)
    fcols0 = maf_pack.solve_t_gradient_columns(made, forward=True)
    fcols1 = maf_pack.solve_t_gradient_columns(made, tri=True, forward=True)
    srcf = maf_pack.final_layer_columns(made)
    z = torch.randn(16, D)
    m64 = made.double()
    with torch.no_grad():
        out = torch.zeros(16, D, dtype=torch.float64)
        for _ in range(D):
            prm = m64(out).view(16, D, 2)
            scale = torch.sigmoid(prm[..., 0] + 2.0) + 1e-3
            out = (z.double() - prm[..., 1]) / scale
        ldref = -torch.log(scale).sum(1)
    x, ld = emulate_inverse(blob, table, z.numpy())
    np.testing.assert_allclose(x, out.numpy(), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(ld, ldref.numpy(), rtol=1e-9, atol=1e-9)
    # format 1 (round 5): regular tiles (<= 8 degrees of <= 4 units) with triangular, statically ordered sequential weights
    assert table1[7] == 1 and np.array_equal(table1[:7], table[:7])
    T = int(table1[4])
    reg = [int(table1[8 + 24 * t + 20]) for t in range(T)]
    plan = maf_pack.plan_tiles(D, made.initial_layer.degrees.numpy())[1]
    assert reg == [1 if maf_pack.is_regular(st) else (2 if maf_pack.extras_prefix(st) else 0) for (_, _, st) in plan]
    if (D, H) == (128, 512):
        assert reg == [2] + [1] * 15                     # config 5: degrees 1-4 own five units -> tile 0 is "regular with extras"
    x1, ld1, S1 = emulate_inverse(blob1, table1, z.numpy(), return_scratch=True)
    np.testing.assert_allclose(x1, out.numpy(), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(ld1, ldref.numpy(), rtol=1e-9, atol=1e-9)
    # The activations the pass publishes are the inputs of MADE's linears at the solution: through solve_t_gradient_columns(forward=True)
    # (nf_maf_scratch_rows on the device) they are the weight-gradient launch's `save` in the training kernels' column order, and
    # h_NB @ flat[final_layer_columns] + bias is MADE's output there -- the implicit backward runs nothing of MADE forward again.
    _, _, S0 = emulate_inverse(blob, table, z.numpy(), return_scratch=True)
    lins = [m64.initial_layer] + [l for b in m64.blocks for l in b.linear_layers]
    with torch.no_grad():
        h = torch.nn.functional.linear(out, lins[0].weight * lins[0].mask, lins[0].bias)
        acts = []
        for b in range(NB):
            acts.append(torch.relu(h))
            t_ = torch.nn.functional.linear(torch.relu(h), lins[1 + 2 * b].weight * lins[1 + 2 * b].mask, lins[1 + 2 * b].bias)
            acts.append(torch.relu(t_))
            h = h + torch.nn.functional.linear(torch.relu(t_), lins[2 + 2 * b].weight * lins[2 + 2 * b].mask, lins[2 + 2 * b].bias)
        acts.append(h)
        prm_ref = m64(out)
    unit_of_col = np.argsort(made.initial_layer.degrees.numpy(), kind="stable")
    for S_, cols in ((S0, fcols0), (S1, fcols1)):
        assert cols.shape == (256 if H <= 256 else 512,) and (cols[H:] == -1).all()
        for l, a in enumerate(acts):
            np.testing.assert_allclose(S_[l][:, cols[:H]], a.numpy()[:, unit_of_col], rtol=1e-9, atol=1e-9, err_msg="layer input %d" % l)
    flat = np.concatenate([np.zeros(1)] + [q.detach().numpy().reshape(-1) for l in lins + [m64.final_layer] for q in (l.weight, l.bias)])
    wf_t = flat[srcf]
    hl = np.zeros((16, srcf.shape[1]))
    hl[:, :H] = acts[-1].numpy()[:, unit_of_col]
    np.testing.assert_allclose(hl @ wf_t.T + m64.final_layer.bias.detach().numpy(), prm_ref.numpy(), rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("D,H,NB,tri", [(40, 100, 2, True), (40, 100, 2, False), (24, 120, 1, True), (128, 512, 2, True)])
def test_position_space_weight_gradient_tables(nfa, D, H, NB, tri):
    """flows/maf_pack.position_wgrad_tables (round 6, nf_made_wgrad_pos): problems, tiles and scatter maps over the one-pass kernels'
    scratch POSITIONS.  A numpy emulation of the launch -- per listed 128 x 128 tile the (negated) product of the two operands as the
    kernel addresses them, then the reduction's masked scatter -- on scratches whose holes hold garbage must reproduce, for arbitrary
    hidden gradients and activations, every masked weight gradient dW_l = (dY_l^T X_l) * mask and bias gradient of MADE
    (nets/made.py:73-81 under autograd) in the flat layout of the ordinary backward pack."""
    from normflows_amd import nets
    from normflows_amd.flows import made_pack, maf_pack
    torch.manual_seed(D + H)
    made = nets.MADE(D, H, num_blocks=NB, output_multiplier=2, use_residual_blocks=True, random_mask=False,
                     activation=torch.nn.functional.relu)
    pw = maf_pack.position_wgrad_tables(made, tri=tri)
    assert pw is not None and pw["positions"] % 128 == 0 and pw["NL"] == 2 * NB + 1
    bwd = made_pack.made_train_structure(made, 2)["bwd"]
    assert pw["nflat"] == bwd["nflat"]
    fslot, vslot, T = maf_pack._unit_positions(made, tri)
    P, NL, B = pw["positions"], pw["NL"], 48
    rng = np.random.default_rng(D)
    Gu = [rng.standard_normal((B, H)) for _ in range(NL)]              # gradient at the output of hidden layer l
    Xu = [np.abs(rng.standard_normal((B, H))) for _ in range(NL - 1)] + [rng.standard_normal((B, H))]     # the linears' inputs
    x, gp = rng.standard_normal((B, D)), rng.standard_normal((B, 2 * D))
    gscr = rng.standard_normal((NL, B, P)) * 1e3                      # garbage in the holes
    fscr = np.abs(rng.standard_normal((NL, B, P))) * 1e3
    for l in range(NL):
        gscr[NL - 1 - l][:, vslot] = -Gu[l]
        fscr[l][:, fslot] = Xu[l]
    Mp, Dx = bwd["Mp"], bwd["Dx"]
    gp_pad, x_pad = np.zeros((B, Mp)), np.zeros((B, Dx))
    gp_pad[:, :2 * D], x_pad[:, :D] = gp, x
    wt, sc, mask = pw["wtable"], pw["stable"], bwd["mask"]
    ntl, npr = int(wt[0]), int(wt[1])
    assert ntl == pw["ntiles"] and npr == 2 + 2 * NB
    grads = np.zeros(pw["nflat"])
    seen_bias = set()
    for t in range(ntl):
        pi, m0, n0, want_bias = (int(v) for v in wt[16 + 8 * npr + 8 * t:16 + 8 * npr + 8 * t + 4])
        pr, ps = wt[16 + 8 * pi:16 + 8 * pi + 8], sc[8 * pi:8 * pi + 8]
        flags = int(pr[7])
        dY = gscr[int(pr[1])][:, m0:m0 + 128] if flags & 1 else {0: gp_pad, 1: x_pad}[int(pr[0])][:, m0:m0 + 128]
        X = fscr[int(pr[4])][:, n0:n0 + 128] if flags & 2 else {0: gp_pad, 1: x_pad}[int(pr[3])][:, n0:n0 + 128]
        assert (int(pr[0]) == 2) == bool(flags & 1) and (int(pr[3]) == 3) == bool(flags & 2)
        if int(pr[6]):
            X = np.maximum(X, 0.0)
        sgn = -1.0 if flags & 4 else 1.0
        part, bias = sgn * dY.T @ X, sgn * dY.sum(0)
        rowmap, colmap = sc[int(ps[3]):int(ps[3]) + 4096], sc[int(ps[4]):int(ps[4]) + 4096]
        for r in range(128):
            row = int(rowmap[m0 + r])
            if row < 0:
                continue
            if want_bias:
                assert (pi, row) not in seen_bias
                seen_bias.add((pi, row))
                grads[int(ps[2]) + row] = bias[r]
            for c in range(128):
                col = int(colmap[n0 + c])
                if col >= 0:
                    dst = int(ps[0]) + row * int(ps[1]) + col
                    if mask[dst]:
                        grads[dst] = part[r, c]
    lins = [made.initial_layer] + [l for b in made.blocks for l in b.linear_layers] + [made.final_layer]
    dys = [Gu[0]] + [Gu[k] for b in range(NB) for k in (2 * b + 1, 2 * b + 2)] + [gp]
    xs = [x] + [Xu[k] for b in range(NB) for k in (2 * b, 2 * b + 1)] + [Xu[2 * NB]]
    for k, (lin, dy, xx, (woff, shape, boff, n)) in enumerate(zip(lins, dys, xs, bwd["offsets"])):
        m = lin.mask.numpy() != 0
        want = (dy.T @ xx) * m
        np.testing.assert_allclose(grads[woff:woff + shape[0] * shape[1]].reshape(shape), want, rtol=1e-10, atol=1e-10, err_msg="dW %d" % k)
        np.testing.assert_allclose(grads[boff:boff + n], dy.sum(0), rtol=1e-10, atol=1e-10, err_msg="db %d" % k)


def test_transposed_pack_marks_regular8_tiles(nfa):
    """pack_made_transposed(tri=True) marks (table entry [21]) the tiles whose FORWARD tile is a format-1 regular tile of exactly 8
    degrees x 4 units -- the ones nf_maf_solve_t_tri runs on the statically unrolled sequential part; their step masks are then the
    compile-time positions tf_step assumes: virtual step s = forward step 7 - s = registers 2 g, 2 g + 1 of both lane-halves."""
    from normflows_amd import nets
    from normflows_amd.flows import maf_pack
    for (D, H), want in (((128, 512), [1] * 15 + [0]), ((40, 100), None), ((64, 252), None)):
        made = nets.MADE(D, H, num_blocks=2, output_multiplier=2, use_residual_blocks=True, random_mask=False,
                         activation=torch.nn.functional.relu)
        _, t0 = maf_pack.pack_made_transposed(made, tri=False)
        _, t1 = maf_pack.pack_made_transposed(made, tri=True)
        T = int(t1[4])
        assert all(int(t0[8 + 24 * t + 21]) == 0 for t in range(T))
        flags = [int(t1[8 + 24 * t + 21]) for t in range(T)]
        plan = maf_pack.plan_tiles(D, made.initial_layer.degrees.numpy())[1]
        assert flags == [1 if (ns == 8 and all(c == 4 for c in st)) else 0 for (_, ns, st) in plan[::-1]]
        if want is not None:
            assert flags == want
        row = lambda r, h: (r & 3) + 8 * (r >> 2) + 4 * h
        for t in range(T):
            if flags[t]:
                for s_ in range(8):
                    g = 7 - s_
                    m = sum(1 << row(2 * g + (i & 1), i >> 1) for i in range(4))
                    assert int(np.uint32(t1[8 + 24 * t + 4 + s_])) == m, (t, s_)


@pytest.mark.parametrize("D,H,K,tails", [(64, 256, 8, "linear"), (9, 40, 4, None), (5, 12, 10, "circular"), (3, 2, 1, "linear")])
def test_arnsf_pack_schedule_matches_d_pass(D, H, K, tails):
    """Rows layout of flows/maf_pack.py (one final-layer block per feature, nf_arnsf_inverse): the emulated schedule
    reproduces the fixed point of the D-pass inverse of neural_spline/autoregressive.py:94-134 (oracle spline, fp64)."""
    import nf_oracle
    from normflows_amd import nets
    from normflows_amd.flows import maf_pack
    from maf_emulator import emulate_inverse
    mult = {"linear": 3 * K - 1, "circular": 3 * K, None: 3 * K + 1}[tails]
    torch.manual_seed(D * K + H)
    made = nets.MADE(features=D, hidden_features=H, num_blocks=2, output_multiplier=mult)
    with torch.no_grad():
        for p in made.parameters():
            p.add_(0.1 * torch.randn_like(p))
    blob, table = maf_pack.pack_made(made, mult=mult, rows=True)
    assert table[5] == mult and blob.size % 4 == 0
    tb = 2.5
    z = (torch.rand(16, D) if tails is None else 2.0 * torch.randn(16, D)).double().numpy()

    def element(prm, zf):
        prm = np.ascontiguousarray(prm)
        y, lad = nf_oracle.rqs_spline(np.ascontiguousarray(zf), prm[:, :K], prm[:, K:2 * K], prm[:, 2 * K:], inverse=True,
                                      tails=tails, tail_bound=tb)
        return y, lad
    m64 = made.double()
    out = np.zeros((16, D))
    with torch.no_grad():
        for _ in range(D):
            prm = m64(torch.from_numpy(out)).view(16, D, mult).numpy()
            cols = [element(prm[:, f], z[:, f]) for f in range(D)]
            out = np.stack([c[0] for c in cols], 1)
        ldref = np.stack([c[1] for c in cols], 1).sum(1)
    x, ld = emulate_inverse(blob, table, z, element=element)
    np.testing.assert_allclose(x, out, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(ld, ldref, rtol=1e-9, atol=1e-9)


def test_maf_pack_rejects_unsupported():
    from normflows_amd import nets
    from normflows_amd.flows import maf_pack
    assert maf_pack.pack_made(nets.MADE(features=12, hidden_features=4, num_blocks=2, output_multiplier=2)) is None
    assert maf_pack.pack_made(nets.MADE(features=8, hidden_features=16, num_blocks=3, output_multiplier=2)) is None
    assert maf_pack.pack_made(nets.MADE(features=8, hidden_features=16, num_blocks=2, output_multiplier=2,
                                        use_residual_blocks=False)) is None


@pytest.mark.parametrize("K", [4, 8, 16])
def test_fused_final_layer_row_order_is_a_bijection(nfa, K):
    """The packed final layer of the fused kernel for K bins: K groups x 3 row-blocks x 32 MFMA rows hold every row of the
    (32 (3 K - 1), hidden) weight exactly once plus 32 padding rows; a lane-half's 48 slots of a group are whole features
    (3 K slots each: 3 K - 1 parameters in order, then the pad); each lane-half sees the 16 transform features of its own
    columns."""
    lib = nfa._lib.lib()
    M, MP, FPL = 3 * K - 1, 3 * K, 16 // K
    seen, pads = set(), 0
    for g in range(K):
        for hh in range(2):
            slots = []
            for rb in range(3):
                for reg in range(16):
                    rho = 8 * (reg >> 2) + 4 * hh + (reg & 3)      # C register reg of lane-half hh is MFMA row rho
                    slots.append(lib.nf_rqs_fused_final_row(K, g, rb, rho))
            assert len(slots) == 48
            for f in range(FPL):
                rows = slots[f * MP:(f + 1) * MP]
                assert rows[-1] == -1 and all(r >= 0 for r in rows[:-1])
                tf = rows[0] // M
                assert rows[:-1] == [tf * M + p for p in range(M)]
                assert (tf % 8) // 4 == hh                        # features 8 Q + 4 hh + j live in lane-half hh
                seen.update(rows[:-1])
                pads += 1
    assert seen == set(range(32 * M)) and pads == 32
    assert lib.nf_rqs_fused_final_row(5, 0, 0, 0) == -95      # NF_ENOTSUP: no instantiation for 5 bins


def test_fused_kernels_have_no_register_spills(nfa):
    """AMDGPU metadata of the built objects (tools/kernel_resources.py): the register-resident fused NSF kernels -- exact
    fp32 and split-bf16, both directions, with and without the fused LU -- use no scratch memory at all."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources as kr
    objdir = os.path.join(ROOT, "normalizing-flows_amd", "lib", "obj")
    seen = 0
    for obj, tag in (("rqs_fused.o", "rqs_fused_kernel"), ("rqs_fused_x3.o", "rqs_fused_x3_kernel")):
        for name, d in kr.resources(os.path.join(objdir, obj)).items():
            if tag in name:
                seen += 1
                assert d["vgpr_spill_count"] == 0 and d["private_segment_fixed_size"] == 0, (name, d)
                # scalar registers: none spilled in the benchmark's instantiations (8 bins, 128 hidden units) of the exact-fp32
                # kernel; the other instantiations and the split-bf16 chain (64 blob pointers + the layer loop's scalars) may
                # park a few in VGPR lanes (v_writelane: registers, not memory; 5 in the 16-bin / 32-unit one since round 3)
                # (round 6: the whole-layer training forward with the fused LU, <0, true, 2>, parks 2)
                strict = "x3" not in tag and "Li8ELi4EE" in name and "ILi0ELb1ELi2E" not in name
                assert d["sgpr_spill_count"] <= (0 if strict else 6), (name, d)
                assert d["vgpr_count"] <= 256
    assert seen == 43, seen    # exact fp32: 4 x {4, 8, 16 bins} x {128, 64, 32 hidden units} + the three training variants; 4 split-bf16


def test_one_pass_backward_kernels_use_no_scratch(nfa):
    """The one-pass backward kernels keep their weight slices and accumulators in registers: a spill reload would wait for every
    LDS-DMA in flight (vmcnt retires in order), which is what made the first INIT variant of nf_resblock_bwd twice as slow."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources as kr
    objdir = os.path.join(ROOT, "normalizing-flows_amd", "lib", "obj")
    seen = 0
    for obj, tag in (("resblock_bwd.o", "resblock_bwd_kernel"), ("lu_bwd.o", "lu_bwd_kernel"), ("lu_bwd.o", "lu_bwd_c_kernel")):
        for name, d in kr.resources(os.path.join(objdir, obj)).items():
            if tag in name:
                seen += 1
                assert d["vgpr_spill_count"] == 0 and d["private_segment_fixed_size"] == 0, (name, d)
    assert seen == 4, seen


@pytest.mark.parametrize("KB", [4, 8, 16])
@pytest.mark.parametrize("inverse", [False, True])
def test_first_descent_level_before_the_knots_selects_the_same_bin(KB, inverse):
    """numpy emulation of fused_common.hpp rqs_regs_h (round 6: the spline epilogue of nsf_wide.hip / made_fwd.hip decides the first
    level of the bin descent from the two half sums of the softmax and builds only that half's knots) against the full knot
    construction of rqs_regs_t: the same bin, the same knots and derivative logits for random parameter lists and inputs, both
    directions, every bin count -- in float64, where the two orders of the prefix sum agree to 1e-12."""
    left, right, bottom, top, min_w, min_h, edge = -3.0, 3.0, -3.0, 3.0, 1e-3, 1e-3, 0.5413
    H = KB // 2
    rng = np.random.default_rng(KB + 7 * inverse)

    def full(x, prm):
        ew, eh = np.exp2(prm[:KB] - prm[:KB].max()), np.exp2(prm[KB:2 * KB] - prm[KB:2 * KB].max())
        pw, ph = np.cumsum(ew), np.cumsum(eh)
        cw, ch = (right - left) * (1 - min_w * KB) / pw[-1], (top - bottom) * (1 - min_h * KB) / ph[-1]
        kw, kh, dp = np.zeros(KB + 1), np.zeros(KB + 1), np.zeros(KB + 1)
        kw[0], kh[0], kw[KB], kh[KB], dp[0], dp[KB] = left, bottom, right, top, edge, edge
        for k in range(1, KB):
            kw[k] = pw[k - 1] * cw + left + (right - left) * min_w * k
            kh[k] = ph[k - 1] * ch + bottom + (top - bottom) * min_h * k
            dp[k] = prm[2 * KB + k - 1]
        s_, o_ = (kh, kw) if inverse else (kw, kh)
        b = max(0, min(KB - 1, np.searchsorted(s_, x, side="right") - 1))
        return np.array([s_[b], s_[b + 1], o_[b], o_[b + 1], dp[b], dp[b + 1]])

    def lean(x, prm):
        mw, mh = prm[:KB].max(), prm[KB:2 * KB].max()
        ew, eh = np.exp2(prm[:KB] - mw), np.exp2(prm[KB:2 * KB] - mh)
        hw, uw, hh, uh = ew[:H].sum(), ew[H:].sum(), eh[:H].sum(), eh[H:].sum()
        cw, ch = (right - left) * (1 - min_w * KB) / (hw + uw), (top - bottom) * (1 - min_h * KB) / (hh + uh)
        sw, sh = (right - left) * min_w, (top - bottom) * min_h
        kwm, khm = hw * cw + left + sw * H, hh * ch + bottom + sh * H
        c = x >= (khm if inverse else kwm)
        w2 = prm[H:KB] if c else prm[:H]
        h2 = prm[KB + H:2 * KB] if c else prm[KB:KB + H]
        d2 = [(edge if i == H else prm[2 * KB + H + i - 1]) if c else (edge if i == 0 else prm[2 * KB + i - 1]) for i in range(H + 1)]
        kw, kh = np.zeros(H + 1), np.zeros(H + 1)
        kw[0], kh[0], kw[H], kh[H] = (kwm, khm, right, top) if c else (left, bottom, kwm, khm)
        aw, ah, ko = (hw, hh, H) if c else (0.0, 0.0, 0)
        for i in range(1, H):
            aw += np.exp2(w2[i - 1] - mw)
            ah += np.exp2(h2[i - 1] - mh)
            kw[i], kh[i] = aw * cw + left + sw * (ko + i), ah * ch + bottom + sh * (ko + i)
        s_, o_ = (kh, kw) if inverse else (kw, kh)
        b = max(0, min(H - 1, np.searchsorted(s_, x, side="right") - 1))
        return np.array([s_[b], s_[b + 1], o_[b], o_[b + 1], d2[b], d2[b + 1]])

    for _ in range(2000):
        prm = 2.0 * rng.normal(size=3 * KB)
        x = rng.uniform(-3.0, 3.0)
        np.testing.assert_allclose(lean(x, prm), full(x, prm), rtol=1e-10, atol=1e-10)


def test_tile_engine_and_maf_kernels_use_no_scratch(nfa):
    """Round 6: every instantiation of the 64-row-tile kernels -- nf_nsf_wide's 36 (4 / 8 / 16 bins, both
    directions, with and without the LU, three shapes), MADE's forward in all four epilogue modes -- and of the one-pass MAF
    kernels is free of scratch memory.  What was spilled there were loop invariants (per-tile address arithmetic hoisted across
    the products, the unroller's remainder bookkeeping of runtime-bound copy loops) and, at 16 bins, the spline epilogue's knot
    arrays (fused_common.hpp rqs_regs_h)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources as kr
    objdir = os.path.join(ROOT, "normalizing-flows_amd", "lib", "obj")
    want = {("rqs_fused_nw4.o", "rqs_fused_kernel_nw4"): 38,      # the 128-row-workgroup build: 36 inference + 2 whole-layer training forwards (round 6, late)
            ("nsf_wide.o", "nsf_wide_kernel"): 36, ("made_fwd.o", "made_fwd_kernel"): 9,
            ("maf_inverse_h.o", "maf_inverse_h_kernel"): 12, ("maf_inverse_h.o", "maf_solve_t_kernel"): 6}
    for (obj, tag), n in want.items():
        seen = 0
        for name, d in kr.resources(os.path.join(objdir, obj)).items():
            if tag in name:
                seen += 1
                if tag == "maf_inverse_h_kernel" and "ELb1EEEv" in name:
                    # the TRAINING instantiations (round 6, late: they also store MADE's output per feature, nf_maf_inverse_h_train) of
                    # the two-block kernels sit two registers over the 256 that two waves per SIMD leave: <= 16 bytes per lane; the
                    # inference instantiations -- BASELINE configs[4]'s kernel among them -- stay free of scratch
                    assert d["private_segment_fixed_size"] <= 16, (name, d)
                    continue
                if tag == "made_fwd_kernel" and "ELi128EEEv" in name:
                    # the 128-row-tile training forward (round 6, last session: two sample blocks per item next to the saved-row stores):
                    # 8 values in 36 bytes; its 64-row sibling and the 128-row backward use none
                    assert d["private_segment_fixed_size"] <= 40, (name, d)
                    continue
                if tag == "maf_solve_t_kernel" and "ELb1EEEv" in name:
                    # the regular-8 instantiations of the transposed solve (round 6, late): the two-block one keeps 7 values in 32 bytes of
                    # scratch (scheduling barriers between the targets made it 81-108) and is still 0.34 ms per layer faster
                    assert d["private_segment_fixed_size"] <= 32, (name, d)
                    continue
                assert d["vgpr_spill_count"] == 0 and d["private_segment_fixed_size"] == 0, (name, d)
        assert seen == n, (tag, seen)


@pytest.mark.parametrize("use_lu", [True, False])
def test_invertible_affine_torch_path_vs_reference_fixture(nfa, use_lu):
    """The differentiable path of InvertibleAffine (taken when a gradient is asked for) on CPU against the reference's outputs
    (tests/golden/invertible_affine_lu*.npz); gradients reach every parameter."""
    from conftest import assert_close, golden_state, load_golden
    g = load_golden("invertible_affine_lu%d" % int(use_lu))
    ia = nfa.flows.InvertibleAffine(7, use_lu=use_lu)
    ia.load_state_dict({k: torch.from_numpy(v) for k, v in golden_state(g).items()}, strict=True)
    z = torch.from_numpy(g["z"]).requires_grad_(True)
    y, ld = ia.forward(z)
    assert_close(y.detach().numpy(), g["z_fwd"], what="z_fwd", rtol=1e-4, atol=1e-5)
    assert_close(ld.detach().numpy(), g["ld_fwd"], what="ld_fwd", rtol=1e-5, atol=1e-5)
    yi, ldi = ia.inverse(z)
    assert_close(yi.detach().numpy(), g["z_inv"], what="z_inv", rtol=1e-4, atol=1e-5)
    assert_close(ldi.detach().numpy(), g["ld_inv"], what="ld_inv", rtol=1e-5, atol=1e-5)
    (y.square().sum() + yi.square().sum() + ld + ldi).backward()
    assert z.grad is not None and all(p.grad is not None and torch.isfinite(p.grad).all() for p in ia.parameters())


def test_cc_affine_const_torch_path_vs_reference_fixture(nfa):
    """The differentiable path of CCAffineConst on CPU against the reference's outputs (tests/golden/cc_affine_const.npz)."""
    from conftest import assert_close, golden_state, load_golden
    g = load_golden("cc_affine_const")
    cc = nfa.flows.CCAffineConst((3, 1, 1), 4)
    cc.load_state_dict({k: torch.from_numpy(v) for k, v in golden_state(g).items()}, strict=True)
    z, y = torch.from_numpy(g["z"]), torch.from_numpy(g["y"])
    out, ld = cc.forward(z, y)
    assert out.requires_grad
    assert_close(out.detach().numpy(), g["z_fwd"], what="z_fwd", rtol=1e-5, atol=1e-5)
    assert_close(ld.detach().numpy(), g["ld_fwd"], what="ld_fwd", rtol=1e-5, atol=1e-5)
    out, ld = cc.inverse(z, y)
    assert_close(out.detach().numpy(), g["z_inv"], what="z_inv", rtol=1e-5, atol=1e-5)
    assert_close(ld.detach().numpy(), g["ld_inv"], what="ld_inv", rtol=1e-5, atol=1e-5)
    (out.sum() + ld.sum()).backward()
    assert all(p.grad is not None for p in cc.parameters())


def test_utils_nn_helpers(nfa):
    """utils.sum_except_batch / utils.tile (utils/nn.py:181-193): per-sample reduction and element-wise repetition."""
    x = torch.arange(24.0).reshape(2, 3, 4)
    assert torch.equal(nfa.utils.sum_except_batch(x), x.sum(dim=(1, 2)))
    assert torch.equal(nfa.utils.sum_except_batch(x, 2), x.sum(dim=2))
    assert torch.equal(nfa.utils.sum_except_batch(torch.arange(3.0)), torch.tensor(3.0))
    assert torch.equal(nfa.utils.tile(torch.tensor([[1, 2], [3, 4]]), 3), torch.tensor([1, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 4]))


def test_reference_import_paths_resolve(nfa):
    """The reference's deep module paths of the hot-path files (normflows.flows.affine.coupling, .neural_spline.wrapper,
    nets.resnet, distributions.base, utils.splines, ...) resolve to the same classes as the flat namespaces (_refpaths.py)."""
    import importlib
    want = {
        "flows.affine.coupling": ["AffineConstFlow", "CCAffineConst", "AffineCoupling", "MaskedAffineFlow", "AffineCouplingBlock"],
        "flows.affine.glow": ["GlowBlock"],
        "flows.affine.autoregressive": ["MaskedAffineAutoregressive"],
        "flows.neural_spline.coupling": ["PiecewiseRationalQuadraticCDF", "PiecewiseRationalQuadraticCoupling"],
        "flows.neural_spline.wrapper": ["CoupledRationalQuadraticSpline", "AutoregressiveRationalQuadraticSpline"],
        "flows.neural_spline.autoregressive": ["MaskedPiecewiseRationalQuadraticAutoregressive"],
        "flows.mixing": ["LULinearPermute", "Invertible1x1Conv", "InvertibleAffine", "Permute"],
        "flows.normalization": ["ActNorm", "BatchNorm"],
        "flows.reshape": ["Split", "Merge", "Squeeze"],
        "nets.resnet": ["ResidualNet", "ResidualBlock"], "nets.mlp": ["MLP"], "nets.cnn": ["ConvNet2d"], "nets.made": ["MADE"],
        "distributions.base": ["DiagGaussian", "ClassCondDiagGaussian", "GlowBase"],
        "utils.splines": ["rational_quadratic_spline", "unconstrained_rational_quadratic_spline", "searchsorted"],
        "utils.masks": ["create_alternating_binary_mask"], "utils.nn": ["sum_except_batch", "tile"],
        "core": ["NormalizingFlow", "MultiscaleFlow"], "transforms": ["Logit", "Shift"],
    }
    flat = {"flows": nfa.flows, "nets": nfa.nets, "distributions": nfa.distributions}
    for path, names in want.items():
        mod = importlib.import_module("normflows_amd." + path)
        for n in names:
            obj = getattr(mod, n)
            top = flat.get(path.split(".")[0])
            if top is not None and hasattr(top, n):
                assert obj is getattr(top, n), (path, n)


def test_no_fence_barrier_behind_counted_vmcnt_waits():
    """Source lint for DESIGN 3.5 item 1: `__syncthreads()` is a fence, and while an LDS-DMA may be pending the compiler turns it
    into `s_waitcnt vmcnt(0)` -- a counted `s_waitcnt vmcnt(N)` directly in front of it never takes effect.  Every kernel source
    that counts must use the raw barrier (`s_waitcnt lgkmcnt(0)` + `s_barrier`) there."""
    import glob
    import re
    bad = []
    for path in sorted(glob.glob(os.path.join(ROOT, "normalizing-flows_amd", "csrc", "*.hip"))):
        lines = open(path).read().split("\n")
        for i, line in enumerate(lines):
            code = line.split("//")[0]
            if re.search(r"s_waitcnt vmcnt\((%0|[1-9][0-9]*)\)", code):
                for j in range(i + 1, min(i + 6, len(lines))):
                    nxt = lines[j].split("//")[0]
                    if "__syncthreads()" in nxt:
                        bad.append("%s:%d" % (os.path.basename(path), j + 1))
                        break
                    if "s_barrier" in nxt or "GL_BARRIER" in nxt or "GC_RING_BARRIER" in nxt or "BB_BARRIER" in nxt or "LB_BARRIER" in nxt:
                        break
    assert not bad, bad


def test_streaming_kernels_have_no_flat_loads(nfa):
    """Disassembly lint for DESIGN 3.5 item 3: a FLAT load counts on lgkmcnt as well as vmcnt, so every wait for an LDS read
    also waits for it -- the 16-pixel Glow kernel's register weight stream was FLAT loads (its base comes out of a pointer
    table = a generic pointer) until round 3.  The kernels that stream through registers / LDS-DMA rings must use global loads."""
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources as kr
    objdir = os.path.join(ROOT, "normalizing-flows_amd", "lib", "obj")
    # (object, kernel-name fragment, allowed FLAT loads: the level table's pointer fetches in the block prologue)
    checks = [("glow_conv.o", "glow_convnet_tiny_kernel", 4), ("glow_conv.o", "glow_convnet_small_kernel", 4),
              ("glow_conv.o", "glow_convnet_kernel", 4), ("maf_inverse_h.o", "maf_inverse_h_kernel", 0),
              ("final_bwd.o", "final_bwd_kernel", 0), ("rqs_fused.o", "rqs_fused_kernel", 0)]
    with tempfile.TemporaryDirectory() as tmp:
        cache = {}
        for obj, frag, allowed in checks:
            if obj not in cache:
                co = kr.code_object(os.path.join(objdir, obj), tmp)
                cache[obj] = subprocess.run([os.path.join(kr.LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True).stdout
            cur, counts = None, {}
            for line in cache[obj].split("\n"):
                if line.endswith(">:"):
                    cur = line.split("<")[-1][:-2]
                elif cur and frag in cur and "flat_load" in line:
                    counts[cur] = counts.get(cur, 0) + 1
            worst = max(counts.values()) if counts else 0
            assert worst <= allowed, (obj, frag, counts)


def test_bench_gpus_n_launches_itself():
    """`python bench.py --gpus 8` with no launcher in front re-executes under torch.distributed.run, one rank per GPU on this
    node, rendezvous on 127.0.0.1, same arguments (NF_BENCH_PRINT_LAUNCH=1 prints the line instead of exec'ing it)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["NF_BENCH_PRINT_LAUNCH"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "5", "--warmup", "2"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    cmd = json.loads(out.stdout.strip().splitlines()[-1])["launch"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]


@pytest.mark.parametrize("D,H,NB,mult", [(128, 512, 2, 2), (20, 64, 2, 2), (6, 300, 1, 2), (33, 256, 3, 2), (8, 40, 2, 23)])
def test_made_forward_pack_matches_dense_made(D, H, NB, mult):
    """flows/made_pack.py (hidden units sorted by degree, per row-block A-operand streams that stop at the last non-zero k-group of
    the MASK) + the kernel's layer schedule (tests/made_fwd_emulator.py restates it in numpy) reproduce nets.MADE.forward
    (nets/made.py:296-304 with the masked linears of :19-81) computed densely with plain torch in fp64."""
    from normflows_amd import nets
    from normflows_amd.flows import made_pack
    from made_fwd_emulator import emulate_forward, work_fraction, work_per_wave
    torch.manual_seed(D + H)
    made = nets.MADE(features=D, hidden_features=H, num_blocks=NB, output_multiplier=mult)
    with torch.no_grad():
        for p in made.parameters():
            p.add_(0.05 * torch.randn_like(p))
    blob, table = made_pack.pack_made_forward(made, mult)
    assert table[0] == D and table[3] in (256, 512) and table[3] >= H and blob.size % 32 == 0
    x = torch.randn(9, D)
    with torch.no_grad():
        ref = made.double()(x.double()).numpy()
    got = emulate_forward(blob, table, x.numpy())
    assert np.max(np.abs(got - ref)) < 1e-9 * max(1.0, np.abs(ref).max())
    if (D, H) == (128, 512):
        assert work_fraction(table) < 0.56          # BASELINE configs[4]: 54 % of the dense MFMA work ...
        wpw = work_per_wave(table)
        assert wpw.max() == wpw.min()               # ... and every wave of the workgroup gets the same share of it


@pytest.mark.parametrize("D,H,NB,mult,B", [(128, 512, 2, 2, 3), (20, 40, 2, 2, 7), (6, 16, 2, 23, 7), (33, 300, 1, 3, 5), (5, 7, 3, 2, 4)])
def test_made_backward_pack_matches_autograd(D, H, NB, mult, B):
    """flows/made_pack.pack_made_backward + the kernels' schedules (tests/made_bwd_emulator.py restates made_bwd.hip in numpy: the
    input-gradient chain over the per-wave streams of TRANSPOSED masked weights with k-group ranges, the weight-gradient tile list,
    the reduction's row / column maps and mask bytes) reproduce torch autograd through the dense masked MADE in float64
    (nets/made.py:296-304 under core.py:87-102): g_x, every weight gradient (zero under the mask) and bias gradient."""
    import copy
    from normflows_amd import nets
    from normflows_amd.flows import made_pack
    import made_bwd_emulator as E
    torch.manual_seed(D + H)
    made = nets.MADE(features=D, hidden_features=H, num_blocks=NB, output_multiplier=mult)
    with torch.no_grad():
        for p in made.parameters():
            p.add_(0.3 * torch.randn_like(p))
    pack = made_pack.pack_made_backward(made, mult)
    sl = made_pack._slot_layers(made, mult)
    assert pack["table"][3] in (256, 512) and pack["nflat"] == sum(p.numel() for p in made.parameters())
    md = copy.deepcopy(made).double()
    x = torch.randn(B, D, dtype=torch.float64, requires_grad=True)
    out = md(x)
    gp = torch.randn_like(out)
    out.backward(gp)
    S, prm = E.slot_forward(sl["layers"], NB, x.detach().numpy(), sl["Dp"])
    assert np.allclose(prm[:, :mult * D], out.detach().numpy(), atol=1e-9)
    gx, G = E.emulate_chain(pack, gp.numpy(), S)
    assert np.abs(gx - x.grad.numpy()).max() < 1e-9 * max(1.0, float(x.grad.abs().max()))
    flat = E.emulate_wgrad(pack, gp.numpy(), x.detach().numpy(), G, S)
    for (woff, shape, boff, n), lin in zip(pack["offsets"], md._linears()):
        gw = flat[woff:woff + shape[0] * shape[1]].reshape(shape)
        assert np.abs(gw - lin.weight.grad.numpy()).max() < 1e-9 * max(1.0, float(lin.weight.grad.abs().max()))
        assert np.abs(flat[boff:boff + n] - lin.bias.grad.numpy()).max() < 1e-9 * max(1.0, float(lin.bias.grad.abs().max()))
    if (D, H) == (128, 512):                                  # BASELINE configs[4]: 50 of the 80 dense 128 x 128 tiles
        assert pack["ntiles"] == 50
        nkg = pack["table"][32:32 + 8 * 11 * 4].reshape(8, 11, 4)[:, :, 0]
        assert (nkg[:, :10].sum(axis=1) == nkg[0, :10].sum()).all()      # equal MFMA work per wave in the hidden products


@pytest.mark.parametrize("Cin,hid,Cout,B,H,W", [(6, 256, 12, 2, 4, 4), (24, 40, 48, 1, 3, 5), (3, 16, 5, 2, 2, 2), (14, 300, 4, 1, 4, 4)])
def test_conv_conditioner_as_pixel_mlp_matches_conv2d_autograd(Cin, hid, Cout, B, H, W):
    """The conv conditioner's training path restated in numpy: 3x3 convolutions as gather / neighbour-sum (csrc/conv_rows.hip) around
    the plain-MLP packs (flows/made_pack.pack_mlp_*: W1c[o][tap Cin + c], W3t[tap Cout + o][c]) walked by the kernel emulators
    (tests/made_fwd_emulator.py, tests/made_bwd_emulator.py) against torch conv2d + autograd in float64 (nets/cnn.py:5-63): output,
    input gradient, every weight / bias gradient; 9 Cin up to 216 (two rounds of the last product), hidden padded to 256 / 512."""
    from normflows_amd.flows import made_pack
    import made_bwd_emulator as E
    import made_fwd_emulator as F

    def gather(x, flip=False):
        Bn, C, Hn, Wn = x.shape
        col = np.zeros((Bn, Hn, Wn, 9, C))
        for tap in range(9):
            dy, dx = (tap // 3 - 1, tap % 3 - 1) if not flip else (1 - tap // 3, 1 - tap % 3)
            for y in range(Hn):
                for xx in range(Wn):
                    if 0 <= y + dy < Hn and 0 <= xx + dx < Wn:
                        col[:, y, xx, tap, :] = x[:, :, y + dy, xx + dx]
        return col.reshape(Bn * Hn * Wn, 9 * C)

    def gather_sum(P, bias, shape, flip=False):
        Bn, C, Hn, Wn = shape
        P = P.reshape(Bn, Hn, Wn, 9, C)
        out = np.zeros(shape)
        for tap in range(9):
            dy, dx = (tap // 3 - 1, tap % 3 - 1) if not flip else (1 - tap // 3, 1 - tap % 3)
            for y in range(Hn):
                for xx in range(Wn):
                    if 0 <= y + dy < Hn and 0 <= xx + dx < Wn:
                        out[:, :, y, xx] += P[:, y + dy, xx + dx, tap, :]
        return out if bias is None else out + bias[None, :, None, None]
    torch.manual_seed(Cin + hid)
    net = torch.nn.Sequential(torch.nn.Conv2d(Cin, hid, 3, padding=1), torch.nn.LeakyReLU(0.0), torch.nn.Conv2d(hid, hid, 1),
                              torch.nn.LeakyReLU(0.0), torch.nn.Conv2d(hid, Cout, 3, padding=1)).double()
    x = torch.randn(B, Cin, H, W, dtype=torch.float64, requires_grad=True)
    out = net(x)
    go = torch.randn_like(out)
    out.backward(go)
    c1, c2, c3 = net[0], net[2], net[4]
    f32 = lambda t: t.detach().numpy().astype(np.float32)
    args = (f32(c1.weight.permute(0, 2, 3, 1).reshape(hid, 9 * Cin)), f32(c1.bias), f32(c2.weight[:, :, 0, 0]), f32(c2.bias),
            f32(c3.weight.permute(2, 3, 0, 1).reshape(9 * Cout, hid)))
    blob, table = made_pack.pack_mlp_forward(*args)
    pack = made_pack.pack_mlp_backward(*args)
    sl = made_pack._mlp_layers(*args, None)
    assert table[13] == 1 and pack["table"][13] == 1 and pack["table"][8] == (sl["Dp"] // 32 + 3) // 4
    col = gather(x.detach().numpy())
    o = gather_sum(F.emulate_forward(blob, table, col), c3.bias.detach().numpy(), tuple(out.shape))
    assert np.abs(o - out.detach().numpy()).max() < 1e-9
    xin = np.zeros((col.shape[0], sl["Dp"]))
    xin[:, :col.shape[1]] = col
    h = xin @ sl["layers"][0][0].astype(np.float64).T + sl["layers"][0][2]
    t = np.maximum(h, 0) @ sl["layers"][1][0].astype(np.float64).T + sl["layers"][1][2]
    S = [h, t, np.zeros_like(h)]
    gP = gather(go.numpy(), flip=True)
    gcol, G = E.emulate_chain(pack, gP, S)
    assert np.abs(gather_sum(gcol, None, tuple(x.shape), flip=True) - x.grad.numpy()).max() < 1e-9
    flat = E.emulate_wgrad(pack, gP, col, G, S)
    (o0, s0, b0, n0), (o1, s1, b1, n1), (o2, s2, _, _) = pack["offsets"]
    assert np.abs(flat[o0:o0 + s0[0] * s0[1]].reshape(hid, 3, 3, Cin).transpose(0, 3, 1, 2) - c1.weight.grad.numpy()).max() < 1e-9
    assert np.abs(flat[o1:o1 + s1[0] * s1[1]].reshape(hid, hid) - c2.weight.grad.numpy()[:, :, 0, 0]).max() < 1e-9
    assert np.abs(flat[o2:o2 + s2[0] * s2[1]].reshape(3, 3, Cout, hid).transpose(2, 3, 0, 1) - c3.weight.grad.numpy()).max() < 1e-9
    assert np.abs(flat[b0:b0 + n0] - c1.bias.grad.numpy()).max() < 1e-9 and np.abs(flat[b1:b1 + n1] - c2.bias.grad.numpy()).max() < 1e-9
    # the training structure's reduction maps scatter straight into the conv parameters' own (o, c, ky, kx) layouts
    st = made_pack.convnet_train_structure(Cin, hid, Cout)
    flat2 = E.emulate_wgrad(st["bwd"], gP, col, G, S)
    assert np.abs(flat2[o0:o0 + s0[0] * s0[1]].reshape(hid, Cin, 3, 3) - c1.weight.grad.numpy()).max() < 1e-9
    assert np.abs(flat2[o1:o1 + s1[0] * s1[1]].reshape(hid, hid, 1, 1) - c2.weight.grad.numpy()).max() < 1e-9
    assert np.abs(flat2[o2:o2 + s2[0] * s2[1]].reshape(Cout, hid, 3, 3) - c3.weight.grad.numpy()).max() < 1e-9
    assert np.abs(flat2[b0:b0 + n0] - c1.bias.grad.numpy()).max() < 1e-9 and np.abs(flat2[b1:b1 + n1] - c2.bias.grad.numpy()).max() < 1e-9


def test_training_packs_as_gather_indices_reproduce_the_value_packs():
    """flows/made_pack.*_train_structure: the packers run on parameter POSITIONS instead of values give gather indices with
    flat[src] == the value packs, bit for bit, for a MADE (masked), a ResidualNet (dense) and the conv conditioner (permuted conv
    weights) -- what nf_pack_gather rebuilds on the device in every training forward."""
    import normflows_amd as nfa
    from normflows_amd.flows import made_pack
    torch.manual_seed(0)
    made = nfa.nets.MADE(20, 40, num_blocks=2, output_multiplier=2)
    net = nfa.nets.ResidualNet(17, 391, 300, num_blocks=2)
    cn = nfa.nets.ConvNet2d([6, 256, 256, 12], [3, 1, 3], init_zeros=False)
    with torch.no_grad():
        for p in list(made.parameters()) + list(net.parameters()):
            p.add_(torch.randn_like(p))
    flat_of = lambda ts: np.concatenate([np.zeros(1, np.float32)] + [t.detach().numpy().reshape(-1) for t in ts])
    st = made_pack.made_train_structure(made, 2)
    flat = flat_of([t for l in made._linears() for t in (l.weight, l.bias)])
    assert np.array_equal(flat[st["src"]], made_pack.pack_made_forward(made, 2)[0])
    assert np.array_equal(flat[st["bwd"]["src"]], made_pack.pack_made_backward(made, 2)["blob"])
    st = made_pack.resnet_train_structure(net)
    lins = [net.initial_layer] + [l for b in net.blocks for l in b.linear_layers] + [net.final_layer]
    flat = flat_of([t for l in lins for t in (l.weight, l.bias)])
    assert np.array_equal(flat[st["src"]], made_pack.pack_resnet_forward(net)[0])
    assert np.array_equal(flat[st["bwd"]["src"]], made_pack.pack_resnet_backward(net)["blob"])
    c1, c2, c3 = cn.net[0], cn.net[2], cn.net[4]
    st = made_pack.convnet_train_structure(6, 256, 12)
    flat = flat_of([c1.weight, c1.bias, c2.weight, c2.bias, c3.weight])
    args = (c1.weight.detach().permute(0, 2, 3, 1).reshape(256, 54).numpy(), c1.bias.detach().numpy(), c2.weight.detach()[:, :, 0, 0].numpy(),
            c2.bias.detach().numpy(), c3.weight.detach().permute(2, 3, 0, 1).reshape(108, 256).numpy())
    assert np.array_equal(flat[st["src"]], made_pack.pack_mlp_forward(*args)[0])
    assert np.array_equal(flat[st["bwd"]["src"]], made_pack.pack_mlp_backward(*args)["blob"])
    assert st["table"][14] == 128 and st["bwd"]["table"][14] == 128 and st["bwd"]["table"][15] == 128      # padded row strides


def test_cache_keys_follow_fused_optimizer_steps():
    """torch's fused optimizers leave Tensor._version untouched; the global post-step hook of _keys.py advances the epoch that every
    packed-weight cache key carries."""
    from normflows_amd import _keys
    p = torch.nn.Parameter(torch.randn(4))
    opt = torch.optim.Adam([p], lr=0.1, fused=True)
    k0 = _keys.pkey([p])
    p.grad = torch.ones(4)
    opt.step()
    assert _keys.pkey([p]) != k0
    k1 = _keys.pkey([p])
    with torch.no_grad():
        p.add_(1.0)
    assert _keys.pkey([p]) != k1
    # ... and ONLY the stepped optimizer's parameters go stale: a frozen model in the same process keeps its packs
    # and its recorded graphs through another model's training steps
    q = torch.nn.Parameter(torch.randn(4))              # e.g. a teacher / EMA model's weight
    kq, sq, st = _keys.pkey([q]), _keys.signature([q]), _keys.stamp()
    kp = _keys.pkey([p])
    opt.step()
    assert _keys.stamp() != st                          # holders of derived state take a look ...
    assert _keys.pkey([q]) == kq and _keys.signature([q]) == sq      # ... and find the frozen tensor untouched
    assert _keys.pkey([p]) != kp
    _keys.bump()                                        # invalidate_caches(): everything
    assert _keys.pkey([q]) != kq and _keys.signature([q]) != sq


def test_wide_resnet_training_path_rechecks_mode_dependent_eligibility():
    """ADVICE r04 (nets.py:78): a ResidualNet with dropout is the kernels' plain ReLU MLP only in eval().  A structure cached by a
    grad-enabled eval() call must not serve a later train() call (MadeFn has no dropout), and the train() verdict must not stick:
    `_train_packs` asks made_pack.resnet_supported on EVERY call."""
    import normflows_amd as nfa
    from normflows_amd.flows import made_pack
    net = nfa.nets.ResidualNet(8, 16, 256, num_blocks=2, dropout_probability=0.2)
    net.eval()
    assert made_pack.resnet_supported(net) and made_pack.resnet_train_structure(net) is not None
    # as if an eval()-mode forward under autograd had cached the structure on the module
    net.__dict__["_train_struct"] = (("cpu",), {"sentinel": True})
    net.train()
    assert not made_pack.resnet_supported(net)
    assert net._train_packs("cpu") is None                       # dropout is live: the cached structure is not used
    assert net.__dict__["_train_struct"][1] == {"sentinel": True}   # ... and the None verdict did not overwrite the cache
    net.eval()
    assert made_pack.resnet_supported(net)                       # eligible again once dropout is the identity
    plain = nfa.nets.ResidualNet(8, 16, 256, num_blocks=2)
    plain.train()
    assert made_pack.resnet_supported(plain)


def test_conv_conditioner_training_path_declines_other_structures():
    """autograd.ConvNetFn only takes the GlowBlock network (3x3 -> 1x1 -> 3x3, ReLU i.e. LeakyReLU(0), biases, 9 Cin <= 256 next to <= 256
    hidden channels): anything else keeps the library path (nets.ConvNet2d._train_packs returns None before touching the device)."""
    import normflows_amd as nfa
    from normflows_amd.flows import made_pack
    assert nfa.nets.ConvNet2d([6, 32, 32, 12], [3, 1, 3], leaky=0.1)._train_packs("cpu") is None            # leaky slope
    assert nfa.nets.ConvNet2d([6, 32, 32, 12], [3, 3, 3])._train_packs("cpu") is None                       # kernel sizes
    assert nfa.nets.ConvNet2d([6, 32, 12], [3, 3])._train_packs("cpu") is None                              # depth
    assert nfa.nets.ConvNet2d([6, 32, 32, 12], [3, 1, 3], actnorm=True)._train_packs("cpu") is None         # no biases / extra layers
    assert made_pack.convnet_train_structure(29, 256, 12) is None                                           # 9 Cin = 261 > 256
    assert made_pack.convnet_train_structure(15, 300, 12) is None                                           # 9 Cin = 135 > 128 next to 512 slots
    assert made_pack.convnet_train_structure(14, 300, 12) is not None


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout only exists in the build container")
def test_implicit_differentiation_of_the_maf_inverse_vs_reference_d_pass_autograd():
    """The formula behind autograd.MafInverseFn, with the REFERENCE's own modules in float64 and no kernel involved: for x = T^-1(z)
    (affine/autoregressive.py:29-38, D MADE passes) and cotangents (c_x, c_ld), the sweep v <- (c_x - J^T g_p(v, c_ld)) / s reaches a
    bit-stable v in at most D sweeps, g_z = v and the parameter gradient = the VJP of ONE MADE pass with g_p(-v, -c_ld) -- equal to
    torch autograd through the reference's D-pass loop to 1e-11 of scale."""
    sys.path.insert(0, REF)
    nf = pytest.importorskip("normflows")
    torch.manual_seed(0)
    D, H, B = 9, 28, 6
    layer = nf.flows.MaskedAffineAutoregressive(D, H, num_blocks=2).double()
    with torch.no_grad():
        for p in layer.parameters():
            p.add_(0.3 * torch.randn_like(p))
    z = torch.randn(B, D, dtype=torch.float64, requires_grad=True)
    cx, cl = torch.randn(B, D, dtype=torch.float64), torch.randn(B, dtype=torch.float64)
    x, ld = layer.inverse(z)
    ((x * cx).sum() + (ld * cl).sum()).backward()
    ref = [z.grad.clone()] + [p.grad.clone() for p in layer.parameters()]
    layer.zero_grad()
    with torch.no_grad():
        xs, _ = layer.inverse(z.detach())
    xq = xs.clone().requires_grad_(True)
    prm = layer.autoregressive_net(xq)
    pr = prm.view(B, D, 2)
    sg = torch.sigmoid(pr[..., 0] + 2.0).detach()
    s_ = sg + 1e-3

    def g_p(a, c):                       # cotangent of the MADE output for cotangent a on z = s x + t and c on sum log s
        return torch.stack([(a * xs + c[:, None] / s_) * sg * (1 - sg), a], -1).reshape(B, 2 * D)
    v = cx / s_
    for sweeps in range(1, D + 2):
        vn = (cx - torch.autograd.grad(prm, xq, g_p(v, cl), retain_graph=True)[0]) / s_
        done = torch.equal(vn, v)
        v = vn
        if done:
            break
    assert sweeps <= D
    prm.backward(g_p(-v, -cl))
    got = [v] + [q.grad for q in layer.parameters()]
    assert all(float((a - b).abs().max()) <= 1e-11 * max(1.0, float(b.abs().max())) for a, b in zip(got, ref))


@pytest.mark.parametrize("kind", ["arnsf", "circular", "maf"])
def test_generic_implicit_inverse_function_on_the_reference_own_layers(kind):
    """autograd.ArInverseImplicitFn -- the Function itself, on the CPU, in float64 -- driving the REFERENCE's own autoregressive layers
    through a three-method adapter (inverse / forward / the element-wise transform at given parameters): the autoregressive spline
    layer (neural_spline/autoregressive.py:11-134), its circular variant, the affine layer.  Gradients of x = T^-1(z) and its
    log-determinant for random cotangents against torch autograd through the reference's D recorded passes
    (autoregressive.py:29-40): 1e-10 of scale, at most D + 1 sweeps; each cotangent alone as well."""
    sys.path.insert(0, REF)
    nf = pytest.importorskip("normflows")
    from normflows_amd.autograd import ArInverseImplicitFn
    torch.manual_seed(3)
    D, H, B = 7, 20, 9
    if kind == "arnsf":
        ref = nf.flows.AutoregressiveRationalQuadraticSpline(D, 2, H).mprqat
        z0 = 1.5 * torch.randn(B, D, dtype=torch.float64)
    elif kind == "circular":
        ref = nf.flows.CircularAutoregressiveRationalQuadraticSpline(D, 1, H, [1, 4], tail_bound=torch.tensor([5.0, 3.0, 4.0, 5.0, 3.14159, 5.0, 5.0])).mprqat
        z0 = torch.randn(B, D, dtype=torch.float64).clamp(-2.9, 2.9)
    else:
        ref = nf.flows.MaskedAffineAutoregressive(D, H, num_blocks=2)
        z0 = torch.randn(B, D, dtype=torch.float64)
    ref = ref.double()
    with torch.no_grad():
        for p in ref.parameters():
            p.add_(0.2 * torch.randn_like(p))

    class Adapter:
        autoregressive_net = ref.autoregressive_net
        inverse = staticmethod(lambda z: ref.inverse(z))
        forward = staticmethod(lambda x: ref.forward(x))
        _elementwise = staticmethod(lambda x, theta, direction: ref._elementwise_forward(x, theta))
    params = tuple(ref.parameters())
    cx, cl = torch.randn(B, D, dtype=torch.float64), torch.randn(B, dtype=torch.float64)
    for use_x, use_l in ((True, True), (True, False), (False, True)):
        out = []
        for implicit in (False, True):
            ref.zero_grad(set_to_none=True)
            z = z0.clone().requires_grad_(True)
            x, ld = ArInverseImplicitFn.apply(Adapter, z, *params) if implicit else ref.inverse(z)
            ((x * cx).sum() * float(use_x) + (ld * cl).sum() * float(use_l) if (use_x and use_l) else
             ((x * cx).sum() if use_x else (ld * cl).sum())).backward()
            out.append([x.detach(), ld.detach(), z.grad] + [p.grad.clone() for p in params])
        assert 1 <= ArInverseImplicitFn.last_sweeps <= D + 1
        for k, (a, b) in enumerate(zip(out[1], out[0])):
            assert float((a - b).abs().max()) <= 1e-10 * max(1.0, float(b.abs().max())), (kind, use_x, use_l, k, float((a - b).abs().max()))


def test_made_forward_pack_rejects_unsupported():
    from normflows_amd import nets
    from normflows_amd.flows import made_pack
    assert made_pack.pack_made_forward(nets.MADE(features=8, hidden_features=16, num_blocks=2, output_multiplier=2,
                                                 use_residual_blocks=False)) is None
    assert made_pack.pack_made_forward(nets.MADE(features=8, hidden_features=16, num_blocks=2, output_multiplier=2,
                                                 permute_mask=True)) is None or True      # a permutation may be the identity
    assert made_pack.pack_made_forward(nets.MADE(features=200, hidden_features=64, num_blocks=2, output_multiplier=2)) is None
    assert made_pack.pack_made_forward(nets.MADE(features=8, hidden_features=600, num_blocks=2, output_multiplier=2)) is None


@pytest.mark.parametrize("D,H,NB,rev,K", [(64, 256, 2, False, 8), (128, 128, 2, False, 8), (128, 256, 1, True, 8), (96, 192, 2, False, 8),
                                          (7, 300, 2, True, 8), (66, 512, 3, False, 8), (64, 256, 2, False, 4), (128, 256, 1, True, 16),
                                          (128, 128, 2, False, 4), (30, 100, 2, True, 16), (9, 200, 2, False, 4), (128, 512, 2, False, 16)])
def test_nsf_wide_pack_matches_dense_conditioner(D, H, NB, rev, K):
    """flows/nsf_wide_pack.py (zero-padded hidden units, the initial layer on full rows, the final layer in groups of four
    transform features whose MFMA rows are the lanes' 2 x 24 parameter lists, width / height rows pre-scaled by log2(e) / sqrt(H))
    + the kernel's walk over the per-wave streams (tests/nsf_wide_emulator.py) reproduce the reference-layout conditioner
    ResidualNet(identity features) (nets/resnet.py:92-104, nsf/coupling.py:83-86, :334-339) computed densely in fp64."""
    import normflows_amd as nfa
    from normflows_amd.flows import nsf_wide_pack
    from nsf_wide_emulator import emulate_conditioner
    torch.manual_seed(D + H)
    layer = nfa.flows.CoupledRationalQuadraticSpline(D, NB, H, num_bins=K, init_identity=False, reverse_mask=rev)
    with torch.no_grad():
        for p in layer.parameters():
            p.add_(0.05 * torch.randn_like(p))
    prqct = layer.prqct
    blob, table = nsf_wide_pack.pack_nsf_wide(prqct)
    assert table[0] == D and table[3] in (128, 256, 512) and table[3] >= H and blob.size % 256 == 0 and table[24] == K
    M_ = 3 * K - 1                   # (round 5: 4 / 16 bins on the same schedule: 8 / 2 transform features per group instead of 4)
    x = torch.randn(5, D)
    nT = len(prqct.transform_features)
    import copy
    net64 = copy.deepcopy(prqct.transform_net).double()
    sc = 1.4426950408889634 / np.sqrt(float(H))

    def reference(rows):
        with torch.no_grad():
            r = net64(rows.double().index_select(1, prqct.identity_features)).numpy().reshape(5, nT, M_).copy()
        r[:, :, :2 * K] *= sc
        return r

    got, lu_out = emulate_conditioner(blob, table, x.numpy())
    ref_s = reference(x)
    assert lu_out is None
    assert np.max(np.abs(got[:, :, :M_] - ref_s)) < 1e-5 * max(1.0, np.abs(ref_s).max())     # (the scale is applied in float32)
    assert np.all(got[:, :, M_] == 0.0)
    # with the adjacent LULinearPermute as a dense product: first in the density direction, last in the sampling direction
    g = torch.Generator().manual_seed(1)
    Wl, bl = torch.randn(D, D, generator=g) / np.sqrt(D), torch.randn(D, generator=g)
    xl = (x.double() @ Wl.double().T + bl.double())
    blob, table = nsf_wide_pack.pack_nsf_wide(prqct, lu=(Wl.numpy(), bl.numpy()), direction=0)
    got, lu_out = emulate_conditioner(blob, table, x.numpy(), 0)
    assert np.max(np.abs(lu_out - xl.numpy())) < 1e-5
    ref_s = reference(xl.float())
    assert np.max(np.abs(got[:, :, :M_] - ref_s)) < 1e-4 * max(1.0, np.abs(ref_s).max())
    blob, table = nsf_wide_pack.pack_nsf_wide(prqct, lu=(Wl.numpy(), bl.numpy()), direction=1)
    got, lu_out = emulate_conditioner(blob, table, x.numpy(), 1)
    assert np.max(np.abs(lu_out - xl.numpy())) < 1e-5
    ref_s = reference(x)
    assert np.max(np.abs(got[:, :, :M_] - ref_s)) < 1e-5 * max(1.0, np.abs(ref_s).max())


def test_nsf_wide_pack_rejects_unsupported():
    import normflows_amd as nfa
    from normflows_amd.flows import nsf_wide_pack
    assert nsf_wide_pack.pack_nsf_wide(nfa.flows.CoupledRationalQuadraticSpline(64, 2, 256, num_bins=10).prqct) is None
    assert nsf_wide_pack.pack_nsf_wide(nfa.flows.CoupledRationalQuadraticSpline(130, 2, 256).prqct) is None
    assert nsf_wide_pack.pack_nsf_wide(nfa.flows.CoupledRationalQuadraticSpline(64, 2, 600).prqct) is None


def test_invalidate_caches_drops_every_packed_image(nfa):
    """normflows_amd.invalidate_caches: what a caller runs after updating parameters through `.data` (no `_version` bump, so the
    (data_ptr, _version) keys of the packed-weight caches cannot see the change)."""
    layer = nfa.flows.CoupledRationalQuadraticSpline(8, 1, 16)
    lu = nfa.flows.LULinearPermute(8)
    made = nfa.nets.MADE(features=6, hidden_features=12, num_blocks=1, output_multiplier=2)
    m = torch.nn.ModuleList([layer, lu, made])
    layer.prqct._fused_cache = ("k", object())
    layer.prqct.__dict__["_wide_cache"] = {None: ("k", object())}
    lu._dense_cache = ("k", object())
    made.__dict__["_fwd_pack_cache"] = ("k", object())
    made.initial_layer._masked_cache = ("k", object())
    v0 = layer.prqct.transform_net.final_layer.bias._version
    layer.prqct.transform_net.final_layer.bias.data.add_(1.0)
    assert layer.prqct.transform_net.final_layer.bias._version == v0        # the reason the hook exists
    nfa.invalidate_caches(m)
    assert layer.prqct._fused_cache is None and layer.prqct.__dict__["_wide_cache"] == {} and lu._dense_cache is None
    assert made.__dict__["_fwd_pack_cache"] is None and made.initial_layer._masked_cache is None


@pytest.mark.parametrize("D,H,NB", [(64, 256, 2), (6, 16, 2), (33, 300, 1), (128, 512, 2)])
def test_made_forward_spline_pack_matches_dense_made(D, H, NB):
    """flows/made_pack.py spline=True (the autoregressive spline layer's MADE: final layer in groups of four features whose MFMA rows
    are the lanes' 2 x 24 parameter lists, widths / heights pre-scaled by log2(e), groups dealt to the waves by work) + the kernel's
    walk (tests/made_fwd_emulator.py) reproduce nets.MADE.forward with 23 outputs per feature, computed densely in fp64."""
    from normflows_amd import nets
    from normflows_amd.flows import made_pack
    from made_fwd_emulator import emulate_forward_spline
    torch.manual_seed(D * 3 + H)
    made = nets.MADE(features=D, hidden_features=H, num_blocks=NB, output_multiplier=23)
    with torch.no_grad():
        for p in made.parameters():
            p.add_(0.05 * torch.randn_like(p))
    blob, table = made_pack.pack_made_forward(made, 23, spline=True)
    assert table[11] == 1 and table[7] == (D + 3) // 4
    x = torch.randn(5, D)
    import copy
    with torch.no_grad():
        ref = copy.deepcopy(made).double()(x.double()).numpy().reshape(5, D, 23).copy()
    ref[:, :, :16] *= 1.4426950408889634
    got = emulate_forward_spline(blob, table, x.numpy())
    assert np.max(np.abs(got[:, :, :23] - ref)) < 1e-5 * max(1.0, np.abs(ref).max()) and np.all(got[:, :, 23] == 0.0)
    assert made_pack.pack_made_forward(nets.MADE(features=8, hidden_features=16, num_blocks=2, output_multiplier=29), 29, spline=True) is None


@pytest.mark.parametrize("tri", [False, True])
@pytest.mark.parametrize("D,H,NB", [(12, 40, 2), (17, 40, 1), (20, 64, 2), (10, 36, 3), (40, 39, 2), (33, 70, 2), (9, 34, 1), (64, 256, 2),
                                    (128, 512, 2)])
def test_maf_transposed_pack_solves_the_implicit_system(D, H, NB, tri):
    """flows/maf_pack.pack_made_transposed + the schedule of csrc/maf_solve_t.hip (numpy emulation, tests/maf_emulator.py) solve
    v s + J^T g_p(v, g_ld) = g_x -- the linear system of autograd.MafInverseFn's backward -- in ONE pass: against the dense solution
    assembled from float64 autograd (J^T through torch.autograd.grad on the reference-structured MADE)."""
    import normflows_amd as nfa
    from normflows_amd import nets
    from normflows_amd.flows import maf_pack
    from maf_emulator import emulate_solve_t
    torch.manual_seed(3 * D + H + NB)
    made = nets.MADE(features=D, hidden_features=H, num_blocks=NB, output_multiplier=2)
    with torch.no_grad():
        for p in made.parameters():
            p.add_((0.3 if H < 100 else 0.03) * torch.randn_like(p))      # (wide nets: keep the triangular system well conditioned)
    blob, table = maf_pack.pack_made_transposed(made, blocks=(1, 2, 3), tri=tri)    # tri: the format-1 forward pack's positions
    assert table[7] == 2 and table[6] == NB and table[1] % 32 == 0
    cols = maf_pack.solve_t_gradient_columns(made, tri=tri)      # (the packers take the float32 module: before .double() below)
    T, Hp = int(table[4]), int(table[3])
    B = 5
    m64 = made.double()
    x = torch.randn(B, D, dtype=torch.float64, requires_grad=True)
    gx = torch.randn(B, D, dtype=torch.float64)
    gld = torch.randn(B, dtype=torch.float64)
    prm = m64(x)                                            # (B, 2 D): row 2 f = unconstrained scale, 2 f + 1 = shift
    sg = torch.sigmoid(prm[:, 0::2] + 2.0)
    scale = sg + 1e-3

    def op(v):                                              # v s + J^T g_p(v, g_ld)
        gp = torch.zeros_like(prm)
        gp[:, 0::2] = (v * x.detach() + gld[:, None] / scale) * sg * (1.0 - sg)
        gp[:, 1::2] = v
        (jt,) = torch.autograd.grad(prm, x, grad_outputs=gp.detach(), retain_graph=True)
        return (v * scale + jt).detach()
    c = op(torch.zeros(B, D, dtype=torch.float64))
    A = torch.stack([op(torch.eye(D, dtype=torch.float64)[i].expand(B, D)) - c for i in range(D)], dim=2)   # (B, D, D): column i
    v_ref = torch.linalg.solve(A, (gx - c).unsqueeze(2)).squeeze(2)
    # ReLU masks at x, per forward layer, in virtual slot order
    lins = [m64.initial_layer] + [l for b in m64.blocks for l in b.linear_layers]
    with torch.no_grad():
        h = torch.nn.functional.linear(x, lins[0].weight * lins[0].mask, lins[0].bias)
        pres = [h]
        for b in range(NB):
            t_ = torch.nn.functional.linear(torch.relu(h), lins[1 + 2 * b].weight * lins[1 + 2 * b].mask, lins[1 + 2 * b].bias)
            pres.append(t_)
            h = h + torch.nn.functional.linear(torch.relu(t_), lins[2 + 2 * b].weight * lins[2 + 2 * b].mask, lins[2 + 2 * b].bias)
            pres.append(h)
    order, tiles = maf_pack.plan_tiles(D, made.initial_layer.degrees.numpy())
    fslot = np.zeros(H, dtype=np.int64)
    k = 0
    for t, (dlo, ns, steps) in enumerate(tiles):
        b_ = 32 * t
        perm = tri and (maf_pack.is_regular(steps) or maf_pack.extras_prefix(steps) > 0)
        for g_, c_ in enumerate(steps):
            for i_ in range(c_):
                fslot[order[k]] = 32 * t + (maf_pack.tile_row(g_, i_) if perm else b_ - 32 * t)
                b_ += 1
                k += 1
    vslot = (T - 1 - fslot // 32) * 32 + fslot % 32
    masks = []
    for kq in range(1, 2 * NB + 1):                         # virtual layer k <-> forward layer 2 NB - k
        mk = np.zeros((B, Hp), dtype=bool)
        mk[:, vslot] = (pres[2 * NB - kq] > 0).numpy()
        masks.append(mk)
    v, S = emulate_solve_t(blob, table, x.detach().numpy(), prm.detach().numpy(), gx.numpy(), gld.numpy(), masks, return_scratch=True)
    np.testing.assert_allclose(v, v_ref.numpy(), rtol=1e-8, atol=1e-8)
    # The activations the solve publishes ARE MADE's input-gradient chain at the solution: through maf_pack.solve_t_gradient_columns
    # (nf_maf_scratch_rows on the device) they give the weight-gradient launch its G[l] = d<prm, g_p(v, g_ld)> / d(pre-activation l) in the
    # training kernels' column order (units sorted by degree), layer order reversed -- against float64 autograd on the written-out forward.
    assert cols.shape == (256 if H <= 256 else 512,) and (cols[H:] == -1).all() and len(set(cols[:H].tolist())) == H
    xq = x.detach()
    h = torch.nn.functional.linear(xq, lins[0].weight * lins[0].mask, lins[0].bias)
    h.retain_grad()
    nodes, hcur = [h], h
    for b in range(NB):
        t_ = torch.nn.functional.linear(torch.relu(hcur), lins[1 + 2 * b].weight * lins[1 + 2 * b].mask, lins[1 + 2 * b].bias)
        t_.retain_grad()
        hcur = hcur + torch.nn.functional.linear(torch.relu(t_), lins[2 + 2 * b].weight * lins[2 + 2 * b].mask, lins[2 + 2 * b].bias)
        hcur.retain_grad()
        nodes += [t_, hcur]
    fin = m64.final_layer
    out = torch.nn.functional.linear(hcur, fin.weight * fin.mask, fin.bias)
    vt = torch.from_numpy(v)
    gp = torch.zeros_like(prm)
    gp[:, 0::2] = ((vt * xq + gld[:, None] / scale) * sg * (1.0 - sg)).detach()
    gp[:, 1::2] = vt
    out.backward(gp.detach())
    unit_of_col = np.argsort(made.initial_layer.degrees.numpy(), kind="stable")
    for l, node in enumerate(nodes):                          # G[l] <-> virtual layer 2 NB - l
        got = S[2 * NB - l][:, cols[:H]]
        np.testing.assert_allclose(got, node.grad.numpy()[:, unit_of_col], rtol=1e-8, atol=1e-8, err_msg="hidden gradient %d" % l)


def test_flat_parameters_and_gradient_destinations(nfa):
    """dp.FlatParameters (round 6): parameters become views of one flat tensor without changing values or state_dict; a backward that
    writes into the registered destinations (_gradbuf.out, as nf_coupling_train_bwd's wrapper does) leaves p.grad AS the slice (no
    copy), any other gradient is copied in by sync(), a missing one zero-filled; Adam on the one flat tensor == Adam on the list."""
    import copy
    from normflows_amd import _gradbuf
    torch.manual_seed(3)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 4), torch.nn.Linear(4, 2))
    ref = copy.deepcopy(net)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    flat = nfa.dp.FlatParameters(net)
    assert all(torch.equal(net.state_dict()[k], v) for k, v in sd.items()) and list(net.state_dict()) == list(sd)
    params = list(net.parameters())
    assert all(p.data_ptr() == flat.param.data_ptr() + 4 * lo for p, (lo, _) in zip(params, flat.offsets))

    class WritesToDestination(torch.autograd.Function):        # stands in for a layer whose kernels take gradient destinations
        @staticmethod
        def forward(ctx, x, w, b):
            ctx.save_for_backward(x, w)
            ctx.b = b
            return x @ w.t() + b

        @staticmethod
        def backward(ctx, g):
            x, w = ctx.saved_tensors
            gw, gb = _gradbuf.out(w), _gradbuf.out(ctx.b)
            torch.matmul(g.t(), x, out=gw)
            torch.sum(g, 0, out=gb)
            return g @ w, gw, gb

    x = torch.randn(7, 6)
    opt = torch.optim.Adam(flat.parameters(), lr=1e-2)
    opt_ref = torch.optim.Adam(ref.parameters(), lr=1e-2)
    for step in range(3):
        flat.zero_grad()
        h = WritesToDestination.apply(x, net[0].weight, net[0].bias)        # layer 0: destinations; layers 2, 3: plain autograd
        out = net[2](torch.tanh(h)) if step else net[3](net[2](torch.tanh(h)))   # step > 0: layer 3 gets NO gradient
        out.pow(2).sum().backward()
        views = dict((id(p), v) for p, v in flat.views)
        assert net[0].weight.grad.data_ptr() == views[id(net[0].weight)].data_ptr()      # adopted without a copy
        assert net[0].bias.grad.data_ptr() == views[id(net[0].bias)].data_ptr()
        assert net[2].weight.grad.data_ptr() != views[id(net[2].weight)].data_ptr()
        n = flat.sync()
        assert n == 4 and flat.param.grad is flat.grad      # two copies + (two copies | two zero fills)
        ref.zero_grad(set_to_none=True)
        h = ref[0](x)
        out = ref[2](torch.tanh(h)) if step else ref[3](ref[2](torch.tanh(h)))
        out.pow(2).sum().backward()
        for (p, v), q in zip(flat.views, ref.parameters()):
            assert torch.allclose(v, q.grad if q.grad is not None else torch.zeros_like(q), atol=1e-6)
        if step:        # layer 3 must see a ZERO gradient in the flat buffer, like "no gradient" in the reference optimizer
            for q in ref[3].parameters():
                q.grad = torch.zeros_like(q)
        opt.step()
        opt_ref.step()
        assert all(torch.allclose(p, q, atol=1e-6) for p, q in zip(net.parameters(), ref.parameters()))
    flat.release()
    assert _gradbuf.target(net[0].weight) is None


def test_pack_keys_follow_flat_and_master_weight_optimizers(nfa):
    """_keys.py: cache keys change when an optimizer steps (a) the parameters themselves, (b) ONE flat tensor the
    parameters are views of -- recorded as an address range, other models' keys untouched --, (c) master copies whose relation to
    the parameters is invisible from the hook -- the process-wide epoch advances."""
    from normflows_amd import _keys
    m, other = torch.nn.Linear(4, 4), torch.nn.Linear(3, 3)
    k_other = _keys.pkey(other.parameters())
    k0 = _keys.pkey(m.parameters())
    for p in m.parameters():
        p.grad = torch.ones_like(p)
    torch.optim.SGD(m.parameters(), lr=0.1).step()
    k1 = _keys.pkey(m.parameters())
    assert k1 != k0 and _keys.pkey(other.parameters()) == k_other
    flat = nfa.dp.FlatParameters(m)
    k2 = _keys.pkey(m.parameters())
    flat.grad.fill_(1.0)
    e = _keys.epoch()
    torch.optim.SGD(flat.parameters(), lr=0.1).step()
    assert _keys.pkey(m.parameters()) != k2 and _keys.epoch() == e and _keys.pkey(other.parameters()) == k_other
    assert _keys.signature(m.parameters()) != _keys.signature(other.parameters())
    masters = [p.detach().clone().requires_grad_() for p in other.parameters()]
    for q in masters:
        q.grad = torch.ones_like(q)
    torch.optim.SGD(masters, lr=0.1).step()
    assert _keys.epoch() == e + 1 and _keys.pkey(other.parameters()) != k_other
    flat.release()
