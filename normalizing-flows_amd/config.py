"""Run-time switches of normflows_amd.

fused_gemm: how the fused NSF coupling-layer kernel evaluates its GEMMs
  "f32"    exact fp32 MFMA (v_mfma_f32_32x32x2_f32), the default;
  "bf16x3" error-compensated split-bf16 MFMA (csrc/rqs_fused_x3.hip): fp32 operands as hi+mid+lo bf16, six products
           accumulated in fp32 -- fp32-equivalent results (same parity tests), ~2x the throughput on gfx950.
"""
fused_gemm = "f32"


def set_fused_gemm(mode):
    global fused_gemm
    if mode not in ("f32", "bf16x3"):
        raise ValueError("fused_gemm must be 'f32' or 'bf16x3'")
    fused_gemm = mode


# Runs of GlowBlocks of one shape inside MultiscaleFlow go out as ONE persistent launch per level (nf_glow_level, Squeeze /
# Split / Merge folded in).  False: one launch per GlowBlock plus the glue kernels (ablation / debugging).
glow_level_chains = True


def set_glow_level_chains(mode=True):
    global glow_level_chains
    glow_level_chains = bool(mode)


def set_roctx(mode=True):
    """roctx ranges per layer type around every layer call of the containers (also NF_ROCTX=1); see _prof.py."""
    from . import _prof
    _prof.enabled = bool(mode)


# The two weight gradients of a residual block as ONE pair launch (nf_linear_wgrad_pair); False = two single launches (ablation).
wgrad_pair = True


def set_wgrad_pair(mode=True):
    global wgrad_pair
    wgrad_pair = bool(mode)


# Consecutive fused [CoupledRQS + LULinearPermute] pairs of one shape as ONE persistent launch (nf_rqs_fused_chain /
# nf_rqs_fused_x3_chain); False = one launch per pair (ablation, tests).
fused_chain = True


# Training forward of an eligible NSF coupling layer as ONE launch (autograd.CouplingTrainFn) instead of per-module Functions.
train_full = True


def set_train_full(mode=True):
    global train_full
    train_full = bool(mode)


# Backward of a residual block (and the initial layer behind the first one) as one pass over the rows (nf_resblock_bwd).
resblock_bwd = True


def set_resblock_bwd(mode=True):
    global resblock_bwd
    resblock_bwd = bool(mode)


# The coupling transform's backward + the final Linear's input gradient as one pass over the rows (nf_final_bwd); False = the
# stand-alone spline backward (nf_rqs_coupling_bwd_p24) + a library GEMM on the padded rows (round 2; ablation / differential tests).
final_bwd_fused = True


def set_final_bwd_fused(mode=True):
    global final_bwd_fused
    final_bwd_fused = bool(mode)


# The whole backward of a benchmark-shaped coupling layer behind one C-ABI call with ONE reduction launch for all of its partial
# tiles (nf_coupling_train_bwd, round 6); False = the kernels issued one by one from autograd.CouplingTrainFn.backward, each with
# its own reduction launch (rounds 3-5; ablation / differential tests: the gradients are bit-identical).
train_bwd_onecall = True


def set_train_bwd_onecall(mode=True):
    global train_bwd_onecall
    train_bwd_onecall = bool(mode)


# A benchmark-shaped [CoupledRQS, LULinearPermute] pair under autograd as ONE forward launch and the composed LU's one-product
# backward (autograd.PairTrainFn, round 6); False = LULinearPermuteFn + CouplingTrainFn (rounds 3-5; ablation / differential tests).
train_pair = True


def set_train_pair(mode=True):
    global train_pair
    train_pair = bool(mode)


# The pair backward's last two launches (the reduction of the partial tiles, the LU's factor gradients: parameter gradients only, 46 us
# of latency-bound work per pair) on a SIDE stream, under the next pair's MFMA-bound kernels (round 6, late; nf_pair_train_bwd_head /
# _tail, _sidestream.py).  Taken only when nothing can read those gradients before the join at the end of the backward pass: every
# parameter of the pair writes into a registered gradient buffer (dp.FlatParameters) whose .grad is unset (autograd adopts the view, no
# accumulation kernel) and carries no hooks other than join-aware ones (dp.OverlappedGradientAverager's).
# OFF by default -- measured (tools/train_bench.py --flat [--async], alternating on one box): without those two launches at all the step
# is 1.46 ms shorter (25.28 -> 23.82 ms), but on the side stream only 0.2-0.3 ms of that came back (25.25 -> 25.0), and nothing once the
# reduction launch issued its longest blocks first (25.1 either way): every heavy kernel of the step allocates the whole register file of
# a CU (2 waves per SIMD x 219-256 registers), so a second kernel gets no wave slot while one runs -- the side stream only fills the
# drain / ramp at kernel boundaries.
train_reduce_async = False


def set_train_reduce_async(mode=True):
    global train_reduce_async
    train_reduce_async = bool(mode)


# Glow's training step (BASELINE configs[3]): the launches of a GlowBlock's backward that only produce PARAMETER gradients -- the conditioner's
# weight gradients (made_wgrad + its reduction + the last bias sum), the 1x1 convolution's weight gradient and its LU factors' -- on the
# side stream (_sidestream.fork), joined at the end of the backward pass.  Unlike the benchmark model's pair (above) the Glow step is
# made of kernels that do not fill the chip: the 8x8 / 4x4 levels give the row-tile kernels 256 / 64 workgroups, the LU-factor and
# reduction kernels are one-workgroup launches.  Taken per Function only when every parameter it feeds has no .grad to accumulate into
# and no hooks (autograd adopts the tensor without a kernel); anything else joins first.
# OFF by default -- measured (tools/glow_leaf_ab.py, alternating on one box, config 4): the step recorded as one hipGraph 44.57 / 44.40 ms
# without, 43.98 / 44.13 ms with (same gradient bits): 12.5 ms of leaf launches per step move to a parallel branch of the graph and 0.4 ms
# comes back -- the replayed graph does not run the branches side by side to any useful degree.
train_leaf_async = False


def set_train_leaf_async(mode=True):
    global train_leaf_async
    train_leaf_async = bool(mode)


# Glow's training step: the Invertible1x1Convs of a level assemble their matrices (and, in the backward, their LU factors' gradients) in
# ONE launch per level (autograd.Inv1x1WeightsFn) instead of one single-workgroup launch per block and direction.  False = per block.
glow_weights_batched = True


def set_glow_weights_batched(mode=True):
    global glow_weights_batched
    glow_weights_batched = bool(mode)


# The 256-slot training kernels (nf_made_forward_train / nf_made_backward: GlowBlock's conv conditioner at the 16x16 level, ResidualNets /
# MADEs of hidden width <= 256 on <= 64 features) on 128-row tiles at batches that are a multiple of 128 rows where the persistent workgroups' rounds come out shorter (csrc/mlp_tile.hpp
# mf_tr128: a weight fragment feeds eight MFMAs instead of four; same bits).  Lives in the library (nf_config_made_tr128).
def set_made_tr128(mode=True):
    """Returns the previous setting.  A forward and its backward must run under the same setting."""
    from . import _lib
    return bool(_lib.lib().nf_config_made_tr128(1 if mode else 0))


# MultiscaleFlow under autograd: the per-layer `log_q += log_det` statements of a level are collected and applied as one launch in the same
# order (flows/affine.lazy_ld, nf_ld_fold_multi: same bits).  False = one launch per layer.
lazy_logdet = True


def set_lazy_logdet(mode=True):
    global lazy_logdet
    lazy_logdet = bool(mode)


# A differentiable density pass of a benchmark-shaped model on a batch that is NOT a multiple of 64 rows (>= 1024) is run on the batch
# padded with zero rows to the next multiple (NormalizingFlow._log_prob_impl) and sliced back: the one-call / pair training kernels need
# whole 64-row tiles, and the slice's backward hands the padding rows a zero cotangent, so they contribute exactly nothing to any
# gradient.  65 537 rows: 44.4 -> 34.4 ms per step of the benchmark model (what 65 600 rows cost).  False = the general kernels.
train_pad_batch = True


def set_train_pad_batch(mode=True):
    global train_pad_batch
    train_pad_batch = bool(mode)


def set_fused_small_batch(mode=True):
    """nf_rqs_fused_chain on 128-row workgroups for batches of at most 32 768 rows (the default; csrc/rqs_fused_nw4.hip: 5.4 -> 3.0 ms
    per pass of the benchmark chain).  False = every batch on the 256-row workgroups (differential tests).  Returns the old setting."""
    from . import _lib as L
    return bool(L.lib().nf_rqs_fused_small_batch(1 if mode else 0))


# LULinearPermute's density-direction backward (D = 64) as one pass over the rows (nf_lu_bwd).
lu_bwd_fused = True


def set_lu_bwd_fused(mode=True):
    global lu_bwd_fused
    lu_bwd_fused = bool(mode)


# Training: every eligible layer's weights packed by ONE launch per kind at the start of a differentiable density pass (_prepack.py).
train_prepack = True


def set_train_prepack(mode=True):
    global train_prepack
    train_prepack = bool(mode)


# MAF inverse (nf_maf_inverse): the mapping with 32 samples per wave and the lane-halves sharing a sample's hidden units (two waves
# per SIMD, csrc/maf_inverse_h.hip); False = round 2's 64-samples-per-wave kernel (csrc/maf_inverse.hip; ablation).
maf_halves = True


def set_maf_halves(mode=True):
    global maf_halves
    maf_halves = bool(mode)


# MAF inverse, round 5: format-1 packs (flows/maf_pack.pack_made(tri=True)) -- regular tiles (<= 8 degrees of <= 4 hidden units) run
# nf_maf_inverse_h_tri's statically unrolled triangular sequential part and 8-deep activation ring; False = format 0 on
# nf_maf_inverse_h (round 3/4; ablation).  Only with maf_halves.
maf_tri = True


def set_maf_tri(mode=True):
    global maf_tri
    maf_tri = bool(mode)


def set_fused_chain(mode=True):
    global fused_chain
    fused_chain = bool(mode)


# LULinearPermute under autograd: its two batch-side products per direction as ONE launch (nf_rows_matvec2); False = two
# nf_rows_matvec launches (ablation).
lu_matvec2 = True


def set_lu_matvec2(mode=True):
    global lu_matvec2
    lu_matvec2 = bool(mode)


# MADE's single pass as ONE launch (nf_made_forward / nf_made_forward_affine, csrc/made_fwd.hip) where flows/made_pack.py takes the
# structure; False = layer by layer (library GEMMs on the pre-masked weights; ablation / differential tests).
made_fused = True


def set_made_fused(mode=True):
    global made_fused
    made_fused = bool(mode)


# MADE under autograd (core.py:87-102 through an autoregressive flow's single-pass direction): forward nf_made_forward_train, backward
# nf_made_backward + nf_made_wgrad (csrc/made_bwd.hip); False = torch autograd through library GEMMs on the pre-masked weights.
made_train = True


def set_made_train(mode=True):
    global made_train
    made_train = bool(mode)


# MaskedAffineAutoregressive.inverse (the reference's DENSITY direction of MAF: D sequential MADE passes, autoregressive.py:29-38) under
# autograd by implicit differentiation (autograd.MafInverseFn): the one-pass inverse kernel forward, the triangular system of the
# backward solved with the MADE input-gradient chain, ONE weight-gradient launch; False = torch autograd through the D-pass loop.
maf_implicit = True
# stop the sweeps once max |v_new - v| <= rtol * max |v| (0.0 = until v stops changing bit for bit: the exact solution of the triangular
# system; 1e-6 saves the last ~15 % of the sweeps at float32-noise-level gradient changes)
maf_implicit_rtol = 0.0


# Autoregressive.inverse under autograd for the other element-wise transforms (AR-NSF sampling, circular splines, MAF structures
# outside the one-pass kernels): implicit differentiation on the layer's own density-direction graph (autograd.ArInverseImplicitFn:
# <= D backward sweeps of the net + one weight-gradient pass).  False = the reference's D recorded passes.
# Restriction: the implicit Functions are once_differentiable -- FIRST-order gradients only.  Double backward through
# an autoregressive layer's inverse (create_graph=True: gradient penalties, Hessian-vector products through sampling) needs the D
# recorded passes, which support it like the reference's loop: `with normflows_amd.config.higher_order_gradients():` (or
# set_ar_implicit(False) / set_maf_implicit(False)) around the forward pass.
ar_implicit = True
# MaskedAffineAutoregressive's element-wise map under autograd as torch formulas (affine/autoregressive.py:98-128 verbatim in torch
# ops: differentiable to any order) instead of nf_maf_affine + nf_maf_affine_bwd; set by higher_order_gradients().
torch_elementwise = False


class higher_order_gradients:
    """Context manager: inside it Autoregressive.inverse / MaskedAffineAutoregressive.inverse run the reference's D recorded passes
    under autograd instead of the once-differentiable implicit Functions, and MADE runs as plain torch modules: as differentiable
    as the element-wise transform's own backward allows (the affine transform of MAF: any order)."""

    def __enter__(self):
        global ar_implicit, maf_implicit, made_train, torch_elementwise
        self.saved = (ar_implicit, maf_implicit, made_train, torch_elementwise)
        # (MADE itself then runs as torch modules on library GEMMs and MAF's element-wise affine map as torch formulas: the
        # hand-written MadeFn / MafAffineFn backwards are sets of kernels, first-order as well)
        ar_implicit = maf_implicit = made_train = False
        torch_elementwise = True
        return self

    def __exit__(self, *exc):
        global ar_implicit, maf_implicit, made_train, torch_elementwise
        ar_implicit, maf_implicit, made_train, torch_elementwise = self.saved
        return False


def set_ar_implicit(mode=True):
    global ar_implicit
    ar_implicit = bool(mode)


def set_maf_implicit(mode=True, rtol=None):
    global maf_implicit, maf_implicit_rtol
    maf_implicit = bool(mode)
    if rtol is not None:
        maf_implicit_rtol = float(rtol)


# CoupledRationalQuadraticSpline beyond the benchmark kernel's shapes (D <= 128, hidden <= 512, 8 bins) as ONE launch (nf_nsf_wide,
# csrc/nsf_wide.hip); False = library GEMMs for the conditioner + nf_rqs_coupling (ablation / differential tests).
nsf_wide = True


def set_nsf_wide(mode=True):
    global nsf_wide
    nsf_wide = bool(mode)


# Debug mode of the element-wise spline API (utils.splines): after every call a check launch (nf_rqs_spline_check)
# sets device-side flags and the shim reads them back -- a host synchronisation per call, which is why it is off by default -- and
# raises what the reference raises: AssertionError for a negative discriminant in the inverse direction (utils/splines.py:181),
# RuntimeError for tails=None inputs outside the domain (the reference's gather index error, utils/splines.py:154-160; without
# debug mode the kernels clamp the bin).
debug_checks = False


def set_debug_checks(mode=True):
    global debug_checks
    debug_checks = bool(mode)


# The implicit backward of the MAF inverse (autograd.MafInverseFn): True = ONE pass of nf_maf_solve_t per layer (round 5: back-
# substitution on the transposed pack, no host read-back); False = round 4's sweeps of nf_made_backward until v stops changing
# (15-25 passes per layer; ablation / cross-check).
maf_onepass = True


def set_maf_onepass(mode=True):
    global maf_onepass
    maf_onepass = bool(mode)


# The weight-gradient launch of the one-pass implicit backward reads MADE's hidden gradients from the SOLVE's scratch (nf_maf_scratch_rows:
# the solve finalises every unit of the transposed network once from final values = the input-gradient chain at the solution) instead of
# running nf_made_backward once more.  False = the extra chain pass.
maf_solve_grads = True


def set_maf_solve_grads(mode=True):
    global maf_solve_grads
    maf_solve_grads = bool(mode)


# Round 6 (late): the one-pass solve runs the regular-8 tiles of a format-1 transposed pack on a statically unrolled sequential part
# (nf_maf_solve_t_tri; bit-identical to the generic part).  False = every tile on the generic part.
maf_solve_fast = True


def set_maf_solve_fast(mode=True):
    global maf_solve_fast
    maf_solve_fast = bool(mode)


# Round 6: that weight-gradient launch reads BOTH scratches where the one-pass kernels left them (nf_made_wgrad_pos: problems, tiles and
# scatter maps over scratch positions; batches that are a multiple of 64 rows, position counts that are a multiple of 128) instead of two
# nf_maf_scratch_rows rearrangements per layer (2 x 245 us and 2.7 GB of traffic per config-5 layer at B = 65 536).  False = rearrange.
maf_wgrad_in_place = True


def set_maf_wgrad_in_place(mode=True):
    global maf_wgrad_in_place
    maf_wgrad_in_place = bool(mode)
