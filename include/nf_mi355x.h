/*
 * nf_mi355x.h -- C ABI of libnf_mi355x.so: the MI355X (gfx950 / CDNA4) implementation of the
 * coupling-layer forward/inverse + log|det J| hot path of normflows 1.7.3.
 *
 * Every entry point replaces one reference function (cited as file:line relative to the normflows
 * repository root).  The library is the drop-in boundary: plain pointers and sizes only, no torch
 * types.  A Python binding (ctypes) lives in normalizing-flows_amd/_lib.py; the stub a normflows
 * maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - All data pointers are DEVICE pointers (HBM) owned by the caller.  Kernels never allocate,
 *     free or retain pointers.  Outputs must not alias inputs unless a function says so.
 *   - `stream` is a hipStream_t passed as void*.  Calls only enqueue work on it and return; no host
 *     synchronisation, no internal streams.  Safe to capture into a hipGraph.
 *   - `dtype`: NF_F32 or NF_F64 selects the element type of every floating-point buffer of the call.
 *   - Return value: NF_OK (0) or a negative errno-style code; nf_strerror() explains it.
 *   - `acc` (log-det accumulation mode): NF_LD_WRITE  logdet[b]  = ld_b
 *                                        NF_LD_ADD    logdet[b] += ld_b   (core.py:193-195 `log_q += log_det`)
 *                                        NF_LD_SUB    logdet[b] -= ld_b   (core.py:177-179 `log_q -= log_det`)
 *   - Direction naming follows normflows: "forward" is generative (z -> x), "inverse" is
 *     normalising (x -> z); log_prob runs every layer's inverse, sample runs every layer's forward.
 */
#ifndef NF_MI355X_H
#define NF_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *nf_stream_t; /* hipStream_t */

enum { NF_OK = 0, NF_EIO = -5, NF_EFAULT = -14, NF_EINVAL = -22, NF_ERANGE = -34, NF_ENOTSUP = -95 };
enum { NF_F32 = 0, NF_F64 = 1 };
enum { NF_LD_WRITE = 0, NF_LD_ADD = 1, NF_LD_SUB = -1 };
enum { NF_TAILS_NONE = 0, NF_TAILS_LINEAR = 1, NF_TAILS_CIRCULAR = 2, NF_TAILS_FEATURE = 3 /* nf_rqs_coupling_ft only */ };
enum { NF_SCALE_EXP = 0, NF_SCALE_SIGMOID = 1, NF_SCALE_SIGMOID_INV = 2, NF_SCALE_NONE = 3 };
enum { NF_RQS_DENSITY = 0, NF_RQS_SAMPLE_IDENTITY = 1, NF_RQS_SAMPLE_TRANSFORM = 2 };

/* Library identification / diagnostics. */
const char *nf_version(void);
const char *nf_strerror(int code);
/* Measurement aid (bench.py): shader clock under fp32-MFMA load.  A chip-filling launch issues iters x 8 independent
 * v_mfma_f32_32x32x2_f32 per wave; out2_u64[0] = shader cycles, [1] = ticks of the 100 MHz wall clock around them. */
int nf_mfma_clock_probe(void *out2_u64, void *sink_f32, int iters, nf_stream_t stream);
/* Largest number of spline bins the compiled kernels accept. */
int nf_max_bins(void);

/* ------------------------------------------------------------------------------------------------
 * Rational-quadratic spline, element-wise.
 * Replaces normflows/utils/splines.py:16-97 (unconstrained_rational_quadratic_spline, tails =
 * linear/circular) and :100-219 (rational_quadratic_spline, tails = NF_TAILS_NONE).
 *
 * N elements; element n reads K unnormalised widths at w + n*ldw, K heights at h + n*ldh and
 * (K-1 | K | K+1) unnormalised derivatives (linear | circular | none) at d + n*ldd (strides in
 * elements, so the three blocks may be slices of one (N, 3K-1) conditioner output).
 * Widths and heights are divided by `wh_div` first (nsf/coupling.py:334-339; pass 1.0 for none).
 * tails != NONE: the spline acts on [-tail_bound, tail_bound]^2, elements outside (and NaN/inf)
 * pass through with logabsdet 0.  tails == NONE: the spline maps [left,right] -> [bottom,top]; the
 * reference raises on out-of-domain inputs (torch.gather index error), here the bin index is clamped.
 * inverse = 0: y = spline(x); inverse = 1: y = spline^{-1}(x).  logabsdet may be NULL.
 */
int nf_rqs_spline(const void *x, const void *w, int64_t ldw, const void *h, int64_t ldh, const void *d,
                  int64_t ldd, void *y, void *logabsdet, int64_t N, int K, int tails, double tail_bound,
                  double left, double right, double bottom, double top, double min_bin_width,
                  double min_bin_height, double min_derivative, double wh_div, int inverse, int dtype,
                  nf_stream_t stream);
/* Debug mode of the element-wise spline: the reference's run-time failures on this path as device-side flags, checked
 * by a SEPARATE launch over one call's inputs x and outputs y -- the transform kernels stay branch-free and never synchronise.
 * ORs into *flags (a device uint32 the caller zeroed): bit 0 = an input outside the domain with tails = NF_TAILS_NONE (the
 * reference's gathers fail on bin index -1 / K there, utils/splines.py:154-160; the kernels clamp the bin), bit 1 = inverse
 * direction, an in-domain input whose output is NaN, i.e. the square root of a negative discriminant (`assert (discriminant >= 0)`,
 * utils/splines.py:181).  The shim launches it and reads the word back only under normflows_amd.config.set_debug_checks(True). */
int nf_rqs_spline_check(const void *x, const void *y, int64_t N, int tails, double tail_bound, double left, double right,
                        double bottom, double top, int inverse, int dtype, void *flags, nf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * NSF coupling transform on (B, D) rows with conditioner outputs materialised in HBM.
 * Replaces normflows/flows/neural_spline/coupling.py:71-128 (Coupling.forward/inverse: index split,
 * coupling transform, unconditional transform of the identity half, scatter), :150-164
 * (PiecewiseCoupling._coupling_transform, row-sum of logabsdet), :221-253
 * (PiecewiseRationalQuadraticCDF._spline, batch-shared parameters), :329-362
 * (PiecewiseRationalQuadraticCoupling._piecewise_cdf) for 2-D inputs.
 *
 *   x, y          (B, D) row-major.  y receives every column the mode owns (see below).
 *   identity_idx  (nI) int64, transform_idx (nT) int64: the `identity_features` /
 *                 `transform_features` buffers (coupling.py:42-47).
 *   cond          (B, nT, M) conditioner output, M = 3K-1 | 3K | 3K+1 (linear | circular | none);
 *                 row layout [w_0..w_{K-1} | h_0..h_{K-1} | d...].  Unused (may be NULL) in
 *                 NF_RQS_SAMPLE_IDENTITY.
 *   uw, uh, ud    unconditional transform parameters (nI,K), (nI,K), (nI,M-2K), or all NULL when the
 *                 layer has no unconditional transform (identity columns are then copied).
 *   mode  NF_RQS_DENSITY           prqct.forward (coupling.py:71-98):  y[:,T] = spline(x[:,T]; cond),
 *                                   y[:,I] = cdf(x[:,I]); ld = sum of both.
 *         NF_RQS_SAMPLE_IDENTITY   first half of prqct.inverse (coupling.py:110-116):
 *                                   y[:,I] = cdf^{-1}(x[:,I]); ld = its row sum; y[:,T] untouched.
 *         NF_RQS_SAMPLE_TRANSFORM  second half (coupling.py:118-128):
 *                                   y[:,T] = spline^{-1}(x[:,T]; cond); ld = its row sum; y[:,I] untouched.
 *   logdet (B): combined with ld according to `acc`.
 */
int nf_rqs_coupling(const void *x, void *y, void *logdet, const void *cond, const void *uw, const void *uh,
                    const void *ud, const int64_t *identity_idx, int nI, const int64_t *transform_idx,
                    int nT, int64_t B, int D, int K, int tails, double tail_bound, double min_bin_width,
                    double min_bin_height, double min_derivative, double wh_div, int mode, int acc,
                    int dtype, nf_stream_t stream);

/* Same transform with tails and / or tail bounds given PER FEATURE (utils/splines.py:48-57 list of tails, :61-66 tensor
 * tail_bound; the circular-coordinate layers of neural_spline/wrapper.py:88-185, 247-330 and coupling.py:283-318):
 *   tails = NF_TAILS_FEATURE: tails_t (nT) / tails_i (nI) int32 hold NF_TAILS_LINEAR or NF_TAILS_CIRCULAR for every
 *     transform / identity feature (in the order of transform_idx / identity_idx); rows carry K+1 derivative logits
 *     (M = 3K+1) whose edge entries are overwritten per type; as in the reference's list branch, inputs outside
 *     their interval produce OUTPUT 0 and log-det 0 (that branch never copies them through).
 *   bound_t (nT) / bound_i (nI): per-feature tail bounds of the data dtype, or NULL for the scalar tail_bound; usable
 *     with any `tails` other than NF_TAILS_NONE.
 * With all four arrays NULL this is nf_rqs_coupling.
 */
int nf_rqs_coupling_ft(const void *x, void *y, void *logdet, const void *cond, const void *uw, const void *uh,
                    const void *ud, const int64_t *identity_idx, int nI, const int64_t *transform_idx,
                    int nT, int64_t B, int D, int K, int tails, double tail_bound, double min_bin_width,
                    double min_bin_height, double min_derivative, double wh_div, int mode, int acc,
                    int dtype, const int32_t *tails_t, const void *bound_t,
                       const int32_t *tails_i, const void *bound_i, nf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Backward (vector-Jacobian product) of nf_rqs_coupling, for training.  The reference differentiates
 * normflows/utils/splines.py:16-219 and nsf/coupling.py:71-128 with PyTorch autograd (training step:
 * core.py:87-102 + loss.backward()); there is no hand-written backward to cite.
 *   grad_y (B, D), grad_logdet (B): upstream gradients of the forward call's outputs (same mode).
 *   grad_x (B, D): written for every column the mode owns.  grad_cond (B, nT, M): per-element
 *   gradients of the conditioner output.  grad_uw/uh/ud: gradients of the batch-shared parameters,
 *   ACCUMULATED (atomics) -- the caller zero-initialises them.
 */
int nf_rqs_coupling_bwd(const void *x, const void *grad_y, const void *grad_logdet, const void *cond,
                        const void *uw, const void *uh, const void *ud, const int64_t *identity_idx, int nI,
                        const int64_t *transform_idx, int nT, int64_t B, int D, int K, int tails,
                        double tail_bound, double min_bin_width, double min_bin_height, double min_derivative,
                        double wh_div, int mode, void *grad_x, void *grad_cond, void *grad_uw, void *grad_uh,
                        void *grad_ud, int dtype, nf_stream_t stream);

/* Backward of nf_rqs_coupling_ft: the same arrays tails_t / bound_t / tails_i / bound_i as the forward call.  In the
 * per-feature branch (tails = NF_TAILS_FEATURE) an input outside its interval produced the constant 0, so its grad_x
 * is 0 (utils/splines.py:48-57 never copies those inputs); the edge derivative logits a feature's tails type
 * overrides receive no gradient (linear) or the gradient of the logit they alias (circular: logit K -> logit 0). */
int nf_rqs_coupling_bwd_ft(const void *x, const void *grad_y, const void *grad_logdet, const void *cond,
                           const void *uw, const void *uh, const void *ud, const int64_t *identity_idx, int nI,
                           const int64_t *transform_idx, int nT, int64_t B, int D, int K, int tails,
                           double tail_bound, double min_bin_width, double min_bin_height, double min_derivative,
                           double wh_div, int mode, void *grad_x, void *grad_cond, void *grad_uw, void *grad_uh,
                           void *grad_ud, int dtype, const int32_t *tails_t, const void *bound_t,
                           const int32_t *tails_i, const void *bound_i, nf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Fully fused NSF coupling layer: ResidualNet conditioner (fp32 MFMA) + spline epilogue, one launch.
 * Replaces the whole of CoupledRationalQuadraticSpline.forward/inverse
 * (normflows/flows/neural_spline/wrapper.py:79-85) for 2-D inputs without context:
 * nets/resnet.py:92-104 (ResidualNet.forward, ReLU, no batch-norm, dropout 0) +
 * nsf/coupling.py:71-128 + utils/splines.py:16-219, tails = linear.
 *
 * Supported shape (anything else returns NF_ENOTSUP and the caller uses nf_rqs_coupling): D = 64 with
 * the alternating mask of wrapper.py:69 (nI = nT = 32), hidden = 128, K = 4 | 8 | 16 (one kernel instantiation each), fp32.
 * nf_rqs_fused / nf_rqs_fused_chain also take hidden = 64 or 32: the blob is still the 128-unit layout (packed with hidden = 128
 * from a conditioner zero-padded by the caller) whose units >= hidden are all-zero; the kernel skips their row-blocks and k-groups.
 * Likewise D = 16 | 32 | 48: rows are still 64 floats wide, columns >= D are the caller's padding (values outside the splines'
 * interval, identity block in the packed LU): the final-layer groups of the all-padding 16-column chunks are skipped.
 *   mask_parity 0: reverse_mask = False (identity = even columns, transform = odd), 1: the opposite.
 * `wpack` is the layer's weights re-laid-out in MFMA operand order by nf_rqs_fused_pack() (device
 * buffer of nf_rqs_fused_pack_size() bytes, 16-byte aligned); weights are torch nn.Linear layout
 * (out, in); w_blocks/b_blocks are HOST arrays of 2*num_blocks DEVICE pointers
 * [blk0.linear0, blk0.linear1, blk1.linear0, ...].  direction 0 = density (wrapper.inverse),
 * 1 = sample (wrapper.forward).
 */
/* Row order of the packed final layer (host-side, for tests and tools): the row of the reference's (32 (3 K - 1), hidden)
 * final-layer weight that MFMA row rho (0..31) of row-block rb (0..2) of group g (0 .. K - 1) holds, or -1 for a padding row. */
int nf_rqs_fused_final_row(int K, int g, int rb, int rho);
int64_t nf_rqs_fused_pack_size(int nI, int nT, int hidden, int num_blocks, int K);
int nf_rqs_fused_pack(void *wpack, const void *w_init, const void *b_init, const void *const *w_blocks,
                      const void *const *b_blocks, const void *w_final, const void *b_final, const void *uw,
                      const void *uh, const void *ud, int nI, int nT, int hidden, int num_blocks, int K,
                      double tail_bound, double min_bin_width, double min_bin_height, double min_derivative,
                      nf_stream_t stream);
/* nf_rqs_fused_pack_lu adds the layer's LULinearPermute (normflows/flows/mixing.py:535-563) to the blob as one
 * dense 64 x 64 matrix per direction (L U P, resp. P^T U^-1 L^-1 and -W b, composed in fp64, rounded to fp32).
 * With fuse_lu = 1, nf_rqs_fused also applies it: density = LULinearPermute.inverse BEFORE the coupling
 * (core.py:193-195 runs flows in reverse), sample = LULinearPermute.forward AFTER it; its constant log|det| is
 * folded into the per-sample log-det. */
int nf_rqs_fused_pack_lu(void *wpack, int num_blocks, const int64_t *perm, const void *lower_entries,
                         const void *upper_entries, const void *unconstrained_upper_diag, const void *bias, int D,
                         double eps, int K, nf_stream_t stream);
int nf_rqs_fused(const void *x, void *y, void *logdet, const void *wpack, int mask_parity, int fuse_lu, int64_t B,
                 int D, int hidden, int num_blocks, int K, double tail_bound, double min_bin_width,
                 double min_bin_height, double min_derivative, int direction, int acc, nf_stream_t stream);
/* A chain of num_layers (<= 64) fused layers of identical shape in ONE persistent launch: replaces num_layers
 * iterations of the container loops normflows/core.py:177-179 / :193-195.  wpacks / mask_parities are HOST arrays in
 * PROCESSING order (for log_prob: last flow first).  The rows stay on chip between layers (x read once, y written once,
 * the accumulated log-det written once). */
int nf_rqs_fused_chain(const void *x, void *y, void *logdet, const void *const *wpacks, const int *mask_parities,
                       int num_layers, int fuse_lu, int64_t B, int D, int hidden, int num_blocks, int K,
                       double tail_bound, double min_bin_width, double min_bin_height, double min_derivative,
                       int direction, int acc, nf_stream_t stream);
/* nf_rqs_fused_chain runs batches of at most 32 768 rows on 4-wave (128-row) workgroups (csrc/rqs_fused_nw4.hip: the same kernel built a
 * second time) -- the 8-wave workgroups own 256 rows for the whole chain, so at <= 32 768 rows half of the CUs have no workgroup; same
 * bits per row.  nf_rqs_fused_small_batch(0) keeps every batch on the 8-wave kernel, (1) restores the default, (-1) only queries; returns
 * the previous setting.  (No counterpart in the reference: a tuning switch of this library, used by the differential tests.) */
int nf_rqs_fused_small_batch(int enable);

/* Training forward of the layer's last stage (core.py:87-102 `forward_kld` over nsf/coupling.py:83-98): final Linear of the
 * conditioner (nets/resnet.py:104) + the density-direction coupling transform in ONE launch of the fused kernel, the hidden
 * activations h2 (B, 128) -- the output of the residual blocks, computed by the autograd-tracked trunk -- read from HBM.
 * x, y (B, 64); logdet (B) per `acc`; cond_out (B, 32, 24): the conditioner output kept for the backward, 23 parameters + 1
 * pad per transform feature, raw scale (nf_rqs_coupling_bwd_p24 reads this layout).  wpack: nf_rqs_fused_pack, or
 * nf_rqs_fused_pack_final (header + final-layer stages + knot tables only: what this entry point reads).
 * Shape: D = 64, hidden = 128, K = 8, linear tails (NF_ENOTSUP otherwise). */
int nf_rqs_fused_pack_final(void *wpack, const void *w_final, const void *b_final, const void *uw, const void *uh,
                            const void *ud, int hidden, int num_blocks, int K, double tail_bound, double min_bin_width,
                            double min_bin_height, double min_derivative, nf_stream_t stream);
int nf_rqs_fused_train_fwd(const void *x, const void *h2, void *y, void *logdet, void *cond_out, const void *wpack,
                           int mask_parity, int64_t B, int D, int hidden, int num_blocks, int K, double tail_bound,
                           double min_bin_width, double min_bin_height, double min_derivative, int acc, nf_stream_t stream);
/* Training forward of the WHOLE layer in one launch (the inference kernel + what the backward needs): act_out
 * (2 num_blocks + 1, B, 128) = h0 (initial layer's output), then per residual block its pre-activation t and its output h
 * (nets/resnet.py:37-50, :92-104); cond_out (B, 32, 24) as above.  wpack: nf_rqs_fused_pack_all (one launch; same layout as
 * nf_rqs_fused_pack without the LU).  D = 64, hidden = 128, K = 8, linear tails.  wfull / wpad / identity_idx (all or none): the
 * same launch also leaves the initial weight transposed on full rows (64, hidden; the identity features' rows written, the caller
 * keeps the rest zero) and, in `wpad` (32 x 24 x hidden floats), the final weight as the 24 A-operand stages nf_final_bwd streams
 * (w_t: stage 3 g + rb = [k-step vv (8)][unit-block quad uq (2)][lane (64)][4]: lane (m = lane & 15, hq = lane >> 4) holds
 * W_final[(8 (g >> 1) + 4 (hq >> 1) + 2 (g & 1) + (hq & 1)) * 23 + 8 rb + vv][16 (4 uq + j) + m], raw scale, zero for the pad
 * step 8 rb + vv = 23: the A operand of v_mfma_f32_16x16x4_f32 whose four k-entries are the group's four transform features). */
int nf_rqs_fused_pack_all(void *wpack, const void *w_init, const void *b_init, const void *const *w_blocks,
                          const void *const *b_blocks, const void *w_final, const void *b_final, const void *uw, const void *uh,
                          const void *ud, int hidden, int num_blocks, int K, double tail_bound, double min_bin_width,
                          double min_bin_height, double min_derivative, void *wfull, void *wpad, const void *identity_idx,
                          nf_stream_t stream);
int nf_rqs_fused_train_full_fwd(const void *x, void *y, void *logdet, void *cond_out, void *act_out, const void *wpack,
                                int mask_parity, int64_t B, int D, int hidden, int num_blocks, int K, double tail_bound,
                                double min_bin_width, double min_bin_height, double min_derivative, int acc, nf_stream_t stream);
/* nf_rqs_fused_pack_all for n_layers layers of one shape in ONE launch (a training step re-packs every layer once).  table
 * (device memory): n_layers rows of 11 + 4 num_blocks pointers -- wpack, w_init, b_init, w_final, b_final, uw, uh, ud, wfull, wpad,
 * identity_idx (the last three NULL together or not at all), the 2 num_blocks hidden weights, the 2 num_blocks hidden biases. */
int nf_rqs_fused_pack_all_multi(const void *table, int n_layers, int hidden, int num_blocks, int K, double tail_bound,
                                double min_bin_width, double min_bin_height, double min_derivative, nf_stream_t stream);
/* Backward of the density-direction coupling transform (nf_rqs_coupling_bwd, mode NF_RQS_DENSITY) on cond / grad_cond rows
 * of 24 floats per transform feature (the layout above; 16-byte aligned).  float32, 8 bins, linear tails. */
int nf_rqs_coupling_bwd_p24(const void *x, const void *grad_y, const void *grad_logdet, const void *cond24, const void *uw,
                            const void *uh, const void *ud, const int64_t *identity_idx, int nI, const int64_t *transform_idx,
                            int nT, int64_t B, int D, double tail_bound, double min_bin_width, double min_bin_height,
                            double min_derivative, double wh_div, void *grad_x, void *grad_cond24, void *grad_uw,
                            void *grad_uh, void *grad_ud, nf_stream_t stream);

/* Backward of the layer's LAST stage in one pass over the rows (final_bwd.hip): nf_rqs_coupling_bwd_p24 AND the input gradient of
 * the conditioner's final Linear (nets/resnet.py:104 under core.py:87-102 `loss.backward()`), grad_h = grad_cond W_final, with the
 * gradient rows fed to the fp32 MFMAs from the registers they were computed in (no library GEMM, no re-read of the 201 MB of
 * rows; they are written once, to grad_cond24, for the final layer's weight gradient).  x, grad_y, grad_x (B, 64); grad_logdet (B);
 * cond24, grad_cond24 (B, 32, 24) (may alias); grad_h (B, 128); w_t: the `wpad` by-product of nf_rqs_fused_pack_all[_multi];
 * wpack: the layer's blob (its knot tables of the batch-shared spline are read).  The batch-shared parameters' gradients leave
 * as knot-space partial sums, one row of 768 floats per workgroup: partials (nf_final_bwd_partials(B) x 768 floats, written,
 * not accumulated); nf_final_bwd_reduce adds them in a fixed order (bit-reproducible, no atomics) and applies the softmax /
 * cumulative-sum / softplus chain once: grad_uw, grad_uh (32, 8), grad_ud (32, 7), written.  D = 64, hidden = 128, K = 8, linear
 * tails, float32 (NF_ENOTSUP otherwise); any B. */
int nf_final_bwd_partials(int64_t B);
int nf_final_bwd(const void *x, const void *grad_y, const void *grad_logdet, const void *cond24, const void *w_t, const void *wpack,
                 void *grad_x, void *grad_cond24, void *grad_h, void *partials, int mask_parity, int64_t B, int D, int hidden,
                 int num_blocks, int K, double tail_bound, double min_bin_width, double min_bin_height, double min_derivative,
                 nf_stream_t stream);
int nf_final_bwd_reduce(const void *partials, int n_partials, const void *uw, const void *uh, const void *ud, void *grad_uw,
                        void *grad_uh, void *grad_ud, int K, double tail_bound, double min_bin_width, double min_bin_height,
                        double min_derivative, nf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * The same fused layer with the GEMMs on the bf16 matrix pipe by error-compensated splitting
 * (fp32 operand = hi + mid + lo bf16, six products accumulated in fp32; csrc/rqs_fused_x3.hip).
 * Results are fp32-equivalent (same parity tests and tolerances as nf_rqs_fused).  The x3 blob is
 * derived from an nf_rqs_fused_pack[_lu] blob of the same layer.
 */
int64_t nf_rqs_fused_x3_pack_size(int nI, int nT, int hidden, int num_blocks, int K);
int nf_rqs_fused_x3_pack(void *x3pack, const void *f32pack, int num_blocks, int has_lu, nf_stream_t stream);
int nf_rqs_fused_x3(const void *x, void *y, void *logdet, const void *x3pack, int mask_parity, int fuse_lu,
                    int64_t B, int D, int hidden, int num_blocks, int K, double tail_bound, double min_bin_width,
                    double min_bin_height, double min_derivative, int direction, int acc, nf_stream_t stream);
/* The split-bf16 counterpart of nf_rqs_fused_chain: up to 64 layers of one shape in ONE persistent launch (rows stay in
 * LDS between the layers, the weight stream runs through the layer boundaries).  x3packs (nf_rqs_fused_x3_pack) and
 * mask_parities in PROCESSING order. */
int nf_rqs_fused_x3_chain(const void *x, void *y, void *logdet, const void *const *x3packs, const int *mask_parities,
                          int num_layers, int fuse_lu, int64_t B, int D, int hidden, int num_blocks, int K, double tail_bound,
                          double min_bin_width, double min_bin_height, double min_derivative, int direction, int acc,
                          nf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * LULinearPermute.  Replaces normflows/flows/mixing.py:535-563 (LULinearPermute), :229-244
 * (_Permutation), :402-473 (_LULinear forward_no_cache / inverse_no_cache), :514-532 (upper_diag,
 * logabsdet).
 *   perm (D) int64 = `permutation._permutation`; lower_entries, upper_entries (D(D-1)/2) in
 *   np.tril_indices(D,-1) / np.triu_indices(D,1) order; unconstrained_upper_diag (D); bias (D).
 *   direction 0 = density  (LULinearPermute.inverse): y = L (U x[:,perm]) + bias, ld = +sum log diag
 *   direction 1 = sample   (LULinearPermute.forward): solve L, U on (x - bias), y[:,perm[j]] = t[:,j],
 *                                                      ld = -sum log diag
 *   diag = softplus(unconstrained_upper_diag) + eps.
 */
int nf_lu_linear_permute(const void *x, void *y, void *logdet, const int64_t *perm, const void *lower_entries,
                         const void *upper_entries, const void *unconstrained_upper_diag, const void *bias,
                         int64_t B, int D, double eps, int direction, int acc, int dtype, nf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * MaskedAffineFlow (RealNVP).  Replaces normflows/flows/affine/coupling.py:209-229.
 *   z, y (B, inner); b (inner) mask; s, t (B, inner) outputs of the scale / translation maps evaluated
 *   on b*z, either may be NULL (= zeros, coupling.py:199-207).  Non-finite s/t become NaN (:212-215).
 *   direction 0 = forward:  y = b z + (1-b)(z e^{s} + t),   ld = +sum (1-b) s
 *   direction 1 = inverse:  y = b z + (1-b)(z - t) e^{-s},  ld = -sum (1-b) s
 */
int nf_masked_affine(const void *z, const void *b, const void *s, const void *t, void *y, void *logdet,
                     int64_t B, int64_t inner, int direction, int acc, int dtype, nf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * AffineCoupling + Split/Merge (AffineCouplingBlock).  Replaces normflows/flows/affine/coupling.py:117-171
 * and :232-267 together with flows/reshape.py:30-33, 57-61 (channel / channel_inv split and merge).
 *   z, y   (B, C, HW) contiguous.  The first `c1` channels (channel mode) or the last `c1` channels
 *          (channel_inv: flip = 1) are z1 (copied), the remaining c2 = C - c1 channels are z2.
 *   param  (B, P, HW): output of param_map(z1); P = 2*c2 interleaved shift = param[:,0::2],
 *          scale_ = param[:,1::2] (coupling.py:131-132) or P = c2 (pure shift) when
 *          scale_map == NF_SCALE_NONE.
 *   direction 0 = forward, 1 = inverse.  ld = per-sample sum over c2*HW elements.
 */
int nf_affine_coupling(const void *z, const void *param, void *y, void *logdet, int64_t B, int C, int c1,
                       int flip, int64_t HW, int scale_map, int direction, int acc, int dtype,
                       nf_stream_t stream);
/* Same with a per-channel bias added to `param` on the fly (param_bias, 2*c2 or c2 values, may be NULL): lets the
 * conditioner's last convolution (nets/cnn.py:51-56) run bias-free in the library while its bias costs no extra pass. */
int nf_affine_coupling_pb(const void *z, const void *param, const void *param_bias, void *y, void *logdet, int64_t B,
                          int C, int c1, int flip, int64_t HW, int scale_map, int direction, int acc, int dtype,
                          nf_stream_t stream);


/* ------------------------------------------------------------------------------------------------
 * AffineConstFlow / ActNorm.  Replaces normflows/flows/affine/coupling.py:38-54 and the data-dependent
 * initialisation of flows/normalization.py:19-39.
 *   z, y (B, C, HW); s, t (C).  direction 0 = forward y = z e^{s} + t, 1 = inverse y = (z - t) e^{-s}.
 *   logdet_scalar (1 element, may be NULL) receives  +-HW * sum(s)  (0-dim log_det of the reference).
 *   logdet (B) (may be NULL) is combined with the same value according to `acc`.
 * nf_actnorm_stats: per-channel mean and UNBIASED std over (B, HW) (torch.std default).
 * nf_actnorm_init: writes s, t from mean/std:  direction 0 (forward-first): s = -log(std+1e-6),
 *   t = -mean e^{s};  direction 1 (inverse-first): s = log(std+1e-6), t = mean.
 */
int nf_actnorm(const void *z, const void *s, const void *t, void *y, void *logdet_scalar, void *logdet,
               int64_t B, int C, int64_t HW, int direction, int acc, int dtype, nf_stream_t stream);
int nf_actnorm_stats(const void *z, void *mean, void *std_unbiased, int64_t B, int C, int64_t HW, int dtype,
                     nf_stream_t stream);
int nf_actnorm_init(const void *mean, const void *std_unbiased, void *s, void *t, int C, int direction,
                    int dtype, nf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Invertible1x1Conv.  Replaces normflows/flows/mixing.py:88-133.
 * nf_inv1x1_assemble: W (C,C) from the LU parametrisation (P, L, U, sign_S, log_S) (:88-104);
 *   inverse = 0:  W = P (tril(L,-1)+I) (triu(U,1)+diag(sign_S e^{log_S}))      (density direction)
 *   inverse = 1:  W = U'^{-1} L'^{-1} P^T with the triangular inverses taken in fp64 (:94-101)
 *   logdet_unit (1 element): +sum(log_S) (inverse=0) or -sum(log_S) (inverse=1).
 * nf_inv1x1_conv: y[b,o,p] = sum_c W[o,c] z[b,c,p];  logdet_scalar = logdet_unit * HW (may be NULL);
 *   logdet (B) (may be NULL) combined according to `acc`.
 */
int nf_inv1x1_assemble(const void *P, const void *L, const void *U, const void *sign_S, const void *log_S,
                       void *W, void *logdet_unit, int C, int inverse, int dtype, nf_stream_t stream);
/* Its VJP in the density direction (W = P Lm Um, log|det| = sum log_S; mixing.py:88-104 under autograd): gL / gU (C x C, the strictly
 * lower / upper parts; zero elsewhere) and g_log_S (C) from the cotangents gW (C x C) and gl (0-dim, may be NULL).  C <= 64. */
int nf_inv1x1_lu_grads(const void *P, const void *L, const void *U, const void *sign_S, const void *log_S, const void *gW,
                       const void *gl, void *gL, void *gU, void *glogS, int C, int dtype, nf_stream_t stream);
/* Both for n layers of one size C in ceil(n / 32) launches (round 6, float32, density direction): a Glow level's K blocks
 * (flows/affine/glow.py:72-84 inside core.py:588-616) assemble their matrices from parameters alone, and their LU factors' gradients
 * need nothing but every block's gW -- one launch per level and direction instead of one per block (96 + 96 single-workgroup launches
 * per training step of BASELINE configs[3]).  Every pointer argument is a HOST array of n device pointers (gl[i] may be NULL). */
int nf_inv1x1_assemble_multi(const void *const *P, const void *const *L, const void *const *U, const void *const *sign_S,
                             const void *const *log_S, void *const *W, void *const *logdet_unit, int n, int C, nf_stream_t stream);
int nf_inv1x1_lu_grads_multi(const void *const *P, const void *const *L, const void *const *U, const void *const *sign_S,
                             const void *const *log_S, const void *const *gW, const void *const *gl, void *const *gL, void *const *gU,
                             void *const *glogS, int n, int C, nf_stream_t stream);
int nf_inv1x1_conv(const void *z, const void *W, const void *logdet_unit, void *y, void *logdet_scalar,
                   void *logdet, int64_t B, int C, int64_t HW, int acc, int dtype, nf_stream_t stream);
/* y = W z + bias per pixel (bias (C) may be NULL): the 1x1 convolution with the neighbouring ActNorm
 * (normalization.py:7-39) folded into W / bias / logdet_unit by the caller -- GlowBlock's [Invertible1x1Conv, ActNorm]
 * pair (affine/glow.py:72-84) as ONE pass over the tensor. */
int nf_inv1x1_conv_affine(const void *z, const void *W, const void *bias, const void *logdet_unit, void *y,
                          void *logdet_scalar, void *logdet, int64_t B, int C, int64_t HW, int acc, int dtype,
                          nf_stream_t stream);
/* y = W^T z per pixel, no log-det: the 1x1 convolution's input gradient under autograd (mixing.py:106-133 through loss.backward())
 * without a transposed copy of W. */
int nf_inv1x1_conv_t(const void *z, const void *W, void *y, int64_t B, int C, int64_t HW, int dtype, nf_stream_t stream);


/* ------------------------------------------------------------------------------------------------
 * DiagGaussian.log_prob.  Replaces normflows/distributions/base.py:94-103.
 *   z (B, d); loc, log_scale (d);  log_p[b] = -d/2 log(2 pi) - sum_j (log_scale_j + 0.5 ((z-loc)/e^{log_scale})^2)
 *   `log_scale_shift` is added to log_scale (temperature, base.py:95-98).  out (B) combined per `acc`.
 */
int nf_diag_gaussian_log_prob(const void *z, const void *loc, const void *log_scale, double log_scale_shift,
                              void *out, int64_t B, int64_t d, int acc, int dtype, nf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Weight and bias gradient of a Linear layer over the batch (training step, core.py:87-102 + loss.backward();
 * the conditioner layers nets/resnet.py:37-50, 92-104): dW[m][n] (+)= sum_b dY[b][m] X[b][n], db[m] (+)= sum_b dY[b][m].
 *   dY (B, M), X (B, N) row-major float32, N <= 128; dW (M, N), db (M) or NULL; accumulate 0 = overwrite, 1 = add.
 *   scratch: nf_linear_wgrad_scratch_floats(B, M, N) floats owned by the caller (per-K-chunk partial tiles, summed in
 *   a fixed order: results are run-to-run deterministic).
 */
int64_t nf_linear_wgrad_scratch_floats(int64_t B, int M, int N);
int nf_linear_wgrad(const void *dY, const void *X, void *dW, void *db, void *scratch, int64_t B, int M, int N,
                    int accumulate, nf_stream_t stream);
/* Same with relu_x = 1: the second operand is relu(X), applied as the values are consumed (the caller keeps only the
 * pre-activation of the block, resnet.py:41-47). */
int nf_linear_wgrad_act(const void *dY, const void *X, void *dW, void *db, void *scratch, int64_t B, int M, int N,
                        int accumulate, int relu_x, nf_stream_t stream);
/* Same with pad rows: when skip_every > 1 (M % skip_every == 0), every skip_every-th row of dY (m % skip_every ==
 * skip_every - 1) is padding and has no output row -- dW is (M - M / skip_every, N), db (M - M / skip_every).  The 24-float
 * parameter rows of nf_rqs_fused_train_fwd / nf_rqs_coupling_bwd_p24 (23 parameters + 1 pad) against the reference's
 * 23-row final layer: the un-padding happens in the reduction, not as two strided copies. */
int nf_linear_wgrad_skip(const void *dY, const void *X, void *dW, void *db, void *scratch, int64_t B, int M, int N,
                         int accumulate, int relu_x, int skip_every, nf_stream_t stream);
/* Two problems of the same shape (B, M, N) in one partial launch and one reduction -- the two weight gradients of a
 * residual block (nets/resnet.py:37-50) become available together.  scratch: 2 x nf_linear_wgrad_scratch_floats(B, M, N);
 * db0 / db1 both given or both NULL; vector / tile paths only (N % 4 == 0), NF_ENOTSUP otherwise. */
int nf_linear_wgrad_pair(const void *dY0, const void *X0, void *dW0, void *db0, const void *dY1, const void *X1, void *dW1,
                         void *db1, void *scratch, int64_t B, int M, int N, int accumulate, int relu_x, nf_stream_t stream);
/* Backward of one residual block of the conditioner (nets/resnet.py:37-50; hidden H = 128) in ONE pass over the rows: the two
 * input-gradient products and both weight / bias gradients: gt = (gh W2)[t > 0], gh_in = gh + (gt W1)[h_in > 0],
 * dW2 = gh^T relu(t), db2 = colsum(gh), dW1 = gt^T relu(h_in), db1 = colsum(gt)  (t = the block's pre-activation, h_in its input,
 * both (B, H) as the training forward saved them).  With x != NULL also the initial Linear layer behind the block
 * (nets/resnet.py:92-104, D = 64): gx (B, D) += gh_in wfull^T (wfull (D, H) = the layer's weight transposed on full rows, zero rows
 * at the transformed features), dW0 (H, D) = gh_in^T x, db0 = colsum(gh_in); gh_in is then not written (may be NULL).
 * col_map (D int32, optional): dW0 keeps only the columns n with col_map[n] >= 0, as (H, n_cols) with column col_map[n] (the
 * identity features: the initial layer's own (H, nI) weight gradient).
 * B a multiple of 64; scratch: nf_resblock_bwd_scratch_floats(B, x != NULL) floats.  Deterministic (fixed-order reduction). */
/* Backward of LULinearPermute's batch side, density direction (mixing.py:535-563: u = U x[perm], y = L u + b), D = 64, in ONE
 * pass over the rows: gx = (gy Lm) Up, dL = gy^T u, db = colsum(gy), dUp = (gy Lm)^T x.  Lm = L, Up = U with permuted columns
 * (nf_lu_factors), u = the forward's intermediate (nf_rows_matvec2); nf_lu_param_grads turns (dL, dUp) into the packed
 * parameter gradients.  B a multiple of 64; scratch: nf_lu_bwd_scratch_floats(B) floats.  Deterministic. */
int64_t nf_lu_bwd_scratch_floats(int64_t B);
int nf_lu_bwd(const void *gy, const void *u, const void *x, const void *Lm, const void *Up, void *gx, void *dL, void *db, void *dUp,
              void *scratch, int64_t B, int D, nf_stream_t stream);
/* ... and its forward on the same tiles: u (B, D) = U x[perm] per row (kept for the backward), y = L u + bias, logdet (op)=
 * ld_sign * *ld_const (acc = NF_LD_WRITE / ADD / SUB; logdet may be NULL).  UpT / LT: the transposed factor images of
 * nf_lu_factors.  D = 64, B a multiple of 64. */
int nf_lu_fwd(const void *x, const void *UpT, const void *LT, const void *bias, void *u, void *y, void *logdet,
              const void *ld_const, double ld_sign, int acc, int64_t B, int D, nf_stream_t stream);
/* Round 6: the LULinearPermute of a benchmark-shaped pair fused into the training forward, and its backward on the composed matrix.
 *   nf_lu_pack_train_multi: once per step (behind nf_rqs_fused_pack_all_multi), n layers in one launch: the density-direction LU
 *   stage, bias and constant log|det| of each layer's training blob (W_d = L U with permuted columns, composed in float64:
 *   mixing.py:402-473, 535-563) and W_d row-major (64, 64) for the backward.  table (device): n rows of 7 pointers perm (int64),
 *   lower_entries, upper_entries, unconstrained_upper_diag, bias, wpack, wd_out.
 *   nf_rqs_fused_train_pair_fwd: nf_rqs_fused_train_full_fwd with LULinearPermute.inverse in front of the coupling in the same
 *   launch (the order NormalizingFlow.log_prob visits a [CoupledRQS, LULinearPermute] pair, core.py:193-195); xlu_out (B, 64) = the
 *   LU's output = the coupling's input, kept for the backward; logdet takes both layers' terms.
 *   nf_lu_bwd_composed: gx (B, 64) = g W_d, dWd (64, 64) = g^T x, db (64) = colsum(g) in one pass over the rows (x = the LU's
 *   INPUT rows); scratch: nf_lu_bwd_composed_scratch_floats(B); B a multiple of 64; deterministic.
 *   nf_lu_param_grads_composed: dWd -> (g_lower, g_upper, g_udiag): dM[:, j] = dWd[:, perm[j]], dL = dM U^T, dU = L^T dM, the
 *   diagonal through softplus' with gl_sum / diag (gld (B): the log-det cotangent, summed inside; may be NULL); Lm, Um: the dense
 *   factors of nf_lu_factors[_multi].  D = 64. */
int nf_lu_pack_train_multi(const void *table, int n_layers, int num_blocks, int D, double eps, nf_stream_t stream);
int nf_rqs_fused_train_pair_fwd(const void *x, void *xlu_out, void *y, void *logdet, void *cond_out, void *act_out, const void *wpack,
                                int mask_parity, int64_t B, int D, int hidden, int num_blocks, int K, double tail_bound,
                                double min_bin_width, double min_bin_height, double min_derivative, int acc, nf_stream_t stream);
int64_t nf_lu_bwd_composed_scratch_floats(int64_t B);
int nf_lu_bwd_composed(const void *g, const void *x, const void *Wd, void *gx, void *dWd, void *db, void *scratch, int64_t B, int D,
                       nf_stream_t stream);
int nf_lu_param_grads_composed(const void *dWd, const void *Lm, const void *Um, const int64_t *perm, const void *gld, int64_t B,
                               const void *unconstrained_upper_diag, double eps, void *g_lower, void *g_upper, void *g_udiag, int D,
                               nf_stream_t stream);
/* The pass of nf_lu_bwd_composed without its reduction (partial tiles [nf_lu_bwd_composed_grid(B)][64 * 64 + 64] stay in scratch), and
 * the whole backward of a [CoupledRQS, LULinearPermute] pair behind ONE call (autograd.PairTrainFn): nf_coupling_train_bwd's four
 * passes on the coupling (x = xlu, the LU's output saved by nf_rqs_fused_train_pair_fwd), nf_lu_bwd_composed_partials on its input
 * gradient, one reduction launch for the partial tiles of BOTH layers, nf_lu_param_grads_composed: seven launches.  Arguments as
 * the two calls it replaces; grad_x_in (B, 64) = the pair's input gradient; scratch: nf_pair_train_bwd_scratch_floats(B, num_blocks). */
int nf_lu_bwd_composed_grid(int64_t B);
int nf_lu_bwd_composed_partials(const void *g, const void *x, const void *Wd, void *gx, void *scratch, int64_t B, int D,
                                nf_stream_t stream);
int64_t nf_pair_train_bwd_scratch_floats(int64_t B, int num_blocks);
int nf_pair_train_bwd(const void *x_in, const void *xlu, const void *grad_y, const void *grad_logdet, const void *cond24,
                      const void *acts, const void *w_t, const void *wpack, const void *wfull_t, const void *const *w_blocks,
                      const void *uw, const void *uh, const void *ud, const void *col_map, int n_cols, const void *Wd, const void *Lm,
                      const void *Um, const int64_t *perm, const void *unconstrained_upper_diag, double lu_eps, void *grad_x_in,
                      void *g_lower, void *g_upper, void *g_udiag, void *g_lbias, void *g_w0, void *g_b0, void *g_wf, void *g_bf,
                      void *g_uw, void *g_uh, void *g_ud, void *const *g_blocks, void *scratch, int mask_parity, int64_t B, int D,
                      int hidden, int num_blocks, int K, double tail_bound, double min_bin_width, double min_bin_height,
                      double min_derivative, nf_stream_t stream);
/* nf_pair_train_bwd in two calls (round 6, late; replaces nothing in the reference -- `loss.backward()` of core.py:87-102 has no such
 * boundary): _head issues the five launches the NEXT pair's backward waits for (the pair's input gradient) and fills `tail`
 * (NF_PAIR_TAIL_BYTES of HOST memory, opaque) with what the last two launches need; _tail issues those two -- the reduction of every
 * partial tile and the LU's factor gradients, which only produce PARAMETER gradients -- on any stream the caller has ordered behind
 * _head's (event fork).  On a side stream they run under the next pair's MFMA-bound kernels (autograd.PairTrainFn).  `scratch`, the
 * saved rows and every gradient destination stay in use until _tail's launches have run; a gradient may be read only on a stream
 * ordered behind them.  _head followed by _tail on one stream is exactly nf_pair_train_bwd (same seven launches, same bits). */
#define NF_PAIR_TAIL_BYTES 2048
int nf_pair_train_bwd_head(const void *x_in, const void *xlu, const void *grad_y, const void *grad_logdet, const void *cond24,
                           const void *acts, const void *w_t, const void *wpack, const void *wfull_t, const void *const *w_blocks,
                           const void *uw, const void *uh, const void *ud, const void *col_map, int n_cols, const void *Wd,
                           const void *Lm, const void *Um, const int64_t *perm, const void *unconstrained_upper_diag, double lu_eps,
                           void *grad_x_in, void *g_lower, void *g_upper, void *g_udiag, void *g_lbias, void *g_w0, void *g_b0,
                           void *g_wf, void *g_bf, void *g_uw, void *g_uh, void *g_ud, void *const *g_blocks, void *scratch,
                           int mask_parity, int64_t B, int D, int hidden, int num_blocks, int K, double tail_bound,
                           double min_bin_width, double min_bin_height, double min_derivative, void *tail, nf_stream_t stream);
int nf_pair_train_bwd_tail(const void *tail, nf_stream_t stream);
int64_t nf_resblock_bwd_scratch_floats(int64_t B, int with_init);
int nf_resblock_bwd(const void *gh, const void *t, const void *h_in, const void *W1, const void *W2, void *gh_in, void *dW1,
                    void *db1, void *dW2, void *db2, const void *x, const void *wfull, void *gx, void *dW0, void *db0,
                    const void *col_map, int n_cols, void *scratch, int64_t B, int H, int D, nf_stream_t stream);
/* The passes over the rows WITHOUT their reduction launches (round 6): the partial tiles stay in `scratch` for a later reduction
 * -- nf_coupling_train_bwd sums all of a layer's partial tiles with ONE launch.
 *   nf_linear_wgrad_partials: scratch = [nf_linear_wgrad_chunks(B, M, N)][M * N + M] (dW tile then the column sums of dY);
 *   nf_resblock_bwd_partials: scratch = [2][nf_resblock_bwd_grid(B)][128 * 128 + 128] ((dW2, db2) then (dW1, db1)), then with x
 *   the initial layer's [grid][128 * 64 + 128]; arguments as nf_resblock_bwd. */
int nf_linear_wgrad_chunks(int64_t B, int M, int N);
int nf_linear_wgrad_partials(const void *dY, const void *X, void *scratch, int64_t B, int M, int N, int relu_x, int want_bias,
                             nf_stream_t stream);
int nf_resblock_bwd_grid(int64_t B);
int nf_resblock_bwd_partials(const void *gh, const void *t, const void *h_in, const void *W1, const void *W2, void *gh_in,
                             const void *x, const void *wfull, void *gx, void *scratch, int64_t B, int H, int D,
                             nf_stream_t stream);
/* The whole backward of a benchmark-shaped coupling layer behind one call (round 6): what `loss.backward()` (core.py:87-102) does
 * for CoupledRationalQuadraticSpline's density direction (nsf/coupling.py:83-98, nets/resnet.py:37-50, 92-104,
 * utils/splines.py:16-219) given what nf_rqs_fused_train_full_fwd saved.  Four passes over the rows (nf_final_bwd, the final
 * layer's weight-gradient partials, nf_resblock_bwd_partials per residual block -- the first block's with the initial layer) and
 * ONE reduction launch for every partial tile of the layer, in the fixed order of the stand-alone reductions (bit-identical
 * gradients, deterministic, no atomics).
 *   x, grad_y (B, 64), grad_logdet (B), cond24 (B, 32, 24), acts (2 num_blocks + 1, B, 128): as saved by the forward;
 *   w_t / wpack / wfull_t: the images nf_rqs_fused_pack_all left (final weight as nf_final_bwd's stages; the blob; the initial
 *   weight transposed on full rows (64, 128)); w_blocks: 2 num_blocks device pointers W1, W2 per block ((128, 128) as stored by
 *   nn.Linear); uw, uh, ud: the batch-shared spline parameters; col_map / n_cols: as nf_resblock_bwd.
 *   Outputs, all written (not accumulated): grad_x (B, 64); g_w0 (128, n_cols), g_b0 (128), g_wf (736, 128), g_bf (736), g_uw,
 *   g_uh (32, 8), g_ud (32, 7); g_blocks: 4 num_blocks device pointers gW1, gb1, gW2, gb2 per block -- any addresses (e.g. views
 *   of one flat gradient buffer).  scratch: nf_coupling_train_bwd_scratch_floats(B, num_blocks) floats.
 *   D = 64, hidden = 128, K = 8, 1 <= num_blocks <= 5, B a multiple of 64, float32; NF_ENOTSUP otherwise. */
int64_t nf_coupling_train_bwd_scratch_floats(int64_t B, int num_blocks);
int nf_coupling_train_bwd(const void *x, const void *grad_y, const void *grad_logdet, const void *cond24, const void *acts,
                          const void *w_t, const void *wpack, const void *wfull_t, const void *const *w_blocks, const void *uw,
                          const void *uh, const void *ud, const void *col_map, int n_cols, void *grad_x, void *g_w0, void *g_b0,
                          void *g_wf, void *g_bf, void *g_uw, void *g_uh, void *g_ud, void *const *g_blocks, void *scratch,
                          int mask_parity, int64_t B, int D, int hidden, int num_blocks, int K, double tail_bound,
                          double min_bin_width, double min_bin_height, double min_derivative, nf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Logit preprocessing transform of image tensors.  Replaces normflows/transforms.py:8-47.
 *   z, y (B, inner) contiguous (inner = C*H*W); beta = 1 - 2 alpha, 0 <= alpha < 0.5.
 *   direction 0 = Logit.forward (:25-32): y = (sigmoid(z) - alpha)/beta,
 *                 ld = -inner log(beta) + sum logsigmoid(z) + sum logsigmoid(-z);
 *   direction 1 = Logit.inverse (:34-47): u = alpha + beta z, y = log u - log(1 - u),
 *                 ld = inner log(beta) - sum log u - sum log(1 - u).
 *   logdet (B) combined according to `acc`.
 */
int nf_logit(const void *z, void *y, void *logdet, int64_t B, int64_t inner, double alpha, int direction, int acc,
             int dtype, nf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Bias + LeakyReLU behind a bias-free convolution of the Glow conditioner (normflows/nets/cnn.py:40-50:
 * Conv2d(bias=True) followed by LeakyReLU(leaky)), in place on y (B, C, H*W) contiguous NCHW:
 *   y = leaky_relu(y + bias[c], negative_slope).
 */
int nf_bias_leaky_relu(void *y, const void *bias, int64_t B, int C, int64_t HW, double negative_slope, int dtype,
                       nf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Diagonal Gaussian log-density with a mean / log-scale ROW per sample.  Replaces
 * normflows/distributions/base.py:326-345 (ClassCondDiagGaussian.log_prob).
 *   loc_rows, log_scale_rows (num_rows, d) row-major: for class labels the transposed parameters
 *   (num_classes, d) with row_idx (B) int64 = labels (an out-of-range label yields NaN for that sample);
 *   for soft labels the blended rows (B, d) with row_idx = NULL (row b belongs to sample b).
 *   out[b] (acc)= -d/2 log(2 pi) - sum_j (ls_j + 0.5 ((z_bj - loc_j) / exp(ls_j))^2), ls = log_scale + log_scale_shift.
 */
int nf_diag_gaussian_log_prob_rows(const void *z, const void *loc_rows, const void *log_scale_rows,
                                   const int64_t *row_idx, int64_t num_rows, double log_scale_shift, void *out,
                                   int64_t B, int64_t d, int acc, int dtype, nf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * A RealNVP-style stack as ONE launch: MaskedAffineFlow layers (affine/coupling.py:209-229) whose s / t
 * conditioners are small MLPs (nets/mlp.py:5-58, Linear + LeakyReLU, widths <= 64), interleaved with
 * AffineConstFlow / ActNorm layers (coupling.py:38-54); the model of examples/real_nvp.ipynb (BASELINE configs[0]).
 *   z, y (B, d) float32, d <= 16; blob: host-packed parameters of the stack (normflows_amd/core.py documents the
 *   layout); hmax = widest MLP layer; direction 0 applies the records in order with the forward formulas,
 *   1 in reverse order with the inverse formulas; logdet (B) combined according to `acc`.
 */
int nf_realnvp_chain(const void *z, void *y, void *logdet, const void *blob, int64_t B, int d, int hmax, int direction,
                     int acc, nf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * MaskedAffineAutoregressive element-wise transform (MAF).  Replaces
 * normflows/flows/affine/autoregressive.py:98-128 (_elementwise_forward / _elementwise_inverse).
 *   params (B, D, 2): MADE output viewed as (unconstrained_scale, shift) per feature (:124-128);
 *   scale = sigmoid(unconstrained_scale + 2) + 1e-3.
 *   direction 0: y = scale x + shift, ld = +sum log scale; 1: y = (x - shift)/scale, ld = -sum log scale.
 *   logdet (B) may be NULL (intermediate passes of the D-step inverse, autoregressive.py:29-38).
 */
int nf_maf_affine(const void *x, const void *params, void *y, void *logdet, int64_t B, int D, int direction, int acc,
                  int dtype, nf_stream_t stream);
/* Its backward under autograd (core.py:87-102): g_x (B, D), g_params (B, D, 2) from the cotangents gy (B, D) and gld (B); either
 * cotangent may be NULL (zero). */
int nf_maf_affine_bwd(const void *x, const void *params, const void *gy, const void *gld, void *gx, void *gparams, int64_t B, int D,
                      int direction, int dtype, nf_stream_t stream);
/* One sweep of the implicit backward of the MAF inverse (the density direction under autograd; normflows_amd.autograd.MafInverseFn):
 * v <- (gx - gxm) / scale in place (gxm NULL on the first sweep), g_p (B, D, 2) = the parameter cotangent for (v, gld) -- the input of
 * the next nf_made_backward --, *changed (int, zeroed by the caller) set when any element of v moved. */
int nf_maf_implicit_sweep(const void *x, const void *params, const void *gx, const void *gld, const void *gxm, void *v, void *gp,
                          void *changed, int64_t B, int D, int dtype, nf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * CoupledRationalQuadraticSpline in ONE launch for the shapes beyond nf_rqs_fused's (D <= 64, hidden <= 128): up to 128 features,
 * up to 512 hidden units, 8 bins (nf_nsf_wide_k: 4 | 8 | 16), linear tails, float32.  Replaces normflows/flows/neural_spline/wrapper.py:79-85 ->
 * nsf/coupling.py:71-128 (split, conditioner, coupling transform, unconditional transform, merge; the direction-dependent order),
 * :150-164, :221-253, :329-362 (parameter rows, the 1/sqrt(hidden) scaling, the batch-shared spline), nets/resnet.py:37-50, 92-104
 * (the ResidualNet conditioner) and utils/splines.py:16-219 (the spline itself); the conditioner output never exists in memory.
 *   blob, table : device copies of the host packer's arrays (normflows_amd/flows/nsf_wide_pack.py; table[3] = hidden_padded).
 *   tabs        : (n_identity, 27) knot tables of the batch-shared spline, written by nf_nsf_wide_tables from the raw
 *                 unnormalized_widths (n_identity, 8), unnormalized_heights (n_identity, 8), unnormalized_derivatives (n_identity, 7).
 *   direction 0 : density direction (wrapper.inverse = prqct.forward); 1: sampling direction (wrapper.forward = prqct.inverse).
 *   lu_logdet   : NULL, or the device scalar sum(log(softplus(u_diag) + eps)) of the adjacent LULinearPermute (mixing.py:535-563,
 *                 :402-473) whose dense matrix the pack carries for THIS direction: the pair [CoupledRationalQuadraticSpline,
 *                 LULinearPermute] then runs as one launch -- density: y = coupling.inverse(LU.inverse(x)) (the order of
 *                 core.py:193-195), sampling: y = LU.forward(coupling.forward(x)) (core.py:177-179); logdet gets both terms.
 *   x, y (B, D); logdet (B) combined according to `acc`.  -EINVAL for min_bin_width * 8 > 1 (utils/splines.py:121-124).
 */
int nf_nsf_wide_tables(const void *uw, const void *uh, const void *ud, void *tabs, int n_identity, int K, double tail_bound,
                       double min_bin_width, double min_bin_height, double min_derivative, nf_stream_t stream);
int nf_nsf_wide(const void *x, void *y, void *logdet, const void *blob, const int32_t *table, const void *tabs,
                const void *lu_logdet, int64_t B, int D, int hidden_padded, int direction, int acc, double tail_bound,
                double min_bin_width, double min_bin_height, double min_derivative, nf_stream_t stream);
/* the same with K = 4 | 8 | 16 bins (`num_bins` of wrapper.py:20-35; round 5): the pack (table[24] = K) and the tables
 * (nf_nsf_wide_tables with the same K: (n_identity, 3 (K + 1)) floats from widths / heights (n_identity, K), derivatives
 * (n_identity, K - 1)) are built for that K; a final-layer group then holds 8 / 4 / 2 transform features.  -ENOTSUP for other K. */
int nf_nsf_wide_k(const void *x, void *y, void *logdet, const void *blob, const int32_t *table, const void *tabs,
                  const void *lu_logdet, int64_t B, int D, int hidden_padded, int K, int direction, int acc, double tail_bound,
                  double min_bin_width, double min_bin_height, double min_derivative, nf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * MADE in ONE launch -- the single-pass direction of the autoregressive flows.  Replaces
 * normflows/nets/made.py:296-304 (MADE.forward: initial MaskedLinear, MaskedResidualBlocks :196-214, final MaskedLinear; every
 * linear is F.linear(x, weight * mask, bias), :80-81) and, for nf_made_forward_affine, also
 * normflows/flows/affine/autoregressive.py:24-27 + :101-110 (Autoregressive.forward + _elementwise_forward:
 * scale = sigmoid(unconstrained_scale + 2) + 1e-3, y = scale x + shift, logdet = sum log scale).
 *   blob, table : device copies of the arrays of the host packer (normflows_amd/flows/made_pack.py): hidden units sorted by
 *                 degree, per 32-row block one stream of MFMA A fragments that stops at the last k-group its MASK reaches
 *                 (structurally zero blocks are neither stored nor multiplied), biases in the same order.
 *   hidden_padded = table[3] (256 or 512), D = table[0] <= 128, float32, ReLU residual blocks, no context / batch norm / dropout /
 *                 permuted input degrees (the packer returns None for anything else and the caller keeps the layer-wise path).
 *   nf_made_forward_affine: x, y (B, D), logdet (B) combined according to `acc`.
 *   nf_made_forward       : params (B, mult D), rows in the reference's order (mult f + p).
 */
int nf_made_forward_affine(const void *x, void *y, void *logdet, const void *blob, const int32_t *table, int64_t B, int D,
                           int hidden_padded, int acc, nf_stream_t stream);
int nf_made_forward(const void *x, void *params, const void *blob, const int32_t *table, int64_t B, int D, int hidden_padded,
                    int mult, nf_stream_t stream);
/* The autoregressive rational-quadratic spline layer's density direction in one launch: replaces
 * normflows/flows/neural_spline/autoregressive.py:94-134 (MaskedPiecewiseRationalQuadraticAutoregressive._elementwise_forward over
 * Autoregressive.forward, flows/affine/autoregressive.py:24-27; what wrapper.py:241-245 AutoregressiveRationalQuadraticSpline.inverse
 * calls): MADE with 23 = 3 * 8 - 1 outputs per feature, then utils/splines.py:16-219 element-wise (8 bins, linear tails, no
 * 1 / sqrt(hidden) scaling: the reference's MADE has no `hidden_features`), row-summed log-det.
 *   blob, table : made_pack.pack_made_forward(made, 23, spline=True); x, y (B, D); logdet (B) combined according to `acc`. */
int nf_made_forward_spline(const void *x, void *y, void *logdet, const void *blob, const int32_t *table, int64_t B, int D,
                           int hidden_padded, int acc, double tail_bound, double min_bin_width, double min_bin_height,
                           double min_derivative, nf_stream_t stream);

/* MADE under autograd.  Replaces what torch autograd records for normflows/nets/made.py:296-304 inside the training loop
 * core.py:87-102 + loss.backward(): per MaskedLinear (:80-81) the products g W, g^T a and the mask multiplication of the weight
 * gradient.  Bp = B rounded up to 64, NB = residual blocks, Hp = hidden_padded.
 *   nf_made_forward_train : nf_made_forward + save ((2 NB + 1) x Bp x Hp floats: the pre-activations h0, t_b, h_(b+1) in slot order,
 *                           rows beyond B zero) + bits ((Bp / 64) x 2 NB x 2 x 512 dwords: the signs of what a ReLU follows).
 *   nf_made_backward      : g_params (B, mult D) -> g_x (B, D) and G ((2 NB + 1) x Bp x Hp: the gradient at every layer's output;
 *                           NULL = not stored: the sweeps of the implicit MAF backward only want g_x);
 *                           blob / table: made_pack.pack_made_backward (transposed masked weights, k-group ranges per row-block).
 *   nf_made_wgrad         : every weight / bias gradient in one launch over the non-zero 128 x 128 tiles + a fixed-order reduction
 *                           (deterministic); grads = flat vector in the packer's layout, zero-filled by the caller, written where
 *                           `mask` (bytes, same layout) is non-zero -- the reference's weight.grad is zero under the mask too;
 *                           gp_pad / x_pad: g_params / x with Bp rows and the row length rounded up to 128 (zeros);
 *                           part: nf_made_wgrad_scratch_floats(B, ntiles) floats. */
int nf_made_forward_train(const void *x, void *params, void *save, void *bits, const void *blob, const int32_t *table, int64_t B,
                          int D, int hidden_padded, int mult, nf_stream_t stream);
int nf_made_backward(const void *g_params, const void *bits, void *g_x, void *G, const void *blob, const int32_t *table, int64_t B,
                     int D, int hidden_padded, int mult, nf_stream_t stream);
/* nf_made_forward_train / nf_made_backward on 128-row tiles where a 256-slot network has <= 64 input features and the batch is a
 * multiple of 128 rows whose rounds of 256 persistent workgroups come out shorter that way (32 768, 65 536, ... rows; round 6: a work item
 * spans two sample blocks like the 512-slot kernels; same bits as 64-row tiles):
 * on (1, default) / off (0), returns the previous setting.  A forward and its backward must run under the same setting. */
int nf_config_made_tr128(int on);
int64_t nf_made_wgrad_scratch_floats(int64_t B, int ntiles);
int nf_made_wgrad(const void *gp_pad, const void *x_pad, const void *G, const void *save, void *grads, const void *mask, void *part,
                  const int32_t *wtable, const int32_t *stable, int ntiles, int64_t B, nf_stream_t stream);
/* The weight streams of a training step from the parameters as they are now: out[i] = flat[src[i]], flat = [0, every parameter of
 * the network flattened], src = the packer's stream with parameter positions in place of values (made_pack.train_structure) -- the
 * reference re-reads its nn.Parameters in every forward; a host-side repack per optimizer step would cost more than the step. */
int nf_pack_gather(const void *flat, const int32_t *src, void *out, int64_t n, nf_stream_t stream);
/* nf_pack_gather from the parameter tensors in place (round 6): params = n_params <= 16 host pointers to device float32 tensors of
 * numels[j] elements; the virtual flat vector is [0, params[0] ..., params[1] ..., ...] as above -- no concatenation per call. */
int nf_pack_gather_multi(const void *const *params, const int64_t *numels, int n_params, const int32_t *src, void *out, int64_t n,
                         nf_stream_t stream);
/* ... for n_modules networks of ONE structure per launch (round 6: the conditioners of a Glow level's K blocks, nets/cnn.py:5-63 inside
 * flows/affine/glow.py:72-84): params = HOST array of n_modules x n_params (<= 8) device pointers, module-major; numels = the n_params
 * element counts shared by the modules; outs = HOST array of n_modules device pointers to n floats each. */
int nf_pack_gather_batch(const void *const *params, const int64_t *numels, int n_params, const int32_t *src, void *const *outs, int64_t n,
                         int n_modules, nf_stream_t stream);

/* GlowBlock's conv conditioner under autograd.  Replaces what torch autograd + the convolution library do for
 * normflows/nets/cnn.py:5-63 (ConvNet2d: Conv2d 3x3 -> LeakyReLU(0) -> Conv2d 1x1 -> LeakyReLU(0) -> Conv2d 3x3, padding 1) inside
 * core.py:87-102 + loss.backward() through flows/affine/glow.py:10-100: a 3x3 convolution over a few channels is a gather of the 9
 * neighbours per pixel (nf_conv3x3_gather: col (B H W, 9 C)) in front of a per-pixel linear layer, or a per-pixel linear layer to 9 C
 * tap products followed by a sum over the 9 neighbours (nf_conv3x3_gather_sum); the per-pixel MLP (9 Cin -> hidden -> hidden -> 9 Cout)
 * runs on nf_made_forward_train / nf_made_backward / nf_made_wgrad with tables in plain-MLP mode (made_pack.pack_mlp_*).  flip = 1
 * negates the offsets: the backward pass's gather of the output cotangent and gather-sum of the column cotangent; batch_stride
 * (elements, >= C H W) lets the gather read a channel split of a wider NCHW tensor in place. */
int nf_conv3x3_gather(const void *in, void *col, int64_t B, int C, int H, int W, int ld, int flip, int64_t batch_stride,
                      nf_stream_t stream);
int nf_conv3x3_gather_sum(const void *P, const void *bias, void *out, int64_t B, int C, int H, int W, int ld, int flip,
                          nf_stream_t stream);
/* out (C) = sum over batch and pixels of g (B, C, H, W) (float32 NCHW): the bias gradient of the conditioner's last convolution
 * (nets/cnn.py:5-63 under loss.backward(); torch: conv2d's bias backward).  One block per channel, fixed order. */
int nf_channel_sum(const void *g, void *out, int64_t B, int C, int64_t HW, nf_stream_t stream);
/* ld (B) float32 = (((ld +- t_0) +- t_1) ...) for n per-sample log-det terms (terms: HOST array of n device pointers, negate[i] != 0:
 * subtracted), in order: what the chain of `log_q += log_det` / `log_q -= log_det` statements in core.py:600-611 / :193-195 computes
 * layer by layer under autograd, as one launch per 120 terms (a Glow level has 96) with the same bits. */
int nf_ld_fold_multi(void *ld, const void *const *terms, const int *negate, int n, int64_t B, nf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * MaskedAffineAutoregressive inverse (MAF sampling direction) in ONE pass.  Replaces the D-pass loop of
 * normflows/flows/affine/autoregressive.py:29-38 over MADE (nets/made.py:217-304, residual blocks :140-214,
 * masks :63-81) together with _elementwise_inverse (:114-128): every hidden unit is finalised once, right after
 * the last feature it may see is known.  Supported structure: 2 residual blocks, ReLU, no context / batch norm /
 * dropout, sequential degrees, every degree 1..D-1 owning 1..32 hidden units, float32.
 *   blob, table : device copies of the arrays produced by the host packer (normflows_amd/flows/maf_pack.py;
 *                 layout documented there); table[3] = hidden_padded.
 *   scratch     : nf_maf_inverse_scratch_floats(B, D, hidden_padded) floats of device memory owned by the caller
 *                 (per-wave activation state, 10.5 KB per sample for D=128, H=512, + 640 B per sample of tile-pair stash);
 *                 contents need not be initialised.
 *   z, y (B, D) row-major; logdet (B) accumulated as `acc` says with -sum log(scale).
 */
int64_t nf_maf_inverse_scratch_floats(int64_t B, int D, int hidden_padded);
int nf_maf_inverse(const void *z, void *y, void *logdet, const void *blob, const int32_t *table, void *scratch,
                   int64_t B, int D, int hidden_padded, int acc, nf_stream_t stream);
/* The same inverse on the second mapping (maf_inverse_h.hip): 32 samples per wave, the lane-halves share each sample's hidden
 * units, two waves per SIMD.  Same blob / table format, arguments and results as nf_maf_inverse, for MADE conditioners of
 * num_blocks = 1, 2 or 3 residual blocks (table[6]; nets/made.py:140-214); scratch: nf_maf_inverse_h_scratch_floats floats. */
int64_t nf_maf_inverse_h_scratch_floats(int64_t B, int D, int hidden_padded, int num_blocks);
int nf_maf_inverse_h(const void *z, void *y, void *logdet, const void *blob, const int32_t *table, void *scratch,
                     int64_t B, int D, int hidden_padded, int num_blocks, int acc, nf_stream_t stream);
/* The same kernel family for FORMAT-1 packs (maf_pack.pack_made(..., tri=True); table[7] == 1; round 5).  REGULAR tiles -- at most 8
 * degrees of at most 4 hidden units each: 15 of the 16 tiles of BASELINE configs[4] (d = 128, hidden 512) -- carry their
 * sequential part's weights triangular and in reading order; they run a statically unrolled sequential part (packed multiply-adds
 * over the registers the mask leaves non-zero, one v_permlane32_swap per target pair) and an 8-deep activation ring.  The other
 * tiles keep the format-0 record and code.  `table_host` is the HOST copy of `table`: the launcher splits the tiles into maximal
 * runs of one kind and issues one launch per run (no device read-back, no host synchronisation); -EINVAL when table_host does not
 * describe this call (D, hidden_padded, num_blocks, format 1), -EFAULT when it is NULL.  Replaces the same reference lines as
 * nf_maf_inverse (affine/autoregressive.py:29-38, :114-128 over nets/made.py:217-304). */
int nf_maf_inverse_h_tri(const void *z, void *y, void *logdet, const void *blob, const int32_t *table, const int32_t *table_host,
                         void *scratch, int64_t B, int D, int hidden_padded, int num_blocks, int acc, nf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * The implicit backward of the MAF inverse in ONE pass (round 5; autograd.MafInverseFn).  The reference differentiates
 * MaskedAffineAutoregressive.inverse by recording its D sequential MADE passes (affine/autoregressive.py:29-38 under
 * core.py:87-102 `forward_kld` + backward).  x = T^-1(z) satisfies T(x) = z, so the cotangents solve the LINEAR system
 * v s + J^T g_p(v, g_ld) = g_x (J = dMADE/dx at x; g_p = the affine transform's parameter cotangent); J^T is strictly triangular in
 * the feature order and MADE's input-gradient chain has, with features and degrees counted from the top, exactly the structure the
 * incremental inverse walks: nf_maf_solve_t back-substitutes it in one launch per layer (round 4: 15-25 sweeps of nf_made_backward
 * with a host read-back every other sweep).
 *   nf_maf_inverse_h_bits : nf_maf_inverse_h (format-0 pack) that also writes the ReLU masks of the pass: ceil(B / 32) * table[4] * 64 *
 *                           num_blocks uint32.
 *   nf_maf_solve_t        : x (B, D) the inverse's result, prm (B, 2 D) = MADE(x) (nf_made_forward_train), gx (B, D), gld (B) or NULL
 *                           -> v (B, D); blob / table from maf_pack.pack_made_transposed (format 2: table[7] == 2);
 *                           scratch: nf_maf_solve_t_scratch_floats floats, contents need not be initialised.  float32. */
int nf_maf_inverse_h_bits(const void *z, void *y, void *logdet, const void *blob, const int32_t *table, void *scratch, void *bits,
                          int64_t B, int D, int hidden_padded, int num_blocks, int acc, nf_stream_t stream);
/* the same on a format-1 pack (table_host as in nf_maf_inverse_h_tri): the masks in that pack's positions, for a transposed pack
 * built with the same option (maf_pack.pack_made_transposed(tri=True)) -- the training forward then runs the fast inverse kernel */
int nf_maf_inverse_h_tri_bits(const void *z, void *y, void *logdet, const void *blob, const int32_t *table, const int32_t *table_host,
                              void *scratch, void *bits, int64_t B, int D, int hidden_padded, int num_blocks, int acc,
                              nf_stream_t stream);
/* both in one entry point (table_host NULL: format-0 pack) that also writes prm (B, 2 D) float32 = MADE's output at the solution,
 * (unconstrained scale, shift) per feature as nets/made.py:296-304 returns it and affine/autoregressive.py:98-128 reads it: what
 * nf_maf_solve_t and nf_maf_affine_bwd take as `prm` -- no second evaluation of the final layer in the backward (round 6) */
int nf_maf_inverse_h_train(const void *z, void *y, void *logdet, const void *blob, const int32_t *table, const int32_t *table_host,
                           void *scratch, void *bits, void *prm, int64_t B, int D, int hidden_padded, int num_blocks, int acc,
                           nf_stream_t stream);
int64_t nf_maf_solve_t_scratch_floats(int64_t B, int D, int hidden_padded, int num_blocks);
int nf_maf_solve_t(const void *x, const void *prm, const void *gx, const void *gld, const void *bits, void *v, const void *blob,
                   const int32_t *table, void *scratch, int64_t B, int D, int hidden_padded, int num_blocks, nf_stream_t stream);
/* nf_maf_solve_t on a transposed pack over the format-1 forward positions (pack_made_transposed(tri=True)); table_host = the HOST copy
 * of `table`: the tiles it marks regular-8 (entry [21]) run a statically unrolled sequential part, launched per run of tiles of one
 * kind (round 6).  Bit for bit the results of nf_maf_solve_t on the same pack. */
int nf_maf_solve_t_tri(const void *x, const void *prm, const void *gx, const void *gld, const void *bits, void *v, const void *blob,
                       const int32_t *table, const int32_t *table_host, void *scratch, int64_t B, int D, int hidden_padded,
                       int num_blocks, nf_stream_t stream);
/* The scratch nf_maf_solve_t leaves behind IS MADE's input-gradient chain at the solution (every virtual unit is finalised once, from
 * final values): the hidden-layer output gradients torch's autograd computes for `F.linear(x, weight * mask, bias)` per MaskedLinear
 * (nets/made.py:73-81) on the way to the weight gradients.  nf_maf_scratch_rows rearranges it from the kernel's tile order into
 * out (2 num_blocks + 1, B rounded up to 64, ldo) row-major float32 -- what nf_made_wgrad reads as G (instead of another
 * nf_made_backward pass): column c = sign * position pos_of_col[c] of the scratch (ldo int32 on the device, < 0: zero column;
 * maf_pack.solve_t_gradient_columns), layer order reversed with reverse_layers, rows >= B zero.  hidden_padded: the pack's table[3]. */
int nf_maf_scratch_rows(const void *scratch, const int32_t *pos_of_col, void *out, int64_t B, int num_blocks, int hidden_padded, int ldo,
                        double sign, int reverse_layers, nf_stream_t stream);
/* Round 6: the weight gradients of the same MaskedLinears (nets/made.py:73-81 under loss.backward(), reached from
 * flows/affine/autoregressive.py:29-38 in core.py:87-102's forward_kld) straight FROM the two scratches -- no rearrangement.
 * nf_made_wgrad_pos = nf_made_wgrad whose hidden operands stay in the one-pass kernels' order over padded positions: gscratch = what
 * nf_maf_solve_t left (the output gradients, negated inside: the solve runs on g_p(v, g_ld), the gradients belong to g_p(-v, -g_ld)),
 * fscratch = what nf_maf_inverse_h_[tri_]bits left (the linears' inputs); wtable / stable = maf_pack.position_wgrad_tables (problems,
 * mask-non-zero 128 x 128 tiles and scatter maps over POSITIONS); gp_pad, x_pad, grads, mask, part as for nf_made_wgrad;
 * num_layers = 2 num_blocks + 1, positions = the packs' table[3].  B a multiple of 64 and positions a multiple of 128, else -ENOTSUP
 * (the caller then rearranges with nf_maf_scratch_rows).  nf_maf_scratch_layer: ONE layer of a scratch as out (Bp, ldo) row-major --
 * the inverse pass's last hidden tensor, from which MADE's output at the solution follows by one product. */
int nf_made_wgrad_pos(const void *gp_pad, const void *x_pad, const void *gscratch, const void *fscratch, void *grads, const void *mask,
                      void *part, const int32_t *wtable, const int32_t *stable, int ntiles, int64_t B, int num_layers, int positions,
                      nf_stream_t stream);
int nf_maf_scratch_layer(const void *scratch, const int32_t *pos_of_col, void *out, int64_t B, int num_blocks, int hidden_padded, int ldo,
                         int layer, nf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * MaskedPiecewiseRationalQuadraticAutoregressive inverse (AR-NSF sampling direction) in ONE pass.  Replaces the D-pass
 * loop of normflows/flows/affine/autoregressive.py:29-38 over MADE with the element-wise inverse spline of
 * neural_spline/autoregressive.py:94-134 (utils/splines.py:16-219); same incremental schedule, same supported MADE
 * structure and the same scratch (nf_maf_inverse_scratch_floats) as nf_maf_inverse.
 *   blob, table : host packer's rows layout (maf_pack.pack_made(made, mult, rows=True)), mult = 3K-1 | 3K | 3K+1 for
 *                 tails linear | circular | none; mult <= 32, otherwise NF_ENOTSUP.
 *   tails / tail_bound / min_*: as nf_rqs_coupling (NF_TAILS_NONE: the unit interval); widths and heights are NOT
 *                 divided by sqrt(hidden) (the reference's MADE has no hidden_features attribute, :107-109).
 *   z, y (B, D) row-major; logdet (B) accumulated as `acc` says with the row sum of the inverse spline's logabsdet.
 */
int nf_arnsf_inverse(const void *z, void *y, void *logdet, const void *blob, const int32_t *table, void *scratch,
                     int64_t B, int D, int hidden_padded, int K, int tails, double tail_bound, double min_bin_width,
                     double min_bin_height, double min_derivative, int acc, nf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * GlowBlock conditioner in one launch.  Replaces ConvNet2d.forward (normflows/nets/cnn.py:5-63) for the network
 * built by normflows/flows/affine/glow.py:41-62: Conv2d(Cin, 256, 3, padding 1) -> LeakyReLU(slope) ->
 * Conv2d(256, 256, 1) -> LeakyReLU(slope) -> Conv2d(256, Cout, 3, padding 1), all with bias, float32.
 *   nf_glow_convnet_layout    : which of the two kernels fits a call: NF_GLOW_CONV_WIDE (256-pixel workgroups, waves of
 *                               32 pixels; needs H W | 256 and pays from ~128 workgroups on) or NF_GLOW_CONV_SMALL
 *                               (64-pixel workgroups, waves of 16 pixels; H W | 64, <= 48 output channels) or
 *                               NF_GLOW_CONV_TINY (16-pixel workgroups whose four waves split the rows of every GEMM;
 *                               H W | 16, for batches too small to fill the chip otherwise); a negative code when none
 *                               applies.  The packed weights are specific to the layout.
 *   nf_glow_convnet_pack_size : bytes of the packed weights (negative error code for an unsupported shape).
 *   nf_glow_convnet_pack      : w1 (256, Cin, 3, 3), b1 (256), w2 (256, 256[, 1, 1]), b2 (256), w3 (Cout, 256, 3, 3),
 *                               b3 (Cout), all contiguous -> wpack (MFMA operand order; repack after a weight update).
 *   nf_glow_convnet           : x = first input channel of image 0; image g starts at x + g * x_image_stride floats
 *                               and holds Cin contiguous (H, W) planes (a channel slice of an NCHW tensor is passed
 *                               without a copy); out (B, Cout, H, W) contiguous.
 * A workgroup processes whole images (the 3x3 zero padding is the image border): H W must divide the workgroup's pixel
 * count (NF_ENOTSUP otherwise; callers fall back to library convolutions).  hidden must be 256.
 */
enum { NF_GLOW_CONV_WIDE = 0, NF_GLOW_CONV_SMALL = 1, NF_GLOW_CONV_TINY = 2 };
int nf_glow_convnet_layout(int64_t B, int H, int W);
int64_t nf_glow_convnet_pack_size(int Cin, int Cout, int hidden);
int nf_glow_convnet_pack(void *wpack, const void *w1, const void *b1, const void *w2, const void *b2, const void *w3,
                         const void *b3, int Cin, int Cout, int hidden, int layout, nf_stream_t stream);
int nf_glow_convnet(const void *x, int64_t x_image_stride, void *out, const void *wpack, int64_t B, int Cin, int H,
                    int W, int Cout, int hidden, double leaky_slope, int layout, nf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * GlowBlocks in the conditioner's launch (inference, parameters frozen between updates), one block (nblocks = 1:
 * replaces GlowBlock.forward / .inverse, normflows/flows/affine/glow.py:72-84, for split_mode "channel", scale = True:
 * AffineCouplingBlock -- coupling.py:232-267: Split, AffineCoupling :117-171, Merge -- + Invertible1x1Conv + ActNorm) or
 * a whole LEVEL of the multi-scale flow in one persistent launch: replaces the inner loops of
 * MultiscaleFlow.log_prob (core.py:600-611: Squeeze.inverse, the level's GlowBlock.inverse calls,
 * Merge.inverse = channel Split) and MultiscaleFlow.sample (core.py:570-582: Merge, GlowBlock.forward
 * calls, Squeeze.forward) for `nblocks` (1..32) consecutive GlowBlocks of one shape (affine/glow.py:72-84),
 * given in PROCESSING order.  A workgroup keeps its whole images in LDS across all blocks; z is read from
 * HBM once and written once; the glue is folded into that load / store:
 *   input : in_squeezed = 1: in0 = (B, C/4, 2H, 2W), read through Squeeze.inverse (reshape.py:122-128);
 *           otherwise channels [0, cin0) from in0 (B, cin0, H, W) and the rest from in1 (B, C - cin0, H, W)
 *           (Merge, reshape.py:88-100; cin0 = C: in1 unused);
 *   output: out_squeezed = 1: out0 = (B, C/4, 2H, 2W), written through Squeeze.forward (reshape.py:116-121);
 *           otherwise channels [0, cout0) to out0 and the rest to out1 (Split "channel", reshape.py:30-34).
 * block_table: DEVICE array of 4 nblocks pointers, block b at [4 b .. 4 b + 3]: nf_glow_convnet_pack of its conditioner
 *   (Cin = ceil(C/2), Cout = 2 floor(C/2)) for `layout` | mix_w (C, C) | mix_b (C) | mix_logdet (device scalar): with
 *   parameters frozen [Invertible1x1Conv, ActNorm] (mixing.py:88-133, normalization.py:19-39) is one per-pixel affine map
 *   m = mix_w z + mix_b with log|det| mix_logdet per pixel, composed by the caller per parameter version.  direction 1
 *   (GlowBlock.inverse, density): mix, conditioner on the mixed identity half, coupling inverse (coupling.py:150-171);
 *   direction 0 (GlowBlock.forward): conditioner on the raw identity half, coupling forward (:117-148), then the mix.
 * logdet (B) combined per `acc` with the sum of the blocks' log-dets (accumulated per image in block order).
 * No input tensor may alias an output tensor.  Returns NF_ERANGE for nblocks outside 1..32, NF_ENOTSUP when
 * the level's working set does not fit one workgroup's LDS.
 */
int nf_glow_level(const void *in0, const void *in1, int cin0, int in_squeezed, void *out0, void *out1, int cout0,
                  int out_squeezed, void *logdet, const void *block_table, int nblocks, int64_t B, int C, int H, int W,
                  int hidden, double leaky_slope, int scale_map, int direction, int acc, int layout, nf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Backward of the affine family for the training path (core.py:87-102 `loss.backward()`; the reference leaves these
 * layers to PyTorch autograd).  Vector-Jacobian products in closed form, one HBM pass each; cotangents gy (like z) and
 * gld (B, may be NULL = zeros) of the layer's outputs (y, log_det per sample); batch reductions in a fixed order.
 *   nf_masked_affine_bwd   : coupling.py:209-229 -> gz, gs, gt (like z; gs / gt NULL when s / t is NULL)
 *   nf_affine_coupling_bwd : coupling.py:117-171 with the channel split / merge -> gz (B, C, HW), gparam like param
 *   nf_actnorm_bwd         : coupling.py:38-54, per-channel s, t, log-det HW sum(s) returned per sample -> gz, gs (C), gt (C)
 *   nf_inv1x1_wgrad        : mixing.py:106-133, y = W z per pixel with per-pixel log|det| ldu -> gW (C, C) = sum over
 *                            pixels gy z^T, gldu (scalar) = HW sum gld; scratch of nf_inv1x1_wgrad_scratch_elems(B, C)
 *                            elements; C <= 64.  (gz = W^T gy is nf_inv1x1_conv_affine on the transposed matrix.)
 */
int nf_masked_affine_bwd(const void *z, const void *b, const void *s, const void *t, const void *gy, const void *gld,
                         void *gz, void *gs, void *gt, int64_t B, int64_t inner, int direction, int dtype,
                         nf_stream_t stream);
int nf_affine_coupling_bwd(const void *z, const void *param, const void *gy, const void *gld, void *gz, void *gparam,
                           int64_t B, int C, int c1, int flip, int64_t HW, int scale_map, int direction, int dtype,
                           nf_stream_t stream);
int64_t nf_actnorm_bwd_scratch_doubles(int64_t B, int C);   /* fp64 partial sums of nf_actnorm_bwd (caller-owned scratch) */
int nf_actnorm_bwd(const void *z, const void *s, const void *t, const void *gy, const void *gld, void *gz, void *gs,
                   void *gt, void *scratch, int64_t B, int C, int64_t HW, int direction, int dtype, nf_stream_t stream);
int64_t nf_inv1x1_wgrad_scratch_elems(int64_t B, int C);
int nf_inv1x1_wgrad(const void *z, const void *gy, const void *gld, void *gW, void *gldu, void *scratch, int64_t B, int C,
                    int64_t HW, int dtype, nf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * y_b = W x_b for every row of a row-major (B, D) float32 matrix, W (D, D) row-major, D <= 128 (nf_rows_matvec2: D <= 64), on exact-fp32 MFMA
 * (HBM-bound: one read and one write of the rows).  The batch-side products of LULinearPermute's backward
 * (mixing.py:535-563 under autograd: u = U x[perm], gu = L^T gy, gx = P U^T gu) with the permutation folded into W.
 * y must not alias x.
 */
int nf_rows_matvec(const void *x, const void *W, void *y, int64_t B, int D, nf_stream_t stream);
/* Same with a bias (D, may be NULL) and, when logdet != NULL, logdet[b] (acc) ld_sign * (*ld_const) for every row. */
int nf_rows_matvec_affine(const void *x, const void *W, const void *bias, void *y, void *logdet, const void *ld_const,
                          double ld_sign, int acc, int64_t B, int D, nf_stream_t stream);
/* Two chained products in one launch: u_b = W1 x_b (written when u != NULL), y_b = W2 u_b + bias (bias may be NULL), logdet as
 * nf_rows_matvec_affine.  LULinearPermute under autograd (mixing.py:535-563): forward u = U x[perm], y = L u + b; backward
 * gu = L^T gy, gx = P U^T gu -- u / gu are kept for the factor gradients and never leave the registers between the products. */
int nf_rows_matvec2(const void *x, const void *W1, const void *W2, const void *bias, void *u, void *y, void *logdet,
                    const void *ld_const, double ld_sign, int acc, int64_t B, int D, nf_stream_t stream);
/* LULinearPermute (mixing.py:402-473, :535-563) composed into one dense matrix per direction (fp64 arithmetic in one
 * workgroup; D <= 64, float32 parameters), once per parameter version: out = Wd (D, D) | Ws (D, D) | bias_d (D) |
 * bias_s (D) | log|det| (1) with  .inverse (density): y = Wd x + bias_d, log_det = +log|det|;  .forward (sample):
 * y = Ws x + bias_s, log_det = -log|det|.  The layer then is ONE nf_rows_matvec_affine launch (HBM-bound). */
/* Training-side helpers of LULinearPermute under autograd (mixing.py:402-473), float32, one launch each:
 * nf_lu_factors: out = L (D, D) | U (D, D) | Up (D, D; Up[:, perm[j]] = U[:, j], i.e. Up x = U x[perm]) | diag (D) |
 *   log|det| (1) | L^T (D, D) | Up^T (D, D): 5 D^2 + D + 1 floats;
 * nf_lu_param_grads: packed parameter gradients (lower_entries, upper_entries, unconstrained_upper_diag) from the dense
 *   factor gradients gL, gU (gU's columns read through perm when perm != NULL: gU = (gu^T x)[:, perm]) and the log-det
 *   cotangent gld (B floats, summed inside the launch; NULL = none), times `sign`. */
int nf_lu_factors(const int64_t *perm, const void *lower_entries, const void *upper_entries,
                  const void *unconstrained_upper_diag, double eps, void *out, int D, nf_stream_t stream);
/* nf_lu_factors for n_layers layers of one width in ONE launch; table (device memory): n_layers rows of 5 pointers -- perm,
 * lower_entries, upper_entries, unconstrained_upper_diag, out. */
int nf_lu_factors_multi(const void *table, int n_layers, double eps, int D, nf_stream_t stream);
int nf_lu_param_grads(const void *gL, const void *gU, const int64_t *perm, const void *gld, int64_t B,
                      const void *unconstrained_upper_diag, double eps, double sign, void *g_lower, void *g_upper, void *g_udiag,
                      int D, nf_stream_t stream);
int nf_lu_compose(const int64_t *perm, const void *lower_entries, const void *upper_entries,
                  const void *unconstrained_upper_diag, const void *bias, double eps, void *out, int D,
                  nf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Training path of the conditioner (nets/resnet.py:37-50 under core.py:87-102), float32.
 * (nf_rows_linear, round 2's single-panel kernel, was removed in round 3: the library GEMM was faster on every shape.)
 */

/* A whole plain residual block (resnet.py:37-50: x + W2 relu(W1 relu(x) + b1) + b2, H <= 128 columns, float32) or its
 * backward in ONE launch:   out1 = mask1( M1 pre1(in) + c1 ),   out2 = in + mask2( M2 pre2(out1) + c2 )
 * with pre = ReLU when relu1 / relu2, mask = zero where the mask source is <= 0 (NULL: none), Mi[j][k] = Mi[j * ldw + k] or,
 * transposed, Mi[k * ldw + j].  forward: in = x, M1 = W1, c1 = b1, M2 = W2, c2 = b2, relu1 = relu2 = 1 -> out1 = t (the
 * pre-activation kept for the backward), out2 = y.  backward: in = gy, M1 = W2 (transposed), mask1 = t, M2 = W1
 * (transposed), mask2 = x -> out1 = gt, out2 = gx.  Both weight panels stay in LDS, out1 feeds the second product from
 * registers; per row one read of `in` and the mask sources, one write of out1 and out2.  Row pitches in floats, multiples
 * of 4; 16-byte aligned origins. */
int nf_rows_block(const void *in, int64_t ldi, const void *M1, int64_t ldw1, int trans1, const void *c1, const void *mask1,
                  int64_t ldm1, void *out1, int64_t ldo1, const void *M2, int64_t ldw2, int trans2, const void *c2,
                  const void *mask2, int64_t ldm2, void *out2, int64_t ldo2, int64_t B, int H, int relu1, int relu2,
                  nf_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Squeeze (flows/reshape.py:116-128).  direction 0 = forward (C,H,W)->(C/4,2H,2W),
 * 1 = inverse (C,H,W)->(4C,H/2,W/2).  z, y contiguous NCHW with the shapes implied.
 */
int nf_squeeze(const void *z, void *y, int64_t B, int C, int H, int W, int direction, int dtype,
               nf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NF_MI355X_H */
