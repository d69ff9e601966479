// train_bwd.hip -- the backward pass of a benchmark-shaped NSF coupling layer (CoupledRationalQuadraticSpline: D = 64, hidden 128,
// 8 bins, linear tails, >= 1 residual block) behind ONE C-ABI call: what `loss.backward()` (core.py:87-102) does for
// nsf/coupling.py:83-98 + nets/resnet.py:37-50, 92-104 + utils/splines.py:16-219.
//
// Round 6.  Until now the host issued the layer's backward kernel by kernel and every kernel that leaves partial tiles was followed
// by its own reduction launch: per layer final_bwd + its reduction, the ring weight gradient + its reduction, two residual-block
// launches + three reductions -- 11 launches, 6 of them 4-10 us reductions on grids too small to pull their partial tiles at more
// than ~2 TB/s, each behind a dependent kernel boundary.  Here the four heavy kernels run back to back and ALL reductions of the
// layer are one launch (nf::layer_reduce_kernel: the routines of train_reduce.hpp, one 256-element group (16-byte loads), one spline
// feature or the cotangent sum per block; the stand-alone reductions' summation order, so the gradients are bit-identical to the
// launch-by-launch path): 5 launches.  nf_pair_train_bwd adds the adjacent LULinearPermute's composed backward (one pass over the
// rows, its partial tiles in the same reduction launch, the factors' gradients on the parameter side): 7 launches for the pair.
// Gradients go straight to the caller's destinations (e.g. views of one flat gradient buffer: no per-parameter tensors).
#include "train_reduce.hpp"
#include <string.h>

extern "C" {
int nf_final_bwd_partials(int64_t B);
int nf_final_bwd(const void *x, const void *grad_y, const void *grad_logdet, const void *cond24, const void *w_t, const void *wpack,
                 void *grad_x, void *grad_cond24, void *grad_h, void *partials, int mask_parity, int64_t B, int D, int hidden,
                 int num_blocks, int K, double tail_bound, double min_bin_width, double min_bin_height, double min_derivative,
                 nf_stream_t stream);
int nf_linear_wgrad_chunks(int64_t B, int M, int N);
int nf_linear_wgrad_partials(const void *dY, const void *X, void *scratch, int64_t B, int M, int N, int relu_x, int want_bias,
                             nf_stream_t stream);
int64_t nf_resblock_bwd_scratch_floats(int64_t B, int with_init);
int nf_resblock_bwd_grid(int64_t B);
int nf_resblock_bwd_partials(const void *gh, const void *t, const void *h_in, const void *W1, const void *W2, void *gh_in,
                             const void *x, const void *wfull, void *gx, void *scratch, int64_t B, int H, int D,
                             nf_stream_t stream);
int nf_lu_bwd_composed_grid(int64_t B);
int nf_lu_bwd_composed_partials(const void *g, const void *x, const void *Wd, void *gx, void *scratch, int64_t B, int D,
                                nf_stream_t stream);
int nf_lu_param_grads_composed(const void *dWd, const void *Lm, const void *Um, const int64_t *perm, const void *gld, int64_t B,
                               const void *unconstrained_upper_diag, double eps, void *g_lower, void *g_upper, void *g_udiag, int D,
                               nf_stream_t stream);
}

namespace nf {

__global__ void __launch_bounds__(64 * RL)
layer_reduce_kernel(ReduceJobs J) {
    __shared__ f32x4 sm4[RL][64];
    const int b = blockIdx.x;
    if (b >= J.nblocks + F_NI) {        // the (B) vector sum: one block, fixed order
        float *smf = reinterpret_cast<float *>(&sm4[0][0]);
        const int tid = threadIdx.x;
        float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f;
        const int64_t n4 = (reinterpret_cast<uintptr_t>(J.vsum) & 15) == 0 ? J.vsum_n / 4 : 0;
        const f32x4 *g4 = reinterpret_cast<const f32x4 *>(J.vsum);
#pragma unroll 16
        for (int64_t i = tid; i < n4; i += 64 * RL) {
            const f32x4 v = g4[i];
            p0 += v[0]; p1 += v[1]; p2 += v[2]; p3 += v[3];
        }
        for (int64_t i = 4 * n4 + tid; i < J.vsum_n; i += 64 * RL) p0 += J.vsum[i];
        const float tot = block_sum((p0 + p1) + (p2 + p3), smf);
        if (tid == 0) *J.vsum_out = tot;
        return;
    }
    if (b >= J.nblocks) {       // the batch-shared spline parameters: one block per identity feature
        float *smf = reinterpret_cast<float *>(&sm4[0][0]);
        final_bwd_reduce_feature(J.fb_part, J.fb_nparts, J.uw, J.uh, J.ud, J.guw, J.guh, J.gud, J.p, b - J.nblocks,
                                 reinterpret_cast<float(*)[24]>(smf), smf + 512);
        return;
    }
    int j = 0;           // (block ranges are not in job order: order_jobs() puts the longest chunk loops first)
#pragma unroll
    for (int i = 1; i < RJ_MAX; ++i)
        if (i < J.nj && b >= J.j[i].block0 && b < J.j[i].block0 + (int)((J.j[i].n + 255) >> 8)) j = i;
    const ReduceJob &q = J.j[j];
    wgrad_reduce_group4(q.part, q.dW, q.db, q.nW, q.n, q.stride, q.chunks, q.N, q.skip_every, q.colmap, q.Nout,
                        (int64_t)(b - q.block0) * 256, sm4);
}

static void add_job(ReduceJobs &J, const float *part, float *dW, float *db, int64_t nW, int M, int chunks, int N, int skip_every,
                    const int *colmap, int Nout) {
    ReduceJob &q = J.j[J.nj++];
    q.part = part; q.dW = dW; q.db = db;
    q.nW = nW; q.n = nW + (db ? M : 0); q.stride = nW + M;
    q.colmap = colmap; q.chunks = chunks; q.N = N; q.skip_every = skip_every; q.Nout = Nout;
    q.block0 = J.nblocks;
    J.nblocks += (int)((q.n + 255) / 256);
}

// Blocks of the jobs with the most chunks first: a residual block's partial tiles come from 256 workgroups, the final layer's from
// 82 row chunks -- a block of the former runs three times as long, and issued last (job order) they were the launch's tail on a chip
// that holds 512 of the ~730 blocks at a time.  Longest first, the short blocks fill in behind them.  (Which block sums what does not
// change a bit of the result.)
static void order_jobs(ReduceJobs &J) {
    int idx[RJ_MAX];
    for (int i = 0; i < J.nj; ++i) idx[i] = i;
    for (int i = 1; i < J.nj; ++i)          // insertion sort, stable, by chunks descending
        for (int k = i; k > 0 && J.j[idx[k]].chunks > J.j[idx[k - 1]].chunks; --k) { const int t = idx[k]; idx[k] = idx[k - 1]; idx[k - 1] = t; }
    int b0 = 0;
    for (int i = 0; i < J.nj; ++i) {
        J.j[idx[i]].block0 = b0;
        b0 += (int)((J.j[idx[i]].n + 255) / 256);
    }
}

constexpr int TB_MP = 24 * F_NI;       // 768: rows of the final layer on the 24-float pitch of cond24

}  // namespace nf

using namespace nf;

// floats of `scratch` for nf_coupling_train_bwd: gradient rows of the final layer (B x 768), two ping-pong hidden-gradient
// tensors (B x 128) and every kernel's partial tiles
extern "C" int64_t nf_coupling_train_bwd_scratch_floats(int64_t B, int num_blocks) {
    if (B < 64 || B % 64 || num_blocks < 1 || 2 * num_blocks + 2 > RJ_MAX) return NF_ENOTSUP;
    const int chunks = nf_linear_wgrad_chunks(B, TB_MP, F_H);
    if (chunks < 0) return chunks;
    int64_t n = B * (int64_t)TB_MP + 2 * B * (int64_t)F_H + (int64_t)nf_final_bwd_partials(B) * FBR_PART +
                (int64_t)chunks * ((int64_t)TB_MP * F_H + TB_MP);
    for (int b = 0; b < num_blocks; ++b) n += nf_resblock_bwd_scratch_floats(B, b == 0);
    return n;
}

// The four passes over the rows; the reduction jobs are appended to J (not launched), *s_end = the first unused float of scratch.
static int coupling_bwd_core(const void *x, const void *grad_y, const void *grad_logdet, const void *cond24, const void *acts,
                             const void *w_t, const void *wpack, const void *wfull_t, const void *const *w_blocks,
                             const void *uw, const void *uh, const void *ud, const void *col_map, int n_cols, void *grad_x,
                             void *g_w0, void *g_b0, void *g_wf, void *g_bf, void *g_uw, void *g_uh, void *g_ud,
                             void *const *g_blocks, void *scratch, int mask_parity, int64_t B, int D, int hidden,
                             int num_blocks, int K, double tail_bound, double min_bin_width, double min_bin_height,
                             double min_derivative, nf_stream_t stream, ReduceJobs &J, float **s_end) {
    if (D != F_D || hidden != F_H || K != F_K || num_blocks < 1 || 2 * num_blocks + 2 > RJ_MAX) return NF_ENOTSUP;
    if (B < 64 || B % 64) return NF_ENOTSUP;
    if (mask_parity != 0 && mask_parity != 1) return NF_EINVAL;
    if (col_map && (n_cols < 1 || n_cols > F_D)) return NF_EINVAL;
    if (!x || !grad_y || !grad_logdet || !cond24 || !acts || !w_t || !wpack || !wfull_t || !w_blocks || !uw || !uh || !ud || !grad_x ||
        !g_w0 || !g_b0 || !g_wf || !g_bf || !g_uw || !g_uh || !g_ud || !g_blocks || !scratch)
        return NF_EFAULT;
    for (int i = 0; i < 2 * num_blocks; ++i)
        if (!w_blocks[i]) return NF_EFAULT;
    for (int i = 0; i < 4 * num_blocks; ++i)
        if (!g_blocks[i]) return NF_EFAULT;
    const int nparts = nf_final_bwd_partials(B), chunks = nf_linear_wgrad_chunks(B, TB_MP, F_H), grid = nf_resblock_bwd_grid(B);
    if (nparts < 0 || chunks < 0 || grid < 0) return NF_ENOTSUP;
    if ((uintptr_t)scratch & 15) return NF_EINVAL;      // (every region below is a multiple of 4 floats: 16-byte reduction loads)
    float *s = (float *)scratch;
    float *gcond = s;                       s += B * (int64_t)TB_MP;
    float *ghA = s;                         s += B * (int64_t)F_H;
    float *ghB = s;                         s += B * (int64_t)F_H;
    float *fb_part = s;                     s += (int64_t)nparts * FBR_PART;
    float *ring_part = s;                   s += (int64_t)chunks * ((int64_t)TB_MP * F_H + TB_MP);
    const float *A = (const float *)acts;
    const int64_t act = B * (int64_t)F_H;

    // spline backward + the final layer's input gradient: gx, gradient rows, gh (the last block's output gradient)
    int rc = nf_final_bwd(x, grad_y, grad_logdet, cond24, w_t, wpack, grad_x, gcond, ghA, fb_part, mask_parity, B, D, hidden,
                          num_blocks, K, tail_bound, min_bin_width, min_bin_height, min_derivative, stream);
    if (rc != NF_OK) return rc;
    // the final layer's weight / bias gradient: dW = gcond^T h_last (pad rows dropped in the reduction)
    rc = nf_linear_wgrad_partials(gcond, A + (int64_t)(2 * num_blocks) * act, ring_part, B, TB_MP, F_H, 0, 1, stream);
    if (rc != NF_OK) return rc;

    J.nj = 0;
    J.nblocks = 0;
    add_job(J, ring_part, (float *)g_wf, (float *)g_bf, (int64_t)TB_MP * F_H, TB_MP, chunks, F_H, 24, nullptr, 0);
    float *gh = ghA, *gh_next = ghB;
    const int64_t bstride = (int64_t)F_H * F_H + F_H;
    for (int b = num_blocks - 1; b >= 0; --b) {
        float *part = s;
        s += nf_resblock_bwd_scratch_floats(B, b == 0);
        const float *t = A + (int64_t)(2 * b + 1) * act, *h_in = A + (int64_t)(2 * b) * act;
        if (b == 0)
            rc = nf_resblock_bwd_partials(gh, t, h_in, w_blocks[0], w_blocks[1], nullptr, x, wfull_t, grad_x, part, B, F_H, F_D, stream);
        else
            rc = nf_resblock_bwd_partials(gh, t, h_in, w_blocks[2 * b], w_blocks[2 * b + 1], gh_next, nullptr, nullptr, nullptr, part, B,
                                          F_H, F_D, stream);
        if (rc != NF_OK) return rc;
        // (dW2, db2) then (dW1, db1); g_blocks: gw1, gb1, gw2, gb2 per block
        add_job(J, part, (float *)g_blocks[4 * b + 2], (float *)g_blocks[4 * b + 3], (int64_t)F_H * F_H, F_H, grid, F_H, 0, nullptr, 0);
        add_job(J, part + (int64_t)grid * bstride, (float *)g_blocks[4 * b], (float *)g_blocks[4 * b + 1], (int64_t)F_H * F_H, F_H, grid,
                F_H, 0, nullptr, 0);
        if (b == 0)
            add_job(J, part + 2 * (int64_t)grid * bstride, (float *)g_w0, (float *)g_b0, (int64_t)F_H * F_D, F_H, grid, F_D, 0,
                    (const int *)col_map, col_map ? n_cols : 0);
        float *tmp = gh; gh = gh_next; gh_next = tmp;
    }
    J.fb_part = fb_part; J.fb_nparts = nparts;
    J.uw = (const float *)uw; J.uh = (const float *)uh; J.ud = (const float *)ud;
    J.guw = (float *)g_uw; J.guh = (float *)g_uh; J.gud = (float *)g_ud;
    J.p = make_rqs_params<float>(K, NF_TAILS_LINEAR, tail_bound, 0, 1, 0, 1, min_bin_width, min_bin_height, min_derivative, 1.0);
    J.vsum = nullptr; J.vsum_out = nullptr; J.vsum_n = 0;
    *s_end = s;
    return NF_OK;
}

extern "C" int nf_coupling_train_bwd(const void *x, const void *grad_y, const void *grad_logdet, const void *cond24, const void *acts,
                                     const void *w_t, const void *wpack, const void *wfull_t, const void *const *w_blocks,
                                     const void *uw, const void *uh, const void *ud, const void *col_map, int n_cols, void *grad_x,
                                     void *g_w0, void *g_b0, void *g_wf, void *g_bf, void *g_uw, void *g_uh, void *g_ud,
                                     void *const *g_blocks, void *scratch, int mask_parity, int64_t B, int D, int hidden,
                                     int num_blocks, int K, double tail_bound, double min_bin_width, double min_bin_height,
                                     double min_derivative, nf_stream_t stream) {
    ReduceJobs J;
    float *s_end = nullptr;
    const int rc = coupling_bwd_core(x, grad_y, grad_logdet, cond24, acts, w_t, wpack, wfull_t, w_blocks, uw, uh, ud, col_map, n_cols,
                                     grad_x, g_w0, g_b0, g_wf, g_bf, g_uw, g_uh, g_ud, g_blocks, scratch, mask_parity, B, D, hidden,
                                     num_blocks, K, tail_bound, min_bin_width, min_bin_height, min_derivative, stream, J, &s_end);
    if (rc != NF_OK) return rc;
    order_jobs(J);
    hipLaunchKernelGGL(layer_reduce_kernel, dim3(J.nblocks + F_NI), dim3(64 * RL), 0, (hipStream_t)stream, J);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// ---- a [CoupledRQS, LULinearPermute] pair: the coupling's backward, then the composed LU's (autograd.PairTrainFn) -------------------
// floats of `scratch` for nf_pair_train_bwd: nf_coupling_train_bwd's + the coupling's input gradient (B x 64), the LU pass's
// partial tiles and the reduced dW_d (64 x 64)
extern "C" int64_t nf_pair_train_bwd_scratch_floats(int64_t B, int num_blocks) {
    const int64_t n = nf_coupling_train_bwd_scratch_floats(B, num_blocks);
    if (n < 0) return n;
    const int g = nf_lu_bwd_composed_grid(B);
    if (g < 0) return g;
    return n + B * (int64_t)F_D + (int64_t)g * (F_D * F_D + F_D) + F_D * F_D + 4;
}

// nf_coupling_train_bwd on the coupling of a pair (its input = xlu, the LU's output saved by nf_rqs_fused_train_pair_fwd), then
// nf_lu_bwd_composed_partials on its input gradient, ONE reduction launch for the partial tiles of both layers, and
// nf_lu_param_grads_composed: seven launches.  x_in: the LU's input rows; Wd: nf_lu_pack_train_multi's (64, 64); Lm, Um: the dense
// factors of nf_lu_factors[_multi]; grad_x_in (B, 64): the pair's input gradient; g_lower, g_upper, g_udiag, g_lbias: written.
// What the pair's last two launches need (nf_pair_train_bwd_tail): the reduction's job list and the LU parameter kernel's arguments.
// A plain host record: the caller keeps it between the two calls (NF_PAIR_TAIL_BYTES of include/nf_mi355x.h).
namespace nf {
struct PairTail {
    ReduceJobs J;
    const float *dWd, *gl_sum, *Lm, *Um, *udiag;
    const int64_t *perm;
    float *g_lower, *g_upper, *g_udiag;
    double lu_eps;
    unsigned magic;
};
constexpr unsigned PAIR_TAIL_MAGIC = 0x50544c36u;
static_assert(sizeof(PairTail) <= 2048, "NF_PAIR_TAIL_BYTES");
}  // namespace nf

static int pair_tail_launch(const nf::PairTail &T, nf_stream_t stream) {
    using namespace nf;
#ifdef NF_ABL_NO_REDUCE       // timing-only ablation: what the step costs without the two launches that only produce parameter gradients
    return NF_OK;
#endif
    hipLaunchKernelGGL(layer_reduce_kernel, dim3(T.J.nblocks + F_NI + 1), dim3(64 * RL), 0, (hipStream_t)stream, T.J);
    NF_CHECK_LAUNCH();
    // (gld = the one-element sum, B = 1: the parameter kernel's own summation loop degenerates to a single load)
    return nf_lu_param_grads_composed(T.dWd, T.Lm, T.Um, T.perm, T.gl_sum, 1, T.udiag, T.lu_eps, T.g_lower, T.g_upper, T.g_udiag, F_D, stream);
}

static int pair_train_bwd_head(const void *x_in, const void *xlu, const void *grad_y, const void *grad_logdet, const void *cond24,
                               const void *acts, const void *w_t, const void *wpack, const void *wfull_t,
                               const void *const *w_blocks, const void *uw, const void *uh, const void *ud, const void *col_map,
                               int n_cols, const void *Wd, const void *Lm, const void *Um, const int64_t *perm,
                               const void *unconstrained_upper_diag, double lu_eps, void *grad_x_in, void *g_lower, void *g_upper,
                               void *g_udiag, void *g_lbias, void *g_w0, void *g_b0, void *g_wf, void *g_bf, void *g_uw, void *g_uh,
                               void *g_ud, void *const *g_blocks, void *scratch, int mask_parity, int64_t B, int D, int hidden,
                               int num_blocks, int K, double tail_bound, double min_bin_width, double min_bin_height,
                               double min_derivative, nf_stream_t stream, nf::PairTail &T) {
    using namespace nf;
    if (!x_in || !Wd || !Lm || !Um || !perm || !unconstrained_upper_diag || !grad_x_in || !g_lower || !g_upper || !g_udiag || !g_lbias ||
        !scratch)
        return NF_EFAULT;
    const int64_t n0 = nf_coupling_train_bwd_scratch_floats(B, num_blocks);
    if (n0 < 0) return (int)n0;
    float *gxl = (float *)scratch + n0;                       // the coupling's input gradient = the LU's output gradient
    float *lu_part = gxl + B * (int64_t)F_D;
    const int lgrid = nf_lu_bwd_composed_grid(B);
    if (lgrid < 0) return NF_ENOTSUP;
    float *dWd = lu_part + (int64_t)lgrid * (F_D * F_D + F_D);
    ReduceJobs &J = T.J;
    float *s_end = nullptr;
    int rc = coupling_bwd_core(xlu, grad_y, grad_logdet, cond24, acts, w_t, wpack, wfull_t, w_blocks, uw, uh, ud, col_map, n_cols, gxl,
                               g_w0, g_b0, g_wf, g_bf, g_uw, g_uh, g_ud, g_blocks, scratch, mask_parity, B, D, hidden, num_blocks, K,
                               tail_bound, min_bin_width, min_bin_height, min_derivative, stream, J, &s_end);
    if (rc != NF_OK) return rc;
    if (s_end > gxl) return NF_EINVAL;
    rc = nf_lu_bwd_composed_partials(gxl, x_in, Wd, grad_x_in, lu_part, B, F_D, stream);
    if (rc != NF_OK) return rc;
    add_job(J, lu_part, dWd, (float *)g_lbias, (int64_t)F_D * F_D, F_D, lgrid, F_D, 0, nullptr, 0);
    float *gl_sum = dWd + F_D * F_D;          // the log-det cotangent's sum, by one more block of the reduction launch
    J.vsum = (const float *)grad_logdet; J.vsum_out = gl_sum; J.vsum_n = B;
    order_jobs(J);
    T.dWd = dWd; T.gl_sum = gl_sum; T.Lm = (const float *)Lm; T.Um = (const float *)Um; T.udiag = (const float *)unconstrained_upper_diag;
    T.perm = perm; T.g_lower = (float *)g_lower; T.g_upper = (float *)g_upper; T.g_udiag = (float *)g_udiag; T.lu_eps = lu_eps;
    T.magic = PAIR_TAIL_MAGIC;
    return NF_OK;
}

extern "C" int nf_pair_train_bwd(const void *x_in, const void *xlu, const void *grad_y, const void *grad_logdet, const void *cond24,
                                 const void *acts, const void *w_t, const void *wpack, const void *wfull_t,
                                 const void *const *w_blocks, const void *uw, const void *uh, const void *ud, const void *col_map,
                                 int n_cols, const void *Wd, const void *Lm, const void *Um, const int64_t *perm,
                                 const void *unconstrained_upper_diag, double lu_eps, void *grad_x_in, void *g_lower, void *g_upper,
                                 void *g_udiag, void *g_lbias, void *g_w0, void *g_b0, void *g_wf, void *g_bf, void *g_uw, void *g_uh,
                                 void *g_ud, void *const *g_blocks, void *scratch, int mask_parity, int64_t B, int D, int hidden,
                                 int num_blocks, int K, double tail_bound, double min_bin_width, double min_bin_height,
                                 double min_derivative, nf_stream_t stream) {
    nf::PairTail T;
    const int rc = pair_train_bwd_head(x_in, xlu, grad_y, grad_logdet, cond24, acts, w_t, wpack, wfull_t, w_blocks, uw, uh, ud, col_map,
                                       n_cols, Wd, Lm, Um, perm, unconstrained_upper_diag, lu_eps, grad_x_in, g_lower, g_upper, g_udiag,
                                       g_lbias, g_w0, g_b0, g_wf, g_bf, g_uw, g_uh, g_ud, g_blocks, scratch, mask_parity, B, D, hidden,
                                       num_blocks, K, tail_bound, min_bin_width, min_bin_height, min_derivative, stream, T);
    return rc != NF_OK ? rc : pair_tail_launch(T, stream);
}

// The same backward in two calls (round 6, late): `_head` issues the five launches the NEXT pair's backward waits for (the coupling's
// four passes and the composed LU's pass: the pair's input gradient) and leaves, in `tail` (host memory, NF_PAIR_TAIL_BYTES), what the
// last two launches need; `_tail` issues those -- the one reduction launch and the LU's factor gradients, which only produce PARAMETER
// gradients -- on any stream the caller has ordered behind `_head`'s.  Off the critical path they cost 46 us per pair of pure
// latency-bound work; on a side stream they run under the next pair's MFMA-bound kernels (autograd.PairTrainFn: fork by event, joined
// before anything reads a gradient).  `scratch` and every gradient destination stay in use until the tail has run.
extern "C" int nf_pair_train_bwd_head(const void *x_in, const void *xlu, const void *grad_y, const void *grad_logdet, const void *cond24,
                                      const void *acts, const void *w_t, const void *wpack, const void *wfull_t,
                                      const void *const *w_blocks, const void *uw, const void *uh, const void *ud, const void *col_map,
                                      int n_cols, const void *Wd, const void *Lm, const void *Um, const int64_t *perm,
                                      const void *unconstrained_upper_diag, double lu_eps, void *grad_x_in, void *g_lower, void *g_upper,
                                      void *g_udiag, void *g_lbias, void *g_w0, void *g_b0, void *g_wf, void *g_bf, void *g_uw,
                                      void *g_uh, void *g_ud, void *const *g_blocks, void *scratch, int mask_parity, int64_t B, int D,
                                      int hidden, int num_blocks, int K, double tail_bound, double min_bin_width, double min_bin_height,
                                      double min_derivative, void *tail, nf_stream_t stream) {
    if (!tail) return NF_EFAULT;
    nf::PairTail T;
    const int rc = pair_train_bwd_head(x_in, xlu, grad_y, grad_logdet, cond24, acts, w_t, wpack, wfull_t, w_blocks, uw, uh, ud, col_map,
                                       n_cols, Wd, Lm, Um, perm, unconstrained_upper_diag, lu_eps, grad_x_in, g_lower, g_upper, g_udiag,
                                       g_lbias, g_w0, g_b0, g_wf, g_bf, g_uw, g_uh, g_ud, g_blocks, scratch, mask_parity, B, D, hidden,
                                       num_blocks, K, tail_bound, min_bin_width, min_bin_height, min_derivative, stream, T);
    if (rc != NF_OK) return rc;
    memcpy(tail, &T, sizeof(T));
    return NF_OK;
}

extern "C" int nf_pair_train_bwd_tail(const void *tail, nf_stream_t stream) {
    if (!tail) return NF_EFAULT;
    nf::PairTail T;
    memcpy(&T, tail, sizeof(T));
    if (T.magic != nf::PAIR_TAIL_MAGIC) return NF_EINVAL;
    return pair_tail_launch(T, stream);
}
