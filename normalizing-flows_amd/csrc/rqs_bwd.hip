// rqs_bwd.hip -- backward (vector-Jacobian product) of the NSF coupling transform, for training
// (the reference differentiates normflows/utils/splines.py:16-219 and nsf/coupling.py:71-128 with PyTorch autograd;
// there is no hand-written backward in the reference).
//
// For every spline element  (y, lad) = f(x; w[K], h[K], d[...])  and upstream gradients (gy, gl) this kernel returns
//   gx      = gy dy/dx + gl dlad/dx
//   g(w_i), g(h_i), g(d_j)  for the raw (unnormalised) parameters.
// The Jacobian of the closed-form bin evaluation w.r.t. (x, knot_lo, knot_hi, other_lo, other_hi, d0, d1) is obtained
// with forward-mode dual numbers (7 tangents) on exactly the forward arithmetic of common.hpp::rqs_eval_bin; the knots'
// dependence on the raw parameters is analytic:
//   knot_j = lo + (hi - lo) (j min + scale C_j),  C_j = sum_{i<j} softmax_i   (knot_0, knot_K pinned: no gradient)
//   d knot_j / d raw_i = (hi - lo) scale softmax_i ([i < j] - C_j) / wh_div
//   d deriv_j / d raw  = sigmoid(raw)  (softplus', torch threshold 20 -> 1)
// Bin selection is piecewise constant (no gradient), elements outside the tails pass gx = gy through.
//
// Transform half: per-element parameter gradients go to dcond (B, nT, M).  Identity half (batch-shared parameters):
// gradients are accumulated per workgroup in LDS and then atomically into (nI, K | K | nd) buffers that the caller
// zero-initialises (summation order across workgroups is not deterministic; fp32 atomics).
#include "common.hpp"
#include "rqs_bwd_common.hpp"

// (the unroll hints below are meant for the compile-time bin count, KT > 0; the build passes -Wno-pass-failed for the
// run-time-K instantiations, where they cannot apply)

namespace nf {

// Softmax probabilities of K logits read through `acc` (already divided by wh_div), written through `put`: the only
// transcendental pass over the widths / heights; everything downstream works on the probabilities.
// KT > 0: bin count known at compile time (loops unrolled: the LDS reads of a row become independent, pipelined accesses).
template <typename T, int KT = 0, typename Acc, typename Put>
__device__ __forceinline__ void rqs_softmax_probs(int Krt, const Acc &acc, const Put &put) {
    const int K = KT ? KT : Krt;
    T m = acc(0);
#pragma unroll
    for (int k = 1; k < K; ++k) m = M<T>::fmax(m, acc(k));
    T e[KT ? KT : 1];
    T s = T(0);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const T v = M<T>::exp(acc(k) - m);
        if (KT) e[k] = v;
        s += v;
    }
    const T inv = T(1) / s;
#pragma unroll
    for (int k = 0; k < K; ++k) put(k, (KT ? e[k] : M<T>::exp(acc(k) - m)) * inv);
}

// Gradient of one spline element w.r.t. x and its raw parameters, given the softmax probabilities pw(k), ph(k) of its
// widths / heights (rqs_softmax_probs) and the raw derivative logits dacc(j).  add_w(i, g), add_h(i, g), add_d(j, g)
// receive the gradients of the RAW parameters (already including 1 / wh_div); add_w(i) / add_h(i) are called after the
// last read of pw(i) / ph(i), before_d() after the last read of dacc: gradients may overwrite the parameters in place.
// Returns gx.
template <typename T, int KT = 0, typename PW, typename PH, typename DAcc, typename AW, typename AH, typename AD, typename ZD>
__device__ __forceinline__ T rqs_element_bwd(const RqsParams<T> &p, T x, T gy_up, T gl_up, const PW &pw, const PH &ph,
                                             const DAcc &dacc, bool inverse, const AW &add_w, const AH &add_h,
                                             const AD &add_d, const ZD &before_d) {
    // identity outside the tails (utils/splines.py:40-41), lad = 0; the per-feature branch (:48-57) leaves zeros there,
    // which do not depend on x
    if (!rqs_inside(p, x)) return p.dfull ? T(0) : gy_up;
    const int K = KT ? KT : p.K;
    // knots of both axes around the bin: knot_j = lo + (hi - lo)(j min + scale C_j), C_j = sum_{i<j} prob_i, ends pinned.
    // searched axis: widths for the forward spline, heights for the inverse.
    const T s_lo = inverse ? p.bottom : p.left, s_hi = inverse ? p.top : p.right;
    const T s_min = inverse ? p.min_h : p.min_w, s_scale = inverse ? p.scale_h : p.scale_w;
    int bin = 0;
    T sk_lo = s_lo, sk_hi = s_lo, Cs_lo = T(0), Cs_hi = T(0);
    {
        T cum = T(0), csm = T(0), knot = s_lo;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const T sm = inverse ? ph(k) : pw(k);
            cum += s_min + s_scale * sm;
            const T next = (k == K - 1) ? s_hi : (s_hi - s_lo) * cum + s_lo;
            if (k == 0 || x >= knot) { bin = k; sk_lo = knot; sk_hi = next; Cs_lo = csm; Cs_hi = csm + sm; }
            knot = next;
            csm += sm;
        }
    }
    const T o_lo = inverse ? p.left : p.bottom, o_hi = inverse ? p.right : p.top;
    const T o_min = inverse ? p.min_w : p.min_h, o_scale = inverse ? p.scale_w : p.scale_h;
    T ok_lo = o_lo, ok_hi = o_lo, Co_lo = T(0), Co_hi = T(0);
    {
        T cum = T(0), csm = T(0), knot = o_lo;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const T sm = inverse ? pw(k) : ph(k);
            cum += o_min + o_scale * sm;
            const T next = (k == K - 1) ? o_hi : (o_hi - o_lo) * cum + o_lo;
            if (k <= bin) { ok_lo = knot; ok_hi = next; Co_lo = csm; Co_hi = csm + sm; }   // same sums as a walk that stops at bin
            knot = next;
            csm += sm;
        }
    }
    const T cw_lo = inverse ? ok_lo : sk_lo, cw_hi = inverse ? ok_hi : sk_hi;
    const T ch_lo = inverse ? sk_lo : ok_lo, ch_hi = inverse ? sk_hi : ok_hi;
    const T Cw_lo = inverse ? Co_lo : Cs_lo, Cw_hi = inverse ? Co_hi : Cs_hi;
    const T Ch_lo = inverse ? Cs_lo : Co_lo, Ch_hi = inverse ? Cs_hi : Co_hi;
    const T r0 = rqs_dlogit(p, dacc, bin), r1 = rqs_dlogit(p, dacc, bin + 1);
    const T d0 = p.min_d + softplus(r0), d1 = p.min_d + softplus(r1);
    T gy[7], gl[7];
    rqs_eval_bin_dual<T>(x, cw_lo, cw_hi, ch_lo, ch_hi, d0, d1, inverse, gy, gl);
    T g[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) g[i] = gy_up * gy[i] + gl_up * gl[i];
    // ---- knots -> raw widths / heights: d knot_j / d raw_i = (hi - lo) scale prob_i ([i < j] - C_j) / wh_div ----
    const T g_cw_lo = bin == 0 ? T(0) : g[1], g_cw_hi = bin == K - 1 ? T(0) : g[2];  // pinned end knots are constants
    const T g_ch_lo = bin == 0 ? T(0) : g[3], g_ch_hi = bin == K - 1 ? T(0) : g[4];
    const T fw = (p.right - p.left) * p.scale_w / p.wh_div, fh = (p.top - p.bottom) * p.scale_h / p.wh_div;
    const T base_w = g_cw_lo * Cw_lo + g_cw_hi * Cw_hi, base_h = g_ch_lo * Ch_lo + g_ch_hi * Ch_hi;
#pragma unroll
    for (int i = 0; i < K; ++i) {
        const T tw = (i < bin ? g_cw_lo : T(0)) + (i < bin + 1 ? g_cw_hi : T(0)) - base_w;
        const T th = (i < bin ? g_ch_lo : T(0)) + (i < bin + 1 ? g_ch_hi : T(0)) - base_h;
        add_w(i, fw * pw(i) * tw);
        add_h(i, fh * ph(i) * th);
    }
    // ---- derivatives -> raw logits (softplus' = sigmoid; padded/edge logits are constants) ----
    auto raw_index = [&](int j) -> int {  // padded logit j -> raw index, or -1 for a constant
        if (p.dfull) {  // K + 1 raw logits per element, edges overridden per tails type (common.hpp::rqs_dlogit)
            if (p.tails == NF_TAILS_LINEAR) return (j == 0 || j == K) ? -1 : j;
            return j == K ? 0 : j;
        }
        if (p.tails == NF_TAILS_LINEAR) return (j == 0 || j == K) ? -1 : j - 1;
        if (p.tails == NF_TAILS_CIRCULAR) return j == K ? 0 : j;
        return j;
    };
    const int j0 = raw_index(bin), j1 = raw_index(bin + 1);
    before_d();   // every read of the raw logits is done: the caller may recycle their storage for the gradients
    if (j0 >= 0) add_d(j0, g[5] * (r0 > T(20) ? T(1) : sigmoid(r0)));
    if (j1 >= 0) add_d(j1, g[6] * (r1 > T(20) ? T(1) : sigmoid(r1)));
    return g[0];
}

// One lane per (sample, feature) element.  mode as in the forward: 0 density (both halves, forward splines),
// 1 identity half with the inverse spline, 2 transform half with the inverse spline.
// Tiles of TS samples: the tile's conditioner rows (TS*nT rows of M numbers, one contiguous span of `cond`) are staged
// into LDS with unit-stride loads, every row is turned into its gradient row IN PLACE (odd pitch: lane-per-row access is
// conflict-free) and leaves with unit-stride stores; a lane touching its 92-byte row directly in HBM costs ~4x.
template <typename T>
__global__ void __launch_bounds__(256)
rqs_coupling_bwd_kernel(const T *__restrict__ x, const T *__restrict__ gy, const T *__restrict__ gld,
                        const T *__restrict__ cond, const T *__restrict__ uw, const T *__restrict__ uh,
                        const T *__restrict__ ud, const int64_t *__restrict__ iidx, int nI,
                        const int64_t *__restrict__ tidx, int nT, int64_t B, int D, RqsParams<T> p, int mode,
                        T *__restrict__ gx, T *__restrict__ gcond, T *__restrict__ guw, T *__restrict__ guh,
                        T *__restrict__ gud, int TS, const int *__restrict__ tails_t, const T *__restrict__ bound_t,
                        const int *__restrict__ tails_i, const T *__restrict__ bound_i) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int K = p.K, nd = p.nd, M = 2 * K + nd;
    const int P = M | 1;                         // LDS row pitch (odd)
    T *s_acc = reinterpret_cast<T *>(smem_raw);  // nI * M block-local accumulators of the shared parameters
    const int PP = (2 * K) | 1;                  // odd pitch: lanes on different features hit different banks
    T *s_prob = s_acc + (size_t)nI * M;          // nI rows of 2K softmax probabilities of the shared widths / heights
    T *s_cond = s_prob + (size_t)nI * PP;        // TS*nT rows, pitch P
    const bool do_t = mode != NF_RQS_SAMPLE_IDENTITY, do_i = mode != NF_RQS_SAMPLE_TRANSFORM;
    const bool inverse = mode != NF_RQS_DENSITY;
    const bool has_uncond = uw != nullptr;
    const bool stage = do_t && nT > 0;
    const float invM = 1.0f / (float)M;
    if (do_i && has_uncond) {
        for (int i = threadIdx.x; i < nI * M; i += blockDim.x) s_acc[i] = T(0);
        // batch-shared parameters: their softmax is computed once per workgroup, not once per element
        for (int j = threadIdx.x; j < nI; j += blockDim.x) {
            const T *wj = uw + (size_t)j * K, *hj = uh + (size_t)j * K;
            T *pj = s_prob + (size_t)j * PP;
            rqs_softmax_probs<T>(K, [=](int k) { return wj[k]; }, [=](int k, T v) { pj[k] = v; });
            rqs_softmax_probs<T>(K, [=](int k) { return hj[k]; }, [=](int k, T v) { pj[K + k] = v; });
        }
    }
    __syncthreads();
    const int64_t ntiles = (B + TS - 1) / TS;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t bt = tile * TS;
        const int ts = (int)((B - bt) < TS ? (B - bt) : TS);
        if (stage) {
            const T *src = cond + bt * nT * (int64_t)M;
            const int n = ts * nT * M;
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                const int r = (int)(((float)i + 0.5f) * invM), k = i - r * M;  // exact for i < 2^22 / M
                s_cond[r * P + k] = src[i];
            }
            __syncthreads();
        }
        // transform elements and identity elements in two separate passes: within a pass every lane runs the same code
        // (one mixed pass made each wave execute both heavy branches)
#ifndef NF_BWD_ABL_NOT
        if (do_t) {
            const T div = p.wh_div;
            for (int el = threadIdx.x; el < ts * nT; el += blockDim.x) {
                const int bl = el / nT, f = el - bl * nT;
                const int64_t b = bt + bl;
                const int col = (int)tidx[f];
                T *row = s_cond + (size_t)el * P;   // parameters in, gradients out: the row is recycled in place
                const T xv = x[b * D + col];
                const RqsParams<T> pf = rqs_feature_params(p, tails_t, bound_t, f);
                if (!rqs_inside(pf, xv)) {           // identity (or zeros, per-feature branch) outside the tails: no parameter gradient
                    for (int k = 0; k < M; ++k) row[k] = T(0);
                    gx[b * D + col] = pf.dfull ? T(0) : gy[b * D + col];
                    continue;
                }
                // the row's raw widths / heights are replaced by their softmax probabilities (one exp pass)
                auto pw = [=](int k) { return row[k]; };
                auto ph = [=](int k) { return row[K + k]; };
                auto dacc = [=](int k) { return row[2 * K + k]; };
                auto aw = [=](int i, T g) { row[i] = g; };
                auto ah = [=](int i, T g) { row[K + i] = g; };
                auto ad = [=](int j, T g) { row[2 * K + j] = g; };
                auto zd = [=]() { for (int k = 2 * K; k < M; ++k) row[k] = T(0); };
                if (K == 8) {   // the default bin count: unrolled walks
                    rqs_softmax_probs<T, 8>(8, [=](int k) { return row[k] / div; }, [=](int k, T v) { row[k] = v; });
                    rqs_softmax_probs<T, 8>(8, [=](int k) { return row[8 + k] / div; }, [=](int k, T v) { row[8 + k] = v; });
                    gx[b * D + col] = rqs_element_bwd<T, 8>(pf, xv, gy[b * D + col], gld[b], pw, ph, dacc, inverse, aw, ah, ad, zd);
                } else {
                    rqs_softmax_probs<T>(K, [=](int k) { return row[k] / div; }, [=](int k, T v) { row[k] = v; });
                    rqs_softmax_probs<T>(K, [=](int k) { return row[K + k] / div; }, [=](int k, T v) { row[K + k] = v; });
                    gx[b * D + col] = rqs_element_bwd<T>(pf, xv, gy[b * D + col], gld[b], pw, ph, dacc, inverse, aw, ah, ad, zd);
                }
            }
        }
#endif
#ifndef NF_BWD_ABL_NOI
        if (do_i) {
            for (int el = threadIdx.x; el < ts * nI; el += blockDim.x) {
                const int bl = el / nI, j = el - bl * nI;
                const int64_t b = bt + bl;
                const int col = (int)iidx[j];
                if (!has_uncond) {
                    gx[b * D + col] = gy[b * D + col];
                    continue;
                }
                RqsParams<T> pu = rqs_feature_params(p, tails_i, bound_i, j);
                pu.wh_div = T(1);  // the unconditional transform is not scaled (nsf/coupling.py:224-232)
                const T *pj = s_prob + (size_t)j * PP, *dj = ud + (size_t)j * nd;
                T *acc = s_acc + (size_t)j * M;
                auto pw = [=](int k) { return pj[k]; };
                auto ph = [=](int k) { return pj[K + k]; };
                auto dacc = [=](int k) { return dj[k]; };
                auto aw = [=](int i, T g) { atomicAdd(acc + i, g); };
                auto ah = [=](int i, T g) { atomicAdd(acc + K + i, g); };
                auto ad = [=](int jj, T g) { atomicAdd(acc + 2 * K + jj, g); };
                if (K == 8)
                    gx[b * D + col] = rqs_element_bwd<T, 8>(pu, x[b * D + col], gy[b * D + col], gld[b], pw, ph, dacc, inverse,
                                                            aw, ah, ad, []() {});
                else
                    gx[b * D + col] = rqs_element_bwd<T>(pu, x[b * D + col], gy[b * D + col], gld[b], pw, ph, dacc, inverse,
                                                         aw, ah, ad, []() {});
            }
        }
#endif
        if (stage) {
            __syncthreads();
            T *dst = gcond + bt * nT * (int64_t)M;
            const int n = ts * nT * M;
            for (int i = threadIdx.x; i < n; i += blockDim.x) {
                const int r = (int)(((float)i + 0.5f) * invM), k = i - r * M;
                dst[i] = s_cond[r * P + k];
            }
            __syncthreads();
        }
    }
    __syncthreads();
    if (do_i && has_uncond) {
        for (int i = threadIdx.x; i < nI * M; i += blockDim.x) {
            const int j = i / M, c = i - j * M;
            const T v = s_acc[i];
            if (v != T(0)) {
                if (c < K) atomicAdd(guw + (size_t)j * K + c, v);
                else if (c < 2 * K) atomicAdd(guh + (size_t)j * K + (c - K), v);
                else atomicAdd(gud + (size_t)j * nd + (c - 2 * K), v);
            }
        }
    }
}


// Wave-private tiles (as rqs_coupling_wave_kernel): SPW samples per pass; conditioner rows in, gradient rows out through
// the wave's own LDS region with unit-stride global accesses, x / grad_y rows staged, gx rows written whole.  The shared
// (identity-half) parameters accumulate in workgroup LDS by atomics as in the tiled kernel.  float32, 8 bins, linear tails.
#ifndef NF_BWD_WAVE_OCC
#define NF_BWD_WAVE_OCC 2   // waves per SIMD the register allocation aims at (3: 168 VGPRs + spills, measured slower)
#endif
#ifndef NF_BWD_WAVE_WAVES
#define NF_BWD_WAVE_WAVES 8  // waves per workgroup: one workgroup per CU = fewest rounds of end-of-kernel global atomics
#endif
// CP: floats per (sample, transform feature) row of cond / grad_cond: 23 = the reference's layout (B, nT * 23); 24 = the
// padded rows the training variant of the fused kernel writes (rqs_fused.hip, 16-byte aligned: read and written as f32x4)
template <int CP>
__global__ void __launch_bounds__(64 * NF_BWD_WAVE_WAVES, NF_BWD_WAVE_OCC)
rqs_coupling_bwd_wave_kernel(const float *__restrict__ x, const float *__restrict__ gy, const float *__restrict__ gld,
                             const float *__restrict__ cond, const float *__restrict__ uw, const float *__restrict__ uh,
                             const float *__restrict__ ud, const int64_t *__restrict__ iidx, int nI,
                             const int64_t *__restrict__ tidx, int nT, int64_t B, int D, RqsParams<float> p, int mode,
                             float *__restrict__ gx, float *__restrict__ gcond, float *__restrict__ guw,
                             float *__restrict__ guh, float *__restrict__ gud, int SPW) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int K = F_K, M = F_M, nd = F_K - 1;
    const int PP = (2 * K) | 1;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nwv = blockDim.x >> 6;
    const int per_wave = SPW * nT * CP + 3 * SPW * D;
    float *s_acc = reinterpret_cast<float *>(smem_raw);           // nI * M
    float *s_prob = s_acc + (size_t)nI * M;                      // nI * PP
    float *s_ud = s_prob + (size_t)nI * PP;                      // nI * nd
    float *s_prm = s_ud + (size_t)nI * nd;                       // nI * 24: shared parameters in rqs_regs order (w, h x log2 e)
    float *w_cond = s_prm + (size_t)nI * 24 + (size_t)wid * per_wave;
    float *w_x = w_cond + (size_t)SPW * nT * CP, *w_gy = w_x + (size_t)SPW * D, *w_gx = w_gy + (size_t)SPW * D;
    // the identity half's per-lane running sums (shared parameters): 64 lanes x 25 floats per wave, conflict-free stride.  In
    // registers they cost 24 VGPRs across the whole loop, which spilled (19 VGPRs, 80 B of scratch per lane at 2 waves / SIMD)
    float *w_racc = s_prm + (size_t)nI * 24 + (size_t)nwv * per_wave + (size_t)wid * 64 * 25 + (size_t)lane * 25;
    int *s_iidx = reinterpret_cast<int *>(s_prm + (size_t)nI * 24 + (size_t)nwv * per_wave + (size_t)nwv * 64 * 25);
    int *s_tidx = s_iidx + nI;
    const bool do_t = mode != NF_RQS_SAMPLE_IDENTITY, do_i = mode != NF_RQS_SAMPLE_TRANSFORM;
    const bool inverse = mode != NF_RQS_DENSITY;
    const bool has_uncond = uw != nullptr;

    for (int j = tid; j < nI; j += blockDim.x) s_iidx[j] = (int)iidx[j];
    for (int j = tid; j < nT; j += blockDim.x) s_tidx[j] = (int)tidx[j];
    if (do_i && has_uncond) {
        for (int i = tid; i < nI * M; i += blockDim.x) s_acc[i] = 0.0f;
        for (int i = tid; i < nI * nd; i += blockDim.x) s_ud[i] = ud[i];
        for (int i = tid; i < nI * 24; i += blockDim.x) {
            const int j = i / 24, c = i - 24 * j;
            s_prm[i] = c < K ? uw[(size_t)j * K + c] * 1.44269504088896340736f
                             : (c < 2 * K ? uh[(size_t)j * K + c - K] * 1.44269504088896340736f
                                          : (c < M ? ud[(size_t)j * nd + c - 2 * K] : 0.0f));
        }
        for (int j = tid; j < nI; j += blockDim.x) {
            const float *wj = uw + (size_t)j * K, *hj = uh + (size_t)j * K;
            float *pj = s_prob + (size_t)j * PP;
            rqs_softmax_probs<float, 8>(8, [=](int k) { return wj[k]; }, [=](int k, float v) { pj[k] = v; });
            rqs_softmax_probs<float, 8>(8, [=](int k) { return hj[k]; }, [=](int k, float v) { pj[K + k] = v; });
        }
    }
    __syncthreads();

    const float sc = 1.44269504088896340736f / p.wh_div, inv_div = 1.0f / p.wh_div;
    const int64_t gw = (int64_t)blockIdx.x * nwv + wid, GW = (int64_t)gridDim.x * nwv;
    // identity half: when one pass covers its elements with one lane each (SPW nI <= 64), a lane keeps the same shared
    // feature for the whole launch: its parameter gradients accumulate in REGISTERS and meet the other lanes once, at the end
    const bool reg_acc = do_i && has_uncond && SPW * nI <= 64;
#pragma unroll
    for (int k = 0; k < 25; ++k) w_racc[k] = 0.0f;
    for (int64_t b0 = gw * SPW; b0 < B; b0 += GW * SPW) {
        const int ns = (int)((B - b0) < SPW ? (B - b0) : SPW);
        for (int i = lane; i < ns * D; i += 64) {
            w_x[i] = x[b0 * D + i];
            w_gy[i] = gy[b0 * D + i];
            w_gx[i] = 0.0f;
        }
        if (do_t) {
            const float *src = cond + b0 * (int64_t)nT * CP;
            for (int i = lane; i < ns * nT * CP; i += 64) w_cond[i] = src[i];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#ifndef NF_BWD_ABL_NOT
        if (do_t) {
            for (int e = lane; e < ns * nT; e += 64) {
                const int s_ = e / nT, j = e - s_ * nT, col = s_tidx[j];
                float *row = w_cond + (size_t)e * CP;
                float prm[24], g[24];
                if constexpr (CP == 24) {
#pragma unroll
                    for (int q = 0; q < 6; ++q) {
                        const f32x4 v = *reinterpret_cast<const f32x4 *>(row + 4 * q);
#pragma unroll
                        for (int r = 0; r < 4; ++r) prm[4 * q + r] = q < 4 ? v[r] * sc : v[r];
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < 2 * K; ++k) prm[k] = row[k] * sc;
#pragma unroll
                    for (int k = 2 * K; k < M; ++k) prm[k] = row[k];
                }
                prm[M] = 0.0f;
                const float xv = w_x[s_ * D + col], gyv = w_gy[s_ * D + col], gl = gld[b0 + s_];
                const float gxv = inverse ? rqs_regs_bwd<true>(p, xv, prm, gyv, gl, g, inv_div)
                                          : rqs_regs_bwd<false>(p, xv, prm, gyv, gl, g, inv_div);
                if constexpr (CP == 24) {
                    g[M] = 0.0f;
#pragma unroll
                    for (int q = 0; q < 6; ++q)
                        *reinterpret_cast<f32x4 *>(row + 4 * q) = f32x4{g[4 * q], g[4 * q + 1], g[4 * q + 2], g[4 * q + 3]};
                } else {
#pragma unroll
                    for (int k = 0; k < M; ++k) row[k] = g[k];
                }
                w_gx[s_ * D + col] = gxv;
            }
        }
#endif
#ifndef NF_BWD_ABL_NOI
        if (do_i) {
            for (int e = lane; e < ns * nI; e += 64) {
                const int s_ = e / nI, j = e - s_ * nI, col = s_iidx[j];
                const float xv = w_x[s_ * D + col], gyv = w_gy[s_ * D + col];
                if (!has_uncond) {
                    w_gx[s_ * D + col] = gyv;
                    continue;
                }
                if (reg_acc) {
                    float prm[24], g[24];
                    const float *pr = s_prm + (size_t)j * 24;
#pragma unroll
                    for (int k = 0; k < 24; ++k) prm[k] = pr[k];
                    const float gl = gld[b0 + s_];
                    w_gx[s_ * D + col] = inverse ? rqs_regs_bwd<true>(p, xv, prm, gyv, gl, g, 1.0f)
                                                 : rqs_regs_bwd<false>(p, xv, prm, gyv, gl, g, 1.0f);
#pragma unroll
                    for (int k = 0; k < M; ++k) w_racc[k] += g[k];      // lane-private LDS words: plain read-add-write
                    continue;
                }
                RqsParams<float> pu = p;
                pu.wh_div = 1.0f;
                const float *pj = s_prob + (size_t)j * PP, *dj = s_ud + (size_t)j * nd;
                float *accj = s_acc + (size_t)j * M;
                auto pwf = [=](int k) { return pj[k]; };
                auto phf = [=](int k) { return pj[K + k]; };
                auto dacc = [=](int k) { return dj[k]; };
                auto aw = [=](int i, float v) { atomicAdd(accj + i, v); };
                auto ah = [=](int i, float v) { atomicAdd(accj + K + i, v); };
                auto ad = [=](int jj, float v) { atomicAdd(accj + 2 * K + jj, v); };
                w_gx[s_ * D + col] = rqs_element_bwd<float, 8>(pu, xv, gyv, gld[b0 + s_], pwf, phf, dacc, inverse, aw, ah, ad,
                                                               []() {});
            }
        }
#endif
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (do_t) {
            float *dst = gcond + b0 * (int64_t)nT * CP;
            for (int i = lane; i < ns * nT * CP; i += 64) dst[i] = w_cond[i];
        }
        if (mode == NF_RQS_DENSITY) {
            for (int i = lane; i < ns * D; i += 64) gx[b0 * D + i] = w_gx[i];
        } else {
            const int nown = do_t ? nT : nI;
            const int *own = do_t ? s_tidx : s_iidx;
            for (int e = lane; e < ns * nown; e += 64) {
                const int s_ = e / nown, j = e - s_ * nown;
                gx[(b0 + s_) * D + own[j]] = w_gx[s_ * D + own[j]];
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (reg_acc && lane < SPW * nI) {
        float *accj = s_acc + (size_t)(lane % nI) * M;
#pragma unroll
        for (int k = 0; k < M; ++k) atomicAdd(accj + k, w_racc[k]);
    }
    __syncthreads();
    if (do_i && has_uncond) {
        for (int i = tid; i < nI * M; i += blockDim.x) {
            const int j = i / M, c = i - j * M;
            const float v = s_acc[i];
            if (v != 0.0f) {
                if (c < K) atomicAdd(guw + (size_t)j * K + c, v);
                else if (c < 2 * K) atomicAdd(guh + (size_t)j * K + (c - K), v);
                else atomicAdd(gud + (size_t)j * nd + (c - 2 * K), v);
            }
        }
    }
}


// ---- the same kernel, software-pipelined, for the benchmark layer (nI = nT = 32, D = 64, density direction, shared identity
// parameters, even B) --------------------------------------------------------------------------------------------------------
// The wave-private kernel above runs load -> compute -> store per pass with nothing but the SIMD's second wave to hide the
// round trips: measured 120 us of staging + 66 us of arithmetic = 186 us, the sum.  Here every pass's inputs (conditioner
// rows, x, grad_y, grad_logdet of its 2 samples) arrive by LDS-DMA (global_load_lds, 16 bytes per lane, no registers) into
// the buffer the PREVIOUS pass is not using, issued before the current pass's arithmetic; a counted s_waitcnt (memory
// operations retire in order: everything younger than the current pass's loads may stay outstanding -- the previous pass's
// S stores and the next pass's L loads) replaces the full drain.  No ordinary global load is left inside the loop, so the
// compiler inserts no vmcnt wait of its own.  Gradient rows leave through 16-byte stores.
template <int CP>
__global__ void __launch_bounds__(64 * NF_BWD_WAVE_WAVES, NF_BWD_WAVE_OCC)
rqs_coupling_bwd_pipe_kernel(const float *__restrict__ x, const float *__restrict__ gy, const float *__restrict__ gld,
                             const float *__restrict__ cond, const float *__restrict__ uw, const float *__restrict__ uh,
                             const float *__restrict__ ud, const int64_t *__restrict__ iidx,
                             const int64_t *__restrict__ tidx, int64_t B, RqsParams<float> p, float *__restrict__ gx,
                             float *__restrict__ gcond, float *__restrict__ guw, float *__restrict__ guh,
                             float *__restrict__ gud) {
    typedef __attribute__((address_space(3))) void *lds_ptr;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int K = F_K, M = F_M, nd = F_K - 1, nI = 32, nT = 32, D = 64, SPW = 2;
    constexpr int CONDF = SPW * nT * CP;            // 1536 | 1472 floats of conditioner rows per pass
    constexpr int XOFF = 1536, GLOFF = XOFF + 2 * SPW * D, BUF = GLOFF + 4;     // cond | x | grad_y | grad_logdet (2 + pad)
    constexpr int NC = (CONDF + 255) / 256;         // 1 KB DMA instructions for the rows
    constexpr int LD_N = NC + 2, ST_N = NC + 1;     // VMEM instructions per pass: loads (rows, x|gy, gld), stores (rows, gx)
    constexpr int PER_WAVE = 2 * BUF + SPW * D;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *s_acc = reinterpret_cast<float *>(smem_raw);           // nI * M: the workgroup's sums of the shared parameters
    float *s_prm = s_acc + nI * M;                               // nI * 24: shared parameters in rqs_regs order (w, h x log2 e)
    float *wbase = s_prm + nI * 24 + (size_t)wid * PER_WAVE;
    float *w_gx = wbase + 2 * BUF;     // per wave: the pass's gx rows
    int *s_iidx = reinterpret_cast<int *>(s_prm + nI * 24 + (size_t)NF_BWD_WAVE_WAVES * PER_WAVE);
    int *s_tidx = s_iidx + nI;
    for (int j = tid; j < nI; j += blockDim.x) s_iidx[j] = (int)iidx[j];
    for (int j = tid; j < nT; j += blockDim.x) s_tidx[j] = (int)tidx[j];
    for (int i = tid; i < nI * M; i += blockDim.x) s_acc[i] = 0.0f;
    for (int i = tid; i < nI * 24; i += blockDim.x) {
        const int j = i / 24, c = i - 24 * j;
        s_prm[i] = c < K ? uw[(size_t)j * K + c] * 1.44269504088896340736f
                         : (c < 2 * K ? uh[(size_t)j * K + c - K] * 1.44269504088896340736f
                                      : (c < M ? ud[(size_t)j * nd + c - 2 * K] : 0.0f));
    }
    __syncthreads();

    const float sc = 1.44269504088896340736f / p.wh_div, inv_div = 1.0f / p.wh_div;
    const int64_t gw = (int64_t)blockIdx.x * NF_BWD_WAVE_WAVES + wid, GW = (int64_t)gridDim.x * NF_BWD_WAVE_WAVES;
    auto issue = [&](int64_t b0, float *buf) {
        const float *csrc = cond + b0 * (int64_t)(nT * CP);
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            if (q < NC - 1 || q * 256 + lane * 4 < CONDF)
                __builtin_amdgcn_global_load_lds(csrc + q * 256 + lane * 4, (lds_ptr)(buf + q * 256), 16, 0, 0);
        }
        const float *xs = lane < 32 ? x + b0 * D + lane * 4 : gy + b0 * D + (lane - 32) * 4;
        __builtin_amdgcn_global_load_lds(xs, (lds_ptr)(buf + XOFF), 16, 0, 0);
        if (lane < SPW) __builtin_amdgcn_global_load_lds(gld + b0 + lane, (lds_ptr)(buf + GLOFF), 4, 0, 0);
    };
    const int s_ = lane >> 5, j = lane & 31;
    const int col_t = s_tidx[j], col_i = s_iidx[j];
    int64_t b0 = gw * SPW;
    float *bufc = wbase, *bufn = wbase + BUF;
    if (b0 < B) issue(b0, bufc);
    bool first = true;
    float racc[24];     // (this specialisation has the registers for it: 172 VGPRs without)
#pragma unroll
    for (int k = 0; k < 24; ++k) racc[k] = 0.0f;
    for (; b0 < B; b0 += GW * SPW) {
        const int64_t b1 = b0 + GW * SPW;
        const bool more = b1 < B;
        if (more) issue(b1, bufn);
        if (first) {
            if (more) NF_WAIT_VMCNT(LD_N);
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            if (more) NF_WAIT_VMCNT(LD_N + ST_N);
            else NF_WAIT_VMCNT(ST_N);
        }
        first = false;
        __builtin_amdgcn_wave_barrier();
        const float gl = bufc[GLOFF + s_];
        {   // transform half: lane = (sample, transform feature); gradient row written over the parameter row
            float *row = bufc + (size_t)lane * CP;
            float prm[24], g[24];
            if constexpr (CP == 24) {
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(row + 4 * q);
#pragma unroll
                    for (int r = 0; r < 4; ++r) prm[4 * q + r] = q < 4 ? v[r] * sc : v[r];
                }
            } else {
#pragma unroll
                for (int k = 0; k < 2 * K; ++k) prm[k] = row[k] * sc;
#pragma unroll
                for (int k = 2 * K; k < M; ++k) prm[k] = row[k];
            }
            prm[M] = 0.0f;
            const float xv = bufc[XOFF + s_ * D + col_t], gyv = bufc[XOFF + SPW * D + s_ * D + col_t];
#ifdef NF_BWD_ABL_NOMATH      // ablation (wrong results): the pass without the spline arithmetic = staging + traffic only
            float gxv = gyv + gl + xv;
#pragma unroll
            for (int k = 0; k < 24; ++k) g[k] = prm[k] * gyv;
#else
            const float gxv = rqs_regs_bwd<false>(p, xv, prm, gyv, gl, g, inv_div);
#endif
            if constexpr (CP == 24) {
                g[M] = 0.0f;
#pragma unroll
                for (int q = 0; q < 6; ++q)
                    *reinterpret_cast<f32x4 *>(row + 4 * q) = f32x4{g[4 * q], g[4 * q + 1], g[4 * q + 2], g[4 * q + 3]};
            } else {
                // rows of 23 floats: the neighbour lane's parameters start right behind this lane's; every lane has read its
                // row before any lane writes (the loads above are complete: g depends on all of them)
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int k = 0; k < M; ++k) row[k] = g[k];
            }
            w_gx[s_ * D + col_t] = gxv;
        }
        {   // identity half: shared parameters, gradients summed per lane
            float prm[24], g[24];
            const float *pr = s_prm + (size_t)j * 24;
#pragma unroll
            for (int k = 0; k < 24; ++k) prm[k] = pr[k];
            const float xv = bufc[XOFF + s_ * D + col_i], gyv = bufc[XOFF + SPW * D + s_ * D + col_i];
#if defined(NF_BWD_ABL_NOMATH) || defined(NF_BWD_ABL_NOIDENT)     // ablation: without the identity half's arithmetic
            w_gx[s_ * D + col_i] = gyv + xv;
#pragma unroll
            for (int k = 0; k < 24; ++k) g[k] = prm[k] * gyv;
#else
            w_gx[s_ * D + col_i] = rqs_regs_bwd<false>(p, xv, prm, gyv, gl, g, 1.0f);
#endif
#pragma unroll
            for (int k = 0; k < M; ++k) racc[k] += g[k];      // a lane keeps its feature for the whole launch: register sums
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        {
            float *dst = gcond + b0 * (int64_t)(nT * CP);
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                if (q < NC - 1 || q * 256 + lane * 4 < CONDF)
                    *reinterpret_cast<f32x4 *>(dst + q * 256 + lane * 4) = *reinterpret_cast<const f32x4 *>(bufc + q * 256 + lane * 4);
            }
            if (lane < 32) *reinterpret_cast<f32x4 *>(gx + b0 * D + lane * 4) = *reinterpret_cast<const f32x4 *>(w_gx + lane * 4);
        }
        __builtin_amdgcn_wave_barrier();
        float *t = bufc; bufc = bufn; bufn = t;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {
        float *accj = s_acc + (size_t)j * M;
#pragma unroll
        for (int k = 0; k < M; ++k) atomicAdd(accj + k, racc[k]);
    }
    __syncthreads();
    for (int i = tid; i < nI * M; i += blockDim.x) {
        const int jj = i / M, c = i - jj * M;
        const float v = s_acc[i];
        if (v != 0.0f) {
            if (c < K) atomicAdd(guw + (size_t)jj * K + c, v);
            else if (c < 2 * K) atomicAdd(guh + (size_t)jj * K + (c - K), v);
            else atomicAdd(gud + (size_t)jj * nd + (c - 2 * K), v);
        }
    }
}

}  // namespace nf

using namespace nf;

template <typename T>
static int launch_bwd(const void *x, const void *gy, const void *gld, const void *cond, const void *uw, const void *uh,
                      const void *ud, const int64_t *iidx, int nI, const int64_t *tidx, int nT, int64_t B, int D,
                      const RqsParams<T> &p, int mode, void *gx, void *gcond, void *guw, void *guh, void *gud,
                      hipStream_t st, const int32_t *tails_t = nullptr, const void *bound_t = nullptr,
                      const int32_t *tails_i = nullptr, const void *bound_i = nullptr) {
    const int M = 2 * p.K + p.nd, P = M | 1;
    const bool stage = mode != NF_RQS_SAMPLE_IDENTITY && nT > 0;
    auto lds_bytes = [&](int ts) {
        return ((size_t)nI * M + (size_t)nI * ((2 * p.K) | 1) + (stage ? (size_t)ts * nT * P : 0)) * sizeof(T) + 16;
    };
    const int nmax = nT > nI ? nT : nI;
    int TS = nmax > 0 && 512 / nmax > 0 ? 512 / nmax : 1;   // samples per tile: ~two elements per lane in each pass
    while (TS > 1 && lds_bytes(TS) > 64 * 1024) TS >>= 1;
    const size_t lds = lds_bytes(TS);
    if (lds > 150 * 1024) return NF_ENOTSUP;
    static LdsOptIn opted = {};
    if (opt_in_lds(reinterpret_cast<const void *>(&rqs_coupling_bwd_kernel<T>), lds, opted) != NF_OK) return NF_ENOTSUP;
    const int grid = grid_for((B + TS - 1) / TS, 1, 512);   // resident workgroups; each one ends with 23 nI global atomics
    hipLaunchKernelGGL(rqs_coupling_bwd_kernel<T>, dim3(grid), dim3(256), lds, st, (const T *)x, (const T *)gy,
                       (const T *)gld, (const T *)cond, (const T *)uw, (const T *)uh, (const T *)ud, iidx, nI, tidx, nT,
                       B, D, p, mode, (T *)gx, (T *)gcond, (T *)guw, (T *)guh, (T *)gud, TS, (const int *)tails_t,
                       (const T *)bound_t, (const int *)tails_i, (const T *)bound_i);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// Launch of the wave-private kernel (float32, 8 bins, linear tails); NF_ENOTSUP when its LDS does not fit.
template <int CP>
static int launch_bwd_wave(const void *x, const void *grad_y, const void *grad_logdet, const void *cond, const void *uw,
                           const void *uh, const void *ud, const int64_t *identity_idx, int nI, const int64_t *transform_idx,
                           int nT, int64_t B, int D, const nf::RqsParams<float> &p, int mode, void *grad_x, void *grad_cond,
                           void *grad_uw, void *grad_uh, void *grad_ud, hipStream_t st) {
    using namespace nf;
#ifndef NF_BWD_NO_PIPE
    if (nI == 32 && nT == 32 && D == 64 && mode == NF_RQS_DENSITY && uw && (B & 1) == 0 && B >= 2 &&
        ((((uintptr_t)x | (uintptr_t)grad_y | (uintptr_t)cond | (uintptr_t)grad_cond | (uintptr_t)grad_x) & 15) == 0) &&
        (((uintptr_t)grad_logdet & 3) == 0)) {
        constexpr int BUFp = 1536 + 2 * 2 * 64 + 4, PERW = 2 * BUFp + 2 * 64;
        const size_t ldsp = ((size_t)32 * F_M + 32 * 24 + (size_t)NF_BWD_WAVE_WAVES * PERW) * sizeof(float) + 64 * sizeof(int) + 16;
        static LdsOptIn opted_p = {};
        if (opt_in_lds(reinterpret_cast<const void *>(&rqs_coupling_bwd_pipe_kernel<CP>), ldsp, opted_p) == NF_OK) {
            const int64_t nw2 = B / 2, gq2 = (nw2 + NF_BWD_WAVE_WAVES - 1) / NF_BWD_WAVE_WAVES;
            const int grid2 = (int)(gq2 < 2048 / NF_BWD_WAVE_WAVES ? gq2 : 2048 / NF_BWD_WAVE_WAVES);
            hipLaunchKernelGGL(rqs_coupling_bwd_pipe_kernel<CP>, dim3(grid2), dim3(64 * NF_BWD_WAVE_WAVES), ldsp, st,
                               (const float *)x, (const float *)grad_y, (const float *)grad_logdet, (const float *)cond,
                               (const float *)uw, (const float *)uh, (const float *)ud, identity_idx, transform_idx, B, p,
                               (float *)grad_x, (float *)grad_cond, (float *)grad_uw, (float *)grad_uh, (float *)grad_ud);
            NF_CHECK_LAUNCH();
            return NF_OK;
        }
    }
#endif
    const int nmax = nT > nI ? nT : nI;
    int SPW = nmax > 0 ? 64 / nmax : 1;
    if (SPW < 1) SPW = 1;
    const size_t per_wave = (size_t)SPW * nT * CP + 3 * (size_t)SPW * D;
    const size_t ldsw = ((size_t)nI * F_M + (size_t)nI * ((2 * F_K) | 1) + (size_t)nI * (F_K - 1) + (size_t)nI * 24 +
                         NF_BWD_WAVE_WAVES * (per_wave + 64 * 25)) * sizeof(float) + (size_t)(nI + nT) * sizeof(int) + 16;
    if (ldsw > 150 * 1024) return NF_ENOTSUP;
    // 8 waves per CU are resident (register-bound): one 8-wave workgroup per CU; more workgroups only add rounds
    // of global atomics on the shared parameters' 23 nI addresses at the end of each
    const int64_t nwaves = (B + SPW - 1) / SPW, gq = (nwaves + NF_BWD_WAVE_WAVES - 1) / NF_BWD_WAVE_WAVES;
    const int grid = (int)(gq < 2048 / NF_BWD_WAVE_WAVES ? gq : 2048 / NF_BWD_WAVE_WAVES);
    static LdsOptIn opted = {};   // one per CP instantiation
    if (opt_in_lds(reinterpret_cast<const void *>(&rqs_coupling_bwd_wave_kernel<CP>), ldsw, opted) != NF_OK) return NF_ENOTSUP;
    hipLaunchKernelGGL(rqs_coupling_bwd_wave_kernel<CP>, dim3(grid), dim3(64 * NF_BWD_WAVE_WAVES), ldsw, st, (const float *)x,
                       (const float *)grad_y, (const float *)grad_logdet, (const float *)cond, (const float *)uw,
                       (const float *)uh, (const float *)ud, identity_idx, nI, transform_idx, nT, B, D, p, mode,
                       (float *)grad_x, (float *)grad_cond, (float *)grad_uw, (float *)grad_uh, (float *)grad_ud, SPW);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

extern "C" int nf_rqs_coupling_bwd_ft(const void *x, const void *grad_y, const void *grad_logdet, const void *cond,
                                      const void *uw, const void *uh, const void *ud, const int64_t *identity_idx, int nI,
                                      const int64_t *transform_idx, int nT, int64_t B, int D, int K, int tails,
                                      double tail_bound, double min_bin_width, double min_bin_height,
                                      double min_derivative, double wh_div, int mode, void *grad_x, void *grad_cond,
                                      void *grad_uw, void *grad_uh, void *grad_ud, int dtype, const int32_t *tails_t,
                                      const void *bound_t, const int32_t *tails_i, const void *bound_i,
                                      nf_stream_t stream) {
    if (K < 1 || K > NF_MAX_BINS) return NF_ERANGE;
    if (tails < NF_TAILS_NONE || tails > NF_TAILS_FEATURE) return NF_EINVAL;
    if (B < 0 || D < 1 || nI < 0 || nT < 0 || nI + nT != D || mode < 0 || mode > 2) return NF_EINVAL;
    if (tails == NF_TAILS_FEATURE && ((nT && mode != NF_RQS_SAMPLE_IDENTITY && !tails_t) ||
                                      (nI && uw && mode != NF_RQS_SAMPLE_TRANSFORM && !tails_i))) return NF_EFAULT;
    if (tails != NF_TAILS_FEATURE && (tails_t || tails_i)) return NF_EINVAL;
    if (tails == NF_TAILS_NONE && (bound_t || bound_i)) return NF_EINVAL;
    if (B == 0) return NF_OK;
    if (!x || !grad_y || !grad_logdet || !grad_x) return NF_EFAULT;
    if (mode != NF_RQS_SAMPLE_IDENTITY && nT && (!cond || !grad_cond || !transform_idx)) return NF_EFAULT;
    if (mode != NF_RQS_SAMPLE_TRANSFORM && nI && !identity_idx) return NF_EFAULT;
    if (uw && (!uh || !ud || !grad_uw || !grad_uh || !grad_ud)) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == NF_F32) {
        auto p = make_rqs_params<float>(K, tails, tail_bound, 0, 1, 0, 1, min_bin_width, min_bin_height, min_derivative,
                                        wh_div);
        if (K == F_K && tails == NF_TAILS_LINEAR && !tails_t && !bound_t && !tails_i && !bound_i) {
            // default parametrisation: wave-private tiles + register-resident element routine
            const int rc = launch_bwd_wave<F_M>(x, grad_y, grad_logdet, cond, uw, uh, ud, identity_idx, nI, transform_idx, nT, B,
                                                D, p, mode, grad_x, grad_cond, grad_uw, grad_uh, grad_ud, st);
            if (rc != NF_ENOTSUP) return rc;
        }
        return launch_bwd<float>(x, grad_y, grad_logdet, cond, uw, uh, ud, identity_idx, nI, transform_idx, nT, B, D, p,
                                 mode, grad_x, grad_cond, grad_uw, grad_uh, grad_ud, st, tails_t, bound_t, tails_i, bound_i);
    }
    if (dtype == NF_F64) {
        auto p = make_rqs_params<double>(K, tails, tail_bound, 0, 1, 0, 1, min_bin_width, min_bin_height,
                                         min_derivative, wh_div);
        return launch_bwd<double>(x, grad_y, grad_logdet, cond, uw, uh, ud, identity_idx, nI, transform_idx, nT, B, D,
                                  p, mode, grad_x, grad_cond, grad_uw, grad_uh, grad_ud, st, tails_t, bound_t, tails_i,
                                  bound_i);
    }
    return NF_ENOTSUP;
}

extern "C" int nf_rqs_coupling_bwd(const void *x, const void *grad_y, const void *grad_logdet, const void *cond,
                                   const void *uw, const void *uh, const void *ud, const int64_t *identity_idx, int nI,
                                   const int64_t *transform_idx, int nT, int64_t B, int D, int K, int tails,
                                   double tail_bound, double min_bin_width, double min_bin_height, double min_derivative,
                                   double wh_div, int mode, void *grad_x, void *grad_cond, void *grad_uw, void *grad_uh,
                                   void *grad_ud, int dtype, nf_stream_t stream) {
    if (tails == NF_TAILS_FEATURE) return NF_EINVAL;
    return nf_rqs_coupling_bwd_ft(x, grad_y, grad_logdet, cond, uw, uh, ud, identity_idx, nI, transform_idx, nT, B, D, K,
                                  tails, tail_bound, min_bin_width, min_bin_height, min_derivative, wh_div, mode, grad_x,
                                  grad_cond, grad_uw, grad_uh, grad_ud, dtype, nullptr, nullptr, nullptr, nullptr, stream);
}

// The density-direction backward on cond / grad_cond rows of 24 floats per transform feature (23 parameters + 1 pad): the
// layout the training variant of the fused kernel writes (nf_rqs_fused_train_fwd).  float32, 8 bins, linear tails.
extern "C" int nf_rqs_coupling_bwd_p24(const void *x, const void *grad_y, const void *grad_logdet, const void *cond24,
                                       const void *uw, const void *uh, const void *ud, const int64_t *identity_idx, int nI,
                                       const int64_t *transform_idx, int nT, int64_t B, int D, double tail_bound,
                                       double min_bin_width, double min_bin_height, double min_derivative, double wh_div,
                                       void *grad_x, void *grad_cond24, void *grad_uw, void *grad_uh, void *grad_ud,
                                       nf_stream_t stream) {
    using namespace nf;
    if (B < 0 || D < 1 || nI < 0 || nT < 1 || nI + nT != D) return NF_EINVAL;
    if (min_bin_width * F_K > 1.0 || min_bin_height * F_K > 1.0) return NF_EINVAL;
    if (B == 0) return NF_OK;
    if (!x || !grad_y || !grad_logdet || !grad_x || !cond24 || !grad_cond24 || !transform_idx || (nI && !identity_idx)) return NF_EFAULT;
    if (uw && (!uh || !ud || !grad_uw || !grad_uh || !grad_ud)) return NF_EFAULT;
    if (((uintptr_t)cond24 | (uintptr_t)grad_cond24) & 15) return NF_EINVAL;
    auto p = make_rqs_params<float>(F_K, NF_TAILS_LINEAR, tail_bound, 0, 1, 0, 1, min_bin_width, min_bin_height, min_derivative,
                                    wh_div);
    return launch_bwd_wave<24>(x, grad_y, grad_logdet, cond24, uw, uh, ud, identity_idx, nI, transform_idx, nT, B, D, p,
                               NF_RQS_DENSITY, grad_x, grad_cond24, grad_uw, grad_uh, grad_ud, (hipStream_t)stream);
}
