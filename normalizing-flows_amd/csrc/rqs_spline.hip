// rqs_spline.hip -- rational-quadratic spline kernels (element-wise and NSF coupling with conditioner
// outputs materialised in HBM).  gfx950 only.
//
// Reference behaviour: normflows/utils/splines.py:11-219, normflows/flows/neural_spline/coupling.py:71-128,
// :150-164, :221-253, :329-362.  The arithmetic lives in common.hpp (rqs_element / rqs_eval_table).
//
// nf_rqs_coupling data movement per workgroup (256 threads, TS samples per tile):
//   HBM -> LDS : the tile's conditioner rows, TS*nT*M contiguous floats, coalesced 16-B loads, w/h
//                pre-divided by sqrt(hidden) on the way in; row pitch padded to an odd number of words
//                so that the per-lane walk over one row's K bins is bank-conflict free;
//   HBM -> LDS : the tile's x rows (TS*D contiguous);
//   once per workgroup: knot tables of the batch-shared unconditional spline (nI * 3(K+1) words);
//   compute    : one lane per (sample, feature) element; results land in the LDS y tile (scatter by the
//                identity/transform index lists happens in LDS, not in HBM);
//   LDS -> HBM : y rows, coalesced; per-sample log-det = row sum of the per-element values.
// Algorithmic HBM bytes per sample (fp32, D=64, nT=32, K=8): 256 (x) + 2944 (cond) + 256 (y) + 8 (ld rmw).
#include "common.hpp"
#include "fused_common.hpp"
#include <type_traits>

namespace nf {

template <typename T>
__global__ void __launch_bounds__(256)
rqs_spline_kernel(const T *__restrict__ x, const T *__restrict__ w, int64_t ldw, const T *__restrict__ h,
                  int64_t ldh, const T *__restrict__ d, int64_t ldd, T *__restrict__ y, T *__restrict__ lad_out,
                  int64_t N, RqsParams<T> p, int inverse) {
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
        const T *wn = w + n * ldw, *hn = h + n * ldh, *dn = d + n * ldd;
        const T div = p.wh_div;
        auto wacc = [=](int k) { return wn[k] / div; };
        auto hacc = [=](int k) { return hn[k] / div; };
        auto dacc = [=](int j) { return dn[j]; };
        T yy, ll;
        rqs_element<T>(p, x[n], wacc, hacc, dacc, inverse != 0, yy, ll);
        y[n] = yy;
        if (lad_out) lad_out[n] = ll;
    }
}

// The same element-wise spline with the parameter rows staged through LDS: a thread reading its own K-float rows of w / h / d touches
// 3 x 64 different cache lines per wave instruction (the kernel above: 434 us for 65 536 x 64 elements, 1.0 TB/s); here the 256
// elements of a pass bring their rows in with unit-stride loads (rows of one tensor are contiguous: ld == row length) into LDS rows of
// an ODD pitch (a lane then walks its own row conflict-free), and the outputs leave unit-stride as before.  FAST (float, 8 bins,
// linear tails -- the default parametrisation): the element runs on registers through the branch-free routine of the fused kernels
// (fused_common.hpp rqs_regs: hardware exp2 / log2 / rcp, <= 1-2 ulp each, the path nf_rqs_coupling's wave kernel takes for the same
// parametrisation); otherwise the libm-accurate rqs_element walks the LDS rows.
template <typename T, bool FAST>
__global__ void __launch_bounds__(256)
rqs_spline_staged_kernel(const T *__restrict__ x, const T *__restrict__ w, const T *__restrict__ h, const T *__restrict__ d,
                         T *__restrict__ y, T *__restrict__ lad_out, int64_t N, RqsParams<T> p, int inverse, int pitch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *rows = reinterpret_cast<T *>(smem_raw);          // [256][pitch]: w (K) | h (K) | d (nd) | pad
    const int K = p.K, nd = p.dfull ? K + 1 : p.nd, tid = threadIdx.x;
    const int64_t npass = (N + 255) / 256;
    for (int64_t pass = blockIdx.x; pass < npass; pass += gridDim.x) {
        const int64_t n0 = pass * 256;
        const int cnt = (int)((N - n0) < 256 ? (N - n0) : 256);
        // unit-stride copies; (row, column) of a thread's next element follow incrementally (one division per tensor and pass)
        auto stage = [&](const T *__restrict__ src, int len, int off) {
            const int total = cnt * len, dr = 256 / len, dc = 256 % len;
            int row = tid / len, col = tid % len;
            const T *sp = src + n0 * len;
#pragma unroll 4
            for (int i = tid; i < total; i += 256) {
                rows[row * pitch + off + col] = sp[i];
                col += dc;
                row += dr;
                if (col >= len) {
                    col -= len;
                    ++row;
                }
            }
        };
        stage(w, K, 0);
        stage(h, K, K);
        stage(d, nd, 2 * K);
        __syncthreads();
        if (tid < cnt) {
            const T *r = rows + tid * pitch;
            T yy, ll;
            if constexpr (FAST) {
                float prm[3 * F_K];
                const float sc = 1.44269504088896340736f / p.wh_div;       // rqs_regs takes exp2 of pre-scaled widths / heights
#pragma unroll
                for (int k = 0; k < 2 * F_K; ++k) prm[k] = r[k] * sc;
#pragma unroll
                for (int k = 0; k < F_K - 1; ++k) prm[2 * F_K + k] = r[2 * F_K + k];
                prm[3 * F_K - 1] = 0.0f;
                if (inverse) rqs_regs<true>(p, x[n0 + tid], prm, yy, ll);
                else rqs_regs<false>(p, x[n0 + tid], prm, yy, ll);
            } else {
                const T div = p.wh_div;
                auto wacc = [=](int k) { return r[k] / div; };
                auto hacc = [=](int k) { return r[K + k] / div; };
                auto dacc = [=](int j) { return r[2 * K + j]; };
                rqs_element<T>(p, x[n0 + tid], wacc, hacc, dacc, inverse != 0, yy, ll);
            }
            y[n0 + tid] = yy;
            if (lad_out) lad_out[n0 + tid] = ll;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
rqs_coupling_kernel(const T *__restrict__ x, T *__restrict__ y, T *__restrict__ logdet, const T *__restrict__ cond,
                    const T *__restrict__ uw, const T *__restrict__ uh, const T *__restrict__ ud,
                    const int64_t *__restrict__ iidx, int nI, const int64_t *__restrict__ tidx, int nT, int64_t B,
                    int D, RqsParams<T> p, int mode, int acc, int TS, int Mp, const int *__restrict__ tails_t,
                    const T *__restrict__ bound_t, const int *__restrict__ tails_i, const T *__restrict__ bound_i) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int K = p.K;
    const int M = 2 * K + p.nd;
    const int TW = 3 * (K + 1);  // table words per identity feature
    T *s_cond = reinterpret_cast<T *>(smem_raw);               // TS * nT * Mp
    T *s_x = s_cond + (size_t)TS * nT * Mp;                    // TS * D
    T *s_y = s_x + (size_t)TS * D;                             // TS * D
    T *s_lad = s_y + (size_t)TS * D;                           // TS * (nT + nI)
    T *s_tab = s_lad + (size_t)TS * (nT + nI);                 // nI * TW
    int *s_iidx = reinterpret_cast<int *>(s_tab + (size_t)nI * TW);  // nI
    int *s_tidx = s_iidx + nI;                                 // nT

    const bool has_uncond = uw != nullptr;
    const bool do_t = mode != NF_RQS_SAMPLE_IDENTITY;
    const bool do_i = mode != NF_RQS_SAMPLE_TRANSFORM;
    const bool inverse = mode != NF_RQS_DENSITY;
    const int tid = threadIdx.x, nth = blockDim.x;

    for (int j = tid; j < nI; j += nth) s_iidx[j] = (int)iidx[j];
    for (int j = tid; j < nT; j += nth) s_tidx[j] = (int)tidx[j];
    if (do_i && has_uncond) {
        for (int j = tid; j < nI; j += nth) {
            // the unconditional transform is not scaled by sqrt(hidden) (nsf/coupling.py:224-232): wh_div is not used here
            const RqsParams<T> pu = rqs_feature_params(p, tails_i, bound_i, j);
            const T *wj = uw + (size_t)j * K, *hj = uh + (size_t)j * K, *dj = ud + (size_t)j * p.nd;
            auto wacc = [=](int k) { return wj[k]; };
            auto hacc = [=](int k) { return hj[k]; };
            auto dacc = [=](int k) { return dj[k]; };
            rqs_build_table<T>(pu, wacc, hacc, dacc, s_tab + (size_t)j * TW);
        }
    }
    __syncthreads();

    const int64_t ntiles = (B + TS - 1) / TS;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t b0 = tile * TS;
        const int ts = (int)((B - b0) < TS ? (B - b0) : TS);
        // ---- stage x rows (contiguous ts*D) ----
        {
            const T *src = x + b0 * D;
            const int n = ts * D;
            for (int i = tid; i < n; i += nth) {
                const T v = src[i];
                s_x[i] = v;
                s_y[i] = v;  // columns this mode does not own are re-written unchanged only if owned below
            }
        }
        // ---- stage conditioner rows (contiguous ts*nT*M) into rows of pitch Mp (odd => conflict-free walks) ----
        if (do_t) {
            const T *src = cond + b0 * (int64_t)nT * M;
            const int n = ts * nT * M;
            if (Mp == M) {
                for (int i = tid; i < n; i += nth) s_cond[i] = src[i];  // straight coalesced copy
            } else {
                const float invM = 1.0f / (float)M;
                for (int i = tid; i < n; i += nth) {
                    const int row = (int)(((float)i + 0.5f) * invM);  // exact for i < 2^22 / M
                    s_cond[(size_t)row * Mp + (i - row * M)] = src[i];
                }
            }
        }
        __syncthreads();
        // ---- transform features: per-element parameters ----
        if (do_t) {
            const int n = ts * nT;
            const T div = p.wh_div;
            for (int e = tid; e < n; e += nth) {
                const int s = e / nT, j = e - s * nT;
                T *row = s_cond + (size_t)e * Mp;
                // nsf/coupling.py:334-339 (division by sqrt(hidden)) and the two softmaxes, once per value, in place
                rqs_softmax_row<T>(row, K, div);
                rqs_softmax_row<T>(row + K, K, div);
                auto pw = [=](int k) { return row[k]; };
                auto ph = [=](int k) { return row[K + k]; };
                auto dacc = [=](int k) { return row[2 * K + k]; };
                T yy, ll;
                rqs_element_probs<T>(rqs_feature_params(p, tails_t, bound_t, j), s_x[s * D + s_tidx[j]], pw, ph, dacc,
                                     inverse, yy, ll);
                s_y[s * D + s_tidx[j]] = yy;
                s_lad[s * (nT + nI) + j] = ll;
            }
        }
        // ---- identity features: batch-shared tables ----
        if (do_i) {
            const int n = ts * nI;
            for (int e = tid; e < n; e += nth) {
                const int s = e / nI, j = e - s * nI;
                T yy = s_x[s * D + s_iidx[j]], ll = T(0);
                if (has_uncond)
                    rqs_eval_table<T>(rqs_feature_params(p, tails_i, bound_i, j), yy, s_tab + (size_t)j * TW, inverse, yy, ll);
                s_y[s * D + s_iidx[j]] = yy;
                s_lad[s * (nT + nI) + nT + j] = ll;
            }
        }
        __syncthreads();
        // ---- write back: only the columns this mode owns (the others keep the caller's y) ----
        if (mode == NF_RQS_DENSITY) {
            T *dst = y + b0 * D;
            const int n = ts * D;
            for (int i = tid; i < n; i += nth) dst[i] = s_y[i];
        } else {
            const int nown = do_t ? nT : nI;
            const int *own = do_t ? s_tidx : s_iidx;
            const int n = ts * nown;
            for (int e = tid; e < n; e += nth) {
                const int s = e / nown, j = e - s * nown;
                y[(b0 + s) * D + own[j]] = s_y[s * D + own[j]];
            }
        }
        {   // per-sample log-det: one wave per sample, lanes stride over the row, fixed-order wave reduction
            const int lane = tid & 63, wv = tid >> 6, nwv = nth >> 6;
            for (int s = wv; s < ts; s += nwv) {
                T a = T(0);
                const T *row = s_lad + s * (nT + nI);
                if (do_t)
                    for (int j = lane; j < nT; j += 64) a += row[j];
                if (do_i)
                    for (int j = lane; j < nI; j += 64) a += row[nT + j];
                a = wave_sum(a);
                if (lane == 0) ld_store(logdet + b0 + s, a, acc);
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------
// NF_RQS_SAMPLE_IDENTITY alone (round 4): the batch-shared spline on the identity columns (nsf/coupling.py:221-253 in the sampling
// direction, :112-114) needs no conditioner rows and no staging -- tables once per workgroup, then P lanes per row (P = the identity
// count rounded up to a power of two, 64 / P rows per wave), each lane one element straight from / to global memory, the row's
// log-det by a butterfly inside its P lanes.  The tiled kernel above spent 56 us on this at B = 65 536, D = 64 (0.08 of the HBM peak).
template <typename T>
__global__ void __launch_bounds__(256)
rqs_identity_kernel(const T *__restrict__ x, T *__restrict__ y, T *__restrict__ logdet, const T *__restrict__ uw,
                    const T *__restrict__ uh, const T *__restrict__ ud, const int64_t *__restrict__ iidx, int nI, int64_t B, int D,
                    RqsParams<T> p, int acc, const int *__restrict__ tails_i, const T *__restrict__ bound_i) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int K = p.K, TW = 3 * (K + 1);
    T *s_tab = reinterpret_cast<T *>(smem_raw);                 // nI * TW
    int *s_iidx = reinterpret_cast<int *>(s_tab + (size_t)nI * TW);
    const int tid = threadIdx.x;
    for (int j = tid; j < nI; j += blockDim.x) {
        s_iidx[j] = (int)iidx[j];
        const RqsParams<T> pu = rqs_feature_params(p, tails_i, bound_i, j);
        const T *wj = uw + (size_t)j * K, *hj = uh + (size_t)j * K, *dj = ud + (size_t)j * p.nd;
        auto wacc = [=](int k) { return wj[k]; };
        auto hacc = [=](int k) { return hj[k]; };
        auto dacc = [=](int k) { return dj[k]; };
        rqs_build_table<T>(pu, wacc, hacc, dacc, s_tab + (size_t)j * TW);
    }
    __syncthreads();
    int P = 1;
    while (P < nI && P < 64) P <<= 1;
    const int rpw = 64 / P, lane = tid & 63, rin = lane / P, i0 = lane - rin * P;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + tid) >> 6, nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r0 = wave * rpw; r0 < B; r0 += nwaves * rpw) {
        const int64_t r = r0 + rin;
        T ld = T(0);
        if (r < B)
            for (int j = i0; j < nI; j += P) {
                const int64_t o = r * D + s_iidx[j];
                T yy, ll;
                rqs_eval_table<T>(rqs_feature_params(p, tails_i, bound_i, j), x[o], s_tab + (size_t)j * TW, true, yy, ll);
                y[o] = yy;
                ld += ll;
            }
        for (int off = P >> 1; off >= 1; off >>= 1) ld += __shfl_xor(ld, off, 64);
        if (r < B && i0 == 0) ld_store(logdet + r, ld, acc);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Same transform with WAVE-private tiles: a wave owns SPW samples at a time (SPW x max(nT, nI) <= ~64 elements per pass),
// stages their conditioner rows / x rows into its own LDS region with unit-stride loads and never meets the other waves
// of the workgroup again after the shared tables are built -- no workgroup barrier in the main loop (the tiled kernel
// above spends most of its time in four barriers per 256 elements).  LDS operations of one wave execute in order, so a
// wave-level fence is all the staging needs.  Per-sample log-det: one lane per sample sums the element terms in feature
// order (fixed order: deterministic).
template <typename T>
__global__ void __launch_bounds__(256)
rqs_coupling_wave_kernel(const T *__restrict__ x, T *__restrict__ y, T *__restrict__ logdet, const T *__restrict__ cond,
                         const T *__restrict__ uw, const T *__restrict__ uh, const T *__restrict__ ud,
                         const int64_t *__restrict__ iidx, int nI, const int64_t *__restrict__ tidx, int nT, int64_t B,
                         int D, RqsParams<T> p, int mode, int acc, int SPW, int Mp, const int *__restrict__ tails_t,
                         const T *__restrict__ bound_t, const int *__restrict__ tails_i, const T *__restrict__ bound_i) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int K = p.K;
    const int M = 2 * K + p.nd;
    const int TW = 3 * (K + 1);
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nwv = blockDim.x >> 6;
    const int per_wave = SPW * nT * Mp + 2 * SPW * D + SPW * (nT + nI);
    T *s_tab = reinterpret_cast<T *>(smem_raw);                         // nI * TW (shared)
    T *w_cond = s_tab + (size_t)nI * TW + (size_t)wid * per_wave;       // SPW * nT * Mp
    T *w_x = w_cond + (size_t)SPW * nT * Mp;                            // SPW * D
    T *w_y = w_x + (size_t)SPW * D;                                     // SPW * D
    T *w_lad = w_y + (size_t)SPW * D;                                   // SPW * (nT + nI)
    int *s_iidx = reinterpret_cast<int *>(s_tab + (size_t)nI * TW + (size_t)nwv * per_wave);
    int *s_tidx = s_iidx + nI;

    const bool has_uncond = uw != nullptr;
    const bool do_t = mode != NF_RQS_SAMPLE_IDENTITY;
    const bool do_i = mode != NF_RQS_SAMPLE_TRANSFORM;
    const bool inverse = mode != NF_RQS_DENSITY;

    for (int j = tid; j < nI; j += blockDim.x) s_iidx[j] = (int)iidx[j];
    for (int j = tid; j < nT; j += blockDim.x) s_tidx[j] = (int)tidx[j];
    if (do_i && has_uncond) {
        for (int j = tid; j < nI; j += blockDim.x) {
            const RqsParams<T> pu = rqs_feature_params(p, tails_i, bound_i, j);
            const T *wj = uw + (size_t)j * K, *hj = uh + (size_t)j * K, *dj = ud + (size_t)j * p.nd;
            rqs_build_table<T>(pu, [=](int k) { return wj[k]; }, [=](int k) { return hj[k]; }, [=](int k) { return dj[k]; },
                               s_tab + (size_t)j * TW);
        }
    }
    __syncthreads();   // the only workgroup barrier

    // the default NSF parametrisation (8 bins, linear tails, float32, launch-wide tails): the branch-free routines of the
    // fused kernel on register-resident parameters (fused_common.hpp: hardware exp2 / log / rcp, <= 1 ulp each) instead of
    // the run-time-K walks over LDS rows
    const bool fast = std::is_same<T, float>::value && K == F_K && p.tails == NF_TAILS_LINEAR && !p.dfull && !tails_t &&
                      !bound_t && !tails_i && !bound_i;
    const float invM = 1.0f / (float)M;
    const int64_t gw = (int64_t)blockIdx.x * nwv + wid, GW = (int64_t)gridDim.x * nwv;
    for (int64_t b0 = gw * SPW; b0 < B; b0 += GW * SPW) {
        const int ns = (int)((B - b0) < SPW ? (B - b0) : SPW);
        {
            const T *src = x + b0 * D;
            for (int i = lane; i < ns * D; i += 64) {
                const T v = src[i];
                w_x[i] = v;
                w_y[i] = v;
            }
        }
        if (do_t) {
            const T *src = cond + b0 * (int64_t)nT * M;
            const int n = ns * nT * M;
            if (Mp == M) {
                for (int i = lane; i < n; i += 64) w_cond[i] = src[i];
            } else {
                for (int i = lane; i < n; i += 64) {
                    const int row = (int)(((float)i + 0.5f) * invM);   // exact for i < 2^22 / M
                    w_cond[(size_t)row * Mp + (i - row * M)] = src[i];
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (do_t) {
            const T div = p.wh_div;
            for (int e = lane; e < ns * nT; e += 64) {
                const int s = e / nT, j = e - s * nT;
                T *row = w_cond + (size_t)e * Mp;
                if constexpr (std::is_same<T, float>::value) {
                    if (fast) {
                        float prm[24];
                        const float sc = 1.44269504088896340736f / p.wh_div;
#pragma unroll
                        for (int k = 0; k < 2 * F_K; ++k) prm[k] = row[k] * sc;
#pragma unroll
                        for (int k = 2 * F_K; k < F_M; ++k) prm[k] = row[k];
                        prm[F_M] = 0.0f;
                        float yy, ll;
                        const float xv = w_x[s * D + s_tidx[j]];
                        if (inverse) rqs_regs<true>(p, xv, prm, yy, ll);
                        else rqs_regs<false>(p, xv, prm, yy, ll);
                        w_y[s * D + s_tidx[j]] = yy;
                        w_lad[s * (nT + nI) + j] = ll;
                        continue;
                    }
                }
                rqs_softmax_row<T>(row, K, div);
                rqs_softmax_row<T>(row + K, K, div);
                auto pw = [=](int k) { return row[k]; };
                auto ph = [=](int k) { return row[K + k]; };
                auto dacc = [=](int k) { return row[2 * K + k]; };
                T yy, ll;
                rqs_element_probs<T>(rqs_feature_params(p, tails_t, bound_t, j), w_x[s * D + s_tidx[j]], pw, ph, dacc, inverse,
                                     yy, ll);
                w_y[s * D + s_tidx[j]] = yy;
                w_lad[s * (nT + nI) + j] = ll;
            }
        }
        if (do_i) {
            for (int e = lane; e < ns * nI; e += 64) {
                const int s = e / nI, j = e - s * nI;
                T yy = w_x[s * D + s_iidx[j]], ll = T(0);
                if constexpr (std::is_same<T, float>::value) {
                    if (fast && has_uncond) {
                        const float xv = yy;
                        if (inverse) rqs_table_fast<true>(p, xv, s_tab + (size_t)j * TW, yy, ll);
                        else rqs_table_fast<false>(p, xv, s_tab + (size_t)j * TW, yy, ll);
                        w_y[s * D + s_iidx[j]] = yy;
                        w_lad[s * (nT + nI) + nT + j] = ll;
                        continue;
                    }
                }
                if (has_uncond)
                    rqs_eval_table<T>(rqs_feature_params(p, tails_i, bound_i, j), yy, s_tab + (size_t)j * TW, inverse, yy, ll);
                w_y[s * D + s_iidx[j]] = yy;
                w_lad[s * (nT + nI) + nT + j] = ll;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (mode == NF_RQS_DENSITY) {
            T *dst = y + b0 * D;
            for (int i = lane; i < ns * D; i += 64) dst[i] = w_y[i];
        } else {
            const int nown = do_t ? nT : nI;
            const int *own = do_t ? s_tidx : s_iidx;
            for (int e = lane; e < ns * nown; e += 64) {
                const int s = e / nown, j = e - s * nown;
                y[(b0 + s) * D + own[j]] = w_y[s * D + own[j]];
            }
        }
        for (int s = lane; s < ns; s += 64) {
            T a = T(0);
            const T *row = w_lad + s * (nT + nI);
            if (do_t)
                for (int j = 0; j < nT; ++j) a += row[j];
            if (do_i)
                for (int j = 0; j < nI; ++j) a += row[nT + j];
            ld_store(logdet + b0 + s, a, acc);
        }
        __builtin_amdgcn_wave_barrier();   // the next pass overwrites the wave's tiles
    }
}

// Software-pipelined specialisation of the wave kernel for the default NSF layer shape (D = 64, 32 identity / 32 transform
// features, 8 bins, linear tails, float32, batch-shared spline on the identity half) -- the layer shapes the fused kernel does
// not take (hidden > 128, context, ...) still have THIS coupling transform behind their library-GEMM conditioner.  The wave
// kernel above runs load -> compute -> store per 2-sample pass (114 us at B = 65 536: 2.0 TB/s); here every pass's inputs (the
// 23-float conditioner rows of its 2 samples, their x rows) arrive by LDS-DMA (global_load_lds, 16 bytes per lane, no
// registers) into the buffer the PREVIOUS pass is not using, issued before the current pass's arithmetic, and a counted
// s_waitcnt (memory operations retire in order) replaces the drain -- the scheme of rqs_coupling_bwd_pipe_kernel.
// MODE: NF_RQS_DENSITY (both halves, forward splines) or NF_RQS_SAMPLE_TRANSFORM (transform half only, inverse spline).
// NT: transform (= identity) features, 32 or 64 (D = 64: two samples per pass and wave, D = 128: one); the pass moves the same
// 1472 + 128 floats either way.
#define NF_FWD_PIPE_WAVES 8
template <int MODE, int NT>
__global__ void __launch_bounds__(64 * NF_FWD_PIPE_WAVES, 2)
rqs_coupling_pipe_kernel(const float *__restrict__ x, float *__restrict__ y, float *__restrict__ logdet,
                         const float *__restrict__ cond, const float *__restrict__ uw, const float *__restrict__ uh,
                         const float *__restrict__ ud, const int64_t *__restrict__ iidx, const int64_t *__restrict__ tidx,
                         int64_t B, RqsParams<float> p, int acc) {
    typedef __attribute__((address_space(3))) void *lds_ptr;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int K = F_K, M = F_M, nI = NT, nT = NT, D = 2 * NT, SPW = 64 / NT, TW = 3 * (K + 1);
    constexpr int CONDF = SPW * nT * M;             // 1472 floats of conditioner rows per pass
    constexpr int XOFF = 1536, BUF = XOFF + SPW * D;  // cond | x
    constexpr int NC = (CONDF + 255) / 256;         // 1 KB DMA instructions for the rows (6)
    constexpr int LD_N = NC + 1, ST_N = 2;          // VMEM instructions per pass: loads (rows, x), stores (y rows, logdet)
    constexpr int PER_WAVE = 2 * BUF + SPW * D;
    constexpr bool DENS = MODE == NF_RQS_DENSITY;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *s_tab = reinterpret_cast<float *>(smem_raw);                          // nI * TW
    float *wbase = s_tab + nI * TW + (size_t)wid * PER_WAVE;
    float *w_y = wbase + 2 * BUF;
    int *s_iidx = reinterpret_cast<int *>(s_tab + nI * TW + (size_t)NF_FWD_PIPE_WAVES * PER_WAVE);
    int *s_tidx = s_iidx + nI;
    for (int j = tid; j < nI; j += blockDim.x) s_iidx[j] = (int)iidx[j];
    for (int j = tid; j < nT; j += blockDim.x) s_tidx[j] = (int)tidx[j];
    if (DENS) {
        for (int j = tid; j < nI; j += blockDim.x) {
            const float *wj = uw + (size_t)j * K, *hj = uh + (size_t)j * K, *dj = ud + (size_t)j * (K - 1);
            rqs_build_table<float>(p, [=](int k) { return wj[k]; }, [=](int k) { return hj[k]; }, [=](int k) { return dj[k]; },
                                   s_tab + (size_t)j * TW);
        }
    }
    __syncthreads();

    const float sc = 1.44269504088896340736f / p.wh_div;
    const int64_t gw = (int64_t)blockIdx.x * NF_FWD_PIPE_WAVES + wid, GW = (int64_t)gridDim.x * NF_FWD_PIPE_WAVES;
    auto issue = [&](int64_t b0, float *buf) {
        const float *csrc = cond + b0 * (int64_t)(nT * M);
#pragma unroll
        for (int q = 0; q < NC; ++q) {
            if (q < NC - 1 || q * 256 + lane * 4 < CONDF)
                __builtin_amdgcn_global_load_lds(csrc + q * 256 + lane * 4, (lds_ptr)(buf + q * 256), 16, 0, 0);
        }
        if (lane < 32) __builtin_amdgcn_global_load_lds(x + b0 * D + lane * 4, (lds_ptr)(buf + XOFF), 16, 0, 0);
    };
    const int s_ = lane / NT, j = lane % NT;
    const int col_t = s_tidx[j], col_i = s_iidx[j];
    int64_t b0 = gw * SPW;
    float *bufc = wbase, *bufn = wbase + BUF;
    if (b0 < B) issue(b0, bufc);
    bool first = true;
    // accumulating into the caller's log-density (acc = add / subtract): its old value is requested FIRST in a pass -- behind the
    // next pass's row requests the read-modify-write's load would be the youngest operation and waiting for it would drain them
    const bool has_old = __builtin_amdgcn_readfirstlane(acc != NF_LD_WRITE ? 1 : 0) != 0;
    for (; b0 < B; b0 += GW * SPW) {
        const int64_t b1 = b0 + GW * SPW;
        const bool more = b1 < B;
        float old = 0.0f;
        if (has_old && j == 0) old = logdet[b0 + s_];
        if (more) issue(b1, bufn);
        // operations younger than this pass's row requests: the previous pass's ST_N stores, this pass's old-value load, the
        // next pass's LD_N requests
        if (has_old) {
            if (first) {
                if (more) NF_WAIT_VMCNT(LD_N + 1);
                else NF_WAIT_VMCNT(1);
            } else {
                if (more) NF_WAIT_VMCNT(LD_N + ST_N + 1);
                else NF_WAIT_VMCNT(ST_N + 1);
            }
        } else {
            if (first) {
                if (more) NF_WAIT_VMCNT(LD_N);
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                if (more) NF_WAIT_VMCNT(LD_N + ST_N);
                else NF_WAIT_VMCNT(ST_N);
            }
        }
        first = false;
        __builtin_amdgcn_wave_barrier();
        float lad;
        {   // transform half: lane = (sample, transform feature)
            const float *row = bufc + (size_t)lane * M;
            float prm[24];
#pragma unroll
            for (int k = 0; k < 2 * K; ++k) prm[k] = row[k] * sc;
#pragma unroll
            for (int k = 2 * K; k < M; ++k) prm[k] = row[k];
            prm[M] = 0.0f;
            const float xv = bufc[XOFF + s_ * D + col_t];
            float yy;
            rqs_regs<!DENS>(p, xv, prm, yy, lad);
            w_y[s_ * D + col_t] = yy;
        }
        {   // identity half: the batch-shared spline (density), or the values passed through (sampling: the caller has them already)
            const float xv = bufc[XOFF + s_ * D + col_i];
            float yy = xv;
            if (DENS) {
                float ll;
                rqs_table_fast<false>(p, xv, s_tab + (size_t)j * TW, yy, ll);
                lad += ll;
            }
            w_y[s_ * D + col_i] = yy;
        }
        // per-sample log-det: the NT lanes of a sample, fixed butterfly order (deterministic)
#pragma unroll
        for (int o = NT / 2; o > 0; o >>= 1) lad += __shfl_xor(lad, o, 64);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (DENS) {
            if (lane < 32) *reinterpret_cast<f32x4 *>(y + b0 * D + lane * 4) = *reinterpret_cast<const f32x4 *>(w_y + lane * 4);
        } else {
            y[(b0 + s_) * D + col_t] = w_y[s_ * D + col_t];       // sampling: only the transform columns are this call's output
        }
        if (j == 0) logdet[b0 + s_] = acc == NF_LD_WRITE ? lad : (acc == NF_LD_ADD ? old + lad : old - lad);
        __builtin_amdgcn_wave_barrier();
        float *t = bufc; bufc = bufn; bufn = t;
    }
}

#ifndef NF_FWD_TS_ELEMS
#define NF_FWD_TS_ELEMS 256   // elements per tile and pass: ~one per lane
#endif

template <typename T>
static int launch_rqs_coupling(const void *x, void *y, void *logdet, const void *cond, const void *uw, const void *uh,
                               const void *ud, const int64_t *iidx, int nI, const int64_t *tidx, int nT, int64_t B,
                               int D, const RqsParams<T> &p, int mode, int acc, hipStream_t st,
                               const int32_t *tails_t = nullptr, const void *bound_t = nullptr,
                               const int32_t *tails_i = nullptr, const void *bound_i = nullptr) {
    const int K = p.K;
    const int M = 2 * K + p.nd;
    const int Mp = M | 1;
    const int nmax = nT > nI ? nT : nI;
    if (mode == NF_RQS_SAMPLE_IDENTITY && uw && nI >= 1) {
        const size_t ldsi = (size_t)nI * 3 * (K + 1) * sizeof(T) + (size_t)nI * sizeof(int) + 16;
        if (ldsi <= 64 * 1024) {
            const int grid = grid_for(B * (int64_t)(nI < 64 ? nI : 64), 256);
            hipLaunchKernelGGL(rqs_identity_kernel<T>, dim3(grid), dim3(256), ldsi, st, (const T *)x, (T *)y, (T *)logdet, (const T *)uw,
                               (const T *)uh, (const T *)ud, iidx, nI, B, D, p, acc, (const int *)tails_i, (const T *)bound_i);
            NF_CHECK_LAUNCH();
            return NF_OK;
        }
    }
#ifndef NF_FWD_NO_PIPE
    if constexpr (std::is_same<T, float>::value) {
        // the default NSF layer shape (D = 64 or 128, alternating halves) on the software-pipelined kernel
        if (K == F_K && p.tails == NF_TAILS_LINEAR && !p.dfull && !tails_t && !bound_t && !tails_i && !bound_i && nI == nT &&
            (nT == 32 || nT == 64) && D == 2 * nT && (mode == NF_RQS_DENSITY || mode == NF_RQS_SAMPLE_TRANSFORM) && uw &&
            B % (64 / nT) == 0 && B >= 1024 && ((((uintptr_t)x | (uintptr_t)y | (uintptr_t)cond) & 15) == 0)) {
            constexpr int BUFp = 1536 + 128, PERW = 2 * BUFp + 128;
            const size_t ldsp = ((size_t)nT * 27 + (size_t)NF_FWD_PIPE_WAVES * PERW) * sizeof(float) + 2 * nT * sizeof(int) + 16;
            const int64_t nw2 = B / (64 / nT), gq2 = (nw2 + NF_FWD_PIPE_WAVES - 1) / NF_FWD_PIPE_WAVES;
            const int grid2 = (int)(gq2 < 4096 / NF_FWD_PIPE_WAVES ? gq2 : 4096 / NF_FWD_PIPE_WAVES);
#define NF_FWD_PIPE_LAUNCH(MODE_, NT_)                                                                                              \
            do {                                                                                                                    \
                static LdsOptIn opt_ = {};                                                                                          \
                if (opt_in_lds(reinterpret_cast<const void *>(&rqs_coupling_pipe_kernel<MODE_, NT_>), ldsp, opt_) == NF_OK) {         \
                    hipLaunchKernelGGL((rqs_coupling_pipe_kernel<MODE_, NT_>), dim3(grid2), dim3(64 * NF_FWD_PIPE_WAVES), ldsp, st,  \
                                       (const float *)x, (float *)y, (float *)logdet, (const float *)cond, (const float *)uw,       \
                                       (const float *)uh, (const float *)ud, iidx, tidx, B, p, acc);                                \
                    NF_CHECK_LAUNCH();                                                                                              \
                    return NF_OK;                                                                                                   \
                }                                                                                                                   \
            } while (0)
            if (mode == NF_RQS_DENSITY) {
                if (nT == 32) NF_FWD_PIPE_LAUNCH(NF_RQS_DENSITY, 32);
                else NF_FWD_PIPE_LAUNCH(NF_RQS_DENSITY, 64);
            } else {
                if (nT == 32) NF_FWD_PIPE_LAUNCH(NF_RQS_SAMPLE_TRANSFORM, 32);
                else NF_FWD_PIPE_LAUNCH(NF_RQS_SAMPLE_TRANSFORM, 64);
            }
#undef NF_FWD_PIPE_LAUNCH
        }
    }
#endif
#ifndef NF_FWD_NO_WAVE_KERNEL
    {   // wave-private tiles when four waves' regions fit 64 KB of LDS
        int SPW = nmax > 0 ? 64 / nmax : 1;
        if (SPW < 1) SPW = 1;
        const size_t per_wave = (size_t)SPW * nT * Mp + 2 * (size_t)SPW * D + (size_t)SPW * (nT + nI);
        const size_t ldsw = ((size_t)nI * 3 * (K + 1) + 4 * per_wave) * sizeof(T) + (size_t)(nI + nT) * sizeof(int) + 16;
        if (ldsw <= 64 * 1024 && mode != NF_RQS_SAMPLE_IDENTITY) {   // (table look-ups alone: the tiled kernel is as fast)
            const int64_t nwaves = (B + SPW - 1) / SPW;
            const int64_t g = (nwaves + 3) / 4;
            const int grid = (int)(g < 2048 ? g : 2048);
            hipLaunchKernelGGL(rqs_coupling_wave_kernel<T>, dim3(grid), dim3(256), ldsw, st, (const T *)x, (T *)y, (T *)logdet,
                               (const T *)cond, (const T *)uw, (const T *)uh, (const T *)ud, iidx, nI, tidx, nT, B, D, p, mode,
                               acc, SPW, Mp, (const int *)tails_t, (const T *)bound_t, (const int *)tails_i,
                               (const T *)bound_i);
            NF_CHECK_LAUNCH();
            return NF_OK;
        }
    }
#endif
    int TS = NF_FWD_TS_ELEMS / (nmax > 0 ? nmax : 1);
    if (TS < 1) TS = 1;
    if (TS > 64) TS = 64;
    auto lds_bytes = [&](int ts) -> size_t {
        size_t words = (size_t)ts * nT * Mp + 2 * (size_t)ts * D + (size_t)ts * (nT + nI) + (size_t)nI * 3 * (K + 1);
        return words * sizeof(T) + (size_t)(nI + nT) * sizeof(int) + 16;
    };
    while (TS > 1 && lds_bytes(TS) > 64 * 1024) TS >>= 1;
    const size_t lds = lds_bytes(TS);
    if (lds > 160 * 1024) return NF_ENOTSUP;
    static LdsOptIn opted = {};
    if (opt_in_lds(reinterpret_cast<const void *>(&rqs_coupling_kernel<T>), lds, opted) != NF_OK) return NF_ENOTSUP;
    const int64_t ntiles = (B + TS - 1) / TS;
    const int grid = (int)(ntiles < 2048 ? ntiles : 2048);
    hipLaunchKernelGGL(rqs_coupling_kernel<T>, dim3(grid), dim3(256), lds, st, (const T *)x, (T *)y, (T *)logdet,
                       (const T *)cond, (const T *)uw, (const T *)uh, (const T *)ud, iidx, nI, tidx, nT, B, D, p, mode,
                       acc, TS, Mp, (const int *)tails_t, (const T *)bound_t, (const int *)tails_i, (const T *)bound_i);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

}  // namespace nf

using namespace nf;

static int check_spline_args(int K, int tails, double min_w, double min_h) {
    if (K < 1 || K > NF_MAX_BINS) return NF_ERANGE;
    if (tails != NF_TAILS_NONE && tails != NF_TAILS_LINEAR && tails != NF_TAILS_CIRCULAR) return NF_EINVAL;
    // utils/splines.py:121-124 raises ValueError
    if (min_w * K > 1.0 || min_h * K > 1.0) return NF_EINVAL;
    return NF_OK;
}

extern "C" int nf_rqs_spline(const void *x, const void *w, int64_t ldw, const void *h, int64_t ldh, const void *d,
                             int64_t ldd, void *y, void *logabsdet, int64_t N, int K, int tails, double tail_bound,
                             double left, double right, double bottom, double top, double min_bin_width,
                             double min_bin_height, double min_derivative, double wh_div, int inverse, int dtype,
                             nf_stream_t stream) {
    int rc = check_spline_args(K, tails, min_bin_width, min_bin_height);
    if (rc) return rc;
    if (N < 0) return NF_EINVAL;
    if (N == 0) return NF_OK;
    if (!x || !w || !h || !d || !y) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    const int grid = grid_for(N, 256);
    // contiguous parameter rows (the usual case: three separate (..., K) tensors): the LDS-staged kernel
    const int nd_rows = tails == NF_TAILS_LINEAR ? K - 1 : (tails == NF_TAILS_CIRCULAR ? K : K + 1);
    if (ldw == K && ldh == K && ldd == nd_rows && (dtype == NF_F32 || dtype == NF_F64)) {
        const int pitch = (2 * K + nd_rows) | 1;
        const size_t esz = dtype == NF_F32 ? 4 : 8;
        const size_t lds = (size_t)256 * pitch * esz;
        const int64_t npass = (N + 255) / 256;
        const int sgrid = (int)(npass < 256 * 8 ? npass : 256 * 8);
        if (lds <= 64 * 1024) {
            if (dtype == NF_F32) {
                auto p = make_rqs_params<float>(K, tails, tail_bound, left, right, bottom, top, min_bin_width, min_bin_height,
                                                min_derivative, wh_div);
                if (K == F_K && tails == NF_TAILS_LINEAR)
                    hipLaunchKernelGGL((rqs_spline_staged_kernel<float, true>), dim3(sgrid), dim3(256), lds, st, (const float *)x,
                                       (const float *)w, (const float *)h, (const float *)d, (float *)y, (float *)logabsdet, N, p,
                                       inverse, pitch);
                else
                    hipLaunchKernelGGL((rqs_spline_staged_kernel<float, false>), dim3(sgrid), dim3(256), lds, st, (const float *)x,
                                       (const float *)w, (const float *)h, (const float *)d, (float *)y, (float *)logabsdet, N, p,
                                       inverse, pitch);
            } else {
                auto p = make_rqs_params<double>(K, tails, tail_bound, left, right, bottom, top, min_bin_width, min_bin_height,
                                                 min_derivative, wh_div);
                hipLaunchKernelGGL((rqs_spline_staged_kernel<double, false>), dim3(sgrid), dim3(256), lds, st, (const double *)x,
                                   (const double *)w, (const double *)h, (const double *)d, (double *)y, (double *)logabsdet, N, p,
                                   inverse, pitch);
            }
            NF_CHECK_LAUNCH();
            return NF_OK;
        }
    }
    if (dtype == NF_F32) {
        auto p = make_rqs_params<float>(K, tails, tail_bound, left, right, bottom, top, min_bin_width, min_bin_height,
                                        min_derivative, wh_div);
        hipLaunchKernelGGL(rqs_spline_kernel<float>, dim3(grid), dim3(256), 0, st, (const float *)x, (const float *)w,
                           ldw, (const float *)h, ldh, (const float *)d, ldd, (float *)y, (float *)logabsdet, N, p,
                           inverse);
    } else if (dtype == NF_F64) {
        auto p = make_rqs_params<double>(K, tails, tail_bound, left, right, bottom, top, min_bin_width, min_bin_height,
                                         min_derivative, wh_div);
        hipLaunchKernelGGL(rqs_spline_kernel<double>, dim3(grid), dim3(256), 0, st, (const double *)x,
                           (const double *)w, ldw, (const double *)h, ldh, (const double *)d, ldd, (double *)y,
                           (double *)logabsdet, N, p, inverse);
    } else
        return NF_ENOTSUP;
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// ---- debug mode: the reference's run-time failures of this path as DEVICE-SIDE flags ----
// utils/splines.py:181 `assert (discriminant >= 0).all()` (inverse direction) and the index error of the gathers behind
// utils/splines.py:154-160 when `tails=None` inputs lie outside the domain (bin index -1 or K).  The transform kernels stay
// branch-free and never synchronise; in debug mode (normflows_amd.config.set_debug_checks) the shim launches this check over
// the call's inputs and outputs and reads the flag word back -- the only host synchronisation, and only in that mode.
//   bit 0: an input outside [lo, hi] with tails=None (NaN counts as outside: the reference's count-based search gives bin -1)
//   bit 1: inverse direction, a non-NaN input inside the domain whose output is NaN: sqrt of a negative (or NaN) discriminant
template <typename T>
__global__ void rqs_debug_check_kernel(const T *__restrict__ x, const T *__restrict__ y, int64_t N, T lo, T hi, int bounded,
                                       int inverse, unsigned int *flags) {
    unsigned int f = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        const T xi = x[i];
        const bool in = xi >= lo && xi <= hi;
        if (bounded && !in) f |= 1u;
        if (inverse && in && y[i] != y[i]) f |= 2u;
    }
    if (f) atomicOr(flags, f);
}

extern "C" int nf_rqs_spline_check(const void *x, const void *y, int64_t N, int tails, double tail_bound, double left, double right,
                                   double bottom, double top, int inverse, int dtype, void *flags, nf_stream_t stream) {
    if (N < 0) return NF_EINVAL;
    if (tails != NF_TAILS_NONE && tails != NF_TAILS_LINEAR && tails != NF_TAILS_CIRCULAR) return NF_EINVAL;
    if (N == 0) return NF_OK;
    if (!x || !y || !flags) return NF_EFAULT;
    const int bounded = tails == NF_TAILS_NONE;
    double lo = inverse ? bottom : left, hi = inverse ? top : right;
    if (!bounded) { lo = -tail_bound; hi = tail_bound; }
    hipStream_t st = (hipStream_t)stream;
    const int grid = grid_for(N, 256);
    if (dtype == NF_F32)
        hipLaunchKernelGGL(rqs_debug_check_kernel<float>, dim3(grid), dim3(256), 0, st, (const float *)x, (const float *)y, N, (float)lo,
                           (float)hi, bounded, inverse, (unsigned int *)flags);
    else if (dtype == NF_F64)
        hipLaunchKernelGGL(rqs_debug_check_kernel<double>, dim3(grid), dim3(256), 0, st, (const double *)x, (const double *)y, N, lo, hi,
                           bounded, inverse, (unsigned int *)flags);
    else
        return NF_ENOTSUP;
    NF_CHECK_LAUNCH();
    return NF_OK;
}

extern "C" int nf_rqs_coupling(const void *x, void *y, void *logdet, const void *cond, const void *uw, const void *uh,
                               const void *ud, const int64_t *identity_idx, int nI, const int64_t *transform_idx,
                               int nT, int64_t B, int D, int K, int tails, double tail_bound, double min_bin_width,
                               double min_bin_height, double min_derivative, double wh_div, int mode, int acc,
                               int dtype, nf_stream_t stream) {
    int rc = check_spline_args(K, tails, min_bin_width, min_bin_height);
    if (rc) return rc;
    if (B < 0 || D < 1 || nI < 0 || nT < 0 || nI + nT != D) return NF_EINVAL;
    if (mode < NF_RQS_DENSITY || mode > NF_RQS_SAMPLE_TRANSFORM) return NF_EINVAL;
    if (acc < NF_LD_SUB || acc > NF_LD_ADD) return NF_EINVAL;
    if (B == 0) return NF_OK;
    if (!x || !y || !logdet || (nI && !identity_idx) || (nT && !transform_idx)) return NF_EFAULT;
    if (mode != NF_RQS_SAMPLE_IDENTITY && nT && !cond) return NF_EFAULT;
    if ((uw || uh || ud) && !(uw && uh && ud)) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == NF_F32) {
        auto p = make_rqs_params<float>(K, tails, tail_bound, 0, 1, 0, 1, min_bin_width, min_bin_height,
                                        min_derivative, wh_div);
        return launch_rqs_coupling<float>(x, y, logdet, cond, uw, uh, ud, identity_idx, nI, transform_idx, nT, B, D, p,
                                          mode, acc, st);
    } else if (dtype == NF_F64) {
        auto p = make_rqs_params<double>(K, tails, tail_bound, 0, 1, 0, 1, min_bin_width, min_bin_height,
                                         min_derivative, wh_div);
        return launch_rqs_coupling<double>(x, y, logdet, cond, uw, uh, ud, identity_idx, nI, transform_idx, nT, B, D,
                                           p, mode, acc, st);
    }
    return NF_ENOTSUP;
}

extern "C" int nf_rqs_coupling_ft(const void *x, void *y, void *logdet, const void *cond, const void *uw, const void *uh,
                               const void *ud, const int64_t *identity_idx, int nI, const int64_t *transform_idx,
                               int nT, int64_t B, int D, int K, int tails, double tail_bound, double min_bin_width,
                               double min_bin_height, double min_derivative, double wh_div, int mode, int acc,
                               int dtype,
                                  const int32_t *tails_t, const void *bound_t, const int32_t *tails_i,
                                  const void *bound_i, nf_stream_t stream) {
    int rc = check_spline_args(K, tails == NF_TAILS_FEATURE ? NF_TAILS_LINEAR : tails, min_bin_width, min_bin_height);
    if (rc) return rc;
    if (tails == NF_TAILS_FEATURE && ((nT && !tails_t) || (nI && uw && !tails_i))) return NF_EFAULT;
    if (tails != NF_TAILS_FEATURE && (tails_t || tails_i)) return NF_EINVAL;
    if (B < 0 || D < 1 || nI < 0 || nT < 0 || nI + nT != D) return NF_EINVAL;
    if (mode < NF_RQS_DENSITY || mode > NF_RQS_SAMPLE_TRANSFORM) return NF_EINVAL;
    if (acc < NF_LD_SUB || acc > NF_LD_ADD) return NF_EINVAL;
    if (B == 0) return NF_OK;
    if (!x || !y || !logdet || (nI && !identity_idx) || (nT && !transform_idx)) return NF_EFAULT;
    if (mode != NF_RQS_SAMPLE_IDENTITY && nT && !cond) return NF_EFAULT;
    if ((uw || uh || ud) && !(uw && uh && ud)) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == NF_F32) {
        auto p = make_rqs_params<float>(K, tails, tail_bound, 0, 1, 0, 1, min_bin_width, min_bin_height,
                                        min_derivative, wh_div);
        return launch_rqs_coupling<float>(x, y, logdet, cond, uw, uh, ud, identity_idx, nI, transform_idx, nT, B, D, p,
                                          mode, acc, st, tails_t, bound_t, tails_i, bound_i);
    } else if (dtype == NF_F64) {
        auto p = make_rqs_params<double>(K, tails, tail_bound, 0, 1, 0, 1, min_bin_width, min_bin_height,
                                         min_derivative, wh_div);
        return launch_rqs_coupling<double>(x, y, logdet, cond, uw, uh, ud, identity_idx, nI, transform_idx, nT, B, D,
                                           p, mode, acc, st, tails_t, bound_t, tails_i, bound_i);
    }
    return NF_ENOTSUP;
}
