// common.hpp -- shared device/host helpers for libnf_mi355x (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "../../include/nf_mi355x.h"

#define NF_MAX_BINS 64
#define NF_WAVE 64

// Hand-counted vector-memory waits (a ring acquire that lets N younger requests stay in flight) rely on the exact number of
// vector-memory instructions the compiler emits between a request and its wait.  -DNF_SAFE_WAITS turns every one of them into a full
// drain: the differential build of tests/test_gpu_hygiene.py::test_counted_waits_equal_full_drains (round-3 ADVICE) -- a count that
// a compiler upgrade made too lax shows up as a bit difference against this build.
#ifdef NF_SAFE_WAITS
#define NF_WAIT_VMCNT(N) asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define NF_WAIT_VMCNT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#endif

// One 16-byte-per-lane LDS-DMA request (global_load_lds_dwordx4) as inline asm instead of __builtin_amdgcn_global_load_lds (round 6).
// Through the builtin hipcc knows an LDS-DMA is pending and answers EVERY later LDS read of the wave with `s_waitcnt lgkmcnt(0)`:
// product loops compile to read -> full wait -> MFMAs whatever look-ahead the source asks for.  The asm form hides the request from that
// bookkeeping; landing is waited for by hand anyway (vmcnt + barrier in the rings' acquire).  base: wave-uniform global pointer (forced
// into an SGPR pair: the "s" constraint alone left a VGPR pair in the instruction when the compiler could not prove uniformity),
// byte_off: the lane's 32-bit offset, lds_dst: wave-uniform LDS address of the 1 KB piece.  m0 is reserved: saved and restored.
#define NF_DMA16(BASE, BYTE_OFF, LDS_DST)                                                                                          \
    do {                                                                                                                           \
        uint32_t m0__;                                                                                                             \
        const uint32_t ldsa__ = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(LDS_DST));                                    \
        const uint64_t b64__ = (uint64_t)(uintptr_t)(BASE);                                                                        \
        const uint64_t sb__ = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b64__ >> 32)) << 32) |                \
                              (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b64__);                                           \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"      \
                     : "=&s"(m0__) : "s"(ldsa__), "v"((uint32_t)(BYTE_OFF)), "s"(sb__) : "memory");                                \
    } while (0)

#define NF_CHECK_LAUNCH()                            \
    do {                                             \
        hipError_t e__ = hipGetLastError();          \
        if (e__ != hipSuccess) return NF_EIO;        \
    } while (0)

namespace nf {

// ---- scalar math, precise variants (parity with the reference's fp32/fp64 CPU path, not fast-math) ----
template <typename T> struct M;
template <> struct M<float> {
#ifdef NF_FAST_F32_MATH   // hardware v_exp_f32 / v_log_f32 (about 1-2 ulp), as the fused kernel's epilogue uses
    static __device__ __forceinline__ float exp(float x) { return __expf(x); }
    static __device__ __forceinline__ float log(float x) { return __logf(x); }
    static __device__ __forceinline__ float log1p(float x) { return __logf(1.0f + x); }
#else
    static __device__ __forceinline__ float exp(float x) { return ::expf(x); }
    static __device__ __forceinline__ float log(float x) { return ::logf(x); }
    static __device__ __forceinline__ float log1p(float x) { return ::log1pf(x); }
#endif
    static __device__ __forceinline__ float sqrt(float x) { return ::sqrtf(x); }
    static __device__ __forceinline__ float fmax(float a, float b) { return ::fmaxf(a, b); }
    static __device__ __forceinline__ bool finite(float x) { return ::isfinite(x); }
    static __device__ __forceinline__ float nan() { return __builtin_nanf(""); }
};
template <> struct M<double> {
    static __device__ __forceinline__ double exp(double x) { return ::exp(x); }
    static __device__ __forceinline__ double log(double x) { return ::log(x); }
    static __device__ __forceinline__ double log1p(double x) { return ::log1p(x); }
    static __device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
    static __device__ __forceinline__ double fmax(double a, double b) { return ::fmax(a, b); }
    static __device__ __forceinline__ bool finite(double x) { return ::isfinite(x); }
    static __device__ __forceinline__ double nan() { return __builtin_nan(""); }
};

// torch.nn.functional.softplus (beta = 1, threshold = 20).
template <typename T> __device__ __forceinline__ T softplus(T x) {
    return x > T(20) ? x : M<T>::log1p(M<T>::exp(x));
}
// torch.sigmoid
template <typename T> __device__ __forceinline__ T sigmoid(T x) { return T(1) / (T(1) + M<T>::exp(-x)); }

// ---- spline constants shared by every element of a launch --------------------------------------------
template <typename T> struct RqsParams {
    int K;
    int tails;         // NF_TAILS_*
    int nd;            // number of derivative logits per element (K-1 | K | K+1)
    T left, right, bottom, top;
    T min_w, min_h, min_d;
    T scale_w, scale_h;  // (1 - min_w K), (1 - min_h K)   (utils/splines.py:127, :141)
    T wh_div;            // sqrt(hidden_features) or 1       (nsf/coupling.py:334-339)
    T edge_logit;        // log(exp(1 - min_d) - 1)          (utils/splines.py:36)
    T eps;               // searchsorted eps = 1e-6          (utils/splines.py:11)
    int dfull;           // 1: per-feature tails (utils/splines.py:48-57): K+1 derivative logits per element, the edge
                         // ones overridden according to the feature's tails type; out-of-interval outputs are 0
};

template <typename T>
static inline RqsParams<T> make_rqs_params(int K, int tails, double tail_bound, double left, double right,
                                           double bottom, double top, double min_w, double min_h, double min_d,
                                           double wh_div) {
    RqsParams<T> p;
    p.K = K;
    p.dfull = tails == NF_TAILS_FEATURE ? 1 : 0;
    p.tails = tails == NF_TAILS_FEATURE ? NF_TAILS_LINEAR : tails;  // per-feature type is set by rqs_feature_params
    p.nd = tails == NF_TAILS_LINEAR ? K - 1 : (tails == NF_TAILS_CIRCULAR ? K : K + 1);
    if (tails != NF_TAILS_NONE) {
        left = -tail_bound; right = tail_bound; bottom = -tail_bound; top = tail_bound;
    }
    p.left = (T)left; p.right = (T)right; p.bottom = (T)bottom; p.top = (T)top;
    p.min_w = (T)min_w; p.min_h = (T)min_h; p.min_d = (T)min_d;
    p.scale_w = (T)(1.0 - min_w * K);
    p.scale_h = (T)(1.0 - min_h * K);
    p.wh_div = (T)wh_div;
    p.edge_logit = (T)log(exp(1.0 - min_d) - 1.0);
    p.eps = (T)1e-6;
    return p;
}

// Inside-interval test of utils/splines.py:28 (false for NaN).
template <typename T> __device__ __forceinline__ bool rqs_inside(const RqsParams<T> &p, T x) {
    return p.tails == NF_TAILS_NONE ? true : (x >= p.left && x <= p.right);
}

// Parameters of one feature when tails / tail_bound are given per feature (utils/splines.py:48-57, :61-66); ft / fb may
// each be null (then the launch-wide value applies).
template <typename T>
__device__ __forceinline__ RqsParams<T> rqs_feature_params(const RqsParams<T> &p, const int *ft, const T *fb, int j) {
    RqsParams<T> q = p;
    if (ft) q.tails = ft[j];
    if (fb) {
        const T tb = fb[j];
        q.left = -tb; q.right = tb; q.bottom = -tb; q.top = tb;
    }
    return q;
}

// Unnormalised derivative logit j in [0, K] after the padding of utils/splines.py:34-44.
template <typename T, typename Acc>
__device__ __forceinline__ T rqs_dlogit(const RqsParams<T> &p, const Acc &dacc, int j) {
    if (p.dfull) {  // utils/splines.py:48-57: the K+1 logits are kept, edges overwritten per tails type
        if (p.tails == NF_TAILS_LINEAR) return (j == 0 || j == p.K) ? p.edge_logit : dacc(j);
        return j == p.K ? dacc(0) : dacc(j);
    }
    if (p.tails == NF_TAILS_LINEAR) return (j == 0 || j == p.K) ? p.edge_logit : dacc(j - 1);
    if (p.tails == NF_TAILS_CIRCULAR) return j == p.K ? dacc(0) : dacc(j);
    return dacc(j);
}

// Closed-form evaluation once the bin is known (utils/splines.py:159-219).
//   cw, bw: left knot and width of the bin on the x axis;  ch, bh: same on the y axis;
//   d0, d1: derivatives at the bin's left/right knot.
template <typename T>
__device__ __forceinline__ void rqs_eval_bin(T x, T cw, T bw, T ch, T bh, T d0, T d1, bool inverse, T &y,
                                             T &lad) {
    const T delta = bh / bw;
    const T dsum = d0 + d1 - T(2) * delta;
    if (!inverse) {
        const T theta = (x - cw) / bw;
        const T t1mt = theta * (T(1) - theta);
        const T num = bh * (delta * theta * theta + d0 * t1mt);
        const T den = delta + dsum * t1mt;
        y = ch + num / den;
        const T omt = T(1) - theta;
        const T dnum = delta * delta * (d1 * theta * theta + T(2) * delta * t1mt + d0 * omt * omt);
        lad = M<T>::log(dnum) - T(2) * M<T>::log(den);
    } else {
        const T dy = x - ch;
        const T a = dy * dsum + bh * (delta - d0);
        const T b = bh * d0 - dy * dsum;
        const T c = -delta * dy;
        const T disc = b * b - T(4) * a * c;
        const T root = (T(2) * c) / (-b - M<T>::sqrt(disc));
        y = root * bw + cw;
        const T t1mt = root * (T(1) - root);
        const T den = delta + dsum * t1mt;
        const T omr = T(1) - root;
        const T dnum = delta * delta * (d1 * root * root + T(2) * delta * t1mt + d0 * omr * omr);
        lad = -(M<T>::log(dnum) - T(2) * M<T>::log(den));
    }
}

// One spline element with per-element parameters read through accessors (K known at run time, no
// register arrays): two softmax passes, then a sequential cumsum that doubles as the bin search
// (utils/splines.py:126-157).  wacc(k), hacc(k) return the conditioner outputs ALREADY divided by
// p.wh_div (nsf/coupling.py:334-339); dacc(j) the raw derivative logits.
template <typename T, typename WAcc, typename HAcc, typename DAcc>
__device__ __forceinline__ void rqs_element(const RqsParams<T> &p, T x, const WAcc &wacc, const HAcc &hacc,
                                            const DAcc &dacc, bool inverse, T &y, T &lad) {
    if (!rqs_inside(p, x)) {  // linear / circular tails: identity outside, also for NaN / inf (:40-41);
        y = p.dfull ? T(0) : x;  // the per-feature branch (:48-57) never copies the outside inputs: they stay 0
        lad = T(0);
        return;
    }
    const int K = p.K;
    T mw = wacc(0), mh = hacc(0);
    for (int k = 1; k < K; ++k) {
        mw = M<T>::fmax(mw, wacc(k));
        mh = M<T>::fmax(mh, hacc(k));
    }
    T sw = T(0), sh = T(0);
    for (int k = 0; k < K; ++k) {
        sw += M<T>::exp(wacc(k) - mw);
        sh += M<T>::exp(hacc(k) - mh);
    }
    // searched axis: x-axis (widths) for the forward spline, y-axis (heights) for the inverse.
    const T s_lo = inverse ? p.bottom : p.left, s_hi = inverse ? p.top : p.right;
    const T o_lo = inverse ? p.left : p.bottom, o_hi = inverse ? p.right : p.top;
    const T s_min = inverse ? p.min_h : p.min_w, s_scale = inverse ? p.scale_h : p.scale_w;
    const T o_min = inverse ? p.min_w : p.min_h, o_scale = inverse ? p.scale_w : p.scale_h;
    const T s_max = inverse ? mh : mw, s_sum = inverse ? sh : sw;
    const T o_max = inverse ? mw : mh, o_sum = inverse ? sw : sh;

    int bin = 0;
    T cum = T(0), knot = s_lo, blo = s_lo, bhi = s_lo;
    for (int k = 0; k < K; ++k) {
        const T raw = inverse ? hacc(k) : wacc(k);
        cum += s_min + s_scale * (M<T>::exp(raw - s_max) / s_sum);
        const T next = (k == K - 1) ? s_hi : (s_hi - s_lo) * cum + s_lo;
        if (k == 0 || x >= knot) {
            bin = k;
            blo = knot;
            bhi = next;
        }
        knot = next;
    }
    cum = T(0);
    knot = o_lo;
    T olo = o_lo, ohi = o_lo;
    for (int k = 0; k <= bin; ++k) {
        const T raw = inverse ? wacc(k) : hacc(k);
        cum += o_min + o_scale * (M<T>::exp(raw - o_max) / o_sum);
        const T next = (k == K - 1) ? o_hi : (o_hi - o_lo) * cum + o_lo;
        olo = knot;
        ohi = next;
        knot = next;
    }
    const T d0 = p.min_d + softplus(rqs_dlogit(p, dacc, bin));
    const T d1 = p.min_d + softplus(rqs_dlogit(p, dacc, bin + 1));
    if (!inverse)
        rqs_eval_bin<T>(x, blo, bhi - blo, olo, ohi - olo, d0, d1, false, y, lad);
    else
        rqs_eval_bin<T>(x, olo, ohi - olo, blo, bhi - blo, d0, d1, true, y, lad);
}

// In-place softmax of K logits stored at row[0..K) (divided by `div` first): ONE exp per logit.
template <typename T> __device__ __forceinline__ void rqs_softmax_row(T *row, int K, T div) {
    T m = row[0] / div;
    for (int k = 1; k < K; ++k) m = M<T>::fmax(m, row[k] / div);
    T s = T(0);
    for (int k = 0; k < K; ++k) {
        const T e = M<T>::exp(row[k] / div - m);
        row[k] = e;
        s += e;
    }
    const T inv = T(1) / s;
    for (int k = 0; k < K; ++k) row[k] *= inv;
}

// rqs_element on softmax PROBABILITIES pw(k), ph(k) (rqs_softmax_row) instead of logits: the knot walk needs no
// transcendental any more (utils/splines.py:126-157 evaluates the softmax once as well).
template <typename T, typename PW, typename PH, typename DAcc>
__device__ __forceinline__ void rqs_element_probs(const RqsParams<T> &p, T x, const PW &pw, const PH &ph,
                                                  const DAcc &dacc, bool inverse, T &y, T &lad) {
    if (!rqs_inside(p, x)) {
        y = p.dfull ? T(0) : x;
        lad = T(0);
        return;
    }
    const int K = p.K;
    const T s_lo = inverse ? p.bottom : p.left, s_hi = inverse ? p.top : p.right;
    const T o_lo = inverse ? p.left : p.bottom, o_hi = inverse ? p.right : p.top;
    const T s_min = inverse ? p.min_h : p.min_w, s_scale = inverse ? p.scale_h : p.scale_w;
    const T o_min = inverse ? p.min_w : p.min_h, o_scale = inverse ? p.scale_w : p.scale_h;
    int bin = 0;
    T cum = T(0), knot = s_lo, blo = s_lo, bhi = s_lo;
    for (int k = 0; k < K; ++k) {
        cum += s_min + s_scale * (inverse ? ph(k) : pw(k));
        const T next = (k == K - 1) ? s_hi : (s_hi - s_lo) * cum + s_lo;
        if (k == 0 || x >= knot) {
            bin = k;
            blo = knot;
            bhi = next;
        }
        knot = next;
    }
    cum = T(0);
    knot = o_lo;
    T olo = o_lo, ohi = o_lo;
    for (int k = 0; k <= bin; ++k) {
        cum += o_min + o_scale * (inverse ? pw(k) : ph(k));
        const T next = (k == K - 1) ? o_hi : (o_hi - o_lo) * cum + o_lo;
        olo = knot;
        ohi = next;
        knot = next;
    }
    const T d0 = p.min_d + softplus(rqs_dlogit(p, dacc, bin));
    const T d1 = p.min_d + softplus(rqs_dlogit(p, dacc, bin + 1));
    if (!inverse)
        rqs_eval_bin<T>(x, blo, bhi - blo, olo, ohi - olo, d0, d1, false, y, lad);
    else
        rqs_eval_bin<T>(x, olo, ohi - olo, blo, bhi - blo, d0, d1, true, y, lad);
}

// Knot table of one batch-shared spline (PiecewiseRationalQuadraticCDF, nsf/coupling.py:221-253): the
// reference expands the (features, K) parameters to the batch and recomputes identical tables B
// times; here they are built once per workgroup.  Layout per feature: cumw[K+1] | cumh[K+1] | deriv[K+1].
template <typename T, typename WAcc, typename HAcc, typename DAcc>
__device__ __forceinline__ void rqs_build_table(const RqsParams<T> &p, const WAcc &wacc, const HAcc &hacc,
                                                const DAcc &dacc, T *tab) {
    const int K = p.K;
    T mw = wacc(0), mh = hacc(0);
    for (int k = 1; k < K; ++k) {
        mw = M<T>::fmax(mw, wacc(k));
        mh = M<T>::fmax(mh, hacc(k));
    }
    T sw = T(0), sh = T(0);
    for (int k = 0; k < K; ++k) {
        sw += M<T>::exp(wacc(k) - mw);
        sh += M<T>::exp(hacc(k) - mh);
    }
    T cw = T(0), ch = T(0);
    tab[0] = p.left;
    tab[K + 1] = p.bottom;
    for (int k = 0; k < K; ++k) {
        cw += p.min_w + p.scale_w * (M<T>::exp(wacc(k) - mw) / sw);
        ch += p.min_h + p.scale_h * (M<T>::exp(hacc(k) - mh) / sh);
        tab[k + 1] = (k == K - 1) ? p.right : (p.right - p.left) * cw + p.left;
        tab[K + 1 + k + 1] = (k == K - 1) ? p.top : (p.top - p.bottom) * ch + p.bottom;
    }
    for (int j = 0; j <= K; ++j) tab[2 * (K + 1) + j] = p.min_d + softplus(rqs_dlogit(p, dacc, j));
}

template <typename T>
__device__ __forceinline__ void rqs_eval_table(const RqsParams<T> &p, T x, const T *tab, bool inverse, T &y,
                                               T &lad) {
    if (!rqs_inside(p, x)) {
        y = p.dfull ? T(0) : x;
        lad = T(0);
        return;
    }
    const int K = p.K;
    const T *cw = tab, *ch = tab + (K + 1), *dv = tab + 2 * (K + 1);
    const T *srch = inverse ? ch : cw;
    int bin = 0;
    for (int k = 1; k < K; ++k)
        if (x >= srch[k]) bin = k;
    rqs_eval_bin<T>(x, cw[bin], cw[bin + 1] - cw[bin], ch[bin], ch[bin + 1] - ch[bin], dv[bin], dv[bin + 1],
                    inverse, y, lad);
}

// ---- wave / block reductions -------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Sum over a 256-thread (or smaller, multiple of 64) block; result valid in every thread.
template <typename T> __device__ __forceinline__ T block_sum(T v, T *scratch /* >= 16 */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) scratch[wid] = v;
    __syncthreads();
    T r = T(0);
    for (int i = 0; i < nw; ++i) r += scratch[i];
    return r;
}

template <typename T> __device__ __forceinline__ void ld_store(T *dst, T v, int acc) {
    if (acc == NF_LD_WRITE) *dst = v;
    else if (acc == NF_LD_ADD) *dst += v;
    else *dst -= v;
}

static inline int grid_for(int64_t work_items, int per_block, int max_blocks = 256 * 8) {
    int64_t g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (int)g;
}

// Dynamic LDS above 64 KB needs a per-kernel opt-in.  hipFuncSetAttribute is a slow, synchronising host call (measured
// in milliseconds when it lands between launches), so each launch site remembers, per device, the largest size it has
// opted in (the attribute belongs to the kernel image of the CURRENT device).
constexpr int NF_MAX_DEVICES = 64;
struct LdsOptIn {
    size_t opted[NF_MAX_DEVICES];  // 0 = nothing beyond the default 64 KB yet
};
inline int opt_in_lds(const void *kernel, size_t lds, LdsOptIn &state) {
    if (lds <= 64 * 1024) return NF_OK;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= NF_MAX_DEVICES) return NF_EIO;
    if (lds <= state.opted[dev]) return NF_OK;
    if (hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return NF_ENOTSUP;
    state.opted[dev] = lds;
    return NF_OK;
}

}  // namespace nf
