// fused_common.hpp -- pieces shared by the fused NSF layer kernels (rqs_fused.hip: exact fp32 MFMA; rqs_fused_x3.hip:
// error-compensated split-bf16 MFMA): packed-blob layout helpers, the branch-free spline epilogue on registers.
#pragma once
#include "common.hpp"

namespace nf {

constexpr int F_MAX_LAYERS = 64;   // layers of one shape a persistent chain launch takes (rqs_fused.hip, rqs_fused_x3.hip)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int F_D = 64;            // features
constexpr int F_NI = 32;           // identity features (= transform features)
constexpr int F_H = 128;           // hidden units
constexpr int F_K = 8;             // bins
constexpr int F_M = 3 * F_K - 1;   // 23 parameters per transform feature
constexpr int F_STAGE = 4096;      // floats per stage (16 KB)
constexpr int F_TABW = 3 * (F_K + 1);  // 27 words per unconditional-spline table row
constexpr int F_HDR = 64;          // header floats

// ---- packed blob layout (floats) ------------------------------------------------------------------------------
//   [0, F_HDR)                         header: [0] = magic, [1] = num_blocks, [2] = has_lu, [3] = lu log|det|
//   small section (copied to LDS at kernel start):
//     bias_init   [4 rowblocks][2 halves][16]                       128
//     bias_hidden [2*nblk][4][2][16]                                 256 * nblk
//     bias_final  [8 groups][3 rowblocks][2][16]                     768
//     tables      [32 identity features][27]                         864
//     bias_lu     [2 directions][2 rowblocks][2][16]                 128
//   stages (16 KB each, 16-byte aligned): init | hidden (8 per block) | final (24) | lu density | lu sample
struct FusedLayout {
    int nblk;
    int K = F_K;    // bins: 4 | 8 | 16 (exact-fp32 kernel; the split-bf16 and training variants are K = 8 only).  A transform
                    // feature occupies MP = 3 K of the lane's slots (3 K - 1 parameters + 1 pad), a group of 3 row-blocks (96
                    // rows, 48 slots per lane) holds 16 / K features per lane-half, so the final layer is K groups = 3 K stages
    __host__ __device__ int ngroups() const { return K; }
    __host__ __device__ int nfinal() const { return 3 * K; }
    __host__ __device__ int tabw() const { return 3 * (K + 1); }
    __host__ __device__ int bias_final_floats() const { return 96 * K; }
    __host__ __device__ int small_floats() const { return 128 + 256 * nblk + bias_final_floats() + F_NI * tabw() + 128; }
    __host__ __device__ int off_bias_init() const { return 0; }
    __host__ __device__ int off_bias_hidden(int lin) const { return 128 + 128 * lin; }
    __host__ __device__ int off_bias_final() const { return 128 + 256 * nblk; }
    __host__ __device__ int off_tables() const { return off_bias_final() + bias_final_floats(); }
    __host__ __device__ int off_bias_lu(int dir) const { return off_tables() + F_NI * tabw() + 64 * dir; }
    __host__ __device__ int lu_stage(int dir) const { return 1 + 8 * nblk + nfinal() + dir; }
    __host__ __device__ int small_padded() const { return (small_floats() + 1023) / 1024 * 1024; }
    __host__ __device__ int off_stages() const { return F_HDR + small_padded(); }  // multiple of 4 floats
    __host__ __device__ int nstages(bool lu) const { return 1 + 8 * nblk + nfinal() + (lu ? 1 : 0); }
    __host__ __device__ int64_t total_floats() const { return (int64_t)off_stages() + (int64_t)(nstages(false) + 2) * F_STAGE; }
};

// Row of the final layer (0 .. 32 (3 K - 1) - 1) held by MFMA row `rho` (0..31) of row-block rb (0..2) of group g (0 .. K - 1),
// or -1 for a padding row.  A lane-half hh ends up with FPL = 16 / K features per group, MP = 3 K slots each (3 K - 1 used):
// feature tf = 8 Q + 4 hh + j, Q = g / (K / 4) the lane's 16-column chunk, j = (g % (K / 4)) FPL + f.
template <int KB>
__host__ __device__ inline int final_row_k(int g, int rb, int rho) {
    constexpr int MP = 3 * KB, M = 3 * KB - 1, FPL = 16 / KB, GQ = KB / 4;
    const int q = rho >> 3, hh = (rho >> 2) & 1, r = rho & 3;
    const int v = 16 * rb + 4 * q + r;  // 0..47: position in the lane's parameter list
    const int f = v / MP, prm = v % MP;
    if (prm >= M) return -1;
    const int tf = 8 * (g / GQ) + 4 * hh + (g % GQ) * FPL + f;
    return tf * M + prm;
}
__host__ __device__ inline int final_row(int g, int rb, int rho) { return final_row_k<F_K>(g, rb, rho); }

// nf_final_bwd (final_bwd.hip): final-layer row (of the (32 * 23, hidden) weight) that k-entry hq (0..3) of k-step v (0..23) of
// group g (0..7) contracts over on v_mfma_f32_16x16x4_f32 -- the four transform features of the group, parameter v; -1: the pad
// slot (v = 23).
__host__ __device__ inline int final_bwd_row(int g, int v, int hq) {
    return v >= F_M ? -1 : (8 * (g >> 1) + 4 * (hq >> 1) + 2 * (g & 1) + (hq & 1)) * F_M + v;
}

// Output column of MFMA row rho (0..31) of LU row-block m (0..1): chosen so that C register `reg` of row-block m is
// the lane's stash slot 16 m + reg (slot c = 8 Q + column-in-chunk, chunk Q = columns [16 Q + 8 hh, +8)).
__host__ __device__ inline int lu_out_col(int m, int rho) {
    const int q = rho >> 3, hh = (rho >> 2) & 1, r = rho & 3;
    return 16 * (2 * m + (q >> 1)) + 8 * hh + 4 * (q & 1) + r;
}
// Input column contracted by k-group s (0..7), k-half hk, element r4: the lane's stash slot 4 s + r4.
__host__ __device__ inline int lu_in_col(int s, int hk, int r4) { return 16 * (s >> 1) + 8 * hk + 4 * (s & 1) + r4; }

// ---- branch-free fp32 math for the epilogue -------------------------------------------------------------------
// The spline evaluations must live in the same basic block as the MFMAs they hide behind, so nothing here may
// branch.  Hardware transcendentals (v_exp_f32 / v_log_f32 / v_rcp_f32 / v_sqrt_f32, <= 1 ulp each) replace the libm
// calls of the unfused kernels; softmax arguments are <= 0 and bounded, log arguments are O(1), so the absolute
// error stays at the 1e-7 level (parity tests: fused vs unfused vs oracle vs reference golden vectors).
__device__ __forceinline__ float fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float flog(float x) { return __builtin_amdgcn_logf(x) * 0.69314718055994530942f; }
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
// softplus = log1p(exp(x)) (threshold 20 like torch); log1p(t) = log(1+t) * t / ((1+t) - 1) keeps full relative
// accuracy for tiny t (Kahan), selected against the t itself when 1+t rounds to 1.
__device__ __forceinline__ float fsoftplus(float x) {
    const float t = fexp(fminf(x, 20.0f));
    const float u = 1.0f + t;
    const float w = u - 1.0f;
    const float l1p = (w == 0.0f) ? t : flog(u) * (t * frcp(w));
    return x > 20.0f ? x : l1p;
}

// utils/splines.py:159-219 once the bin is known; same formulas as rqs_eval_bin, branch-free.
template <bool INVERSE>
__device__ __forceinline__ void rqs_eval_bin_fast(float x, float cw, float bw, float ch, float bh, float d0, float d1,
                                                  float &y, float &lad) {
    const float delta = bh * frcp(bw);
    const float dsum = d0 + d1 - 2.0f * delta;
    float theta, den;
    if (!INVERSE) {
        theta = (x - cw) * frcp(bw);
        const float t1mt = theta * (1.0f - theta);
        const float num = bh * (delta * theta * theta + d0 * t1mt);
        den = delta + dsum * t1mt;
        y = ch + num * frcp(den);
    } else {
        const float dy = x - ch;
        const float a = dy * dsum + bh * (delta - d0);
        const float b = bh * d0 - dy * dsum;
        const float c = -delta * dy;
        const float disc = b * b - 4.0f * a * c;
        theta = (2.0f * c) * frcp(-b - fsqrt(disc));
        y = theta * bw + cw;
        den = delta + dsum * (theta * (1.0f - theta));
    }
    const float omt = 1.0f - theta;
    const float dnum = delta * delta * (d1 * theta * theta + 2.0f * delta * (theta * omt) + d0 * omt * omt);
    const float l = flog(dnum) - 2.0f * flog(den);
    lad = INVERSE ? -l : l;
}

// ---- spline on register-resident parameters (KB bins, linear tails), static indexing only, branch-free ------
// prm[0..7] raw widths, prm[8..15] raw heights, prm[16..22] raw derivative logits.
template <bool INVERSE, int KB = F_K>
__device__ __forceinline__ void rqs_regs(const RqsParams<float> &p, float x, const float (&prm)[3 * KB], float &y,
                                         float &lad) {
    // prm[0..7] / prm[8..15] arrive pre-multiplied by log2(e)/sqrt(hidden) (pack_final_kernel): softmax = exp2(. - max)/sum.
    const bool inside = x >= p.left && x <= p.right;  // false for NaN (utils/splines.py:28)
    float mw = prm[0], mh = prm[KB];
#pragma unroll
    for (int k = 1; k < KB; ++k) {
        mw = fmaxf(mw, prm[k]);
        mh = fmaxf(mh, prm[KB + k]);
    }
    // inclusive prefix sums of the un-normalised softmax terms; knot_k = lo + (hi - lo) (k min + scale P_{k-1} / P_7)
    float pw[KB], ph[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) {
        const float ew = __builtin_amdgcn_exp2f(prm[k] - mw), eh = __builtin_amdgcn_exp2f(prm[KB + k] - mh);
        pw[k] = k == 0 ? ew : pw[k - 1] + ew;
        ph[k] = k == 0 ? eh : ph[k - 1] + eh;
    }
    const float cw = (p.right - p.left) * p.scale_w * frcp(pw[KB - 1]);
    const float ch = (p.top - p.bottom) * p.scale_h * frcp(ph[KB - 1]);
    float kw[KB + 1], kh[KB + 1];
    kw[0] = p.left;
    kh[0] = p.bottom;
    kw[KB] = p.right;
    kh[KB] = p.top;
#pragma unroll
    for (int k = 1; k < KB; ++k) {
        kw[k] = fmaf(pw[k - 1], cw, p.left + (p.right - p.left) * p.min_w * (float)k);
        kh[k] = fmaf(ph[k - 1], ch, p.bottom + (p.top - p.bottom) * p.min_h * (float)k);
    }
    int bin = 0;
    float slo = INVERSE ? kh[0] : kw[0], shi = INVERSE ? kh[1] : kw[1];
    float olo = INVERSE ? kw[0] : kh[0], ohi = INVERSE ? kw[1] : kh[1];
#pragma unroll
    for (int k = 1; k < KB; ++k) {
        const bool ge = x >= (INVERSE ? kh[k] : kw[k]);
        bin = ge ? k : bin;
        slo = ge ? (INVERSE ? kh[k] : kw[k]) : slo;
        shi = ge ? (INVERSE ? kh[k + 1] : kw[k + 1]) : shi;
        olo = ge ? (INVERSE ? kw[k] : kh[k]) : olo;
        ohi = ge ? (INVERSE ? kw[k + 1] : kh[k + 1]) : ohi;
    }
    float dl0 = p.edge_logit, dl1 = p.edge_logit;
#pragma unroll
    for (int k = 0; k < KB - 1; ++k) {
        dl0 = (bin == k + 1) ? prm[2 * KB + k] : dl0;  // padded logit j = bin  -> raw index bin - 1
        dl1 = (bin == k) ? prm[2 * KB + k] : dl1;      // padded logit j = bin+1 -> raw index bin
    }
    const float d0 = p.min_d + fsoftplus(dl0), d1 = p.min_d + fsoftplus(dl1);
    float yy, ll;
    if (!INVERSE)
        rqs_eval_bin_fast<false>(x, slo, shi - slo, olo, ohi - olo, d0, d1, yy, ll);
    else
        rqs_eval_bin_fast<true>(x, olo, ohi - olo, slo, shi - slo, d0, d1, yy, ll);
    y = inside ? yy : x;       // linear tails: identity outside, also for NaN / +-inf (utils/splines.py:40-41)
    lad = inside ? ll : 0.0f;
}

// ---- round 5: TWO spline elements per call (the fused kernels hand a lane the parameter lists of two features at a time) ----
// Same arithmetic as rqs_regs, element by element, arranged so that (a) every add / multiply / fused multiply-add works on a
// (feature 0, feature 1) register pair = one packed instruction (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: gfx950 issues them at
// the rate of the scalar forms), and (b) the bin is found by a 3-level binary descent that carries the candidate knots, heights and
// derivative logits along (3 compares + 30 selects per element) instead of a linear scan (19 compares + 49 selects).  Knots are
// non-decreasing (prefix sums of non-negative terms through a monotone fma), so the descent lands in the bin the scan finds.
// Compiled stand-alone (tools/ubench/epilogue_count.sh): 558 -> 388 vector instructions per element PAIR (density direction).
typedef float f32x2e __attribute__((ext_vector_type(2)));
typedef int i32x2e __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2e pk_fma(f32x2e a, f32x2e b, f32x2e c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2e pk_rcp(f32x2e a) { return f32x2e{frcp(a[0]), frcp(a[1])}; }
__device__ __forceinline__ f32x2e pk_log(f32x2e a) { return f32x2e{flog(a[0]), flog(a[1])}; }
__device__ __forceinline__ f32x2e pk_bc(float a) { return f32x2e{a, a}; }

// descent over N bins: s = search knots, o = the other axis' knots, d = padded derivative logits (N + 1 candidates each)
template <int N>
__device__ __forceinline__ void rqs_descend2(f32x2e x, const f32x2e (&s)[N + 1], const f32x2e (&o)[N + 1], const f32x2e (&d)[N + 1],
                                             f32x2e &slo, f32x2e &shi, f32x2e &olo, f32x2e &ohi, f32x2e &dl0, f32x2e &dl1) {
    if constexpr (N == 1) {
        slo = s[0]; shi = s[1]; olo = o[0]; ohi = o[1]; dl0 = d[0]; dl1 = d[1];
    } else {
        constexpr int H = N / 2;
        const i32x2e c = x >= s[H];
        f32x2e s2[H + 1], o2[H + 1], d2[H + 1];
#pragma unroll
        for (int i = 0; i <= H; ++i) {
            s2[i] = c ? s[H + i] : s[i];
            o2[i] = c ? o[H + i] : o[i];
            d2[i] = c ? d[H + i] : d[i];
        }
        rqs_descend2<H>(x, s2, o2, d2, slo, shi, olo, ohi, dl0, dl1);
    }
}

template <bool INVERSE>
__device__ __forceinline__ void rqs_eval_bin_fast2(f32x2e x, f32x2e cw, f32x2e bw, f32x2e ch, f32x2e bh, f32x2e d0, f32x2e d1,
                                                   f32x2e &y, f32x2e &lad) {
    const f32x2e delta = bh * pk_rcp(bw);
    const f32x2e dsum = d0 + d1 - 2.0f * delta;
    f32x2e theta, den;
    if (!INVERSE) {
        theta = (x - cw) * pk_rcp(bw);
        const f32x2e t1mt = theta * (1.0f - theta);
        const f32x2e num = bh * (delta * theta * theta + d0 * t1mt);
        den = delta + dsum * t1mt;
        y = ch + num * pk_rcp(den);
    } else {
        const f32x2e dy = x - ch;
        const f32x2e a = dy * dsum + bh * (delta - d0);
        const f32x2e b = bh * d0 - dy * dsum;
        const f32x2e c = -delta * dy;
        const f32x2e disc = b * b - 4.0f * a * c;
        theta = (2.0f * c) * pk_rcp(-b - f32x2e{fsqrt(disc[0]), fsqrt(disc[1])});
        y = theta * bw + cw;
        den = delta + dsum * (theta * (1.0f - theta));
    }
    const f32x2e omt = 1.0f - theta;
    const f32x2e dnum = delta * delta * (d1 * theta * theta + 2.0f * delta * (theta * omt) + d0 * omt * omt);
    const f32x2e l = pk_log(dnum) - 2.0f * pk_log(den);
    lad = INVERSE ? -l : l;
}

// elements (x0, prm0) and (x1, prm1); prm as in rqs_regs.  KB a power of two.
template <bool INVERSE, int KB = F_K>
__device__ __forceinline__ void rqs_regs2(const RqsParams<float> &p, float x0, float x1, const float (&prm0)[3 * KB],
                                          const float (&prm1)[3 * KB], float &y0, float &y1, float &lad0, float &lad1) {
    static_assert((KB & (KB - 1)) == 0, "binary descent");
    const f32x2e x = {x0, x1};
    const i32x2e inside = (x >= p.left) & (x <= p.right);          // false for NaN (utils/splines.py:28)
    f32x2e mw = {prm0[0], prm1[0]}, mh = {prm0[KB], prm1[KB]};
#pragma unroll
    for (int k = 1; k < KB; ++k) {
        mw = f32x2e{fmaxf(mw[0], prm0[k]), fmaxf(mw[1], prm1[k])};
        mh = f32x2e{fmaxf(mh[0], prm0[KB + k]), fmaxf(mh[1], prm1[KB + k])};
    }
    f32x2e pw[KB], ph[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) {
        const f32x2e aw = f32x2e{prm0[k], prm1[k]} - mw, ah = f32x2e{prm0[KB + k], prm1[KB + k]} - mh;
        const f32x2e ew = {__builtin_amdgcn_exp2f(aw[0]), __builtin_amdgcn_exp2f(aw[1])};
        const f32x2e eh = {__builtin_amdgcn_exp2f(ah[0]), __builtin_amdgcn_exp2f(ah[1])};
        pw[k] = k == 0 ? ew : pw[k - 1] + ew;
        ph[k] = k == 0 ? eh : ph[k - 1] + eh;
    }
    const f32x2e cw = ((p.right - p.left) * p.scale_w) * pk_rcp(pw[KB - 1]);
    const f32x2e ch = ((p.top - p.bottom) * p.scale_h) * pk_rcp(ph[KB - 1]);
    f32x2e kw[KB + 1], kh[KB + 1], dp[KB + 1];
    kw[0] = pk_bc(p.left);
    kh[0] = pk_bc(p.bottom);
    kw[KB] = pk_bc(p.right);
    kh[KB] = pk_bc(p.top);
    dp[0] = dp[KB] = pk_bc(p.edge_logit);
#pragma unroll
    for (int k = 1; k < KB; ++k) {
        kw[k] = pk_fma(pw[k - 1], cw, pk_bc(p.left + (p.right - p.left) * p.min_w * (float)k));
        kh[k] = pk_fma(ph[k - 1], ch, pk_bc(p.bottom + (p.top - p.bottom) * p.min_h * (float)k));
        dp[k] = f32x2e{prm0[2 * KB + k - 1], prm1[2 * KB + k - 1]};
    }
    f32x2e slo, shi, olo, ohi, dl0, dl1;
    if (!INVERSE)
        rqs_descend2<KB>(x, kw, kh, dp, slo, shi, olo, ohi, dl0, dl1);
    else
        rqs_descend2<KB>(x, kh, kw, dp, slo, shi, olo, ohi, dl0, dl1);
    const f32x2e d0 = p.min_d + f32x2e{fsoftplus(dl0[0]), fsoftplus(dl0[1])};
    const f32x2e d1 = p.min_d + f32x2e{fsoftplus(dl1[0]), fsoftplus(dl1[1])};
    f32x2e yy, ll;
    if (!INVERSE)
        rqs_eval_bin_fast2<false>(x, slo, shi - slo, olo, ohi - olo, d0, d1, yy, ll);
    else
        rqs_eval_bin_fast2<true>(x, olo, ohi - olo, slo, shi - slo, d0, d1, yy, ll);
    const f32x2e yr = inside ? yy : x;       // linear tails: identity outside, also for NaN / +-inf (utils/splines.py:40-41)
    const f32x2e lr = inside ? ll : pk_bc(0.0f);
    y0 = yr[0]; y1 = yr[1]; lad0 = lr[0]; lad1 = lr[1];
}

// ONE element with the binary bin descent of rqs_regs2 and scalar arithmetic: for kernels that have no registers to spare for a
// pair (nsf_wide.hip, made_fwd.hip: the pair version spilled 11-23 registers there); 279 -> 245 vector instructions per element.
template <int N>
__device__ __forceinline__ void rqs_descend1(float x, const float (&s)[N + 1], const float (&o)[N + 1], const float (&d)[N + 1],
                                             float &slo, float &shi, float &olo, float &ohi, float &dl0, float &dl1) {
    if constexpr (N == 1) {
        slo = s[0]; shi = s[1]; olo = o[0]; ohi = o[1]; dl0 = d[0]; dl1 = d[1];
    } else {
        constexpr int H = N / 2;
        const bool c = x >= s[H];
        float s2[H + 1], o2[H + 1], d2[H + 1];
#pragma unroll
        for (int i = 0; i <= H; ++i) {
            s2[i] = c ? s[H + i] : s[i];
            o2[i] = c ? o[H + i] : o[i];
            d2[i] = c ? d[H + i] : d[i];
        }
        rqs_descend1<H>(x, s2, o2, d2, slo, shi, olo, ohi, dl0, dl1);
    }
}

template <bool INVERSE, int KB = F_K>
__device__ __forceinline__ void rqs_regs_t(const RqsParams<float> &p, float x, const float (&prm)[3 * KB], float &y, float &lad) {
    static_assert((KB & (KB - 1)) == 0, "binary descent");
    const bool inside = x >= p.left && x <= p.right;
    float mw = prm[0], mh = prm[KB];
#pragma unroll
    for (int k = 1; k < KB; ++k) {
        mw = fmaxf(mw, prm[k]);
        mh = fmaxf(mh, prm[KB + k]);
    }
    float pw[KB], ph[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) {
        const float ew = __builtin_amdgcn_exp2f(prm[k] - mw), eh = __builtin_amdgcn_exp2f(prm[KB + k] - mh);
        pw[k] = k == 0 ? ew : pw[k - 1] + ew;
        ph[k] = k == 0 ? eh : ph[k - 1] + eh;
    }
    const float cw = (p.right - p.left) * p.scale_w * frcp(pw[KB - 1]);
    const float ch = (p.top - p.bottom) * p.scale_h * frcp(ph[KB - 1]);
    float kw[KB + 1], kh[KB + 1], dp[KB + 1];
    kw[0] = p.left;
    kh[0] = p.bottom;
    kw[KB] = p.right;
    kh[KB] = p.top;
    dp[0] = dp[KB] = p.edge_logit;
#pragma unroll
    for (int k = 1; k < KB; ++k) {
        kw[k] = fmaf(pw[k - 1], cw, p.left + (p.right - p.left) * p.min_w * (float)k);
        kh[k] = fmaf(ph[k - 1], ch, p.bottom + (p.top - p.bottom) * p.min_h * (float)k);
        dp[k] = prm[2 * KB + k - 1];
    }
    float slo, shi, olo, ohi, dl0, dl1;
    if (!INVERSE)
        rqs_descend1<KB>(x, kw, kh, dp, slo, shi, olo, ohi, dl0, dl1);
    else
        rqs_descend1<KB>(x, kh, kw, dp, slo, shi, olo, ohi, dl0, dl1);
    const float d0 = p.min_d + fsoftplus(dl0), d1 = p.min_d + fsoftplus(dl1);
    float yy, ll;
    if (!INVERSE)
        rqs_eval_bin_fast<false>(x, slo, shi - slo, olo, ohi - olo, d0, d1, yy, ll);
    else
        rqs_eval_bin_fast<true>(x, olo, ohi - olo, slo, shi - slo, d0, d1, yy, ll);
    y = inside ? yy : x;
    lad = inside ? ll : 0.0f;
}

// rqs_regs_t with the FIRST level of the bin descent taken before any knot is materialised (round 6, nsf_wide.hip's K = 16 / K = 8
// epilogues: 19-35 / 3-12 spilled registers with rqs_regs_t): one pass over the raw parameters leaves the two softmax totals and the
// two first-half sums -- enough for the middle knot --, the comparison picks a half, and only that half's KB / 2 widths, heights and
// KB / 2 + 1 derivatives are kept (selects on the RAW values) and run through the knot construction with the half's base sums.  Live
// arrays halve (2 KB + 3 (KB + 1) -> KB + 3 (KB / 2 + 1) floats) for KB more v_exp_f32 per element.  Same arithmetic per knot up to the
// order of the prefix sum (second half: first-half sum + prefix instead of one running prefix).
template <bool INVERSE, int KB = F_K>
__device__ __forceinline__ void rqs_regs_h(const RqsParams<float> &p, float x, const float (&prm)[3 * KB], float &y, float &lad) {
    static_assert((KB & (KB - 1)) == 0 && KB >= 4, "binary descent");
    constexpr int H = KB / 2;
    const bool inside = x >= p.left && x <= p.right;
    float mw = prm[0], mh = prm[KB];
#pragma unroll
    for (int k = 1; k < KB; ++k) {
        mw = fmaxf(mw, prm[k]);
        mh = fmaxf(mh, prm[KB + k]);
    }
    float hw = 0.0f, hh = 0.0f, uw = 0.0f, uh = 0.0f;      // sums of the first / second half
#pragma unroll
    for (int k = 0; k < KB; ++k) {
        const float ew = __builtin_amdgcn_exp2f(prm[k] - mw), eh = __builtin_amdgcn_exp2f(prm[KB + k] - mh);
        if (k < H) { hw += ew; hh += eh; } else { uw += ew; uh += eh; }
    }
    const float cw = (p.right - p.left) * p.scale_w * frcp(hw + uw);
    const float ch = (p.top - p.bottom) * p.scale_h * frcp(hh + uh);
    const float sw = (p.right - p.left) * p.min_w, sh = (p.top - p.bottom) * p.min_h;
    const float kwm = fmaf(hw, cw, p.left + sw * (float)H), khm = fmaf(hh, ch, p.bottom + sh * (float)H);
    const bool c = x >= (INVERSE ? khm : kwm);
    float w2[H], h2[H], d2[H + 1];
#pragma unroll
    for (int i = 0; i < H; ++i) {
        w2[i] = c ? prm[H + i] : prm[i];
        h2[i] = c ? prm[KB + H + i] : prm[KB + i];
    }
#pragma unroll
    for (int i = 0; i <= H; ++i) {
        const float lo = i == 0 ? p.edge_logit : prm[2 * KB + i - 1];
        const float hi = i == H ? p.edge_logit : prm[2 * KB + H + i - 1];
        d2[i] = c ? hi : lo;
    }
    const float bw = c ? hw : 0.0f, bh = c ? hh : 0.0f, ko = c ? (float)H : 0.0f;
    float kw[H + 1], kh[H + 1];
    kw[0] = c ? kwm : p.left;
    kh[0] = c ? khm : p.bottom;
    kw[H] = c ? p.right : kwm;
    kh[H] = c ? p.top : khm;
    float aw = bw, ah = bh;
#pragma unroll
    for (int i = 1; i < H; ++i) {
        aw += __builtin_amdgcn_exp2f(w2[i - 1] - mw);
        ah += __builtin_amdgcn_exp2f(h2[i - 1] - mh);
        kw[i] = fmaf(aw, cw, p.left + sw * (ko + (float)i));
        kh[i] = fmaf(ah, ch, p.bottom + sh * (ko + (float)i));
    }
    float slo, shi, olo, ohi, dl0, dl1;
    if (!INVERSE)
        rqs_descend1<H>(x, kw, kh, d2, slo, shi, olo, ohi, dl0, dl1);
    else
        rqs_descend1<H>(x, kh, kw, d2, slo, shi, olo, ohi, dl0, dl1);
    const float d0 = p.min_d + fsoftplus(dl0), d1 = p.min_d + fsoftplus(dl1);
    float yy, ll;
    if (!INVERSE)
        rqs_eval_bin_fast<false>(x, slo, shi - slo, olo, ohi - olo, d0, d1, yy, ll);
    else
        rqs_eval_bin_fast<true>(x, olo, ohi - olo, slo, shi - slo, d0, d1, yy, ll);
    y = inside ? yy : x;
    lad = inside ? ll : 0.0f;
}

// Batch-shared spline from its LDS knot table (cumw[9] | cumh[9] | deriv[9]), branch-free.
template <bool INVERSE, int KB = F_K>
__device__ __forceinline__ void rqs_table_fast(const RqsParams<float> &p, float x, const float *tab, float &y, float &lad) {
    const bool inside = x >= p.left && x <= p.right;
    const float *srch = INVERSE ? tab + (KB + 1) : tab;
    int bin = 0;
#pragma unroll
    for (int k = 1; k < KB; ++k) bin = (x >= srch[k]) ? k : bin;
    const float cw0 = tab[bin], cw1 = tab[bin + 1], ch0 = tab[KB + 1 + bin], ch1 = tab[KB + 2 + bin];
    const float d0 = tab[2 * (KB + 1) + bin], d1 = tab[2 * (KB + 1) + bin + 1];
    float yy, ll;
    rqs_eval_bin_fast<INVERSE>(x, cw0, cw1 - cw0, ch0, ch1 - ch0, d0, d1, yy, ll);
    y = inside ? yy : x;
    lad = inside ? ll : 0.0f;
}

__device__ __forceinline__ f32x16 load_bias16(const float *src) {
    const f32x4 a = *reinterpret_cast<const f32x4 *>(src), b = *reinterpret_cast<const f32x4 *>(src + 4),
                c = *reinterpret_cast<const f32x4 *>(src + 8), d = *reinterpret_cast<const f32x4 *>(src + 12);
    f32x16 v;
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
    v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
    v[8] = c[0]; v[9] = c[1]; v[10] = c[2]; v[11] = c[3];
    v[12] = d[0]; v[13] = d[1]; v[14] = d[2]; v[15] = d[3];
    return v;
}


}  // namespace nf
