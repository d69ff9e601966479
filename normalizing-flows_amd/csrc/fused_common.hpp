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
