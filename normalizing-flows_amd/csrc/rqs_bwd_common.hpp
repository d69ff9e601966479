// rqs_bwd_common.hpp -- device routines shared by the spline backward kernels (rqs_bwd.hip) and the fused final-layer
// backward (final_bwd.hip): forward-mode dual numbers on the closed-form bin evaluation (utils/splines.py:159-219) and the
// register-resident backward of one element of the default parametrisation (8 bins, linear tails, float32).
#pragma once
#include "fused_common.hpp"

namespace nf {

template <typename T, int N> struct Dual {
    T v;
    T d[N];
    __device__ __forceinline__ Dual() {}
    __device__ __forceinline__ Dual(T c) : v(c) {
#pragma unroll
        for (int i = 0; i < N; ++i) d[i] = T(0);
    }
    static __device__ __forceinline__ Dual var(T c, int idx) {
        Dual r(c);
        r.d[idx] = T(1);
        return r;
    }
};
#define DU template <typename T, int N> __device__ __forceinline__ Dual<T, N>
DU operator+(const Dual<T, N> &a, const Dual<T, N> &b) { Dual<T, N> r; r.v = a.v + b.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
DU operator-(const Dual<T, N> &a, const Dual<T, N> &b) { Dual<T, N> r; r.v = a.v - b.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
DU operator-(const Dual<T, N> &a) { Dual<T, N> r; r.v = -a.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = -a.d[i]; return r; }
DU operator*(const Dual<T, N> &a, const Dual<T, N> &b) { Dual<T, N> r; r.v = a.v * b.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
DU operator/(const Dual<T, N> &a, const Dual<T, N> &b) { Dual<T, N> r; const T inv = T(1) / b.v; r.v = a.v * inv;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * inv; return r; }
DU dlog(const Dual<T, N> &a) { Dual<T, N> r; r.v = M<T>::log(a.v); const T inv = T(1) / a.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * inv; return r; }
DU dsqrt(const Dual<T, N> &a) { Dual<T, N> r; r.v = M<T>::sqrt(a.v); const T h = T(0.5) / r.v;
#pragma unroll
    for (int i = 0; i < N; ++i) r.d[i] = a.d[i] * h; return r; }
#undef DU

// Same arithmetic as common.hpp::rqs_eval_bin on dual numbers.
// variables: 0 = x, 1 = cw (x-axis knot lo), 2 = cw_hi, 3 = ch (y-axis knot lo), 4 = ch_hi, 5 = d0, 6 = d1
template <typename T>
__device__ __forceinline__ void rqs_eval_bin_dual(T x, T cw, T cwh, T ch, T chh, T d0, T d1, bool inverse, T (&gy)[7],
                                                  T (&gl)[7]) {
    typedef Dual<T, 7> D7;
    const D7 X = D7::var(x, 0), CW = D7::var(cw, 1), CWH = D7::var(cwh, 2), CH = D7::var(ch, 3), CHH = D7::var(chh, 4),
             D0 = D7::var(d0, 5), D1 = D7::var(d1, 6);
    const D7 bw = CWH - CW, bh = CHH - CH;
    const D7 delta = bh / bw;
    const D7 two(T(2)), one(T(1)), four(T(4));
    const D7 dsum = D0 + D1 - two * delta;
    D7 y, lad;
    if (!inverse) {
        const D7 theta = (X - CW) / bw;
        const D7 omt = one - theta;
        const D7 t1mt = theta * omt;
        const D7 num = bh * (delta * theta * theta + D0 * t1mt);
        const D7 den = delta + dsum * t1mt;
        y = CH + num / den;
        const D7 dnum = delta * delta * (D1 * theta * theta + two * delta * t1mt + D0 * omt * omt);
        lad = dlog(dnum) - two * dlog(den);
    } else {
        const D7 dy = X - CH;
        const D7 a = dy * dsum + bh * (delta - D0);
        const D7 b = bh * D0 - dy * dsum;
        const D7 c = -(delta * dy);
        const D7 disc = b * b - four * a * c;
        const D7 root = (two * c) / (-b - dsqrt(disc));
        y = root * bw + CW;
        const D7 omr = one - root;
        const D7 t1mt = root * omr;
        const D7 den = delta + dsum * t1mt;
        const D7 dnum = delta * delta * (D1 * root * root + two * delta * t1mt + D0 * omr * omr);
        lad = -(dlog(dnum) - two * dlog(den));
    }
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        gy[i] = y.d[i];
        gl[i] = lad.d[i];
    }
}

// Vector-Jacobian product of the forward bin evaluation (utils/splines.py:184-219: y and logabsdet from x, the bin's knots
// cw < cwh on the x axis, ch < chh on the y axis and the derivatives d0, d1 at its ends) with the cotangents (gy, gl) of
// (y, logabsdet): ONE reverse sweep over the forward's ~25 operations instead of carrying seven forward-mode partials through
// every one of them (rqs_eval_bin_dual: ~500 instructions per element; this: ~90).  g[0..6] = d/d(x, cw, cwh, ch, chh, d0, d1).
// Branch-free, hardware reciprocals (<= 1 ulp) like the fused forward's epilogue.
__device__ __forceinline__ void rqs_eval_bin_vjp(float x, float cw, float cwh, float ch, float chh, float d0, float d1, float gy,
                                                 float gl, float (&g)[7]) {
    const float bw = cwh - cw, bh = chh - ch, ibw = frcp(bw);
    const float delta = bh * ibw;
    const float xm = x - cw, theta = xm * ibw, omt = 1.0f - theta, t1mt = theta * omt, th2 = theta * theta;
    const float dsum = d0 + d1 - 2.0f * delta;
    const float P = delta * th2 + d0 * t1mt;                      // num = bh P
    const float den = delta + dsum * t1mt, iden = frcp(den);
    const float Q = d1 * th2 + 2.0f * delta * t1mt + d0 * omt * omt;   // dnum = delta^2 Q
    // y = ch + bh P / den;  lad = log(delta^2 Q) - 2 log(den)
    const float num_b = gy * iden;                                 // cotangent of num
    const float den_b = -(num_b * bh * P + 2.0f * gl) * iden;      // cotangent of den
    const float Q_b = gl * frcp(Q);                                // lad = 2 log(delta) + log(Q) - 2 log(den)
    const float P_b = num_b * bh;
    float delta_b = 2.0f * gl * frcp(delta) + Q_b * 2.0f * t1mt + den_b + P_b * th2;
    const float dsum_b = den_b * t1mt;
    float t1mt_b = Q_b * 2.0f * delta + den_b * dsum + P_b * d0;
    const float d1_b = Q_b * th2 + dsum_b;
    const float d0_b = Q_b * omt * omt + P_b * t1mt + dsum_b;
    delta_b -= 2.0f * dsum_b;
    const float omt_b = Q_b * 2.0f * d0 * omt + t1mt_b * theta;
    const float theta_b = Q_b * 2.0f * d1 * theta + P_b * 2.0f * delta * theta + t1mt_b * omt - omt_b;
    float bh_b = num_b * P + delta_b * ibw;
    const float x_b = theta_b * ibw;
    const float ibw_b = theta_b * xm + delta_b * bh;
    const float bw_b = -ibw_b * ibw * ibw;
    g[0] = x_b;
    g[1] = -x_b - bw_b;
    g[2] = bw_b;
    g[3] = gy - bh_b;
    g[4] = bh_b;
    g[5] = d0_b;
    g[6] = d1_b;
}

// ---- default parametrisation (8 bins, linear tails, float32): everything in registers, static indexing ----------------
// Same derivation as rqs_element_bwd on the register layout of fused_common.hpp::rqs_regs: prm[0..7] / prm[8..15] arrive
// multiplied by log2(e) / wh_div (softmax through exp2), prm[16..22] are the raw derivative logits.  Writes the gradients
// of the RAW parameters to g[0..22] and returns gx.
template <bool INVERSE>
__device__ __forceinline__ float rqs_regs_bwd(const RqsParams<float> &p, float x, const float (&prm)[24], float gy_up,
                                              float gl_up, float (&g)[24], float inv_div) {
    const bool inside = x >= p.left && x <= p.right;
    float mw = prm[0], mh = prm[F_K];
#pragma unroll
    for (int k = 1; k < F_K; ++k) {
        mw = fmaxf(mw, prm[k]);
        mh = fmaxf(mh, prm[F_K + k]);
    }
    float ew[F_K], eh[F_K], pw[F_K], ph[F_K];
#pragma unroll
    for (int k = 0; k < F_K; ++k) {
        ew[k] = __builtin_amdgcn_exp2f(prm[k] - mw);
        eh[k] = __builtin_amdgcn_exp2f(prm[F_K + k] - mh);
        pw[k] = k == 0 ? ew[k] : pw[k - 1] + ew[k];
        ph[k] = k == 0 ? eh[k] : ph[k - 1] + eh[k];
    }
    const float rsw = frcp(pw[F_K - 1]), rsh = frcp(ph[F_K - 1]);
    const float cw = (p.right - p.left) * p.scale_w * rsw, ch = (p.top - p.bottom) * p.scale_h * rsh;
    float kw[F_K + 1], kh[F_K + 1];
    kw[0] = p.left; kh[0] = p.bottom; kw[F_K] = p.right; kh[F_K] = p.top;
#pragma unroll
    for (int k = 1; k < F_K; ++k) {
        kw[k] = fmaf(pw[k - 1], cw, p.left + (p.right - p.left) * p.min_w * (float)k);
        kh[k] = fmaf(ph[k - 1], ch, p.bottom + (p.top - p.bottom) * p.min_h * (float)k);
    }
    // bin on the searched axis (x axis for the forward spline, y axis for the inverse); knots and cumulative softmax
    // C_bin, C_{bin+1} of BOTH axes at that bin
    int bin = 0;
    float xlo = kw[0], xhi = kw[1], ylo = kh[0], yhi = kh[1];
    float Cw_lo = 0.0f, Cw_hi = pw[0] * rsw, Ch_lo = 0.0f, Ch_hi = ph[0] * rsh;
#pragma unroll
    for (int k = 1; k < F_K; ++k) {
        const bool ge = x >= (INVERSE ? kh[k] : kw[k]);
        bin = ge ? k : bin;
        xlo = ge ? kw[k] : xlo; xhi = ge ? kw[k + 1] : xhi;
        ylo = ge ? kh[k] : ylo; yhi = ge ? kh[k + 1] : yhi;
        Cw_lo = ge ? pw[k - 1] * rsw : Cw_lo; Cw_hi = ge ? pw[k] * rsw : Cw_hi;
        Ch_lo = ge ? ph[k - 1] * rsh : Ch_lo; Ch_hi = ge ? ph[k] * rsh : Ch_hi;
    }
    float dl0 = p.edge_logit, dl1 = p.edge_logit;
#pragma unroll
    for (int k = 0; k < F_K - 1; ++k) {
        dl0 = (bin == k + 1) ? prm[2 * F_K + k] : dl0;
        dl1 = (bin == k) ? prm[2 * F_K + k] : dl1;
    }
    const float d0 = p.min_d + fsoftplus(dl0), d1 = p.min_d + fsoftplus(dl1);
    float gv[7];
    if constexpr (!INVERSE) {
        rqs_eval_bin_vjp(x, xlo, xhi, ylo, yhi, d0, d1, gy_up, gl_up, gv);
    } else {
        float jy[7], jl[7];
        rqs_eval_bin_dual<float>(x, xlo, xhi, ylo, yhi, d0, d1, true, jy, jl);
#pragma unroll
        for (int i = 0; i < 7; ++i) gv[i] = gy_up * jy[i] + gl_up * jl[i];
    }
    const float g_cw_lo = bin == 0 ? 0.0f : gv[1], g_cw_hi = bin == F_K - 1 ? 0.0f : gv[2];   // pinned end knots
    const float g_ch_lo = bin == 0 ? 0.0f : gv[3], g_ch_hi = bin == F_K - 1 ? 0.0f : gv[4];
    const float fw = (p.right - p.left) * p.scale_w * inv_div, fh = (p.top - p.bottom) * p.scale_h * inv_div;
    const float base_w = g_cw_lo * Cw_lo + g_cw_hi * Cw_hi, base_h = g_ch_lo * Ch_lo + g_ch_hi * Ch_hi;
#pragma unroll
    for (int i = 0; i < F_K; ++i) {
        const float tw = (i < bin ? g_cw_lo : 0.0f) + (i <= bin ? g_cw_hi : 0.0f) - base_w;
        const float th = (i < bin ? g_ch_lo : 0.0f) + (i <= bin ? g_ch_hi : 0.0f) - base_h;
        g[i] = inside ? fw * (ew[i] * rsw) * tw : 0.0f;
        g[F_K + i] = inside ? fh * (eh[i] * rsh) * th : 0.0f;
    }
    const float s0 = dl0 > 20.0f ? 1.0f : sigmoid(dl0), s1 = dl1 > 20.0f ? 1.0f : sigmoid(dl1);
#pragma unroll
    for (int k = 0; k < F_K - 1; ++k) {
        const float a = (bin == k + 1 ? gv[5] * s0 : 0.0f) + (bin == k ? gv[6] * s1 : 0.0f);
        g[2 * F_K + k] = inside ? a : 0.0f;
    }
    g[F_M] = 0.0f;
    return inside ? gv[0] : gy_up;
}

// Two elements at once (the two transform features a lane-half owns in a final-layer group): the same arithmetic as
// rqs_regs_bwd<false>, statement by statement for both elements, so that the instruction stream of a wave that is alone on its SIMD
// carries two independent dependency chains (a dependent v_fma issues every ~4 cycles, independent ones every ~2; transcendentals
// have longer latencies still).  prm / g: [element][24]; returns gx of both.
__device__ __forceinline__ void rqs_regs_bwd_pair(const RqsParams<float> &p, const float (&x)[2], const float (&prm)[2][24],
                                                  const float (&gy_up)[2], float gl_up, float (&g)[2][24], float inv_div,
                                                  float (&gx)[2]) {
    bool inside[2];
    float mw[2], mh[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        inside[e] = x[e] >= p.left && x[e] <= p.right;
        mw[e] = prm[e][0];
        mh[e] = prm[e][F_K];
    }
#pragma unroll
    for (int k = 1; k < F_K; ++k) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            mw[e] = fmaxf(mw[e], prm[e][k]);
            mh[e] = fmaxf(mh[e], prm[e][F_K + k]);
        }
    }
    float ew[2][F_K], eh[2][F_K], pw[2][F_K], ph[2][F_K];
#pragma unroll
    for (int k = 0; k < F_K; ++k) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            ew[e][k] = __builtin_amdgcn_exp2f(prm[e][k] - mw[e]);
            eh[e][k] = __builtin_amdgcn_exp2f(prm[e][F_K + k] - mh[e]);
            pw[e][k] = k == 0 ? ew[e][k] : pw[e][k - 1] + ew[e][k];
            ph[e][k] = k == 0 ? eh[e][k] : ph[e][k - 1] + eh[e][k];
        }
    }
    float rsw[2], rsh[2], cw[2], ch[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        rsw[e] = frcp(pw[e][F_K - 1]);
        rsh[e] = frcp(ph[e][F_K - 1]);
        cw[e] = (p.right - p.left) * p.scale_w * rsw[e];
        ch[e] = (p.top - p.bottom) * p.scale_h * rsh[e];
    }
    float kw[2][F_K + 1], kh[2][F_K + 1];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        kw[e][0] = p.left; kh[e][0] = p.bottom; kw[e][F_K] = p.right; kh[e][F_K] = p.top;
    }
#pragma unroll
    for (int k = 1; k < F_K; ++k) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            kw[e][k] = fmaf(pw[e][k - 1], cw[e], p.left + (p.right - p.left) * p.min_w * (float)k);
            kh[e][k] = fmaf(ph[e][k - 1], ch[e], p.bottom + (p.top - p.bottom) * p.min_h * (float)k);
        }
    }
    int bin[2];
    float xlo[2], xhi[2], ylo[2], yhi[2], Cw_lo[2], Cw_hi[2], Ch_lo[2], Ch_hi[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        bin[e] = 0;
        xlo[e] = kw[e][0]; xhi[e] = kw[e][1]; ylo[e] = kh[e][0]; yhi[e] = kh[e][1];
        Cw_lo[e] = 0.0f; Cw_hi[e] = pw[e][0]; Ch_lo[e] = 0.0f; Ch_hi[e] = ph[e][0];     // un-normalised: x rs below
    }
#pragma unroll
    for (int k = 1; k < F_K; ++k) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const bool ge = x[e] >= kw[e][k];
            bin[e] = ge ? k : bin[e];
            xlo[e] = ge ? kw[e][k] : xlo[e]; xhi[e] = ge ? kw[e][k + 1] : xhi[e];
            ylo[e] = ge ? kh[e][k] : ylo[e]; yhi[e] = ge ? kh[e][k + 1] : yhi[e];
            Cw_lo[e] = ge ? pw[e][k - 1] : Cw_lo[e]; Cw_hi[e] = ge ? pw[e][k] : Cw_hi[e];
            Ch_lo[e] = ge ? ph[e][k - 1] : Ch_lo[e]; Ch_hi[e] = ge ? ph[e][k] : Ch_hi[e];
        }
    }
    float dl0[2], dl1[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) { dl0[e] = p.edge_logit; dl1[e] = p.edge_logit; }
#pragma unroll
    for (int k = 0; k < F_K - 1; ++k) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            dl0[e] = (bin[e] == k + 1) ? prm[e][2 * F_K + k] : dl0[e];
            dl1[e] = (bin[e] == k) ? prm[e][2 * F_K + k] : dl1[e];
        }
    }
    // softplus and its derivative (sigmoid) from one exponential: t = e^l, softplus = log1p(t), sigmoid = t / (1 + t)
    float d0[2], d1[2], s0[2], s1[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const float t0 = fexp(fminf(dl0[e], 20.0f)), t1 = fexp(fminf(dl1[e], 20.0f));
        const float u0 = 1.0f + t0, u1 = 1.0f + t1, w0 = u0 - 1.0f, w1 = u1 - 1.0f;
        const float r0 = frcp(u0), r1 = frcp(u1);
        const float l0 = (w0 == 0.0f) ? t0 : flog(u0) * (t0 * frcp(w0)), l1 = (w1 == 0.0f) ? t1 : flog(u1) * (t1 * frcp(w1));
        d0[e] = p.min_d + (dl0[e] > 20.0f ? dl0[e] : l0);
        d1[e] = p.min_d + (dl1[e] > 20.0f ? dl1[e] : l1);
        s0[e] = dl0[e] > 20.0f ? 1.0f : t0 * r0;
        s1[e] = dl1[e] > 20.0f ? 1.0f : t1 * r1;
    }
    float gv[2][7];
    rqs_eval_bin_vjp(x[0], xlo[0], xhi[0], ylo[0], yhi[0], d0[0], d1[0], gy_up[0], gl_up, gv[0]);
    rqs_eval_bin_vjp(x[1], xlo[1], xhi[1], ylo[1], yhi[1], d0[1], d1[1], gy_up[1], gl_up, gv[1]);
    const float fw = (p.right - p.left) * p.scale_w * inv_div, fh = (p.top - p.bottom) * p.scale_h * inv_div;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
#pragma unroll
        for (int i = 1; i < 7; ++i) gv[e][i] = inside[e] ? gv[e][i] : 0.0f;     // (outside the interval the bin evaluation is garbage: select, never multiply)
        const float g_cw_lo = bin[e] == 0 ? 0.0f : gv[e][1], g_cw_hi = bin[e] == F_K - 1 ? 0.0f : gv[e][2];   // pinned end knots
        const float g_ch_lo = bin[e] == 0 ? 0.0f : gv[e][3], g_ch_hi = bin[e] == F_K - 1 ? 0.0f : gv[e][4];
        const float base_w = (g_cw_lo * Cw_lo[e] + g_cw_hi * Cw_hi[e]) * rsw[e], base_h = (g_ch_lo * Ch_lo[e] + g_ch_hi * Ch_hi[e]) * rsh[e];
        const float fws = fw * rsw[e], fhs = fh * rsh[e];
#pragma unroll
        for (int i = 0; i < F_K; ++i) {
            const float tw = (i < bin[e] ? g_cw_lo : 0.0f) + (i <= bin[e] ? g_cw_hi : 0.0f) - base_w;
            const float th = (i < bin[e] ? g_ch_lo : 0.0f) + (i <= bin[e] ? g_ch_hi : 0.0f) - base_h;
            // (select, not multiply: NaN parameters of an element outside the interval must not leak into its zero gradient)
            g[e][i] = inside[e] ? fws * ew[e][i] * tw : 0.0f;
            g[e][F_K + i] = inside[e] ? fhs * eh[e][i] * th : 0.0f;
        }
        const float a0 = inside[e] ? gv[e][5] * s0[e] : 0.0f, a1 = inside[e] ? gv[e][6] * s1[e] : 0.0f;
#pragma unroll
        for (int k = 0; k < F_K - 1; ++k) g[e][2 * F_K + k] = (bin[e] == k + 1 ? a0 : 0.0f) + (bin[e] == k ? a1 : 0.0f);
        g[e][F_M] = 0.0f;
        gx[e] = inside[e] ? gv[e][0] : gy_up[e];
    }
}

}  // namespace nf
