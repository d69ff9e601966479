/*
 * nf_oracle_impl.h -- body of the CPU oracle, included twice by nf_oracle.c (REAL = float / double).
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the normflows 1.7.3 algorithms on the
 * coupling-layer hot path, written to mirror the reference's order of operations (arrays of K bins,
 * sequential cumsum, count-based searchsorted), NOT the structure of the HIP kernels.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.  It is pinned against golden
 * vectors produced by the real reference (tests/golden/make_golden.py, tests/test_oracle_golden.py).
 *
 * Each function cites the reference file:line (relative to the normflows repo root) it follows.
 */

#ifndef NFO_MAXK
#define NFO_MAXK 64
#endif

/* torch.nn.functional.softplus, beta=1, threshold=20 */
static inline REAL FN(softplus)(REAL x) { return x > (REAL)20 ? x : LOG1P(EXP(x)); }

/* F.softmax over K entries followed by utils/splines.py:126-136 (or :140-152): returns knots[K+1], binsz[K]. */
static void FN(knots)(const REAL *unnorm, int K, REAL div, REAL min_bin, REAL lo, REAL hi, REAL *knots, REAL *binsz) {
    REAL v[NFO_MAXK], m, s = 0, c = 0;
    int k;
    for (k = 0; k < K; ++k) v[k] = unnorm[k] / div; /* nsf/coupling.py:334-339 (in-place /= sqrt(hidden)) */
    m = v[0];
    for (k = 1; k < K; ++k) if (v[k] > m) m = v[k];
    for (k = 0; k < K; ++k) { v[k] = EXP(v[k] - m); s += v[k]; }
    for (k = 0; k < K; ++k) v[k] = v[k] / s;                       /* softmax            :126 */
    {
        const REAL scale = (REAL)(1.0 - (double)min_bin * K);      /* python-double scalar, cast to tensor dtype */
        for (k = 0; k < K; ++k) v[k] = min_bin + scale * v[k];     /*                    :127 */
    }
    knots[0] = 0;
    for (k = 0; k < K; ++k) { c += v[k]; knots[k + 1] = c; }       /* cumsum + pad       :128-129 */
    for (k = 0; k <= K; ++k) knots[k] = (hi - lo) * knots[k] + lo; /*                    :133 */
    knots[0] = lo;                                                 /*                    :134 */
    knots[K] = hi;                                                 /*                    :135 */
    for (k = 0; k < K; ++k) binsz[k] = knots[k + 1] - knots[k];    /*                    :136 */
}

/* utils/splines.py:100-219 rational_quadratic_spline for ONE element. derivs_unnorm has K+1 entries. */
static void FN(rqs_one)(REAL x, const REAL *uw, const REAL *uh, const REAL *ud_padded, int K, REAL div, int inverse,
                        REAL left, REAL right, REAL bottom, REAL top, REAL min_w, REAL min_h, REAL min_d, REAL *y,
                        REAL *lad) {
    REAL cumw[NFO_MAXK + 1], cumh[NFO_MAXK + 1], w[NFO_MAXK], h[NFO_MAXK], dv[NFO_MAXK + 1];
    int k, bin, cnt = 0;
    FN(knots)(uw, K, div, min_w, left, right, cumw, w);
    FN(knots)(uh, K, div, min_h, bottom, top, cumh, h);
    for (k = 0; k <= K; ++k) dv[k] = min_d + FN(softplus)(ud_padded[k]); /* :138 */
    {   /* searchsorted :11-13, :154-157 : last knot += eps, count of (x >= knot) - 1 */
        REAL *srch = inverse ? cumh : cumw;
        srch[K] += (REAL)1e-6;
        for (k = 0; k <= K; ++k) cnt += (x >= srch[k]) ? 1 : 0;
        bin = cnt - 1;
        if (bin < 0) bin = 0;        /* reference raises (gather index -1); clamp documented in nf_mi355x.h */
        if (bin > K - 1) bin = K - 1;
    }
    {
        const REAL icw = cumw[bin], ibw = w[bin], ich = cumh[bin], ih = h[bin];
        const REAL delta = h[bin] / w[bin];
        const REAL d0 = dv[bin], d1 = dv[bin + 1];
        if (inverse) { /* :171-198 */
            const REAL a = (x - ich) * (d0 + d1 - 2 * delta) + ih * (delta - d0);
            const REAL b = ih * d0 - (x - ich) * (d0 + d1 - 2 * delta);
            const REAL c = -delta * (x - ich);
            const REAL disc = b * b - 4 * a * c;
            const REAL root = (2 * c) / (-b - SQRT(disc));
            const REAL t1mt = root * (1 - root);
            const REAL den = delta + ((d0 + d1 - 2 * delta) * t1mt);
            const REAL dnum = (delta * delta) * (d1 * (root * root) + 2 * delta * t1mt + d0 * ((1 - root) * (1 - root)));
            *y = root * ibw + icw;
            *lad = -(LOG(dnum) - 2 * LOG(den));
        } else { /* :199-219 */
            const REAL theta = (x - icw) / ibw;
            const REAL t1mt = theta * (1 - theta);
            const REAL num = ih * (delta * (theta * theta) + d0 * t1mt);
            const REAL den = delta + ((d0 + d1 - 2 * delta) * t1mt);
            const REAL dnum = (delta * delta) * (d1 * (theta * theta) + 2 * delta * t1mt + d0 * ((1 - theta) * (1 - theta)));
            *y = ich + num / den;
            *lad = LOG(dnum) - 2 * LOG(den);
        }
    }
}

/* utils/splines.py:16-97 unconstrained_rational_quadratic_spline (tails linear / circular) or the bare
 * bounded spline (tails none) for ONE element; ud has K-1 | K | K+1 raw derivative logits. */
static void FN(urqs_one)(REAL x, const REAL *uw, const REAL *uh, const REAL *ud, int K, int tails, REAL tail_bound,
                         REAL left, REAL right, REAL bottom, REAL top, REAL div, int inverse, REAL min_w, REAL min_h,
                         REAL min_d, REAL *y, REAL *lad) {
    REAL dpad[NFO_MAXK + 1];
    int k;
    if (tails == 0) {
        for (k = 0; k <= K; ++k) dpad[k] = ud[k];
        FN(rqs_one)(x, uw, uh, dpad, K, div, inverse, left, right, bottom, top, min_w, min_h, min_d, y, lad);
        return;
    }
    if (!((x >= -tail_bound) && (x <= tail_bound))) { /* :28-29, :40-41 (NaN compares false -> outside) */
        *y = x;
        *lad = 0;
        return;
    }
    if (tails == 1) { /* linear :33-38 */
        const REAL cst = (REAL)log(exp(1.0 - (double)min_d) - 1.0);
        dpad[0] = cst;
        for (k = 0; k < K - 1; ++k) dpad[k + 1] = ud[k];
        dpad[K] = cst;
    } else { /* circular :42-44 */
        for (k = 0; k < K; ++k) dpad[k] = ud[k];
        dpad[K] = dpad[0];
    }
    FN(rqs_one)(x, uw, uh, dpad, K, div, inverse, -tail_bound, tail_bound, -tail_bound, tail_bound, min_w, min_h,
                min_d, y, lad);
}

void FN(nfo_rqs_spline)(const REAL *x, const REAL *w, int64_t ldw, const REAL *h, int64_t ldh, const REAL *d,
                        int64_t ldd, REAL *y, REAL *lad, int64_t N, int K, int tails, double tail_bound, double left,
                        double right, double bottom, double top, double min_w, double min_h, double min_d,
                        double wh_div, int inverse) {
    int64_t n;
#pragma omp parallel for schedule(static)
    for (n = 0; n < N; ++n) {
        REAL yy, ll;
        FN(urqs_one)(x[n], w + n * ldw, h + n * ldh, d + n * ldd, K, tails, (REAL)tail_bound, (REAL)left, (REAL)right,
                     (REAL)bottom, (REAL)top, (REAL)wh_div, inverse, (REAL)min_w, (REAL)min_h, (REAL)min_d, &yy, &ll);
        y[n] = yy;
        if (lad) lad[n] = ll;
    }
}

/* nsf/coupling.py:71-128 Coupling.forward / inverse on (B, D) rows with the conditioner output given.
 * mode 0 = density (forward, :71-98), 1 = sample/identity half (:110-116), 2 = sample/transform half (:118-128). */
void FN(nfo_rqs_coupling)(const REAL *x, REAL *y, REAL *logdet, const REAL *cond, const REAL *uw, const REAL *uh,
                          const REAL *ud, const int64_t *iidx, int nI, const int64_t *tidx, int nT, int64_t B, int D,
                          int K, int tails, double tail_bound, double min_w, double min_h, double min_d, double wh_div,
                          int mode, int acc) {
    const int nd = tails == 1 ? K - 1 : (tails == 2 ? K : K + 1);
    const int Mrow = 2 * K + nd;
    const int inverse = mode != 0;
    int64_t b;
#pragma omp parallel for schedule(static)
    for (b = 0; b < B; ++b) {
        REAL ld = 0;
        int j;
        if (mode != 1) { /* transform features: _piecewise_cdf :329-362 then sum_except_batch :164 */
            REAL s = 0;
            for (j = 0; j < nT; ++j) {
                const REAL *row = cond + ((size_t)b * nT + j) * Mrow;
                REAL yy, ll;
                FN(urqs_one)(x[b * D + tidx[j]], row, row + K, row + 2 * K, K, tails, (REAL)tail_bound, 0, 1, 0, 1,
                             (REAL)wh_div, inverse, (REAL)min_w, (REAL)min_h, (REAL)min_d, &yy, &ll);
                y[b * D + tidx[j]] = yy;
                s += ll;
            }
            ld += s;
        }
        if (mode != 2) { /* identity features: unconditional CDF :221-253 (batch-shared, not scaled) */
            REAL s = 0;
            for (j = 0; j < nI; ++j) {
                REAL yy = x[b * D + iidx[j]], ll = 0;
                if (uw)
                    FN(urqs_one)(yy, uw + (size_t)j * K, uh + (size_t)j * K, ud + (size_t)j * nd, K, tails,
                                 (REAL)tail_bound, 0, 1, 0, 1, (REAL)1, inverse, (REAL)min_w, (REAL)min_h, (REAL)min_d,
                                 &yy, &ll);
                y[b * D + iidx[j]] = yy;
                s += ll;
            }
            ld += s;
        }
        if (acc == 0) logdet[b] = ld;
        else if (acc > 0) logdet[b] += ld;
        else logdet[b] -= ld;
    }
}

/* nets/resnet.py:92-104 ResidualNet.forward (ReLU, no context / batch-norm / dropout) with :37-50 blocks.
 * x (B, ldx), gathered columns idx[in_f] (NULL = leading columns); weights are torch nn.Linear layout (out, in). */
/* y[i] = bias[i] + sum_k Wt[k][i] * x[k], accumulated in k order per output (same rounding sequence as a
 * sequential dot product); the inner loop runs over outputs so the compiler can vectorise it without
 * re-associating any sum. */
static void FN(dense)(const REAL *Wt, const REAL *bias, const REAL *x, REAL *y, int in_f, int out_f) {
    int i, k;
    for (i = 0; i < out_f; ++i) y[i] = 0;
    for (k = 0; k < in_f; ++k) {
        const REAL xk = x[k];
        const REAL *wr = Wt + (size_t)k * out_f;
        for (i = 0; i < out_f; ++i) y[i] += wr[i] * xk;
    }
    for (i = 0; i < out_f; ++i) y[i] += bias[i];
}

static REAL *FN(transpose)(const REAL *W, int rows, int cols) { /* (rows, cols) -> (cols, rows) */
    REAL *T = (REAL *)malloc((size_t)rows * cols * sizeof(REAL));
    int r, c;
    for (r = 0; r < rows; ++r)
        for (c = 0; c < cols; ++c) T[(size_t)c * rows + r] = W[(size_t)r * cols + c];
    return T;
}

void FN(nfo_resnet_mlp)(const REAL *x, int64_t ldx, const int64_t *idx, int in_f, const REAL *w_init,
                        const REAL *b_init, const REAL *const *w_blocks, const REAL *const *b_blocks, int num_blocks,
                        const REAL *w_final, const REAL *b_final, int hidden, int out_f, REAL *out, int64_t B) {
    REAL *Wi = FN(transpose)(w_init, hidden, in_f), *Wf = FN(transpose)(w_final, out_f, hidden);
    REAL **Wb = (REAL **)malloc(sizeof(REAL *) * 2 * (num_blocks > 0 ? num_blocks : 1));
    int64_t b;
    int q;
    for (q = 0; q < 2 * num_blocks; ++q) Wb[q] = FN(transpose)(w_blocks[q], hidden, hidden);
#pragma omp parallel for schedule(static)
    for (b = 0; b < B; ++b) {
        REAL xin[1024], t0[1024], t1[1024], t2[1024];
        int i, k, blk;
        for (k = 0; k < in_f; ++k) xin[k] = x[b * ldx + (idx ? idx[k] : k)];   /* nsf/coupling.py:80 gather */
        FN(dense)(Wi, b_init, xin, t0, in_f, hidden);                          /* initial_layer  resnet.py:98 */
        for (blk = 0; blk < num_blocks; ++blk) {                               /* ResidualBlock  resnet.py:37-50 */
            for (i = 0; i < hidden; ++i) t1[i] = t0[i] > 0 ? t0[i] : 0;
            FN(dense)(Wb[2 * blk], b_blocks[2 * blk], t1, t2, hidden, hidden);
            for (i = 0; i < hidden; ++i) t2[i] = t2[i] > 0 ? t2[i] : 0;
            FN(dense)(Wb[2 * blk + 1], b_blocks[2 * blk + 1], t2, t1, hidden, hidden);
            for (i = 0; i < hidden; ++i) t0[i] = t0[i] + t1[i];
        }
        FN(dense)(Wf, b_final, t0, out + (size_t)b * out_f, hidden, out_f);    /* final_layer    resnet.py:103 */
    }
    for (q = 0; q < 2 * num_blocks; ++q) free(Wb[q]);
    free(Wb); free(Wi); free(Wf);
}

/* mixing.py:535-563 LULinearPermute; :402-412 _create_lower_upper; :414-473 forward/inverse_no_cache;
 * :514-532 upper_diag / logabsdet.  direction 0 = density (.inverse), 1 = sample (.forward). */
void FN(nfo_lu_linear_permute)(const REAL *x, REAL *y, REAL *logdet, const int64_t *perm, const REAL *lower_entries,
                               const REAL *upper_entries, const REAL *udiag_raw, const REAL *bias, int64_t B, int D,
                               double eps, int direction, int acc) {
    REAL *Lm = (REAL *)calloc((size_t)D * D, sizeof(REAL)), *Um = (REAL *)calloc((size_t)D * D, sizeof(REAL));
    REAL lad = 0;
    int r, c;
    int64_t b;
    size_t li = 0, ui = 0;
    for (r = 0; r < D; ++r)
        for (c = 0; c < D; ++c) {
            if (c < r) Lm[r * D + c] = lower_entries[li++];           /* np.tril_indices(D,-1): row-major */
            else if (c == r) { Lm[r * D + c] = 1; Um[r * D + c] = FN(softplus)(udiag_raw[r]) + (REAL)eps; }
        }
    for (r = 0; r < D; ++r)
        for (c = r + 1; c < D; ++c) Um[r * D + c] = upper_entries[ui++]; /* np.triu_indices(D,1): row-major */
    for (r = 0; r < D; ++r) lad += LOG(Um[r * D + r]);
    if (direction) lad = -lad;
#pragma omp parallel for schedule(static)
    for (b = 0; b < B; ++b) {
        REAL t[1024], u[1024];
        int i, j;
        if (direction == 0) {
            for (j = 0; j < D; ++j) t[j] = x[b * D + perm[j]];                 /* index_select :239 */
            for (i = 0; i < D; ++i) { REAL a = 0; for (j = 0; j < D; ++j) a += Um[i * D + j] * t[j]; u[i] = a; }
            for (i = 0; i < D; ++i) { REAL a = 0; for (j = 0; j < D; ++j) a += Lm[i * D + j] * u[j]; y[b * D + i] = a + bias[i]; }
        } else {
            for (j = 0; j < D; ++j) t[j] = x[b * D + j] - bias[j];
            for (i = 0; i < D; ++i) { REAL a = t[i]; for (j = 0; j < i; ++j) a -= Lm[i * D + j] * u[j]; u[i] = a; }
            for (i = D - 1; i >= 0; --i) { REAL a = u[i]; for (j = i + 1; j < D; ++j) a -= Um[i * D + j] * t[j]; t[i] = a / Um[i * D + i]; }
            for (j = 0; j < D; ++j) y[b * D + perm[j]] = t[j];                  /* inverse permutation :246-247 */
        }
        if (acc == 0) logdet[b] = lad;
        else if (acc > 0) logdet[b] += lad;
        else logdet[b] -= lad;
    }
    free(Lm);
    free(Um);
}

/* affine/coupling.py:209-229 MaskedAffineFlow */
void FN(nfo_masked_affine)(const REAL *z, const REAL *bm, const REAL *s, const REAL *t, REAL *y, REAL *logdet,
                           int64_t B, int64_t inner, int direction, int acc) {
    int64_t b;
    for (b = 0; b < B; ++b) {
        REAL ld = 0;
        int64_t i;
        for (i = 0; i < inner; ++i) {
            const int64_t o = b * inner + i;
            REAL si = s ? s[o] : 0, ti = t ? t[o] : 0;
            const REAL zm = bm[i] * z[o];
            if (!isfinite(si)) si = (REAL)NAN;
            if (!isfinite(ti)) ti = (REAL)NAN;
            if (direction == 0) y[o] = zm + (1 - bm[i]) * (z[o] * EXP(si) + ti);
            else y[o] = zm + (1 - bm[i]) * (z[o] - ti) * EXP(-si);
            ld += (1 - bm[i]) * si;
        }
        if (direction) ld = -ld;
        if (acc == 0) logdet[b] = ld;
        else if (acc > 0) logdet[b] += ld;
        else logdet[b] -= ld;
    }
}

/* affine/coupling.py:117-171 AffineCoupling + reshape.py:30-33,57-61 channel split / merge. */
void FN(nfo_affine_coupling)(const REAL *z, const REAL *param, REAL *y, REAL *logdet, int64_t B, int C, int c1,
                             int flip, int64_t HW, int scale_map, int direction, int acc) {
    const int c2 = C - c1;
    const int z1o = flip ? c2 : 0, z2o = flip ? 0 : c1;
    const int P = scale_map == 3 ? c2 : 2 * c2;
    int64_t b;
    for (b = 0; b < B; ++b) {
        const REAL *zr = z + b * (int64_t)C * HW, *pr = param + b * (int64_t)P * HW;
        REAL *yr = y + b * (int64_t)C * HW;
        REAL ld = 0;
        int64_t i;
        int c;
        for (i = 0; i < (int64_t)c1 * HW; ++i) yr[z1o * HW + i] = zr[z1o * HW + i];
        for (c = 0; c < c2; ++c)
            for (i = 0; i < HW; ++i) {
                const REAL v = zr[(z2o + c) * HW + i];
                REAL o;
                if (scale_map == 3) {
                    o = direction == 0 ? v + pr[c * HW + i] : v - pr[c * HW + i];
                } else {
                    const REAL sh = pr[(2 * c) * HW + i], sc = pr[(2 * c + 1) * HW + i];
                    if (scale_map == 0) {
                        o = direction == 0 ? v * EXP(sc) + sh : (v - sh) * EXP(-sc);
                        ld += direction == 0 ? sc : -sc;
                    } else {
                        const REAL sg = 1 / (1 + EXP(-(sc + 2)));
                        if (scale_map == 1) {
                            o = direction == 0 ? v / sg + sh : (v - sh) * sg;
                            ld += direction == 0 ? -LOG(sg) : LOG(sg);
                        } else {
                            o = direction == 0 ? v * sg + sh : (v - sh) / sg;
                            ld += direction == 0 ? LOG(sg) : -LOG(sg);
                        }
                    }
                }
                yr[(z2o + c) * HW + i] = o;
            }
        if (acc == 0) logdet[b] = ld;
        else if (acc > 0) logdet[b] += ld;
        else logdet[b] -= ld;
    }
}

/* affine/coupling.py:38-54 AffineConstFlow.forward / inverse for s,t of shape (1,C,1,..,1) */
void FN(nfo_actnorm)(const REAL *z, const REAL *s, const REAL *t, REAL *y, REAL *logdet_scalar, REAL *logdet,
                     int64_t B, int C, int64_t HW, int direction, int acc) {
    REAL ssum = 0, ldv;
    int64_t i, n = B * (int64_t)C * HW;
    int c;
    for (c = 0; c < C; ++c) ssum += s[c];
    ldv = (direction == 0 ? 1 : -1) * (REAL)HW * ssum;
    for (i = 0; i < n; ++i) {
        c = (int)((i / HW) % C);
        y[i] = direction == 0 ? z[i] * EXP(s[c]) + t[c] : (z[i] - t[c]) * EXP(-s[c]);
    }
    if (logdet_scalar) *logdet_scalar = ldv;
    if (logdet)
        for (i = 0; i < B; ++i) {
            if (acc == 0) logdet[i] = ldv;
            else if (acc > 0) logdet[i] += ldv;
            else logdet[i] -= ldv;
        }
}

/* torch.mean / torch.std (unbiased) over (B, HW) per channel, normalization.py:23-26, :35-37 */
void FN(nfo_actnorm_stats)(const REAL *z, REAL *mean, REAL *stdu, int64_t B, int C, int64_t HW) {
    int c;
    for (c = 0; c < C; ++c) {
        double acc = 0, acc2 = 0, mu;
        int64_t b, p, n = B * HW;
        for (b = 0; b < B; ++b)
            for (p = 0; p < HW; ++p) acc += (double)z[(b * C + c) * HW + p];
        mu = acc / (double)n;
        for (b = 0; b < B; ++b)
            for (p = 0; p < HW; ++p) { const double dl = (double)z[(b * C + c) * HW + p] - mu; acc2 += dl * dl; }
        mean[c] = (REAL)mu;
        stdu[c] = (REAL)sqrt(acc2 / (double)(n - 1));
    }
}

void FN(nfo_actnorm_init)(const REAL *mean, const REAL *stdu, REAL *s, REAL *t, int C, int direction) {
    int c;
    for (c = 0; c < C; ++c) {
        if (direction == 0) { s[c] = -LOG(stdu[c] + (REAL)1e-6); t[c] = -mean[c] * EXP(s[c]); } /* normalization.py:23-27 */
        else { s[c] = LOG(stdu[c] + (REAL)1e-6); t[c] = mean[c]; }                              /* :35-37 */
    }
}

/* mixing.py:88-104 _assemble_W */
void FN(nfo_inv1x1_assemble)(const REAL *P, const REAL *L, const REAL *U, const REAL *sign_S, const REAL *log_S,
                             REAL *W, REAL *logdet_unit, int C, int inverse) {
    const int n = C * C;
    REAL *Lp = (REAL *)calloc(n, sizeof(REAL)), *Up = (REAL *)calloc(n, sizeof(REAL)), *T1 = (REAL *)calloc(n, sizeof(REAL));
    double *Ld = (double *)calloc(n, sizeof(double)), *Ud = (double *)calloc(n, sizeof(double));
    double *Li = (double *)calloc(n, sizeof(double)), *Ui = (double *)calloc(n, sizeof(double));
    int r, c, k;
    REAL ls = 0;
    for (r = 0; r < C; ++r) ls += log_S[r];
    if (logdet_unit) *logdet_unit = inverse ? -ls : ls;
    for (r = 0; r < C; ++r)
        for (c = 0; c < C; ++c) {
            Lp[r * C + c] = c < r ? L[r * C + c] : (c == r ? (REAL)1 : (REAL)0);
            Up[r * C + c] = c > r ? U[r * C + c] : (c == r ? sign_S[r] * EXP(log_S[r]) : (REAL)0);
            Ld[r * C + c] = Lp[r * C + c];
            Ud[r * C + c] = Up[r * C + c];
        }
    if (!inverse) { /* W = P @ L @ U */
        for (r = 0; r < C; ++r)
            for (c = 0; c < C; ++c) { REAL a = 0; for (k = 0; k < C; ++k) a += P[r * C + k] * Lp[k * C + c]; T1[r * C + c] = a; }
        for (r = 0; r < C; ++r)
            for (c = 0; c < C; ++c) { REAL a = 0; for (k = 0; k < C; ++k) a += T1[r * C + k] * Up[k * C + c]; W[r * C + c] = a; }
    } else { /* torch.inverse(L.double()), torch.inverse(U.double()) -> dtype; W = U_inv @ L_inv @ P.t() */
        for (c = 0; c < C; ++c) {
            for (r = c; r < C; ++r) {
                double a = r == c ? 1.0 : 0.0;
                for (k = c; k < r; ++k) a -= Ld[r * C + k] * Li[k * C + c];
                Li[r * C + c] = a / Ld[r * C + r];
            }
            for (r = c; r >= 0; --r) {
                double a = r == c ? 1.0 : 0.0;
                for (k = r + 1; k <= c; ++k) a -= Ud[r * C + k] * Ui[k * C + c];
                Ui[r * C + c] = a / Ud[r * C + r];
            }
        }
        for (r = 0; r < n; ++r) { Lp[r] = (REAL)Li[r]; Up[r] = (REAL)Ui[r]; }
        for (r = 0; r < C; ++r)
            for (c = 0; c < C; ++c) { REAL a = 0; for (k = 0; k < C; ++k) a += Up[r * C + k] * Lp[k * C + c]; T1[r * C + c] = a; }
        for (r = 0; r < C; ++r)
            for (c = 0; c < C; ++c) { REAL a = 0; for (k = 0; k < C; ++k) a += T1[r * C + k] * P[c * C + k]; W[r * C + c] = a; }
    }
    free(Lp); free(Up); free(T1); free(Ld); free(Ud); free(Li); free(Ui);
}

/* mixing.py:106-133: conv2d(z, W.view(C,C,1,1)); log_det = logdet_unit * H * W */
void FN(nfo_inv1x1_conv)(const REAL *z, const REAL *W, const REAL *logdet_unit, REAL *y, REAL *logdet_scalar,
                         REAL *logdet, int64_t B, int C, int64_t HW, int acc) {
    int64_t b, p;
    int o, c;
    const REAL ldv = logdet_unit ? (*logdet_unit) * (REAL)HW : 0;
    for (b = 0; b < B; ++b)
        for (o = 0; o < C; ++o)
            for (p = 0; p < HW; ++p) {
                REAL a = 0;
                for (c = 0; c < C; ++c) a += W[o * C + c] * z[(b * C + c) * HW + p];
                y[(b * C + o) * HW + p] = a;
            }
    if (logdet_scalar) *logdet_scalar = ldv;
    if (logdet)
        for (b = 0; b < B; ++b) {
            if (acc == 0) logdet[b] = ldv;
            else if (acc > 0) logdet[b] += ldv;
            else logdet[b] -= ldv;
        }
}

/* distributions/base.py:94-103 DiagGaussian.log_prob */
void FN(nfo_diag_gaussian_log_prob)(const REAL *z, const REAL *loc, const REAL *log_scale, double ls_shift, REAL *out,
                                    int64_t B, int64_t d, int acc) {
    const REAL cst = (REAL)(-0.5 * (double)d * log(2.0 * M_PI));
    int64_t b, j;
    for (b = 0; b < B; ++b) {
        REAL a = 0;
        for (j = 0; j < d; ++j) {
            const REAL ls = log_scale[j] + (REAL)ls_shift;
            const REAL q = (z[b * d + j] - loc[j]) / EXP(ls);
            a += ls + (REAL)0.5 * (q * q);
        }
        if (acc == 0) out[b] = cst - a;
        else if (acc > 0) out[b] += cst - a;
        else out[b] -= cst - a;
    }
}

/* distributions/base.py:326-345 ClassCondDiagGaussian.log_prob: sample b uses row row_idx[b] (or b when NULL) of the
 * (num_rows, d) mean / log-scale tables (the transposed (*shape, num_classes) parameters, or the blended rows). */
void FN(nfo_diag_gaussian_log_prob_rows)(const REAL *z, const REAL *loc, const REAL *log_scale, const int64_t *row_idx,
                                         double ls_shift, REAL *out, int64_t B, int64_t d) {
    const REAL cst = (REAL)(-0.5 * (double)d * log(2.0 * M_PI));
    int64_t b, j;
    for (b = 0; b < B; ++b) {
        const int64_t row = row_idx ? row_idx[b] : b;
        REAL a = 0;
        for (j = 0; j < d; ++j) {
            const REAL ls = log_scale[row * d + j] + (REAL)ls_shift;
            const REAL q = (z[b * d + j] - loc[row * d + j]) / EXP(ls);
            a += ls + (REAL)0.5 * (q * q);
        }
        out[b] = cst - a;
    }
}

/* transforms.py:25-47 Logit.forward (direction 0) / Logit.inverse (direction 1); logsigmoid(v) = -softplus(-v) */
void FN(nfo_logit)(const REAL *z, REAL *y, REAL *logdet, int64_t B, int64_t inner, double alpha, int direction) {
    const REAL al = (REAL)alpha, beta = (REAL)(1.0 - 2.0 * alpha), log_beta = (REAL)log(1.0 - 2.0 * alpha);
    int64_t b, j;
    for (b = 0; b < B; ++b) {
        REAL a = 0;
        for (j = 0; j < inner; ++j) {
            const REAL v = z[b * inner + j];
            if (direction == 0) {
                const REAL spn = (-v > (REAL)20) ? -v : LOG1P(EXP(-v));
                const REAL spp = (v > (REAL)20) ? v : LOG1P(EXP(v));
                a -= spn + spp;
                y[b * inner + j] = ((REAL)1 / ((REAL)1 + EXP(-v)) - al) / beta;
            } else {
                const REAL u = al + beta * v;
                const REAL lu = LOG(u), l1 = LOG((REAL)1 - u);
                a -= lu + l1;
                y[b * inner + j] = lu - l1;
            }
        }
        logdet[b] = a + (direction == 0 ? -log_beta : log_beta) * (REAL)inner;
    }
}

/* reshape.py:116-128 Squeeze.forward (direction 0) / inverse (direction 1) */
void FN(nfo_squeeze)(const REAL *z, REAL *y, int64_t B, int C, int H, int W, int direction) {
    int64_t b;
    int c, h, w, i, j;
    if (direction == 0) {
        const int Co = C / 4;
        for (b = 0; b < B; ++b)
            for (c = 0; c < Co; ++c)
                for (i = 0; i < 2; ++i)
                    for (j = 0; j < 2; ++j)
                        for (h = 0; h < H; ++h)
                            for (w = 0; w < W; ++w)
                                y[((b * Co + c) * (2 * H) + 2 * h + i) * (2 * W) + 2 * w + j] =
                                    z[((b * C + 4 * c + 2 * i + j) * H + h) * W + w];
    } else {
        const int Ho = H / 2, Wo = W / 2;
        for (b = 0; b < B; ++b)
            for (c = 0; c < C; ++c)
                for (h = 0; h < Ho; ++h)
                    for (i = 0; i < 2; ++i)
                        for (w = 0; w < Wo; ++w)
                            for (j = 0; j < 2; ++j)
                                y[((b * 4 * C + 4 * c + 2 * i + j) * Ho + h) * Wo + w] =
                                    z[((b * C + c) * H + 2 * h + i) * W + 2 * w + j];
    }
}


/* Y[s][i] = bias[i] + sum_k Wt[k][i] X[s][k] for a chunk of n rows: the k-ordered accumulation per output is the same
 * rounding sequence as FN(dense); blocking over outputs keeps the Y tile in L1 and reads each weight once per chunk. */
static void FN(dense_batch)(const REAL *Wt, const REAL *bias, const REAL *X, int ldx, REAL *Y, int ldy, int n, int in_f,
                            int out_f, int relu_in) {
    int i0, s, k, i;
    for (i0 = 0; i0 < out_f; i0 += 128) {
        const int nb = (out_f - i0) < 128 ? (out_f - i0) : 128;
        for (s = 0; s < n; ++s)
            for (i = 0; i < nb; ++i) Y[(size_t)s * ldy + i0 + i] = 0;
        for (k = 0; k < in_f; ++k) {
            const REAL *wr = Wt + (size_t)k * out_f + i0;
            for (s = 0; s < n; ++s) {
                REAL xk = X[(size_t)s * ldx + k];
                REAL *yr = Y + (size_t)s * ldy + i0;
                if (relu_in && !(xk > 0)) xk = 0;
                for (i = 0; i < nb; ++i) yr[i] += wr[i] * xk;
            }
        }
        for (s = 0; s < n; ++s)
            for (i = 0; i < nb; ++i) Y[(size_t)s * ldy + i0 + i] += bias[i0 + i];
    }
}

/* Whole-flow log_prob of core.py:182-197 for a stack of L [CoupledRationalQuadraticSpline, LULinearPermute] pairs and
 * a DiagGaussian base, parallel over row chunks (each thread runs the complete 2L-layer chain on its rows).  Same
 * per-layer arithmetic as the functions above; used for the timed CPU baseline.
 * ptrs: per pair (9 + 4 nblk + 5) pointers:
 *   [ii, ti, w0, b0, (w, b) x 2 nblk, wf, bf, uw, uh, ud, perm, lower, upper, udiag, bias]   (w* pre-transposed = NULL) */
void FN(nfo_nsf_log_prob)(const REAL *x, REAL *logq, int64_t B, int D, int L, const void *const *ptrs, int nblk, int hidden,
                          int K, double tail_bound, const REAL *loc, const REAL *log_scale, double eps) {
    const int per = 9 + 4 * nblk + 5;
    const int nI = (D + 1) / 2, nT = D / 2, Mrow = 3 * K - 1, out_f = nT * Mrow;
    /* pre-transposed conditioner weights and assembled LU factors, once */
    REAL **Wi = (REAL **)malloc(sizeof(REAL *) * L), **Wf = (REAL **)malloc(sizeof(REAL *) * L);
    REAL **Wb = (REAL **)malloc(sizeof(REAL *) * L * 2 * (nblk > 0 ? nblk : 1));
    REAL **Lm = (REAL **)malloc(sizeof(REAL *) * L), **Um = (REAL **)malloc(sizeof(REAL *) * L);
    REAL *lad = (REAL *)malloc(sizeof(REAL) * L);
    int l, q, r, c;
    int64_t c0;
    for (l = 0; l < L; ++l) {
        const void *const *P = ptrs + (size_t)l * per;
        Wi[l] = FN(transpose)((const REAL *)P[2], hidden, nI);
        for (q = 0; q < 2 * nblk; ++q) Wb[l * 2 * nblk + q] = FN(transpose)((const REAL *)P[4 + 2 * q], hidden, hidden);
        Wf[l] = FN(transpose)((const REAL *)P[4 + 4 * nblk], out_f, hidden);
        {
            const REAL *lo = (const REAL *)P[per - 4], *up = (const REAL *)P[per - 3], *ud = (const REAL *)P[per - 2];
            size_t li = 0, ui = 0;
            REAL a = 0;
            Lm[l] = (REAL *)calloc((size_t)D * D, sizeof(REAL));
            Um[l] = (REAL *)calloc((size_t)D * D, sizeof(REAL));
            for (r = 0; r < D; ++r)
                for (c = 0; c < D; ++c) {
                    if (c < r) Lm[l][r * D + c] = lo[li++];
                    else if (c == r) { Lm[l][r * D + c] = 1; Um[l][r * D + c] = FN(softplus)(ud[r]) + (REAL)eps; }
                }
            for (r = 0; r < D; ++r)
                for (c = r + 1; c < D; ++c) Um[l][r * D + c] = up[ui++];
            for (r = 0; r < D; ++r) a += LOG(Um[l][r * D + r]);
            lad[l] = a;
        }
    }
#pragma omp parallel for schedule(dynamic, 1)
    for (c0 = 0; c0 < B; c0 += 64) {
        const int n = (int)((B - c0) < 64 ? (B - c0) : 64);
        REAL *z = (REAL *)malloc(sizeof(REAL) * 64 * D), *z2 = (REAL *)malloc(sizeof(REAL) * 64 * D);
        REAL *xin = (REAL *)malloc(sizeof(REAL) * 64 * nI), *t0 = (REAL *)malloc(sizeof(REAL) * 64 * hidden);
        REAL *t1 = (REAL *)malloc(sizeof(REAL) * 64 * hidden), *t2 = (REAL *)malloc(sizeof(REAL) * 64 * hidden);
        REAL *cond = (REAL *)malloc(sizeof(REAL) * 64 * out_f);
        REAL lq[64];
        int s, i, j, k, ll, blk;
        memcpy(z, x + c0 * D, sizeof(REAL) * n * D);
        for (s = 0; s < n; ++s) lq[s] = 0;
        for (ll = L - 1; ll >= 0; --ll) {
            const void *const *P = ptrs + (size_t)ll * per;
            const int64_t *ii = (const int64_t *)P[0], *ti = (const int64_t *)P[1], *perm = (const int64_t *)P[per - 5];
            const REAL *b0 = (const REAL *)P[3], *bfin = (const REAL *)P[5 + 4 * nblk];
            const REAL *uw = (const REAL *)P[6 + 4 * nblk], *uh = (const REAL *)P[7 + 4 * nblk], *ud = (const REAL *)P[8 + 4 * nblk];
            const REAL *bias = (const REAL *)P[per - 1];
            for (s = 0; s < n; ++s) { /* LULinearPermute.inverse (mixing.py:560-563) */
                REAL t[1024], u[1024];
                REAL *zr = z + s * D, *yr = z2 + s * D;
                for (j = 0; j < D; ++j) t[j] = zr[perm[j]];
                for (i = 0; i < D; ++i) { REAL a = 0; for (j = i; j < D; ++j) a += Um[ll][i * D + j] * t[j]; u[i] = a; }
                for (i = 0; i < D; ++i) { REAL a = 0; for (j = 0; j <= i; ++j) a += Lm[ll][i * D + j] * u[j]; yr[i] = a + bias[i]; }
                lq[s] += lad[ll];
                for (k = 0; k < nI; ++k) xin[s * nI + k] = yr[ii[k]];
            }
            /* ResidualNet conditioner (nets/resnet.py:92-104) on the chunk */
            FN(dense_batch)(Wi[ll], b0, xin, nI, t0, hidden, n, nI, hidden, 0);
            for (blk = 0; blk < nblk; ++blk) {
                FN(dense_batch)(Wb[ll * 2 * nblk + 2 * blk], (const REAL *)P[5 + 4 * blk], t0, hidden, t2, hidden, n, hidden, hidden, 1);
                FN(dense_batch)(Wb[ll * 2 * nblk + 2 * blk + 1], (const REAL *)P[7 + 4 * blk], t2, hidden, t1, hidden, n, hidden, hidden, 1);
                for (i = 0; i < n * hidden; ++i) t0[i] = t0[i] + t1[i];
            }
            FN(dense_batch)(Wf[ll], bfin, t0, hidden, cond, out_f, n, hidden, out_f, 0);
            for (s = 0; s < n; ++s) { /* CoupledRationalQuadraticSpline.inverse = prqct.forward (nsf/coupling.py:71-98) */
                REAL *zr = z + s * D, *yr = z2 + s * D;
                REAL st = 0, si = 0;
                for (j = 0; j < nT; ++j) {
                    const REAL *row = cond + (size_t)s * out_f + (size_t)j * Mrow;
                    REAL yy, l2;
                    FN(urqs_one)(yr[ti[j]], row, row + K, row + 2 * K, K, 1, (REAL)tail_bound, 0, 1, 0, 1,
                                 SQRT((REAL)hidden), 0, (REAL)1e-3, (REAL)1e-3, (REAL)1e-3, &yy, &l2);
                    zr[ti[j]] = yy;
                    st += l2;
                }
                for (j = 0; j < nI; ++j) {
                    REAL yy, l2;
                    FN(urqs_one)(yr[ii[j]], uw + (size_t)j * K, uh + (size_t)j * K, ud + (size_t)j * (K - 1), K, 1,
                                 (REAL)tail_bound, 0, 1, 0, 1, (REAL)1, 0, (REAL)1e-3, (REAL)1e-3, (REAL)1e-3, &yy, &l2);
                    zr[ii[j]] = yy;
                    si += l2;
                }
                lq[s] += st + si;
            }
        }
        for (s = 0; s < n; ++s) { /* DiagGaussian.log_prob (distributions/base.py:94-103) */
            const REAL cst = (REAL)(-0.5 * (double)D * log(2.0 * M_PI));
            REAL a = 0;
            int j2;
            for (j2 = 0; j2 < D; ++j2) {
                const REAL qv = (z[s * D + j2] - loc[j2]) / EXP(log_scale[j2]);
                a += log_scale[j2] + (REAL)0.5 * (qv * qv);
            }
            logq[c0 + s] = lq[s] + (cst - a);
        }
        free(z); free(z2); free(xin); free(t0); free(t1); free(t2); free(cond);
    }
    for (l = 0; l < L; ++l) {
        free(Wi[l]); free(Wf[l]); free(Lm[l]); free(Um[l]);
        for (q = 0; q < 2 * nblk; ++q) free(Wb[l * 2 * nblk + q]);
    }
    free(Wi); free(Wf); free(Wb); free(Lm); free(Um); free(lad);
}


/* affine/autoregressive.py:98-128 MaskedAffineAutoregressive._elementwise_forward (direction 0) / _inverse (1);
 * params (B, D, 2) = (unconstrained_scale, shift). */
void FN(nfo_maf_affine)(const REAL *x, const REAL *params, REAL *y, REAL *logdet, int64_t B, int D, int direction) {
    int64_t b;
    for (b = 0; b < B; ++b) {
        REAL a = 0;
        int j;
        for (j = 0; j < D; ++j) {
            const REAL u = params[(b * D + j) * 2], sh = params[(b * D + j) * 2 + 1];
            const REAL scale = 1 / (1 + EXP(-(u + 2))) + (REAL)1e-3;
            y[b * D + j] = direction == 0 ? scale * x[b * D + j] + sh : (x[b * D + j] - sh) / scale;
            a += LOG(scale);
        }
        if (logdet) logdet[b] = direction == 0 ? a : -a;
    }
}

/* nets/cnn.py:5-63 (ConvNet2d): one Conv2d(Cin, Cout, k, padding = k / 2, stride 1) with bias, NCHW, optionally followed by
 * LeakyReLU(slope) (torch.nn.LeakyReLU: x if x > 0 else slope x).  w (Cout, Cin, k, k), cross-correlation as torch.conv2d. */
void FN(nfo_conv2d_same)(const REAL *x, const REAL *w, const REAL *b, REAL *y, int64_t B, int Cin, int H, int W, int Cout,
                         int k, int leaky_on, REAL slope) {
    const int pad = k / 2;
    int64_t n;
#pragma omp parallel for schedule(static)
    for (n = 0; n < B * Cout; ++n) {
        const int64_t img = n / Cout;
        const int co = (int)(n - img * Cout);
        int yy, xx, ci, ky, kx;
        for (yy = 0; yy < H; ++yy)
            for (xx = 0; xx < W; ++xx) {
                REAL a = b ? b[co] : 0;
                for (ci = 0; ci < Cin; ++ci)
                    for (ky = 0; ky < k; ++ky) {
                        const int sy = yy + ky - pad;
                        if (sy < 0 || sy >= H) continue;
                        for (kx = 0; kx < k; ++kx) {
                            const int sx = xx + kx - pad;
                            if (sx < 0 || sx >= W) continue;
                            a += w[((co * (int64_t)Cin + ci) * k + ky) * k + kx] * x[((img * Cin + ci) * H + sy) * W + sx];
                        }
                    }
                if (leaky_on && !(a > 0)) a = a * slope;
                y[((img * Cout + co) * H + yy) * W + xx] = a;
            }
    }
}
