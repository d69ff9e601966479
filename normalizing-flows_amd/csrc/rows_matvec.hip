// rows_matvec.hip -- y_b = W x_b for every row b of a row-major (B, D) matrix, D <= 64 (one product also D <= 128), on exact-fp32 MFMA.
//
// The training path's products of batch rows with D x D parameter matrices -- the input gradient of LULinearPermute
// (mixing.py:535-563 under `loss.backward()`, core.py:87-102: gx = P U^T L^T gy and the triangular factors' inputs
// u = U x[perm], gu = L^T gy) -- are GEMMs of a (65 536 x 64) matrix with a 64 x 64 one: 33 MB of traffic around 0.5 GFLOP,
// HBM-bound.  One wave owns 32 rows: Y^T = W X^T with v_mfma_f32_32x32x2_f32, the contraction ordered so that lane-half hh
// contracts over columns [32 hh, 32 hh + 32) of its own row (eight 16-byte loads per lane, no LDS round trip for the
// activations); W is staged once per workgroup in LDS in A-operand order; the C registers of a lane are four 16-byte runs
// of its output row.  Rows beyond B and columns beyond D are zero padding.
#include "fused_common.hpp"

namespace nf {

constexpr int RM_D = 64;
constexpr int RM_NW = 4;

// LULinearPermute (mixing.py:402-473, :535-563) as ONE dense D x D matrix per direction, composed in fp64 by a single
// workgroup (parameter-side work, once per parameter version):
//   density (.inverse): y = L (U x[perm]) + b              -> Wd[i][perm[j]] = (L U)[i][j],            bias_d = b
//   sample  (.forward): y[perm[j]] = (U^-1 L^-1 (x - b))_j  -> Ws[perm[j]][k] = (U^-1 L^-1)[j][k],     bias_s = -Ws b
// out: Wd (D x D) | Ws (D x D) | bias_d (D) | bias_s (D) | log|det| = sum log(softplus(u_diag) + eps) (1), row-major fp32.
__global__ void __launch_bounds__(256)
lu_compose_kernel(const int64_t *__restrict__ perm, const float *__restrict__ lower_entries,
                  const float *__restrict__ upper_entries, const float *__restrict__ udiag_raw,
                  const float *__restrict__ bias, float eps, float *__restrict__ out, int D) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lu_raw[];
    const int N = D * D;
    double *A = reinterpret_cast<double *>(lu_raw), *Bm = A + N, *Cm = Bm + N, *Dm = Cm + N;
    __shared__ float sred[16];
    const int tid = threadIdx.x;
    float *Wd = out, *Ws = out + N, *bd = out + 2 * N, *bs = bd + D, *lad = bs + D;
    for (int i = tid; i < N; i += 256) {
        const int r = i / D, c = i - r * D;
        double l = 0.0, u = 0.0;
        if (c < r) l = (double)lower_entries[r * (r - 1) / 2 + c];
        else if (c == r) { l = 1.0; u = (double)(softplus(udiag_raw[r]) + eps); }
        else u = (double)upper_entries[r * (D - 1) - r * (r - 1) / 2 + (c - r - 1)];
        A[i] = l;
        Bm[i] = u;
    }
    float part = 0.0f;
    for (int i = tid; i < D; i += 256) part += logf(softplus(udiag_raw[i]) + eps);  // mixing.py:514-532
    const float ladv = block_sum(part, sred);
    if (tid == 0) *lad = ladv;
    __syncthreads();
    for (int i = tid; i < N; i += 256) {            // density: (L U)[r][c] -> column perm[c]
        const int r = i / D, c = i - r * D;
        double a = 0.0;
        for (int k = 0; k <= (r < c ? r : c); ++k) a += A[r * D + k] * Bm[k * D + c];
        Wd[r * D + (int)perm[c]] = (float)a;
    }
    for (int i = tid; i < D; i += 256) bd[i] = bias[i];
    for (int i = tid; i < N; i += 256) { Cm[i] = 0.0; Dm[i] = 0.0; }
    __syncthreads();
    for (int c = tid; c < D; c += 256) {            // Cm = L^-1, Dm = U^-1 (column solves)
        for (int r = c; r < D; ++r) {
            double a = (r == c) ? 1.0 : 0.0;
            for (int k = c; k < r; ++k) a -= A[r * D + k] * Cm[k * D + c];
            Cm[r * D + c] = a;
        }
        for (int r = c; r >= 0; --r) {
            double a = (r == c) ? 1.0 : 0.0;
            for (int k = r + 1; k <= c; ++k) a -= Bm[r * D + k] * Dm[k * D + c];
            Dm[r * D + c] = a / Bm[r * D + r];
        }
    }
    __syncthreads();
    for (int i = tid; i < N; i += 256) {            // A := U^-1 L^-1
        const int r = i / D, c = i - r * D;
        double a = 0.0;
        for (int k = (r > c ? r : c); k < D; ++k) a += Dm[r * D + k] * Cm[k * D + c];
        A[i] = a;
    }
    __syncthreads();
    for (int i = tid; i < N; i += 256) {
        const int j = i / D, k = i - j * D;
        Ws[(int)perm[j] * D + k] = (float)A[i];
    }
    for (int j = tid; j < D; j += 256) {
        double a = 0.0;
        for (int k = 0; k < D; ++k) a += A[j * D + k] * (double)bias[k];
        bs[(int)perm[j]] = (float)(-a);
    }
}

// Training-side helpers of LULinearPermute (mixing.py:402-473 under autograd): one launch instead of a dozen element-wise /
// indexing launches each.  nf_lu_factors: out = L (D x D, unit diagonal) | U (D x D, diagonal softplus(u) + eps) | Up (D x D,
// Up[:, perm[j]] = U[:, j], so that Up x = U x[perm]) | diag (D) | log|det| = sum log diag (1) | L^T (D x D) | Up^T (D x D)
// (the backward's two row mat-vecs take the transposes: emitted here instead of two transpose-copy launches per layer).
__device__ __forceinline__ void lu_factors_body(const int64_t *__restrict__ perm, const float *__restrict__ lower_entries,
                                                const float *__restrict__ upper_entries, const float *__restrict__ udiag_raw,
                                                float eps, float *__restrict__ out, int D) {
    __shared__ float sred[16];
    const int N = D * D, tid = threadIdx.x;
    float *Lm = out, *Um = out + N, *Up = out + 2 * N, *dg = out + 3 * N, *lad = dg + D, *LT = lad + 1, *UpT = LT + N;
    for (int i = blockIdx.x * 256 + tid; i < N; i += gridDim.x * 256) {      // one element per thread at D = 64 (16 workgroups)
        const int r = i / D, c = i - r * D;
        float l = 0.0f, u = 0.0f;
        if (c < r) l = lower_entries[r * (r - 1) / 2 + c];
        else if (c == r) { l = 1.0f; u = softplus(udiag_raw[r]) + eps; }
        else u = upper_entries[r * (D - 1) - r * (r - 1) / 2 + (c - r - 1)];
        Lm[i] = l;
        Um[i] = u;
        Up[r * D + (int)perm[c]] = u;
        LT[c * D + r] = l;
        UpT[(int)perm[c] * D + r] = u;
    }
    if (blockIdx.x != 0) return;
    float part = 0.0f;
    for (int i = tid; i < D; i += 256) {
        const float d = softplus(udiag_raw[i]) + eps;
        dg[i] = d;
        part += logf(d);
    }
    const float s = block_sum(part, sred);
    if (tid == 0) *lad = s;
}

__global__ void __launch_bounds__(256)
lu_factors_kernel(const int64_t *__restrict__ perm, const float *__restrict__ lower_entries,
                  const float *__restrict__ upper_entries, const float *__restrict__ udiag_raw, float eps,
                  float *__restrict__ out, int D) {
    lu_factors_body(perm, lower_entries, upper_entries, udiag_raw, eps, out, D);
}

// n layers in one launch (blockIdx.y = layer); table: n rows of 5 device pointers perm, lower, upper, udiag, out
__global__ void __launch_bounds__(256)
lu_factors_multi_kernel(const void *const *__restrict__ table, float eps, int D) {
    const void *const *row = table + (size_t)blockIdx.y * 5;
    lu_factors_body((const int64_t *)row[0], (const float *)row[1], (const float *)row[2], (const float *)row[3], eps,
                    (float *)row[4], D);
}

// Parameter gradients from the dense factor gradients gL, gU (D x D; gU's columns taken through perm when given) and the
// log-det cotangent gld (B, summed here): g_lower / g_upper = sign * the strictly triangular entries (packed row-major as the parameters are),
// g_udiag = sign * (gU[r][r] + gl_sum / diag[r]) * softplus'(u_r).
__global__ void __launch_bounds__(1024)
lu_param_grads_kernel(const float *__restrict__ gL, const float *__restrict__ gU, const int64_t *__restrict__ perm,
                      const float *__restrict__ gld, int64_t B, const float *__restrict__ udiag_raw, float eps, float sign,
                      float *__restrict__ g_lower, float *__restrict__ g_upper, float *__restrict__ g_udiag, int D) {
    __shared__ float sred[16];
    const int N = D * D;
    float gl = 0.0f;
    if (gld) {      // every workgroup sums the (B) cotangent itself, in the same order: no reduction launch, same value everywhere
        float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f;
        const int64_t B4 = (reinterpret_cast<uintptr_t>(gld) & 15) == 0 ? B / 4 : 0;
        const f32x4 *g4 = reinterpret_cast<const f32x4 *>(gld);
#pragma unroll 8
        for (int64_t b = threadIdx.x; b < B4; b += 1024) {
            const f32x4 v = g4[b];
            p0 += v[0]; p1 += v[1]; p2 += v[2]; p3 += v[3];
        }
        for (int64_t b = 4 * B4 + threadIdx.x; b < B; b += 1024) p0 += gld[b];
        gl = block_sum((p0 + p1) + (p2 + p3), sred);
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        const int r = i / D, c = i - r * D;
        if (c < r) { g_lower[r * (r - 1) / 2 + c] = sign * gL[i]; continue; }
        const float gu = perm ? gU[r * D + (int)perm[c]] : gU[i];      // column select of gu^T x folded in
        if (c > r) g_upper[r * (D - 1) - r * (r - 1) / 2 + (c - r - 1)] = sign * gu;
        else {
            const float u = udiag_raw[r], d = softplus(u) + eps;
            const float sg = u > 20.0f ? 1.0f : sigmoid(u);
            g_udiag[r] = sign * (gu + gl / d) * sg;
        }
    }
}

// y_b = W x_b (+ bias); optionally logdet[b] (op)= ld_sign * (*ld_const) for every row (LULinearPermute's constant log-det).
__global__ void __launch_bounds__(64 * RM_NW)
rows_matvec_kernel(const float *__restrict__ x, const float *__restrict__ W, float *__restrict__ y, int64_t B, int D,
                   const float *__restrict__ bias, float *__restrict__ logdet, const float *__restrict__ ld_const,
                   float ld_sign, int acc) {
    // Wl[m][s4][lane][4]: W[32 m + (lane & 31)][4 s4 + r + 32 (lane >> 5)], s4 = 0..7  (2 x 8 x 64 x 4 floats = 16 KB)
    __shared__ __attribute__((aligned(16))) float Wl[2 * 8 * 64 * 4];
    const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5;
    for (int i = tid; i < 2 * 8 * 64 * 4; i += 64 * RM_NW) {
        const int r = i & 3, l = (i >> 2) & 63, s4 = (i >> 8) & 7, m = i >> 11;
        const int row = 32 * m + (l & 31), col = 4 * s4 + r + 32 * (l >> 5);
        Wl[i] = (row < D && col < D) ? W[row * D + col] : 0.0f;
    }
    const int64_t row = ((int64_t)blockIdx.x * RM_NW + (tid >> 6)) * 32 + (lane & 31);
    float xv[32];
    if (D == RM_D && row < B) {
        const float *src = x + row * RM_D + 32 * hh;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(src + 4 * q);
            xv[4 * q] = v[0]; xv[4 * q + 1] = v[1]; xv[4 * q + 2] = v[2]; xv[4 * q + 3] = v[3];
        }
    } else {
#pragma unroll
        for (int k = 0; k < 32; ++k) xv[k] = (row < B && 32 * hh + k < D) ? x[row * D + 32 * hh + k] : 0.0f;
    }
    __syncthreads();
    f32x16 o0 = {0}, o1 = {0};
    if (bias) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int r0 = 8 * (c >> 2) + 4 * hh + (c & 3);
            o0[c] = r0 < D ? bias[r0] : 0.0f;
            o1[c] = 32 + r0 < D ? bias[32 + r0] : 0.0f;
        }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        f32x16 &o = m == 0 ? o0 : o1;
#pragma unroll
        for (int s4 = 0; s4 < 8; ++s4) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(Wl + ((m * 8 + s4) * 64 + lane) * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) o = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], xv[4 * s4 + r], o, 0, 0, 0);
        }
    }
    if (row < B) {
        if (logdet && hh == 0) ld_store(logdet + row, ld_sign * (*ld_const), acc);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const f32x16 &o = m == 0 ? o0 : o1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c0 = 32 * m + 8 * q + 4 * hh;   // C register 4 q + r = output row c0 + r of the product
                if (D == RM_D) {
                    f32x4 v = {o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]};
                    *reinterpret_cast<f32x4 *>(y + row * RM_D + c0) = v;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (c0 + r < D) y[row * D + c0 + r] = o[4 * q + r];
                }
            }
        }
    }
}

// The same product for 64 < D <= 128 (LULinearPermute of wider flows, mixing.py:535-563: round 2 sent them to the LDS-column kernel
// at 0.07 of the HBM peak): four 32-row blocks of W (zero-padded to 128 x 128, 64 KB of LDS in A-operand order), lane-half hh
// contracts over columns [64 hh, 64 hh + 64) of its own row (sixteen 16-byte loads), 64 accumulator registers.
__global__ void __launch_bounds__(64 * RM_NW)
rows_matvec128_kernel(const float *__restrict__ x, const float *__restrict__ W, float *__restrict__ y, int64_t B, int D,
                      const float *__restrict__ bias, float *__restrict__ logdet, const float *__restrict__ ld_const,
                      float ld_sign, int acc) {
    // Wl[m][s4][lane][4]: W[32 m + (lane & 31)][4 s4 + r + 64 (lane >> 5)], m = 0..3, s4 = 0..15
    extern __shared__ __attribute__((aligned(16))) float Wl[];
    const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5;
    for (int i = tid; i < 4 * 16 * 64 * 4; i += 64 * RM_NW) {
        const int r = i & 3, l = (i >> 2) & 63, s4 = (i >> 8) & 15, m = i >> 12;
        const int row = 32 * m + (l & 31), col = 4 * s4 + r + 64 * (l >> 5);
        Wl[i] = (row < D && col < D) ? W[row * D + col] : 0.0f;
    }
    const int64_t row = ((int64_t)blockIdx.x * RM_NW + (tid >> 6)) * 32 + (lane & 31);
    float xv[64];
    if (D == 128 && row < B) {
        const float *src = x + row * 128 + 64 * hh;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(src + 4 * q);
            xv[4 * q] = v[0]; xv[4 * q + 1] = v[1]; xv[4 * q + 2] = v[2]; xv[4 * q + 3] = v[3];
        }
    } else {
#pragma unroll
        for (int k = 0; k < 64; ++k) xv[k] = (row < B && 64 * hh + k < D) ? x[row * D + 64 * hh + k] : 0.0f;
    }
    __syncthreads();
    f32x16 o[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int r0 = 32 * m + 8 * (c >> 2) + 4 * hh + (c & 3);
            o[m][c] = (bias && r0 < D) ? bias[r0] : 0.0f;
        }
    }
#pragma unroll
    for (int s4 = 0; s4 < 16; ++s4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const float a = Wl[((m * 16 + s4) * 64 + lane) * 4 + r];
                o[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, xv[4 * s4 + r], o[m], 0, 0, 0);
            }
        }
    }
    if (row < B) {
        if (logdet && hh == 0) ld_store(logdet + row, ld_sign * (*ld_const), acc);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c0 = 32 * m + 8 * q + 4 * hh;   // C register 4 q + r = output row c0 + r of the product
                if (D == 128) {
                    *reinterpret_cast<f32x4 *>(y + row * 128 + c0) = f32x4{o[m][4 * q], o[m][4 * q + 1], o[m][4 * q + 2], o[m][4 * q + 3]};
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (c0 + r < D) y[row * D + c0 + r] = o[m][4 * q + r];
                }
            }
        }
    }
}

// u_b = W1 x_b, y_b = W2 u_b (+ bias) in ONE launch: LULinearPermute's two batch-side products (forward: u = U x[perm] kept for
// the backward, y = L u + b; backward: gu = L^T gy kept for the factor gradients, gx = P U^T gu).  The first product's C
// registers are the second product's B operand: lane-half hh of C register c of row-block m1 holds unit
// k = 32 m1 + 8 (c >> 2) + 4 hh + (c & 3), so W2 is laid out in LDS in that contraction order (the trick of rows_block / the fused
// layer) and u never leaves the registers between the products.
__global__ void __launch_bounds__(64 * RM_NW)
rows_matvec2_kernel(const float *__restrict__ x, const float *__restrict__ W1, const float *__restrict__ W2,
                    float *__restrict__ u, float *__restrict__ y, int64_t B, int D, const float *__restrict__ bias,
                    float *__restrict__ logdet, const float *__restrict__ ld_const, float ld_sign, int acc) {
    __shared__ __attribute__((aligned(16))) float Wl1[2 * 8 * 64 * 4];   // [m][s4][lane][4]: W1[32 m + (lane & 31)][4 s4 + r + 32 hh]
    __shared__ __attribute__((aligned(16))) float Wl2[2 * 8 * 64 * 4];   // [m][c4][lane][4]: W2[32 m + (lane & 31)][32 (c4 >> 2) + 8 (c4 & 3) + 4 hh + r]
    const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5;
    for (int i = tid; i < 2 * 8 * 64 * 4; i += 64 * RM_NW) {
        const int r = i & 3, l = (i >> 2) & 63, s4 = (i >> 8) & 7, m = i >> 11;
        const int row = 32 * m + (l & 31);
        const int c1 = 4 * s4 + r + 32 * (l >> 5);
        const int c2 = 32 * (s4 >> 2) + 8 * (s4 & 3) + 4 * (l >> 5) + r;
        Wl1[i] = (row < D && c1 < D) ? W1[row * D + c1] : 0.0f;
        Wl2[i] = (row < D && c2 < D) ? W2[row * D + c2] : 0.0f;
    }
    const int64_t row = ((int64_t)blockIdx.x * RM_NW + (tid >> 6)) * 32 + (lane & 31);
    float xv[32];
    if (D == RM_D && row < B) {
        const float *src = x + row * RM_D + 32 * hh;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(src + 4 * q);
            xv[4 * q] = v[0]; xv[4 * q + 1] = v[1]; xv[4 * q + 2] = v[2]; xv[4 * q + 3] = v[3];
        }
    } else {
#pragma unroll
        for (int k = 0; k < 32; ++k) xv[k] = (row < B && 32 * hh + k < D) ? x[row * D + 32 * hh + k] : 0.0f;
    }
    __syncthreads();
    f32x16 t0 = {0}, t1 = {0};
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        f32x16 &o = m == 0 ? t0 : t1;
#pragma unroll
        for (int s4 = 0; s4 < 8; ++s4) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(Wl1 + ((m * 8 + s4) * 64 + lane) * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) o = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], xv[4 * s4 + r], o, 0, 0, 0);
        }
    }
    auto store_rows = [&](float *dst, const f32x16 &o0, const f32x16 &o1) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const f32x16 &o = m == 0 ? o0 : o1;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c0 = 32 * m + 8 * q + 4 * hh;   // C register 4 q + r = output row c0 + r of the product
                if (D == RM_D) {
                    *reinterpret_cast<f32x4 *>(dst + row * RM_D + c0) = f32x4{o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]};
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (c0 + r < D) dst[row * D + c0 + r] = o[4 * q + r];
                }
            }
        }
    };
    if (row < B && u) store_rows(u, t0, t1);
    f32x16 o0 = {0}, o1 = {0};
    if (bias) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int r0 = 8 * (c >> 2) + 4 * hh + (c & 3);
            o0[c] = r0 < D ? bias[r0] : 0.0f;
            o1[c] = 32 + r0 < D ? bias[32 + r0] : 0.0f;
        }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        f32x16 &o = m == 0 ? o0 : o1;
#pragma unroll
        for (int c4 = 0; c4 < 8; ++c4) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(Wl2 + ((m * 8 + c4) * 64 + lane) * 4);
            const f32x16 &t = (c4 >> 2) == 0 ? t0 : t1;
#pragma unroll
            for (int r = 0; r < 4; ++r) o = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], t[4 * (c4 & 3) + r], o, 0, 0, 0);
        }
    }
    if (row < B) {
        if (logdet && hh == 0) ld_store(logdet + row, ld_sign * (*ld_const), acc);
        store_rows(y, o0, o1);
    }
}

}  // namespace nf

using namespace nf;

extern "C" int nf_rows_matvec_affine(const void *x, const void *W, const void *bias, void *y, void *logdet,
                                     const void *ld_const, double ld_sign, int acc, int64_t B, int D, nf_stream_t stream) {
    if (B < 0 || D < 1 || acc < NF_LD_SUB || acc > NF_LD_ADD) return NF_EINVAL;
    if (D > 2 * RM_D) return NF_ENOTSUP;
    if (B == 0) return NF_OK;
    if (!x || !W || !y || (logdet && !ld_const)) return NF_EFAULT;
    const int64_t grid = (B + 32 * RM_NW - 1) / (32 * RM_NW);
    if (grid > 0x7fffffff) return NF_ERANGE;
    if (D > RM_D) {
        const size_t lds = (size_t)4 * 16 * 64 * 4 * sizeof(float);
        static LdsOptIn opted = {};
        if (opt_in_lds(reinterpret_cast<const void *>(&rows_matvec128_kernel), lds, opted) != NF_OK) return NF_ENOTSUP;
        hipLaunchKernelGGL(rows_matvec128_kernel, dim3((unsigned)grid), dim3(64 * RM_NW), lds, (hipStream_t)stream, (const float *)x,
                           (const float *)W, (float *)y, B, D, (const float *)bias, (float *)logdet, (const float *)ld_const,
                           (float)ld_sign, acc);
        NF_CHECK_LAUNCH();
        return NF_OK;
    }
    hipLaunchKernelGGL(rows_matvec_kernel, dim3((unsigned)grid), dim3(64 * RM_NW), 0, (hipStream_t)stream, (const float *)x,
                       (const float *)W, (float *)y, B, D, (const float *)bias, (float *)logdet, (const float *)ld_const,
                       (float)ld_sign, acc);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

extern "C" int nf_rows_matvec2(const void *x, const void *W1, const void *W2, const void *bias, void *u, void *y, void *logdet,
                               const void *ld_const, double ld_sign, int acc, int64_t B, int D, nf_stream_t stream) {
    if (B < 0 || D < 1 || acc < NF_LD_SUB || acc > NF_LD_ADD) return NF_EINVAL;
    if (D > RM_D) return NF_ENOTSUP;
    if (B == 0) return NF_OK;
    if (!x || !W1 || !W2 || !y || (logdet && !ld_const)) return NF_EFAULT;
    const int64_t grid = (B + 32 * RM_NW - 1) / (32 * RM_NW);
    if (grid > 0x7fffffff) return NF_ERANGE;
    hipLaunchKernelGGL(rows_matvec2_kernel, dim3((unsigned)grid), dim3(64 * RM_NW), 0, (hipStream_t)stream, (const float *)x,
                       (const float *)W1, (const float *)W2, (float *)u, (float *)y, B, D, (const float *)bias, (float *)logdet,
                       (const float *)ld_const, (float)ld_sign, acc);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

extern "C" int nf_rows_matvec(const void *x, const void *W, void *y, int64_t B, int D, nf_stream_t stream) {
    return nf_rows_matvec_affine(x, W, nullptr, y, nullptr, nullptr, 0.0, NF_LD_WRITE, B, D, stream);
}

extern "C" int nf_lu_factors(const int64_t *perm, const void *lower_entries, const void *upper_entries,
                             const void *unconstrained_upper_diag, double eps, void *out, int D, nf_stream_t stream) {
    if (D < 1) return NF_EINVAL;
    if (!perm || !unconstrained_upper_diag || !out || (D > 1 && (!lower_entries || !upper_entries))) return NF_EFAULT;
    hipLaunchKernelGGL(lu_factors_kernel, dim3((D * D + 255) / 256), dim3(256), 0, (hipStream_t)stream, perm, (const float *)lower_entries,
                       (const float *)upper_entries, (const float *)unconstrained_upper_diag, (float)eps, (float *)out, D);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

extern "C" int nf_lu_factors_multi(const void *table, int n_layers, double eps, int D, nf_stream_t stream) {
    if (D < 2 || n_layers < 0 || n_layers > 65535) return NF_EINVAL;
    if (n_layers == 0) return NF_OK;
    if (!table) return NF_EFAULT;
    hipLaunchKernelGGL(lu_factors_multi_kernel, dim3((D * D + 255) / 256, n_layers), dim3(256), 0, (hipStream_t)stream,
                       (const void *const *)table, (float)eps, D);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

extern "C" int nf_lu_param_grads(const void *gL, const void *gU, const int64_t *perm, const void *gld, int64_t B,
                                 const void *unconstrained_upper_diag, double eps, double sign, void *g_lower, void *g_upper,
                                 void *g_udiag, int D, nf_stream_t stream) {
    if (D < 1) return NF_EINVAL;
    if (!gL || !gU || !unconstrained_upper_diag || !g_udiag || (D > 1 && (!g_lower || !g_upper))) return NF_EFAULT;
    const int grid = (D * D + 1023) / 1024;
    hipLaunchKernelGGL(lu_param_grads_kernel, dim3(grid < 64 ? grid : 64), dim3(1024), 0, (hipStream_t)stream, (const float *)gL,
                       (const float *)gU, perm, (const float *)gld, B, (const float *)unconstrained_upper_diag, (float)eps, (float)sign,
                       (float *)g_lower, (float *)g_upper, (float *)g_udiag, D);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

extern "C" int nf_lu_compose(const int64_t *perm, const void *lower_entries, const void *upper_entries,
                             const void *unconstrained_upper_diag, const void *bias, double eps, void *out, int D,
                             nf_stream_t stream) {
    if (D < 1) return NF_EINVAL;
    if (D > RM_D) return NF_ENOTSUP;
    if (!perm || !unconstrained_upper_diag || !bias || !out || (D > 1 && (!lower_entries || !upper_entries))) return NF_EFAULT;
    const size_t lds = (size_t)4 * D * D * sizeof(double);
    static LdsOptIn opted = {};
    if (opt_in_lds(reinterpret_cast<const void *>(&lu_compose_kernel), lds, opted) != NF_OK) return NF_ENOTSUP;
    hipLaunchKernelGGL(lu_compose_kernel, dim3(1), dim3(256), lds, (hipStream_t)stream, perm, (const float *)lower_entries,
                       (const float *)upper_entries, (const float *)unconstrained_upper_diag, (const float *)bias, (float)eps,
                       (float *)out, D);
    NF_CHECK_LAUNCH();
    return NF_OK;
}
