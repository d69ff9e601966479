// affine_bwd.hip -- backward (vector-Jacobian products) of the affine family for the training path: MaskedAffineFlow,
// AffineCoupling (+ channel Split / Merge), AffineConstFlow / ActNorm, the per-pixel C x C product of Invertible1x1Conv.
// gfx950 only.  The reference differentiates these layers with PyTorch autograd (core.py:87-102 `loss.backward()` over
// affine/coupling.py:38-54, :117-171, :209-229, mixing.py:106-133); here every gradient is a closed form evaluated in one
// HBM-bound pass: one read of every input / cotangent element, one write of every gradient element.  Reductions over the
// batch (ActNorm's s, t; the 1x1 matrix) are done in a fixed order (deterministic run to run).
#include "common.hpp"

namespace nf {

// ---- MaskedAffineFlow (coupling.py:209-229) ------------------------------------------------------------------------------
//   direction 0: y = b z + (1 - b)(z e^s + t), ld = sum (1 - b) s
//   direction 1: y = b z + (1 - b)(z - t) e^-s, ld = -sum (1 - b) s
template <typename T>
__global__ void __launch_bounds__(256)
masked_affine_bwd_kernel(const T *__restrict__ z, const T *__restrict__ b, const T *__restrict__ s, const T *__restrict__ t,
                         const T *__restrict__ gy, const T *__restrict__ gld, T *__restrict__ gz, T *__restrict__ gs,
                         T *__restrict__ gt, int64_t B, int64_t inner, int direction) {
    const int64_t N = B * inner;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < N; o += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = o / inner, i = o - r * inner;
        const T bi = b[i], nb = T(1) - bi, zi = z[o], g = gy[o], gl = gld ? gld[r] : T(0);
        T si = s ? s[o] : T(0), ti = t ? t[o] : T(0);
        if (!M<T>::finite(si)) si = M<T>::nan();
        if (!M<T>::finite(ti)) ti = M<T>::nan();
        if (direction == 0) {
            const T e = M<T>::exp(si);
            gz[o] = g * (bi + nb * e);
            if (gs) gs[o] = nb * (g * zi * e + gl);
            if (gt) gt[o] = nb * g;
        } else {
            const T e = M<T>::exp(-si);
            gz[o] = g * (bi + nb * e);
            if (gs) gs[o] = -nb * (g * (zi - ti) * e + gl);
            if (gt) gt[o] = -nb * g * e;
        }
    }
}

// ---- AffineCoupling with the channel split / merge folded in (coupling.py:117-171, reshape.py:30-85) -------------------
// z (B, C, HW): z1 = c1 identity channels (copied: gz = gy there), z2 transformed with the interleaved parameter planes
// param (B, P, HW), P = 2 (C - c1) (shift at 2 c, scale at 2 c + 1) or C - c1 (scale_map NONE: shift only).
template <typename T>
__global__ void __launch_bounds__(256)
affine_coupling_bwd_kernel(const T *__restrict__ z, const T *__restrict__ param, const T *__restrict__ gy,
                           const T *__restrict__ gld, T *__restrict__ gz, T *__restrict__ gparam, int64_t B, int C, int c1,
                           int flip, int64_t HW, int scale_map, int direction) {
    const int c2 = C - c1;
    const int z1_off = flip ? c2 : 0, z2_off = flip ? 0 : c1;
    const int P = scale_map == NF_SCALE_NONE ? c2 : 2 * c2;
    const int64_t n1 = (int64_t)c1 * HW, n2 = (int64_t)c2 * HW;
    for (int64_t r = blockIdx.x; r < B; r += gridDim.x) {
        const T *zr = z + r * (int64_t)C * HW, *gyr = gy + r * (int64_t)C * HW, *pr = param + r * (int64_t)P * HW;
        T *gzr = gz + r * (int64_t)C * HW, *gpr = gparam + r * (int64_t)P * HW;
        const T gl = gld ? gld[r] : T(0);
        for (int64_t i = threadIdx.x; i < n1; i += blockDim.x) gzr[(int64_t)z1_off * HW + i] = gyr[(int64_t)z1_off * HW + i];
        for (int64_t i = threadIdx.x; i < n2; i += blockDim.x) {
            const int64_t c = i / HW, p = i - c * HW;
            const T v = zr[(int64_t)z2_off * HW + i], g = gyr[(int64_t)z2_off * HW + i];
            T gv;
            if (scale_map == NF_SCALE_NONE) {
                gv = g;
                gpr[i] = direction == 0 ? g : -g;
            } else {
                const T sh = pr[(2 * c) * HW + p], sc = pr[(2 * c + 1) * HW + p];
                T gsh, gsc;
                if (scale_map == NF_SCALE_EXP) {
                    if (direction == 0) {          // y = v e^sc + sh, ld = +sum sc
                        const T e = M<T>::exp(sc);
                        gv = g * e; gsh = g; gsc = g * v * e + gl;
                    } else {                       // y = (v - sh) e^-sc, ld = -sum sc
                        const T e = M<T>::exp(-sc);
                        gv = g * e; gsh = -g * e; gsc = -(g * (v - sh) * e + gl);
                    }
                } else {
                    const T sg = sigmoid(sc + T(2)), om = T(1) - sg;   // d sg / d sc = sg (1 - sg), d log sg / d sc = 1 - sg
                    const bool mult = (scale_map == NF_SCALE_SIGMOID_INV) == (direction == 0);   // y uses * sg (else / sg)
                    if (direction == 0) {
                        if (mult) { gv = g * sg; gsh = g; gsc = g * v * sg * om + gl * om; }            // ld = +sum log sg
                        else { gv = g / sg; gsh = g; gsc = -(g * v * om / sg + gl * om); }              // ld = -sum log sg
                    } else {
                        if (mult) { gv = g * sg; gsh = -g * sg; gsc = g * (v - sh) * sg * om + gl * om; }   // ld = +sum log sg
                        else { gv = g / sg; gsh = -g / sg; gsc = -(g * (v - sh) * om / sg + gl * om); }     // ld = -sum log sg
                    }
                }
                gpr[(2 * c) * HW + p] = gsh;
                gpr[(2 * c + 1) * HW + p] = gsc;
            }
            gzr[(int64_t)z2_off * HW + i] = gv;
        }
    }
}

// ---- AffineConstFlow / ActNorm with per-channel s, t (coupling.py:38-54) ---------------------------------------------
//   direction 0: y = z e^s + t, ld (every sample) = +HW sum s;   direction 1: y = (z - t) e^-s, ld = -HW sum s.
// Grid (C, nsplit): workgroup (c, j) handles images j, j + nsplit, ... of channel c: gz in the same pass as the partial
// per-channel sums (fp64); the second kernel adds the nsplit partials in a fixed order (deterministic).
template <typename T>
__global__ void __launch_bounds__(256)
actnorm_bwd_kernel(const T *__restrict__ z, const T *__restrict__ s, const T *__restrict__ t, const T *__restrict__ gy,
                   T *__restrict__ gz, double *__restrict__ partial, int64_t B, int C, int64_t HW, int direction) {
    __shared__ double sred[16];
    const int c = blockIdx.x, j = blockIdx.y, nsplit = gridDim.y;
    const T sc = s[c], tc = t[c];
    const T e = M<T>::exp(direction == 0 ? sc : -sc);
    double as = 0.0, at = 0.0;
    const int64_t nimg = (B - j + nsplit - 1) / nsplit;       // images j, j + nsplit, ...
    if constexpr (sizeof(T) == 4) {
        if ((HW & 3) == 0) {
            // float32 planes of a multiple of 4 pixels (round 4): a thread owns FOUR consecutive pixels per step (16-byte loads of z
            // and gy, a 16-byte store of gz), (image, pixel quad) follow incrementally
            typedef float v4 __attribute__((ext_vector_type(4)));
            const int64_t Q = HW >> 2;
            int64_t k = threadIdx.x / Q, q = threadIdx.x - k * Q;
            const int64_t dk = blockDim.x / Q, dq = blockDim.x - dk * Q;
            for (int64_t e_ = threadIdx.x; e_ < nimg * Q; e_ += blockDim.x) {
                const int64_t o = ((j + k * nsplit) * C + c) * HW + 4 * q;
                const v4 g = *reinterpret_cast<const v4 *>(reinterpret_cast<const float *>(gy) + o);
                const v4 v = *reinterpret_cast<const v4 *>(reinterpret_cast<const float *>(z) + o);
                v4 out;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    out[i] = g[i] * (float)e;
                    if (direction == 0) { as += (double)(g[i] * v[i] * (float)e); at += (double)g[i]; }
                    else { as -= (double)(g[i] * (v[i] - (float)tc) * (float)e); at -= (double)(g[i] * (float)e); }
                }
                *reinterpret_cast<v4 *>(reinterpret_cast<float *>(gz) + o) = out;
                k += dk;
                q += dq;
                if (q >= Q) {
                    q -= Q;
                    ++k;
                }
            }
            as = block_sum(as, sred);
            at = block_sum(at, sred);
            if (threadIdx.x == 0) {
                partial[((int64_t)j * C + c) * 2] = as;
                partial[((int64_t)j * C + c) * 2 + 1] = at;
            }
            return;
        }
    }
    // (image, pixel) of a thread's next element follow incrementally: one 64-bit division per thread instead of one per element
    int64_t k = threadIdx.x / HW, p = threadIdx.x - k * HW;
    const int64_t dk = blockDim.x / HW, dp = blockDim.x - dk * HW;
    for (int64_t e_ = threadIdx.x; e_ < nimg * HW; e_ += blockDim.x) {
        const int64_t o = ((j + k * nsplit) * C + c) * HW + p;
        const T g = gy[o], v = z[o];
        gz[o] = g * e;
        if (direction == 0) { as += (double)(g * v * e); at += (double)g; }
        else { as -= (double)(g * (v - tc) * e); at -= (double)(g * e); }
        k += dk;
        p += dp;
        if (p >= HW) {
            p -= HW;
            ++k;
        }
    }
    as = block_sum(as, sred);
    at = block_sum(at, sred);
    if (threadIdx.x == 0) {
        partial[((int64_t)j * C + c) * 2] = as;
        partial[((int64_t)j * C + c) * 2 + 1] = at;
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
actnorm_bwd_reduce_kernel(const double *__restrict__ partial, const T *__restrict__ gld, T *__restrict__ gs, T *__restrict__ gt,
                          int nsplit, int64_t B, int C, int64_t HW, int direction) {
    __shared__ double sred[16];
    double gl = 0.0;   // d ld / d s_c = +-HW for every sample
    if (gld)
        for (int64_t r = threadIdx.x; r < B; r += blockDim.x) gl += (double)gld[r];
    gl = block_sum(gl, sred);
    __shared__ double glb;
    if (threadIdx.x == 0) glb = gl;
    __syncthreads();
    // thread = (channel c0, slice q of the splits): Q slices per channel summed through LDS in slice order (deterministic).  Round 5:
    // one thread per channel walked all ~2048 / C splits alone -- 26 us per call for 12-48 channels, 96 calls per Glow step.
    __shared__ double pas[256], pat[256];
    const int Cq = C < 256 ? C : 256, Q = 256 / Cq;
    const int q = threadIdx.x / Cq, c0 = threadIdx.x - q * Cq;
    for (int cb = 0; cb < C; cb += Cq) {
        const int c = cb + c0;
        double as = 0.0, at = 0.0;
        if (q < Q && c < C)
            for (int j = q; j < nsplit; j += Q) { as += partial[((int64_t)j * C + c) * 2]; at += partial[((int64_t)j * C + c) * 2 + 1]; }
        pas[threadIdx.x] = as;
        pat[threadIdx.x] = at;
        __syncthreads();
        if (q == 0 && c < C) {
            for (int k = 1; k < Q; ++k) { as += pas[k * Cq + c0]; at += pat[k * Cq + c0]; }
            gs[c] = (T)(as + (direction == 0 ? 1.0 : -1.0) * (double)HW * glb);
            gt[c] = (T)at;
        }
        __syncthreads();
    }
}

// ---- per-pixel C x C product y = W z (mixing.py:106-133): gW = sum over pixels of gy z^T -------------------------------
// Workgroup w handles images [w ipw, (w + 1) ipw): the C x C partial sums in registers (thread = entry (i, j), strided),
// pixels staged through LDS; a second kernel adds the partials in a fixed order.  C <= 64.
template <typename T>
__global__ void __launch_bounds__(256)
inv1x1_wgrad_partial_kernel(const T *__restrict__ z, const T *__restrict__ gy, T *__restrict__ partial, int64_t B, int C,
                            int64_t HW, int ipw) {
    extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
    T *zs = reinterpret_cast<T *>(raw);          // [C][64]
    T *gs = zs + C * 64;                         // [C][64]
    const int tid = threadIdx.x, CC = C * C;
    T acc[16];                                   // entries tid, tid + 256, ... (C <= 64 -> <= 16 per thread)
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = T(0);
    const int64_t b0 = (int64_t)blockIdx.x * ipw, b1 = b0 + ipw < B ? b0 + ipw : B;
    for (int64_t bb = b0; bb < b1; ++bb) {
        for (int64_t p0 = 0; p0 < HW; p0 += 64) {
            const int np = (int)(HW - p0 < 64 ? HW - p0 : 64);
            __syncthreads();
            for (int e = tid; e < C * 64; e += 256) {
                const int c = e >> 6, p = e & 63;
                const int64_t o = (bb * C + c) * HW + p0 + p;
                zs[e] = p < np ? z[o] : T(0);
                gs[e] = p < np ? gy[o] : T(0);
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int ent = tid + 256 * k;
                if (ent < CC) {
                    const int i = ent / C, j = ent - i * C;
                    T a = T(0);
                    for (int p = 0; p < 64; ++p) a += gs[i * 64 + p] * zs[j * 64 + p];
                    acc[k] += a;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int ent = tid + 256 * k;
        if (ent < CC) partial[(int64_t)blockIdx.x * CC + ent] = acc[k];
    }
}

// The same partial sums on fp32 MFMA (round 4; float32, HW a multiple of 16): gW = Gy Z^T is a (C x C) x pixels product.  The
// scalar kernel above reads two LDS words per multiply-add, leaves most threads idle for C = 12 (144 entries) and meets two
// barriers per 64 pixels: 327 / 721 us at (8192, 12, 16, 16) / (8192, 48, 4, 4) = 0.08 / 0.01 of the HBM peak.  Here every WAVE
// streams its own 16-pixel chunks straight into MFMA operands -- lane (m = lane & 15, kq = lane >> 4) loads the four consecutive
// pixels 4 kq .. 4 kq + 3 of row 16 ib + m with ONE 16-byte load and uses element s as its k-entry of step s (the contraction
// order over the chunk's pixels is free as long as both operands use the same one) -- no LDS, no barrier, four chunks of loads
// in flight per wave; NB^2 accumulators of 4 registers; one partial per wave, summed in a fixed order by the reduce kernel.
template <int NB>
__global__ void __launch_bounds__(256)
inv1x1_wgrad_mfma_kernel(const float *__restrict__ z, const float *__restrict__ gy, float *__restrict__ partial, int64_t B, int C,
                         int64_t HW) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, m = lane & 15, kq = lane >> 4;
    const int64_t gw = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int64_t cpi = HW >> 4, nchunks = B * cpi;           // 16-pixel chunks per image
    f4 acc[NB * NB];
#pragma unroll
    for (int q = 0; q < NB * NB; ++q) acc[q] = f4{0.0f, 0.0f, 0.0f, 0.0f};
    auto load = [&](const float *__restrict__ src, int64_t ch, f4 (&v)[NB]) {
        const int64_t bb = ch / cpi, p = (ch - bb * cpi) * 16 + 4 * kq;
#pragma unroll
        for (int ib = 0; ib < NB; ++ib) {
            const int c = 16 * ib + m;
            v[ib] = c < C ? *reinterpret_cast<const f4 *>(src + (bb * C + c) * HW + p) : f4{0.0f, 0.0f, 0.0f, 0.0f};
        }
    };
    for (int64_t ch0 = gw; ch0 < nchunks; ch0 += 4 * nw) {
        f4 a[4][NB], b[4][NB];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t ch = ch0 + u * nw;
            if (ch < nchunks) {
                load(gy, ch, a[u]);
                load(z, ch, b[u]);
            } else {
#pragma unroll
                for (int ib = 0; ib < NB; ++ib) a[u][ib] = b[u][ib] = f4{0.0f, 0.0f, 0.0f, 0.0f};
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int ib = 0; ib < NB; ++ib)
#pragma unroll
                for (int jb = 0; jb < NB; ++jb)
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4)
                        acc[ib * NB + jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][ib][s4], b[u][jb][s4], acc[ib * NB + jb], 0, 0, 0);
    }
    float *out = partial + gw * C * C;      // C register r of a block: row 4 (lane >> 4) + r, column lane & 15
#pragma unroll
    for (int ib = 0; ib < NB; ++ib)
#pragma unroll
        for (int jb = 0; jb < NB; ++jb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = 16 * ib + 4 * kq + r, jj = 16 * jb + m;
                if (i < C && jj < C) out[i * C + jj] = acc[ib * NB + jb][r];
            }
}

// 16 entries per workgroup; thread (entry e = tid & 15, slice q = tid >> 4) adds the partials q, q + 16, ... in order, the 16 slices
// are then added in order by the entry's first thread: a fixed summation order (deterministic) without one thread walking all the
// partials alone (256 and more serial loads per entry in round 3).
template <typename T>
__global__ void __launch_bounds__(256)
inv1x1_wgrad_reduce_kernel(const T *__restrict__ partial, const T *__restrict__ gld, T *__restrict__ gW, T *__restrict__ gl,
                           int nparts, int CC, int64_t B, int64_t HW) {
    __shared__ double sred[16];
    __shared__ double slices[16][17];
    const int e = threadIdx.x & 15, q = threadIdx.x >> 4;
    const int ent = blockIdx.x * 16 + e;
    double a = 0.0;
    if (ent < CC) {      // four loads in flight (round 6, late: one accumulator = one load at a time, 10 us per call at 1024 partials)
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int w = q;
        for (; w + 48 < nparts; w += 64) {
            a0 += (double)partial[(int64_t)w * CC + ent];
            a1 += (double)partial[(int64_t)(w + 16) * CC + ent];
            a2 += (double)partial[(int64_t)(w + 32) * CC + ent];
            a3 += (double)partial[(int64_t)(w + 48) * CC + ent];
        }
        for (; w < nparts; w += 16) a0 += (double)partial[(int64_t)w * CC + ent];
        a = (a0 + a1) + (a2 + a3);
    }
    slices[q][e] = a;
    __syncthreads();
    if (q == 0 && ent < CC) {
        double t = 0.0;
        for (int k = 0; k < 16; ++k) t += slices[k][e];
        gW[ent] = (T)t;
    }
    if (blockIdx.x == 0 && gl) {   // ld (every sample) = HW ldu  ->  d / d ldu = HW sum_b gld
        double g = 0.0;
        if (gld)
            for (int64_t r = threadIdx.x; r < B; r += blockDim.x) g += (double)gld[r];
        g = block_sum(g, sred);
        if (threadIdx.x == 0) *gl = (T)((double)HW * g);
    }
}


// ---- MaskedAffineAutoregressive element-wise transform (affine/autoregressive.py:98-128; forward: affine.hip maf_affine_kernel) -------
//   scale = sigmoid(u + 2) + 1e-3 (u = params[.., 0], shift = params[.., 1])
//   direction 0: y = scale x + shift, ld = +sum log scale;   direction 1: y = (x - shift) / scale, ld = -sum log scale
// one element per thread-iteration: g_x, g_params (B, D, 2) from the cotangents gy (B, D) and gld (B), either may be NULL (= 0).
template <typename T>
__global__ void __launch_bounds__(256)
maf_affine_bwd_kernel(const T *__restrict__ x, const T *__restrict__ params, const T *__restrict__ gy, const T *__restrict__ gld,
                      T *__restrict__ gx, T *__restrict__ gparams, int64_t B, int D, int direction) {
    const int64_t N = B * D;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < N; o += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = o / D;
        const T u = params[2 * o], sh = params[2 * o + 1];
        const T sg = sigmoid(u + T(2));
        const T scale = sg + T(1e-3);
        const T g = gy ? gy[o] : T(0), gl = gld ? gld[r] : T(0);
        const T xv = x[o];
        T gscale, gshift;
        if (direction == 0) {
            gx[o] = g * scale;
            gshift = g;
            gscale = g * xv + gl / scale;
        } else {
            const T inv = T(1) / scale;
            gx[o] = g * inv;
            gshift = -g * inv;
            gscale = -(g * (xv - sh) * inv + gl) * inv;
        }
        gparams[2 * o] = gscale * sg * (T(1) - sg);
        gparams[2 * o + 1] = gshift;
    }
}


// ---- Invertible1x1Conv's LU parametrisation (mixing.py:88-104), density direction: W = P (tril(L, -1) + I) (triu(U, 1) + diag(sign_S
// exp(log_S))), log|det| per pixel = sum log_S.  VJP of (W, log|det|) -> (L, U, log_S) for the cotangents gW (C x C), gl (0-dim): what
// torch autograd does with ~15 tiny tril / triu / diag / exp / matmul launches per layer and step.  C <= 64, one workgroup per row:
//   gLm = P^T gW Um^T -> strictly lower part;  gUm = (P Lm)^T gW -> strictly upper part;  g_log_S = diag(gUm) sign_S exp(log_S) + gl
template <typename T>
__device__ __forceinline__ void inv1x1_lu_grads_body(const T *__restrict__ P, const T *__restrict__ L, const T *__restrict__ U,
                                                     const T *__restrict__ sign_S, const T *__restrict__ log_S, const T *__restrict__ gW,
                                                     const T *__restrict__ gl, T *__restrict__ gL, T *__restrict__ gU,
                                                     T *__restrict__ glogS, int C) {
    // one workgroup per output row i, thread j per column: row i of A = P^T gW and column i of P Lm through LDS.  Round 5: P, gW, U and
    // column i of Lm are staged in LDS first (coalesced) -- the loops below were ~6 C dependent global loads per thread, 18 us per call
    extern __shared__ __attribute__((aligned(16))) unsigned char lg_raw[];
    T *Ps = reinterpret_cast<T *>(lg_raw), *Gs = Ps + C * C, *Us = Gs + C * C, *lmc = Us + C * C;
    __shared__ T a[64], plc[64];
    const int i = blockIdx.x, j = threadIdx.x;
    for (int e = j; e < C * C; e += 64) { Ps[e] = P[e]; Gs[e] = gW[e]; Us[e] = U[e]; }
    for (int k = j; k < C; k += 64) lmc[k] = k > i ? L[k * C + i] : (k == i ? T(1) : T(0));     // Lm[k][i]
    __syncthreads();
    if (j < C) {
        T av = T(0), pl = T(0);
        for (int k = 0; k < C; ++k) {
            av += Ps[k * C + i] * Gs[k * C + j];                                      // A[i][j]
            pl += Ps[j * C + k] * lmc[k];                                             // (P Lm)[j][i]
        }
        a[j] = av;
        plc[j] = pl;
    }
    __syncthreads();
    if (j < C) {
        T glm = T(0), gum = T(0);
        for (int k = 0; k < C; ++k) {
            const T um = k > j ? Us[j * C + k] : (k == j ? sign_S[j] * M<T>::exp(log_S[j]) : T(0));     // Um[j][k]
            glm += a[k] * um;                            // (A Um^T)[i][j]
            gum += plc[k] * Gs[k * C + j];               // ((P Lm)^T gW)[i][j]
        }
        gL[i * C + j] = i > j ? glm : T(0);
        gU[i * C + j] = j > i ? gum : T(0);
        if (i == j) glogS[i] = gum * sign_S[i] * M<T>::exp(log_S[i]) + (gl ? *gl : T(0));
    }
}

template <typename T>
__global__ void __launch_bounds__(64)
inv1x1_lu_grads_kernel(const T *__restrict__ P, const T *__restrict__ L, const T *__restrict__ U, const T *__restrict__ sign_S,
                       const T *__restrict__ log_S, const T *__restrict__ gW, const T *__restrict__ gl, T *__restrict__ gL,
                       T *__restrict__ gU, T *__restrict__ glogS, int C) {
    inv1x1_lu_grads_body<T>(P, L, U, sign_S, log_S, gW, gl, gL, gU, glogS, C);
}

// Round 6 (late): up to 32 layers of one size per launch (grid = rows x layers): the LU factors' gradients of a whole Glow level once
// every block's gW exists, instead of one 14 us launch per block inside the backward pass.
constexpr int I1G_MULTI = 32;
struct Inv1x1LuGradsMulti {
    const void *P[I1G_MULTI], *L[I1G_MULTI], *U[I1G_MULTI], *sign_S[I1G_MULTI], *log_S[I1G_MULTI], *gW[I1G_MULTI], *gl[I1G_MULTI];
    void *gL[I1G_MULTI], *gU[I1G_MULTI], *glogS[I1G_MULTI];
};

__global__ void __launch_bounds__(64)
inv1x1_lu_grads_multi_kernel(Inv1x1LuGradsMulti m, int C) {
    const int b = blockIdx.y;
    inv1x1_lu_grads_body<float>((const float *)m.P[b], (const float *)m.L[b], (const float *)m.U[b], (const float *)m.sign_S[b],
                                (const float *)m.log_S[b], (const float *)m.gW[b], (const float *)m.gl[b], (float *)m.gL[b],
                                (float *)m.gU[b], (float *)m.glogS[b], C);
}


// ---- one sweep of the implicit backward of MaskedAffineAutoregressive.inverse (autograd.MafInverseFn) ----------------------------------
//   v <- (g_x - gxm) / scale   (first sweep: gxm = NULL -> v = g_x / scale);   *changed |= any element of v differs bitwise from before;
//   g_p (B, D, 2) = the parameter cotangent of z = scale x + shift, sum log scale for the cotangents (v, g_ld): the next chain's input.
template <typename T>
__global__ void __launch_bounds__(256)
maf_implicit_sweep_kernel(const T *__restrict__ x, const T *__restrict__ params, const T *__restrict__ gx, const T *__restrict__ gld,
                          const T *__restrict__ gxm, T *__restrict__ v, T *__restrict__ gp, int *__restrict__ changed, int64_t B, int D) {
    const int64_t N = B * D;
    int diff = 0;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < N; o += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = o / D;
        const T sg = sigmoid(params[2 * o] + T(2));
        const T scale = sg + T(1e-3);
        const T vn = ((gx ? gx[o] : T(0)) - (gxm ? gxm[o] : T(0))) / scale;
        const T vo = v[o];
        diff |= !(vn == vo) && !(vn != vn && vo != vo);        // (NaN == NaN counts as unchanged: a diverged solve must still stop)
        v[o] = vn;
        gp[2 * o] = (vn * x[o] + (gld ? gld[r] : T(0)) / scale) * sg * (T(1) - sg);
        gp[2 * o + 1] = vn;
    }
    if (__any(diff) && (threadIdx.x & 63) == 0) atomicOr(changed, 1);
}

}  // namespace nf

using namespace nf;

#define NF_DISPATCH(dtype, CALL_F32, CALL_F64) \
    do {                                       \
        if ((dtype) == NF_F32) { CALL_F32; }   \
        else if ((dtype) == NF_F64) { CALL_F64; } \
        else return NF_ENOTSUP;                \
    } while (0)

extern "C" int nf_masked_affine_bwd(const void *z, const void *b, const void *s, const void *t, const void *gy,
                                    const void *gld, void *gz, void *gs, void *gt, int64_t B, int64_t inner, int direction,
                                    int dtype, nf_stream_t stream) {
    if (B < 0 || inner < 1 || (direction != 0 && direction != 1)) return NF_EINVAL;
    if (B == 0) return NF_OK;
    if (!z || !b || !gy || !gz) return NF_EFAULT;
    if ((gs && !s) || (gt && !t)) return NF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int grid = grid_for(B * inner, 256 * 4);
    NF_DISPATCH(dtype,
                hipLaunchKernelGGL(masked_affine_bwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float *)z,
                                   (const float *)b, (const float *)s, (const float *)t, (const float *)gy,
                                   (const float *)gld, (float *)gz, (float *)gs, (float *)gt, B, inner, direction),
                hipLaunchKernelGGL(masked_affine_bwd_kernel<double>, dim3(grid), dim3(256), 0, st, (const double *)z,
                                   (const double *)b, (const double *)s, (const double *)t, (const double *)gy,
                                   (const double *)gld, (double *)gz, (double *)gs, (double *)gt, B, inner, direction));
    NF_CHECK_LAUNCH();
    return NF_OK;
}

extern "C" int nf_affine_coupling_bwd(const void *z, const void *param, const void *gy, const void *gld, void *gz,
                                      void *gparam, int64_t B, int C, int c1, int flip, int64_t HW, int scale_map,
                                      int direction, int dtype, nf_stream_t stream) {
    if (B < 0 || C < 1 || c1 < 0 || c1 >= C || HW < 1 || (direction != 0 && direction != 1)) return NF_EINVAL;
    if (scale_map < NF_SCALE_EXP || scale_map > NF_SCALE_NONE) return NF_EINVAL;
    if (B == 0) return NF_OK;
    if (!z || !param || !gy || !gz || !gparam) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    const int grid = grid_for(B, 1, 256 * 16);
    NF_DISPATCH(dtype,
                hipLaunchKernelGGL(affine_coupling_bwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float *)z,
                                   (const float *)param, (const float *)gy, (const float *)gld, (float *)gz, (float *)gparam,
                                   B, C, c1, flip, HW, scale_map, direction),
                hipLaunchKernelGGL(affine_coupling_bwd_kernel<double>, dim3(grid), dim3(256), 0, st, (const double *)z,
                                   (const double *)param, (const double *)gy, (const double *)gld, (double *)gz,
                                   (double *)gparam, B, C, c1, flip, HW, scale_map, direction));
    NF_CHECK_LAUNCH();
    return NF_OK;
}

static inline int actnorm_bwd_nsplit(int64_t B, int C) {
    int64_t n = (2048 + C - 1) / C;     // ~2048 workgroups in all
    if (n > B) n = B;
    return (int)(n < 1 ? 1 : n);
}

extern "C" int64_t nf_actnorm_bwd_scratch_doubles(int64_t B, int C) {
    if (B < 0 || C < 1) return NF_EINVAL;
    return (int64_t)actnorm_bwd_nsplit(B, C) * C * 2;
}

extern "C" int nf_actnorm_bwd(const void *z, const void *s, const void *t, const void *gy, const void *gld, void *gz,
                              void *gs, void *gt, void *scratch, int64_t B, int C, int64_t HW, int direction, int dtype,
                              nf_stream_t stream) {
    if (B < 1 || C < 1 || HW < 1 || (direction != 0 && direction != 1)) return NF_EINVAL;
    if (!s || !t || !gs || !gt || !scratch || !z || !gy || !gz) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    const int nsplit = actnorm_bwd_nsplit(B, C);
    NF_DISPATCH(dtype,
                hipLaunchKernelGGL(actnorm_bwd_kernel<float>, dim3(C, nsplit), dim3(256), 0, st, (const float *)z,
                                   (const float *)s, (const float *)t, (const float *)gy, (float *)gz, (double *)scratch, B, C,
                                   HW, direction);
                hipLaunchKernelGGL(actnorm_bwd_reduce_kernel<float>, dim3(1), dim3(256), 0, st, (const double *)scratch,
                                   (const float *)gld, (float *)gs, (float *)gt, nsplit, B, C, HW, direction),
                hipLaunchKernelGGL(actnorm_bwd_kernel<double>, dim3(C, nsplit), dim3(256), 0, st, (const double *)z,
                                   (const double *)s, (const double *)t, (const double *)gy, (double *)gz, (double *)scratch, B,
                                   C, HW, direction);
                hipLaunchKernelGGL(actnorm_bwd_reduce_kernel<double>, dim3(1), dim3(256), 0, st, (const double *)scratch,
                                   (const double *)gld, (double *)gs, (double *)gt, nsplit, B, C, HW, direction));
    NF_CHECK_LAUNCH();
    return NF_OK;
}

extern "C" int64_t nf_inv1x1_wgrad_scratch_elems(int64_t B, int C) {
    if (B < 0 || C < 1 || C > 64) return NF_EINVAL;
    const int64_t parts = B < 256 ? (B > 0 ? B : 1) : 256;
    return 4 * parts * (int64_t)C * C;        // the MFMA kernel leaves one partial per wave: 4 per workgroup
}

extern "C" int nf_inv1x1_wgrad(const void *z, const void *gy, const void *gld, void *gW, void *gldu, void *scratch, int64_t B,
                               int C, int64_t HW, int dtype, nf_stream_t stream) {
    if (B < 1 || C < 1 || HW < 1) return NF_EINVAL;
    if (C > 64) return NF_ENOTSUP;
    if (!z || !gy || !gW || !scratch) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    const int parts = (int)(B < 256 ? B : 256);
    const int ipw = (int)((B + parts - 1) / parts);
    const int nparts = (int)((B + ipw - 1) / ipw);
    const size_t esz = dtype == NF_F64 ? 8 : 4;
    const size_t lds = (size_t)2 * C * 64 * esz;
    if (dtype == NF_F32 && (HW & 15) == 0 && (((uintptr_t)z | (uintptr_t)gy) & 15) == 0) {
        const int NB = (C + 15) / 16;
        const int64_t nchunks = B * (HW >> 4);
        int64_t wg = (nchunks + 15) / 16;                     // >= 4 chunks per wave
        if (wg > parts) wg = parts;                           // <= 256 workgroups = 1024 wave partials (nf_inv1x1_wgrad_scratch_elems)
        if (wg < 1) wg = 1;
#define NF_W1(NBV)                                                                                                            \
    hipLaunchKernelGGL((inv1x1_wgrad_mfma_kernel<NBV>), dim3((unsigned)wg), dim3(256), 0, st, (const float *)z, (const float *)gy, \
                       (float *)scratch, B, C, HW)
        if (NB == 1) NF_W1(1); else if (NB == 2) NF_W1(2); else if (NB == 3) NF_W1(3); else NF_W1(4);
#undef NF_W1
        hipLaunchKernelGGL(inv1x1_wgrad_reduce_kernel<float>, dim3((C * C + 15) / 16), dim3(256), 0, st, (const float *)scratch,
                           (const float *)gld, (float *)gW, (float *)gldu, (int)(4 * wg), C * C, B, HW);
        NF_CHECK_LAUNCH();
        return NF_OK;
    }
    NF_DISPATCH(dtype,
                hipLaunchKernelGGL(inv1x1_wgrad_partial_kernel<float>, dim3(nparts), dim3(256), lds, st, (const float *)z,
                                   (const float *)gy, (float *)scratch, B, C, HW, ipw);
                hipLaunchKernelGGL(inv1x1_wgrad_reduce_kernel<float>, dim3((C * C + 15) / 16), dim3(256), 0, st,
                                   (const float *)scratch, (const float *)gld, (float *)gW, (float *)gldu, nparts, C * C, B, HW),
                hipLaunchKernelGGL(inv1x1_wgrad_partial_kernel<double>, dim3(nparts), dim3(256), lds, st, (const double *)z,
                                   (const double *)gy, (double *)scratch, B, C, HW, ipw);
                hipLaunchKernelGGL(inv1x1_wgrad_reduce_kernel<double>, dim3((C * C + 15) / 16), dim3(256), 0, st,
                                   (const double *)scratch, (const double *)gld, (double *)gW, (double *)gldu, nparts, C * C, B,
                                   HW));
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// Backward of nf_maf_affine: g_x (B, D) and g_params (B, D, 2) from the cotangents gy (B, D) / gld (B) (either may be NULL).
extern "C" int nf_maf_affine_bwd(const void *x, const void *params, const void *gy, const void *gld, void *gx, void *gparams,
                                 int64_t B, int D, int direction, int dtype, nf_stream_t stream) {
    if (B < 0 || D < 1 || (direction != 0 && direction != 1)) return NF_EINVAL;
    if (B == 0) return NF_OK;
    if (!x || !params || !gx || !gparams) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    const int grid = grid_for(B * D, 256);
    NF_DISPATCH(dtype,
                hipLaunchKernelGGL(maf_affine_bwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float *)x, (const float *)params,
                                   (const float *)gy, (const float *)gld, (float *)gx, (float *)gparams, B, D, direction),
                hipLaunchKernelGGL(maf_affine_bwd_kernel<double>, dim3(grid), dim3(256), 0, st, (const double *)x,
                                   (const double *)params, (const double *)gy, (const double *)gld, (double *)gx, (double *)gparams,
                                   B, D, direction));
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// VJP of Invertible1x1Conv's LU parametrisation in the density direction (mixing.py:88-104): gL, gU (C x C: strictly lower / upper
// parts, zero elsewhere), g_log_S (C) from gW (C x C) and gl (0-dim, may be NULL).  C <= 64.
extern "C" int nf_inv1x1_lu_grads(const void *P, const void *L, const void *U, const void *sign_S, const void *log_S, const void *gW,
                                  const void *gl, void *gL, void *gU, void *glogS, int C, int dtype, nf_stream_t stream) {
    if (C < 1) return NF_EINVAL;
    if (C > 64) return NF_ENOTSUP;
    if (!P || !L || !U || !sign_S || !log_S || !gW || !gL || !gU || !glogS) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    {   // (float64 at C > 52 stages more than the default 64 KB of dynamic LDS)
        static LdsOptIn opt_f = {}, opt_d = {};
        const int rc = dtype == NF_F64
            ? opt_in_lds(reinterpret_cast<const void *>(&inv1x1_lu_grads_kernel<double>), (size_t)(3 * C * C + C) * sizeof(double), opt_d)
            : opt_in_lds(reinterpret_cast<const void *>(&inv1x1_lu_grads_kernel<float>), (size_t)(3 * C * C + C) * sizeof(float), opt_f);
        if (rc != NF_OK) return NF_ENOTSUP;
    }
    NF_DISPATCH(dtype,
                hipLaunchKernelGGL(inv1x1_lu_grads_kernel<float>, dim3(C), dim3(64), (size_t)(3 * C * C + C) * sizeof(float), st, (const float *)P, (const float *)L,
                                   (const float *)U, (const float *)sign_S, (const float *)log_S, (const float *)gW, (const float *)gl,
                                   (float *)gL, (float *)gU, (float *)glogS, C),
                hipLaunchKernelGGL(inv1x1_lu_grads_kernel<double>, dim3(C), dim3(64), (size_t)(3 * C * C + C) * sizeof(double), st, (const double *)P, (const double *)L,
                                   (const double *)U, (const double *)sign_S, (const double *)log_S, (const double *)gW,
                                   (const double *)gl, (double *)gL, (double *)gU, (double *)glogS, C));
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// nf_inv1x1_lu_grads (float32) for n layers of one size: every argument a HOST array of n device pointers (gl[i] may be NULL).
extern "C" int nf_inv1x1_lu_grads_multi(const void *const *P, const void *const *L, const void *const *U, const void *const *sign_S,
                                        const void *const *log_S, const void *const *gW, const void *const *gl, void *const *gL,
                                        void *const *gU, void *const *glogS, int n, int C, nf_stream_t stream) {
    if (n < 0 || C < 1) return NF_EINVAL;
    if (C > 64) return NF_ENOTSUP;
    if (n == 0) return NF_OK;
    if (!P || !L || !U || !sign_S || !log_S || !gW || !gl || !gL || !gU || !glogS) return NF_EFAULT;
    for (int i = 0; i < n; ++i)
        if (!P[i] || !L[i] || !U[i] || !sign_S[i] || !log_S[i] || !gW[i] || !gL[i] || !gU[i] || !glogS[i]) return NF_EFAULT;
    static LdsOptIn opt_m = {};
    const size_t lds = (size_t)(3 * C * C + C) * sizeof(float);
    if (opt_in_lds(reinterpret_cast<const void *>(&inv1x1_lu_grads_multi_kernel), lds, opt_m) != NF_OK) return NF_ENOTSUP;
    for (int i0 = 0; i0 < n; i0 += I1G_MULTI) {
        const int m = n - i0 < I1G_MULTI ? n - i0 : I1G_MULTI;
        Inv1x1LuGradsMulti t = {};
        for (int i = 0; i < m; ++i) {
            t.P[i] = P[i0 + i]; t.L[i] = L[i0 + i]; t.U[i] = U[i0 + i]; t.sign_S[i] = sign_S[i0 + i]; t.log_S[i] = log_S[i0 + i];
            t.gW[i] = gW[i0 + i]; t.gl[i] = gl[i0 + i]; t.gL[i] = gL[i0 + i]; t.gU[i] = gU[i0 + i]; t.glogS[i] = glogS[i0 + i];
        }
        hipLaunchKernelGGL(inv1x1_lu_grads_multi_kernel, dim3(C, m), dim3(64), lds, (hipStream_t)stream, t, C);
        NF_CHECK_LAUNCH();
    }
    return NF_OK;
}

// One sweep of autograd.MafInverseFn's triangular solve: v (B, D) updated in place from gx, gxm (NULL on the first sweep), the next
// chain's input g_p (B, D, 2) written, *changed (int, zeroed by the caller) set when any element of v moved.
extern "C" int nf_maf_implicit_sweep(const void *x, const void *params, const void *gx, const void *gld, const void *gxm, void *v,
                                     void *gp, void *changed, int64_t B, int D, int dtype, nf_stream_t stream) {
    if (B < 0 || D < 1) return NF_EINVAL;
    if (B == 0) return NF_OK;
    if (!x || !params || !v || !gp || !changed) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    const int grid = grid_for(B * D, 256);
    NF_DISPATCH(dtype,
                hipLaunchKernelGGL(maf_implicit_sweep_kernel<float>, dim3(grid), dim3(256), 0, st, (const float *)x, (const float *)params,
                                   (const float *)gx, (const float *)gld, (const float *)gxm, (float *)v, (float *)gp, (int *)changed,
                                   B, D),
                hipLaunchKernelGGL(maf_implicit_sweep_kernel<double>, dim3(grid), dim3(256), 0, st, (const double *)x,
                                   (const double *)params, (const double *)gx, (const double *)gld, (const double *)gxm, (double *)v,
                                   (double *)gp, (int *)changed, B, D));
    NF_CHECK_LAUNCH();
    return NF_OK;
}
