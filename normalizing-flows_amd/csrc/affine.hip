// affine.hip -- affine family: MaskedAffineFlow, AffineCoupling(+Split/Merge), AffineConstFlow/ActNorm,
// DiagGaussian.log_prob, Squeeze.  gfx950 only.  All kernels are HBM-bound streams: one read of every input
// element, one write of every output element, one fp value of log-det per sample.
//
// Reference behaviour: normflows/flows/affine/coupling.py:38-54, :117-171, :209-229, :232-267;
// flows/normalization.py:19-39; flows/reshape.py:30-33, :57-61, :116-128; distributions/base.py:94-103.
#include "common.hpp"

namespace nf {

// ---------------------------------------------------------------------------------------------------------
// MaskedAffineFlow: one workgroup per sample row-chunk; per-sample reduction of (1-b)*s.
template <typename T>
__global__ void __launch_bounds__(256)
masked_affine_kernel(const T *__restrict__ z, const T *__restrict__ b, const T *__restrict__ s,
                     const T *__restrict__ t, T *__restrict__ y, T *__restrict__ logdet, int64_t B, int64_t inner,
                     int direction, int acc) {
    __shared__ T sred[16];
    if constexpr (sizeof(T) == 4) {
        if (inner <= 256 && (inner & 3) == 0) {
            // rows of up to 256 floats, a multiple of 4 long (round 4): a lane owns FOUR consecutive elements (16-byte loads and
            // stores; 4-byte accesses reached 0.38 of the HBM peak at (65 536, 64)), a wave holds 64 / P whole rows (P = lanes per row
            // rounded up to a power of two), the per-sample sum is a butterfly inside the row's P lanes
            typedef float v4 __attribute__((ext_vector_type(4)));
            const int L4 = (int)inner >> 2;
            int P = 1;
            while (P < L4) P <<= 1;
            const int rpw = 64 / P, lane = threadIdx.x & 63, rin = lane / P, i = lane - rin * P;
            const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
            const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
            const bool col_on = i < L4;
            v4 bi = {0.0f, 0.0f, 0.0f, 0.0f};
            if (col_on) bi = *reinterpret_cast<const v4 *>(reinterpret_cast<const float *>(b) + 4 * i);
            for (int64_t r0 = wave * rpw; r0 < B; r0 += nwaves * rpw) {
                const int64_t r = r0 + rin;
                const bool on = r < B && col_on;
                float ld = 0.0f;
                if (on) {
                    const int64_t o = r * inner + 4 * i;
                    const v4 zi = *reinterpret_cast<const v4 *>(reinterpret_cast<const float *>(z) + o);
                    v4 si = {0.0f, 0.0f, 0.0f, 0.0f}, ti = si, yo;
                    if (s) si = *reinterpret_cast<const v4 *>(reinterpret_cast<const float *>(s) + o);
                    if (t) ti = *reinterpret_cast<const v4 *>(reinterpret_cast<const float *>(t) + o);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float se = si[e], te = ti[e];
                        if (!M<float>::finite(se)) se = M<float>::nan();
                        if (!M<float>::finite(te)) te = M<float>::nan();
                        const float zm = bi[e] * zi[e];
                        if (direction == 0) yo[e] = zm + (1.0f - bi[e]) * (zi[e] * M<float>::exp(se) + te);
                        else yo[e] = zm + (1.0f - bi[e]) * (zi[e] - te) * M<float>::exp(-se);
                        ld += (1.0f - bi[e]) * se;
                    }
                    *reinterpret_cast<v4 *>(reinterpret_cast<float *>(y) + o) = yo;
                }
                for (int off = P >> 1; off >= 1; off >>= 1) ld += __shfl_xor(ld, off, 64);
                if (on && i == 0) ld_store(reinterpret_cast<float *>(logdet) + r, direction == 0 ? ld : -ld, acc);
            }
            return;
        }
    }
    if (inner <= 64) {
        // short rows: a wave holds 64 / P whole rows (P = inner rounded up to a power of two), lane = element -- unit-stride
        // loads and stores -- and the per-sample sum is a butterfly inside the row's P lanes
        int P = 1;
        while (P < (int)inner) P <<= 1;
        const int rpw = 64 / P, lane = threadIdx.x & 63, rin = lane / P, i = lane - rin * P;
        const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
        for (int64_t r0 = wave * rpw; r0 < B; r0 += nwaves * rpw) {
            const int64_t r = r0 + rin;
            const bool on = r < B && i < (int)inner;
            T ld = T(0);
            if (on) {
                const int64_t o = r * inner + i;
                const T bi = b[i], zi = z[o];
                T si = s ? s[o] : T(0), ti = t ? t[o] : T(0);
                if (!M<T>::finite(si)) si = M<T>::nan();
                if (!M<T>::finite(ti)) ti = M<T>::nan();
                const T zm = bi * zi;
                if (direction == 0) y[o] = zm + (T(1) - bi) * (zi * M<T>::exp(si) + ti);
                else y[o] = zm + (T(1) - bi) * (zi - ti) * M<T>::exp(-si);
                ld = (T(1) - bi) * si;
            }
            for (int off = P >> 1; off >= 1; off >>= 1) ld += __shfl_xor(ld, off, 64);
            if (on && i == 0) ld_store(logdet + r, direction == 0 ? ld : -ld, acc);
        }
        return;
    }
    for (int64_t r = blockIdx.x; r < B; r += gridDim.x) {
        T ld = T(0);
        for (int64_t i = threadIdx.x; i < inner; i += blockDim.x) {
            const int64_t o = r * inner + i;
            const T bi = b[i], zi = z[o];
            T si = s ? s[o] : T(0), ti = t ? t[o] : T(0);
            if (!M<T>::finite(si)) si = M<T>::nan();
            if (!M<T>::finite(ti)) ti = M<T>::nan();
            const T zm = bi * zi;
            if (direction == 0) y[o] = zm + (T(1) - bi) * (zi * M<T>::exp(si) + ti);
            else y[o] = zm + (T(1) - bi) * (zi - ti) * M<T>::exp(-si);
            ld += (T(1) - bi) * si;
        }
        ld = block_sum(ld, sred);
        if (threadIdx.x == 0) ld_store(logdet + r, direction == 0 ? ld : -ld, acc);
    }
}

// ---------------------------------------------------------------------------------------------------------
// AffineCouplingBlock: one workgroup per sample.  z1 channels are copied, z2 channels transformed with the
// interleaved (shift, scale_) parameter planes.
template <typename T>
__global__ void __launch_bounds__(256)
affine_coupling_kernel(const T *__restrict__ z, const T *__restrict__ param, T *__restrict__ y,
                       T *__restrict__ logdet, int64_t B, int C, int c1, int flip, int64_t HW, int scale_map,
                       int direction, int acc, const T *__restrict__ pbias) {
    __shared__ T sred[16];
    const int c2 = C - c1;
    const int z1_off = flip ? c2 : 0, z2_off = flip ? 0 : c1;  // channel offsets inside a sample
    const int P = scale_map == NF_SCALE_NONE ? c2 : 2 * c2;
    const int64_t n1 = (int64_t)c1 * HW, n2 = (int64_t)c2 * HW;
    for (int64_t r = blockIdx.x; r < B; r += gridDim.x) {
        const T *zr = z + r * (int64_t)C * HW;
        T *yr = y + r * (int64_t)C * HW;
        const T *pr = param + r * (int64_t)P * HW;
        for (int64_t i = threadIdx.x; i < n1; i += blockDim.x) yr[(int64_t)z1_off * HW + i] = zr[(int64_t)z1_off * HW + i];
        T ld = T(0);
        for (int64_t i = threadIdx.x; i < n2; i += blockDim.x) {
            const int64_t c = i / HW, p = i - c * HW;
            const T v = zr[(int64_t)z2_off * HW + i];
            T o;
            if (scale_map == NF_SCALE_NONE) {
                const T sh = pr[i] + (pbias ? pbias[c] : T(0));
                o = direction == 0 ? v + sh : v - sh;
            } else {
                // pbias: bias of the conditioner's last (bias-free) convolution, per parameter channel
                const T sh = pr[(2 * c) * HW + p] + (pbias ? pbias[2 * c] : T(0));
                const T sc = pr[(2 * c + 1) * HW + p] + (pbias ? pbias[2 * c + 1] : T(0));
                if (scale_map == NF_SCALE_EXP) {
                    o = direction == 0 ? v * M<T>::exp(sc) + sh : (v - sh) * M<T>::exp(-sc);
                    ld += sc;  // +sum (forward) / -sum (inverse)
                } else {
                    const T sg = sigmoid(sc + T(2));
                    const T lg = M<T>::log(sg);
                    if (scale_map == NF_SCALE_SIGMOID) {
                        o = direction == 0 ? v / sg + sh : (v - sh) * sg;
                        ld -= lg;  // forward: -sum log scale; inverse: +sum log scale
                    } else {
                        o = direction == 0 ? v * sg + sh : (v - sh) / sg;
                        ld += lg;
                    }
                }
            }
            yr[(int64_t)z2_off * HW + i] = o;
        }
        ld = block_sum(ld, sred);
        if (threadIdx.x == 0) ld_store(logdet + r, direction == 0 ? ld : -ld, acc);
    }
}

// ---------------------------------------------------------------------------------------------------------
// AffineConstFlow / ActNorm apply.  logdet value = +-HW * sum(s) is recomputed per workgroup (C words).
template <typename T>
__global__ void __launch_bounds__(256)
actnorm_kernel(const T *__restrict__ z, const T *__restrict__ s, const T *__restrict__ t, T *__restrict__ y,
               T *__restrict__ logdet_scalar, T *__restrict__ logdet, int64_t B, int C, int64_t HW, int direction,
               int acc) {
    __shared__ T sred[16];
    T part = T(0);
    for (int c = threadIdx.x; c < C; c += blockDim.x) part += s[c];
    T ssum = block_sum(part, sred);
    const T ldv = (direction == 0 ? T(1) : T(-1)) * (T)HW * ssum;
    const int64_t N = B * (int64_t)C * HW;
    const int64_t gstride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gstride) {
        const int c = (int)((i / HW) % C);
        const T v = z[i];
        y[i] = direction == 0 ? v * M<T>::exp(s[c]) + t[c] : (v - t[c]) * M<T>::exp(-s[c]);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && logdet_scalar) *logdet_scalar = ldv;
    if (logdet)
        for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < B; r += gstride) ld_store(logdet + r, ldv, acc);
}

// Per-channel mean and unbiased std over (B, HW): one workgroup per channel, fp64 accumulation of the sum
// then of the squared deviations (two passes, like torch.std's numerically safe formulation).
template <typename T>
__global__ void __launch_bounds__(256)
actnorm_stats_kernel(const T *__restrict__ z, T *__restrict__ mean, T *__restrict__ stdu, int64_t B, int C, int64_t HW) {
    __shared__ double sred[16];
    const int c = blockIdx.x;
    const int64_t n = B * HW;
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const int64_t b = i / HW, p = i - b * HW;
        acc += (double)z[(b * C + c) * HW + p];
    }
    const double mu = block_sum(acc, sred) / (double)n;
    double acc2 = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        const int64_t b = i / HW, p = i - b * HW;
        const double dlt = (double)z[(b * C + c) * HW + p] - mu;
        acc2 += dlt * dlt;
    }
    const double var = block_sum(acc2, sred) / (double)(n - 1);  // n == 1 -> nan, as torch
    if (threadIdx.x == 0) {
        mean[c] = (T)mu;
        stdu[c] = (T)::sqrt(var);
    }
}

template <typename T>
__global__ void actnorm_init_kernel(const T *__restrict__ mean, const T *__restrict__ stdu, T *__restrict__ s,
                                    T *__restrict__ t, int C, int direction) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (direction == 0) {  // normalization.py:23-27
        const T sv = -M<T>::log(stdu[c] + T(1e-6));
        s[c] = sv;
        t[c] = -mean[c] * M<T>::exp(sv);
    } else {  // normalization.py:35-37
        s[c] = M<T>::log(stdu[c] + T(1e-6));
        t[c] = mean[c];
    }
}

// ---------------------------------------------------------------------------------------------------------
// DiagGaussian.log_prob: one wave per sample row for d >= 64, one lane per row otherwise.
template <typename T>
__global__ void __launch_bounds__(256)
diag_gaussian_kernel(const T *__restrict__ z, const T *__restrict__ loc, const T *__restrict__ log_scale,
                     T ls_shift, T *__restrict__ out, int64_t B, int64_t d, T cst, int acc) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    if constexpr (sizeof(T) == 4) {
        if (d <= 256 && (d & 3) == 0) {
            // rows of up to 256 floats, a multiple of 4 long (round 4): a lane owns four consecutive columns -- their 1 / scale and
            // log-scale are computed ONCE per launch, not per element -- and 64 / P whole rows share a wave (P = lanes per row rounded
            // up to a power of two): 16-byte loads, a log2(P)-step butterfly instead of the full wave's six steps per row
            typedef float v4 __attribute__((ext_vector_type(4)));
            const int L4 = (int)d >> 2;
            int P = 1;
            while (P < L4) P <<= 1;
            const int rpw = 64 / P, rin = lane / P, i = lane - rin * P;
            const bool col_on = i < L4;
            v4 mu = {0.0f, 0.0f, 0.0f, 0.0f}, inv = mu;
            float lsum = 0.0f;
            if (col_on) {
                mu = *reinterpret_cast<const v4 *>(reinterpret_cast<const float *>(loc) + 4 * i);
                const v4 lsv = *reinterpret_cast<const v4 *>(reinterpret_cast<const float *>(log_scale) + 4 * i);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float ls = lsv[e] + (float)ls_shift;
                    inv[e] = 1.0f / M<float>::exp(ls);
                    lsum += ls;
                }
            }
            for (int64_t r0 = wave * rpw; r0 < B; r0 += nwaves * rpw) {
                const int64_t r = r0 + rin;
                float a = 0.0f;
                if (r < B && col_on) {
                    const v4 zv = *reinterpret_cast<const v4 *>(reinterpret_cast<const float *>(z) + r * d + 4 * i);
                    a = lsum;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float q = (zv[e] - mu[e]) * inv[e];
                        a += 0.5f * q * q;
                    }
                }
                for (int off = P >> 1; off >= 1; off >>= 1) a += __shfl_xor(a, off, 64);
                if (r < B && i == 0) ld_store(reinterpret_cast<float *>(out) + r, (float)cst - a, acc);
            }
            return;
        }
    }
    for (int64_t r = wave; r < B; r += nwaves) {
        T a = T(0);
        for (int64_t j = lane; j < d; j += 64) {
            const T ls = log_scale[j] + ls_shift;
            const T q = (z[r * d + j] - loc[j]) / M<T>::exp(ls);
            a += ls + T(0.5) * q * q;
        }
        a = wave_sum(a);
        if (lane == 0) ld_store(out + r, cst - a, acc);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Squeeze (reshape.py:116-128).  forward: (C,H,W)->(C/4,2H,2W): out[c, 2h+i, 2w+j] = in[4c+2i+j, h, w];
// inverse is the opposite mapping.
template <typename T>
__global__ void __launch_bounds__(256)
squeeze_kernel(const T *__restrict__ z, T *__restrict__ y, int64_t B, int C, int H, int W, int direction) {
    const int64_t N = B * (int64_t)C * H * W;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        // index the OUTPUT element i, gather from the input
        if (direction == 0) {
            const int Co = C / 4, Ho = 2 * H, Wo = 2 * W;
            int64_t r = i;
            const int wo = (int)(r % Wo); r /= Wo;
            const int ho = (int)(r % Ho); r /= Ho;
            const int co = (int)(r % Co); r /= Co;
            const int ci = 4 * co + 2 * (ho & 1) + (wo & 1);
            y[i] = z[((r * C + ci) * H + (ho >> 1)) * W + (wo >> 1)];
        } else {
            const int Co = 4 * C, Ho = H / 2, Wo = W / 2;
            int64_t r = i;
            const int wo = (int)(r % Wo); r /= Wo;
            const int ho = (int)(r % Ho); r /= Ho;
            const int co = (int)(r % Co); r /= Co;
            const int ci = co >> 2, di = (co >> 1) & 1, dj = co & 1;
            y[i] = z[((r * C + ci) * H + (2 * ho + di)) * W + (2 * wo + dj)];
        }
    }
}


// ---------------------------------------------------------------------------------------------------------
// MaskedAffineAutoregressive element-wise transform (normflows/flows/affine/autoregressive.py:98-128):
// params (B, D, 2) = (unconstrained_scale, shift) per feature; scale = sigmoid(u + 2) + 1e-3.
//   direction 0 (_elementwise_forward): y = scale x + shift,      ld = +sum log scale
//   direction 1 (_elementwise_inverse): y = (x - shift) / scale,  ld = -sum log scale
// One wave per row (lanes stride over features), fixed-order wave reduction for the log-det.
template <typename T>
__global__ void __launch_bounds__(256)
maf_affine_kernel(const T *__restrict__ x, const T *__restrict__ params, T *__restrict__ y, T *__restrict__ logdet,
                  int64_t B, int D, int direction, int acc) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < B; r += nwaves) {
        T a = T(0);
        for (int j = lane; j < D; j += 64) {
            const T u = params[(r * D + j) * 2], sh = params[(r * D + j) * 2 + 1];
            const T scale = sigmoid(u + T(2)) + T(1e-3);
            const T xv = x[r * D + j];
            y[r * D + j] = direction == 0 ? scale * xv + sh : (xv - sh) / scale;
            a += M<T>::log(scale);
        }
        a = wave_sum(a);
        if (lane == 0 && logdet) ld_store(logdet + r, direction == 0 ? a : -a, acc);
    }
}

}  // namespace nf

using namespace nf;

#define NF_DISPATCH(dtype, CALL_F32, CALL_F64) \
    do {                                       \
        if ((dtype) == NF_F32) { CALL_F32; }   \
        else if ((dtype) == NF_F64) { CALL_F64; } \
        else return NF_ENOTSUP;                \
    } while (0)

extern "C" int nf_masked_affine(const void *z, const void *b, const void *s, const void *t, void *y, void *logdet,
                                int64_t B, int64_t inner, int direction, int acc, int dtype, nf_stream_t stream) {
    if (B < 0 || inner < 1 || (direction != 0 && direction != 1) || acc < NF_LD_SUB || acc > NF_LD_ADD) return NF_EINVAL;
    if (B == 0) return NF_OK;
    if (!z || !b || !y || !logdet) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    int grid = inner <= 64 ? grid_for(B * inner, 256 * 4) : grid_for(B, 1, 256 * 16);
    if (dtype == NF_F32 && inner <= 256 && (inner & 3) == 0) grid = grid_for(B * inner / 4, 256);
    NF_DISPATCH(dtype,
                hipLaunchKernelGGL(masked_affine_kernel<float>, dim3(grid), dim3(256), 0, st, (const float *)z,
                                   (const float *)b, (const float *)s, (const float *)t, (float *)y, (float *)logdet, B,
                                   inner, direction, acc),
                hipLaunchKernelGGL(masked_affine_kernel<double>, dim3(grid), dim3(256), 0, st, (const double *)z,
                                   (const double *)b, (const double *)s, (const double *)t, (double *)y,
                                   (double *)logdet, B, inner, direction, acc));
    NF_CHECK_LAUNCH();
    return NF_OK;
}

extern "C" int nf_affine_coupling_pb(const void *z, const void *param, const void *param_bias, void *y, void *logdet,
                                     int64_t B, int C, int c1, int flip, int64_t HW, int scale_map, int direction,
                                     int acc, int dtype, nf_stream_t stream) {
    if (B < 0 || C < 1 || c1 < 0 || c1 >= C || HW < 1 || (direction != 0 && direction != 1)) return NF_EINVAL;
    if (scale_map < NF_SCALE_EXP || scale_map > NF_SCALE_NONE || acc < NF_LD_SUB || acc > NF_LD_ADD) return NF_EINVAL;
    if (B == 0) return NF_OK;
    if (!z || !param || !y || !logdet) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    const int grid = grid_for(B, 1, 256 * 16);
    NF_DISPATCH(dtype,
                hipLaunchKernelGGL(affine_coupling_kernel<float>, dim3(grid), dim3(256), 0, st, (const float *)z,
                                   (const float *)param, (float *)y, (float *)logdet, B, C, c1, flip, HW, scale_map,
                                   direction, acc, (const float *)param_bias),
                hipLaunchKernelGGL(affine_coupling_kernel<double>, dim3(grid), dim3(256), 0, st, (const double *)z,
                                   (const double *)param, (double *)y, (double *)logdet, B, C, c1, flip, HW, scale_map,
                                   direction, acc, (const double *)param_bias));
    NF_CHECK_LAUNCH();
    return NF_OK;
}

extern "C" int nf_affine_coupling(const void *z, const void *param, void *y, void *logdet, int64_t B, int C, int c1,
                                  int flip, int64_t HW, int scale_map, int direction, int acc, int dtype,
                                  nf_stream_t stream) {
    return nf_affine_coupling_pb(z, param, nullptr, y, logdet, B, C, c1, flip, HW, scale_map, direction, acc, dtype, stream);
}

extern "C" int nf_actnorm(const void *z, const void *s, const void *t, void *y, void *logdet_scalar, void *logdet,
                          int64_t B, int C, int64_t HW, int direction, int acc, int dtype, nf_stream_t stream) {
    if (B < 0 || C < 1 || HW < 1 || (direction != 0 && direction != 1) || acc < NF_LD_SUB || acc > NF_LD_ADD)
        return NF_EINVAL;
    if (!s || !t) return NF_EFAULT;
    if (B > 0 && (!z || !y)) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    const int grid = grid_for(B * (int64_t)C * HW, 256 * 4);
    NF_DISPATCH(dtype,
                hipLaunchKernelGGL(actnorm_kernel<float>, dim3(grid), dim3(256), 0, st, (const float *)z,
                                   (const float *)s, (const float *)t, (float *)y, (float *)logdet_scalar,
                                   (float *)logdet, B, C, HW, direction, acc),
                hipLaunchKernelGGL(actnorm_kernel<double>, dim3(grid), dim3(256), 0, st, (const double *)z,
                                   (const double *)s, (const double *)t, (double *)y, (double *)logdet_scalar,
                                   (double *)logdet, B, C, HW, direction, acc));
    NF_CHECK_LAUNCH();
    return NF_OK;
}

extern "C" int nf_actnorm_stats(const void *z, void *mean, void *std_unbiased, int64_t B, int C, int64_t HW, int dtype,
                                nf_stream_t stream) {
    if (B < 1 || C < 1 || HW < 1) return NF_EINVAL;
    if (!z || !mean || !std_unbiased) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    NF_DISPATCH(dtype,
                hipLaunchKernelGGL(actnorm_stats_kernel<float>, dim3(C), dim3(256), 0, st, (const float *)z,
                                   (float *)mean, (float *)std_unbiased, B, C, HW),
                hipLaunchKernelGGL(actnorm_stats_kernel<double>, dim3(C), dim3(256), 0, st, (const double *)z,
                                   (double *)mean, (double *)std_unbiased, B, C, HW));
    NF_CHECK_LAUNCH();
    return NF_OK;
}

extern "C" int nf_actnorm_init(const void *mean, const void *std_unbiased, void *s, void *t, int C, int direction,
                               int dtype, nf_stream_t stream) {
    if (C < 1 || (direction != 0 && direction != 1)) return NF_EINVAL;
    if (!mean || !std_unbiased || !s || !t) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    const int grid = (C + 255) / 256;
    NF_DISPATCH(dtype,
                hipLaunchKernelGGL(actnorm_init_kernel<float>, dim3(grid), dim3(256), 0, st, (const float *)mean,
                                   (const float *)std_unbiased, (float *)s, (float *)t, C, direction),
                hipLaunchKernelGGL(actnorm_init_kernel<double>, dim3(grid), dim3(256), 0, st, (const double *)mean,
                                   (const double *)std_unbiased, (double *)s, (double *)t, C, direction));
    NF_CHECK_LAUNCH();
    return NF_OK;
}

extern "C" int nf_diag_gaussian_log_prob(const void *z, const void *loc, const void *log_scale, double log_scale_shift,
                                         void *out, int64_t B, int64_t d, int acc, int dtype, nf_stream_t stream) {
    if (B < 0 || d < 1 || acc < NF_LD_SUB || acc > NF_LD_ADD) return NF_EINVAL;
    if (B == 0) return NF_OK;
    if (!z || !loc || !log_scale || !out) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    const double cst = -0.5 * (double)d * log(2.0 * M_PI);  // base.py:99
    int grid = grid_for(B, 4);
    if (dtype == NF_F32 && d <= 256 && (d & 3) == 0) grid = grid_for(B * d / 4, 256);
    NF_DISPATCH(dtype,
                hipLaunchKernelGGL(diag_gaussian_kernel<float>, dim3(grid), dim3(256), 0, st, (const float *)z,
                                   (const float *)loc, (const float *)log_scale, (float)log_scale_shift, (float *)out,
                                   B, d, (float)cst, acc),
                hipLaunchKernelGGL(diag_gaussian_kernel<double>, dim3(grid), dim3(256), 0, st, (const double *)z,
                                   (const double *)loc, (const double *)log_scale, log_scale_shift, (double *)out, B,
                                   d, cst, acc));
    NF_CHECK_LAUNCH();
    return NF_OK;
}

extern "C" int nf_squeeze(const void *z, void *y, int64_t B, int C, int H, int W, int direction, int dtype,
                          nf_stream_t stream) {
    if (B < 0 || C < 1 || H < 1 || W < 1 || (direction != 0 && direction != 1)) return NF_EINVAL;
    if (direction == 0 && (C % 4)) return NF_EINVAL;
    if (direction == 1 && ((H % 2) || (W % 2))) return NF_EINVAL;
    if (B == 0) return NF_OK;
    if (!z || !y) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    const int grid = grid_for(B * (int64_t)C * H * W, 256 * 4);
    NF_DISPATCH(dtype,
                hipLaunchKernelGGL(squeeze_kernel<float>, dim3(grid), dim3(256), 0, st, (const float *)z, (float *)y, B,
                                   C, H, W, direction),
                hipLaunchKernelGGL(squeeze_kernel<double>, dim3(grid), dim3(256), 0, st, (const double *)z,
                                   (double *)y, B, C, H, W, direction));
    NF_CHECK_LAUNCH();
    return NF_OK;
}

extern "C" int nf_maf_affine(const void *x, const void *params, void *y, void *logdet, int64_t B, int D, int direction,
                             int acc, int dtype, nf_stream_t stream) {
    if (B < 0 || D < 1 || (direction != 0 && direction != 1) || acc < NF_LD_SUB || acc > NF_LD_ADD) return NF_EINVAL;
    if (B == 0) return NF_OK;
    if (!x || !params || !y) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    const int grid = grid_for(B, 4);
    NF_DISPATCH(dtype,
                hipLaunchKernelGGL(maf_affine_kernel<float>, dim3(grid), dim3(256), 0, st, (const float *)x,
                                   (const float *)params, (float *)y, (float *)logdet, B, D, direction, acc),
                hipLaunchKernelGGL(maf_affine_kernel<double>, dim3(grid), dim3(256), 0, st, (const double *)x,
                                   (const double *)params, (double *)y, (double *)logdet, B, D, direction, acc));
    NF_CHECK_LAUNCH();
    return NF_OK;
}
