// Image-side glue of the Glow path for gfx950:
//   nf_logit                       normflows/transforms.py:8-47   (Logit.forward / Logit.inverse + log-det)
//   nf_diag_gaussian_log_prob_rows normflows/distributions/base.py:273-345 (ClassCondDiagGaussian.log_prob: every
//                                  sample has its own mean / log-scale row, selected by class label or blended)
// Both are one read (+ one write) per element with a per-sample wave reduction: HBM-bound.
#include "common.hpp"

namespace nf {

// One wave per sample; `inner` = C*H*W elements.
//   direction 0 (Logit.forward, transforms.py:25-32):  y = (sigmoid(z) - alpha)/beta,
//        ld = -log(beta) n + sum logsigmoid(z) + sum logsigmoid(-z)
//   direction 1 (Logit.inverse, :34-47): u = alpha + beta z; y = log u - log(1 - u); ld = log(beta) n - sum log u - sum log(1-u)
template <typename T>
__global__ void __launch_bounds__(256)
logit_kernel(const T *__restrict__ z, T *__restrict__ y, T *__restrict__ logdet, int64_t B, int64_t inner, T alpha,
             T log_beta, int direction, int acc) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const T beta = T(1) - T(2) * alpha;
    for (int64_t r = wave; r < B; r += nwaves) {
        T a = T(0);
        for (int64_t j = lane; j < inner; j += 64) {
            const T v = z[r * inner + j];
            if (direction == 0) {
                a -= softplus(-v) + softplus(v);  // logsigmoid(v) + logsigmoid(-v)
                y[r * inner + j] = (sigmoid(v) - alpha) / beta;
            } else {
                const T u = alpha + beta * v;
                const T lu = M<T>::log(u), l1 = M<T>::log(T(1) - u);
                a -= lu + l1;
                y[r * inner + j] = lu - l1;
            }
        }
        a = wave_sum(a);
        if (lane == 0) ld_store(logdet + r, a + (direction == 0 ? -log_beta : log_beta) * (T)inner, acc);
    }
}

// out[b] (acc)= -d/2 log(2 pi) - sum_j (ls_j + 0.5 ((z_bj - loc_j)/exp(ls_j))^2) with loc/ls rows selected per sample:
// row = row_idx[b] (class label; rows are the transposed (num_classes, d) parameters) or b itself (blended rows).
template <typename T>
__global__ void __launch_bounds__(256)
diag_gaussian_rows_kernel(const T *__restrict__ z, const T *__restrict__ loc, const T *__restrict__ log_scale,
                          const int64_t *__restrict__ row_idx, int64_t num_rows, T ls_shift, T *__restrict__ out,
                          int64_t B, int64_t d, T cst, int acc) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t r = wave; r < B; r += nwaves) {
        int64_t row = row_idx ? row_idx[r] : r;
        const bool bad = row < 0 || row >= num_rows;   // out-of-range label: NaN instead of a wild read
        if (bad) row = 0;
        const T *lc = loc + row * d, *lsr = log_scale + row * d;
        T a = T(0);
        for (int64_t j = lane; j < d; j += 64) {
            const T ls = lsr[j] + ls_shift;
            const T q = (z[r * d + j] - lc[j]) / M<T>::exp(ls);
            a += ls + T(0.5) * q * q;
        }
        a = wave_sum(a);
        if (lane == 0) ld_store(out + r, bad ? M<T>::nan() : cst - a, acc);
    }
}

// y[b][c][hw] = leaky_relu(y[b][c][hw] + bias[c]) in place: the two element-wise passes that follow a bias-free
// library convolution in ConvNet2d (nets/cnn.py:40-50), as one read + one write.
template <typename T>
__global__ void __launch_bounds__(256)
bias_leaky_relu_kernel(T *__restrict__ y, const T *__restrict__ bias, int64_t n, int C, int64_t HW, T slope) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)((i / HW) % C);
        const T v = y[i] + bias[c];
        y[i] = v > T(0) ? v : v * slope;
    }
}

}  // namespace nf

#define NF_DISPATCH(dtype, CALL_F32, CALL_F64) \
    do {                                       \
        if ((dtype) == NF_F32) { CALL_F32; }   \
        else if ((dtype) == NF_F64) { CALL_F64; } \
        else return NF_ENOTSUP;                \
    } while (0)

extern "C" int nf_logit(const void *z, void *y, void *logdet, int64_t B, int64_t inner, double alpha, int direction,
                        int acc, int dtype, nf_stream_t stream) {
    if (B < 0 || inner < 1 || (direction != 0 && direction != 1) || acc < NF_LD_SUB || acc > NF_LD_ADD) return NF_EINVAL;
    if (!(alpha >= 0.0 && alpha < 0.5)) return NF_EINVAL;
    if (B == 0) return NF_OK;
    if (!z || !y || !logdet) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    const double log_beta = log(1.0 - 2.0 * alpha);
    const int grid = nf::grid_for(B, 4);
    NF_DISPATCH(dtype,
                hipLaunchKernelGGL(nf::logit_kernel<float>, dim3(grid), dim3(256), 0, st, (const float *)z, (float *)y,
                                   (float *)logdet, B, inner, (float)alpha, (float)log_beta, direction, acc),
                hipLaunchKernelGGL(nf::logit_kernel<double>, dim3(grid), dim3(256), 0, st, (const double *)z, (double *)y,
                                   (double *)logdet, B, inner, alpha, log_beta, direction, acc));
    NF_CHECK_LAUNCH();
    return NF_OK;
}

extern "C" int nf_diag_gaussian_log_prob_rows(const void *z, const void *loc_rows, const void *log_scale_rows,
                                              const int64_t *row_idx, int64_t num_rows, double log_scale_shift,
                                              void *out, int64_t B, int64_t d, int acc, int dtype, nf_stream_t stream) {
    if (B < 0 || d < 1 || num_rows < 1 || acc < NF_LD_SUB || acc > NF_LD_ADD) return NF_EINVAL;
    if (!row_idx && num_rows < B) return NF_EINVAL;
    if (B == 0) return NF_OK;
    if (!z || !loc_rows || !log_scale_rows || !out) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    const double cst = -0.5 * (double)d * log(2.0 * M_PI);  // base.py:341
    const int grid = nf::grid_for(B, 4);
    NF_DISPATCH(dtype,
                hipLaunchKernelGGL(nf::diag_gaussian_rows_kernel<float>, dim3(grid), dim3(256), 0, st, (const float *)z,
                                   (const float *)loc_rows, (const float *)log_scale_rows, row_idx, num_rows,
                                   (float)log_scale_shift, (float *)out, B, d, (float)cst, acc),
                hipLaunchKernelGGL(nf::diag_gaussian_rows_kernel<double>, dim3(grid), dim3(256), 0, st, (const double *)z,
                                   (const double *)loc_rows, (const double *)log_scale_rows, row_idx, num_rows,
                                   log_scale_shift, (double *)out, B, d, cst, acc));
    NF_CHECK_LAUNCH();
    return NF_OK;
}

namespace nf {
// ld[b] = (((ld[b] +- t_0[b]) +- t_1[b]) ...): the per-layer log-det terms of a chain folded into the accumulator in the chain's own
// order -- bit for bit what one `ld += t_i` launch per layer gives (round 6, late: 96 three-microsecond launches per Glow level).
constexpr int LDF_MAX = 120;
struct LdFoldTerms {
    const float *t[LDF_MAX];
    unsigned neg[(LDF_MAX + 31) / 32];      // bit i: term i is subtracted
};

__global__ void __launch_bounds__(256)
ld_fold_multi_kernel(float *__restrict__ ld, LdFoldTerms m, int n, int64_t B) {
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < B; b += (int64_t)gridDim.x * blockDim.x) {
        float a = ld[b];
        int i = 0;
        for (; i + 8 <= n; i += 8) {         // eight independent loads, then the adds in order (one load at a time: 40 us for 96 terms)
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = m.t[i + j][b];
#pragma unroll
            for (int j = 0; j < 8; ++j) a = (m.neg[(i + j) >> 5] >> ((i + j) & 31)) & 1u ? a - v[j] : a + v[j];
        }
        for (; i < n; ++i) {
            const float v = m.t[i][b];
            a = (m.neg[i >> 5] >> (i & 31)) & 1u ? a - v : a + v;
        }
        ld[b] = a;
    }
}
}  // namespace nf

// ld (B) float32 += / -= the n terms (HOST array of n device pointers to (B) float32 vectors; negate[i] != 0: subtracted), in order.
extern "C" int nf_ld_fold_multi(void *ld, const void *const *terms, const int *negate, int n, int64_t B, nf_stream_t stream) {
    if (n < 0 || B < 0) return NF_EINVAL;
    if (n == 0 || B == 0) return NF_OK;
    if (!ld || !terms || !negate) return NF_EFAULT;
    for (int i = 0; i < n; ++i)
        if (!terms[i]) return NF_EFAULT;
    const int grid = nf::grid_for(B, 256);
    for (int i0 = 0; i0 < n; i0 += nf::LDF_MAX) {
        const int m = n - i0 < nf::LDF_MAX ? n - i0 : nf::LDF_MAX;
        nf::LdFoldTerms t = {};
        for (int i = 0; i < m; ++i) {
            t.t[i] = (const float *)terms[i0 + i];
            if (negate[i0 + i]) t.neg[i >> 5] |= 1u << (i & 31);
        }
        hipLaunchKernelGGL(nf::ld_fold_multi_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (float *)ld, t, m, B);
        NF_CHECK_LAUNCH();
    }
    return NF_OK;
}

extern "C" int nf_bias_leaky_relu(void *y, const void *bias, int64_t B, int C, int64_t HW, double negative_slope, int dtype,
                                  nf_stream_t stream) {
    if (B < 0 || C < 1 || HW < 1) return NF_EINVAL;
    if (B == 0) return NF_OK;
    if (!y || !bias) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    const int64_t n = B * C * HW;
    const int grid = nf::grid_for(n, 256 * 4);
    NF_DISPATCH(dtype,
                hipLaunchKernelGGL(nf::bias_leaky_relu_kernel<float>, dim3(grid), dim3(256), 0, st, (float *)y,
                                   (const float *)bias, n, C, HW, (float)negative_slope),
                hipLaunchKernelGGL(nf::bias_leaky_relu_kernel<double>, dim3(grid), dim3(256), 0, st, (double *)y,
                                   (const double *)bias, n, C, HW, negative_slope));
    NF_CHECK_LAUNCH();
    return NF_OK;
}
