// rows_matvec.hip -- y_b = W x_b for every row b of a row-major (B, D) matrix, D <= 64, on exact-fp32 MFMA.
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

__global__ void __launch_bounds__(64 * RM_NW)
rows_matvec_kernel(const float *__restrict__ x, const float *__restrict__ W, float *__restrict__ y, int64_t B, int D) {
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

}  // namespace nf

using namespace nf;

extern "C" int nf_rows_matvec(const void *x, const void *W, void *y, int64_t B, int D, nf_stream_t stream) {
    if (B < 0 || D < 1) return NF_EINVAL;
    if (D > RM_D) return NF_ENOTSUP;
    if (B == 0) return NF_OK;
    if (!x || !W || !y) return NF_EFAULT;
    const int64_t grid = (B + 32 * RM_NW - 1) / (32 * RM_NW);
    if (grid > 0x7fffffff) return NF_ERANGE;
    hipLaunchKernelGGL(rows_matvec_kernel, dim3((unsigned)grid), dim3(64 * RM_NW), 0, (hipStream_t)stream, (const float *)x,
                       (const float *)W, (float *)y, B, D);
    NF_CHECK_LAUNCH();
    return NF_OK;
}
