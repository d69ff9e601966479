// Weight / bias gradient of a Linear layer over a large batch: dW[M][N] = sum_b dY[b][m] * X[b][n], db[m] = sum_b dY[b][m]
// (the backward of the conditioner's nn.Linear layers, nets/resnet.py:37-50, 92-104, in the training step core.py:87-102).
//
// Shape: M, N <= a few hundred, K = batch = 65 536: a split-K problem.  Both operands of v_mfma_f32_32x32x2_f32 come
// straight from the row-major activations: for a k-pair (two batch rows) lane (i = lane & 31, h = lane >> 5) supplies
// A = dY[b + h][m0 + i] and B = X[b + h][n0 + i] -- 128-byte coalesced rows, no LDS, no transposes.  A wave owns one
// 32-row m-tile and up to four n-tiles (64 accumulator registers); a workgroup = 4 m-tiles over one K-chunk, so the X rows
// are shared through L1.  Partial tiles go to a caller-owned scratch [chunk][M][N] and a second kernel sums the chunks in
// a fixed order: deterministic, unlike atomics.
#include "common.hpp"
#include "fused_common.hpp"

namespace nf {

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

constexpr int WG_MT = 4;  // m-tiles (waves) per workgroup

template <int NT>
__global__ void __launch_bounds__(64 * WG_MT)
wgrad_partial_kernel(const float *__restrict__ dY, const float *__restrict__ X, float *__restrict__ part,
                     float *__restrict__ bpart, int64_t B, int M, int N, int chunk_rows) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int mt = blockIdx.y * WG_MT + wid;     // m-tile of this wave
    const int m = mt * 32 + i;
    const bool mval = m < M;
    const int64_t b0 = (int64_t)blockIdx.x * chunk_rows;
    int64_t b1 = b0 + chunk_rows;
    if (b1 > B) b1 = B;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x16{0};
    float bsum = 0.0f;
    if (mt * 32 < M) {
        const float *pa = dY + (mval ? m : 0);
        bool nval[NT];
        const float *pb[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            nval[t] = t * 32 + i < N;
            pb[t] = X + (nval[t] ? t * 32 + i : 0);
        }
        // four k-pairs per trip, all loads of a trip issued before its MFMAs
        int64_t b = b0;
        for (; b + 8 <= b1; b += 8) {
            float a[4], x[4][NT];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t r = b + 2 * j + h;
                a[j] = mval ? pa[r * M] : 0.0f;
#pragma unroll
                for (int t = 0; t < NT; ++t) x[j][t] = nval[t] ? pb[t][r * N] : 0.0f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bsum += a[j];
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = MFMA32(a[j], x[j][t], acc[t]);
            }
        }
        for (; b < b1; b += 2) {   // tail: row pairs, the odd last row is zero-filled
            const int64_t r = b + h;
            const bool rv = r < b1;
            const float a = (mval && rv) ? pa[r * M] : 0.0f;
            bsum += a;
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = MFMA32(a, (nval[t] && rv) ? pb[t][r * N] : 0.0f, acc[t]);
        }
        // C layout: row = (reg & 3) + 8 (reg >> 2) + 4 h (m within the tile), col = lane & 31 (n within the tile)
        float *out = part + (size_t)blockIdx.x * M * N;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = t * 32 + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mm = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (mm < M && n < N) out[(size_t)mm * N + n] = acc[t][r];
            }
        }
        if (bpart) {
            bsum += __shfl_xor(bsum, 32);
            if (h == 0 && mval) bpart[(size_t)blockIdx.x * M + m] = bsum;
        }
    }
}

// out[e] (=|+=) sum over chunks of part[c][e], fixed order.
__global__ void __launch_bounds__(256)
wgrad_reduce_kernel(const float *__restrict__ part, float *__restrict__ out, int64_t n, int chunks, int accumulate) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
        int c = 0;
        for (; c + 4 <= chunks; c += 4) {
            s0 += part[(size_t)c * n + e];
            s1 += part[(size_t)(c + 1) * n + e];
            s2 += part[(size_t)(c + 2) * n + e];
            s3 += part[(size_t)(c + 3) * n + e];
        }
        for (; c < chunks; ++c) s0 += part[(size_t)c * n + e];
        const float s = (s0 + s1) + (s2 + s3);
        out[e] = accumulate ? out[e] + s : s;
    }
}

static int wgrad_chunks(int64_t B, int M) {
    // aim at >= 1024 waves in flight: (#chunks) x (m-tiles) >= 1024, chunk rows a multiple of 8, at least 64
    const int mtiles = (M + 31) / 32;
    int64_t want = (1024 + mtiles - 1) / mtiles;
    int64_t rows = (B + want - 1) / want;
    rows = (rows + 7) / 8 * 8;
    if (rows < 64) rows = 64;
    return (int)rows;
}

}  // namespace nf

extern "C" int64_t nf_linear_wgrad_scratch_floats(int64_t B, int M, int N) {
    if (B < 0 || M < 1 || N < 1) return NF_EINVAL;
    const int rows = nf::wgrad_chunks(B, M);
    const int64_t chunks = (B + rows - 1) / rows;
    return chunks * ((int64_t)M * N + M);
}

extern "C" int nf_linear_wgrad(const void *dY, const void *X, void *dW, void *db, void *scratch, int64_t B, int M, int N,
                               int accumulate, nf_stream_t stream) {
    if (B < 1 || M < 1 || N < 1 || (accumulate != 0 && accumulate != 1)) return NF_EINVAL;
    if (N > 128) return NF_ENOTSUP;  // four 32-column tiles of accumulators per wave
    if (!dY || !X || !dW || !scratch) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    const int rows = nf::wgrad_chunks(B, M);
    const int chunks = (int)((B + rows - 1) / rows);
    float *part = (float *)scratch;
    float *bpart = db ? part + (size_t)chunks * M * N : nullptr;
    const int mtiles = (M + 31) / 32;
    dim3 grid(chunks, (mtiles + nf::WG_MT - 1) / nf::WG_MT);
    const int nt = (N + 31) / 32;
#define NF_WGRAD_LAUNCH(NT)                                                                                         \
    hipLaunchKernelGGL(nf::wgrad_partial_kernel<NT>, grid, dim3(64 * nf::WG_MT), 0, st, (const float *)dY,           \
                       (const float *)X, part, bpart, B, M, N, rows)
    if (nt == 1) NF_WGRAD_LAUNCH(1);
    else if (nt == 2) NF_WGRAD_LAUNCH(2);
    else if (nt == 3) NF_WGRAD_LAUNCH(3);
    else NF_WGRAD_LAUNCH(4);
#undef NF_WGRAD_LAUNCH
    NF_CHECK_LAUNCH();
    const int64_t n = (int64_t)M * N;
    hipLaunchKernelGGL(nf::wgrad_reduce_kernel, dim3(nf::grid_for(n, 256)), dim3(256), 0, st, part, (float *)dW, n, chunks,
                       accumulate);
    NF_CHECK_LAUNCH();
    if (db) {
        hipLaunchKernelGGL(nf::wgrad_reduce_kernel, dim3(nf::grid_for(M, 256)), dim3(256), 0, st, bpart, (float *)db,
                           (int64_t)M, chunks, accumulate);
        NF_CHECK_LAUNCH();
    }
    return NF_OK;
}
