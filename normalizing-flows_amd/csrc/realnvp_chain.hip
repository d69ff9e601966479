// A whole RealNVP-style stack -- MaskedAffineFlow layers whose s / t conditioners are small MLPs, interleaved with
// ActNorm / AffineConstFlow layers -- as ONE launch for gfx950 (BASELINE configs[0]: 4 x [MaskedAffineFlow + ActNorm] on
// 2-D data, batch 1024: ~50 eager kernels per layer in the reference, latency-bound).
//
// Reference semantics: normflows/flows/affine/coupling.py:209-229 (MaskedAffineFlow: z_masked = b z, scale = s(z_masked),
// trans = t(z_masked), non-finite scale / trans -> NaN, forward z' = z_masked + (1 - b)(z e^scale + trans),
// log_det = sum (1 - b) scale; inverse z' = z_masked + (1 - b)(z - trans) e^-scale, log_det = -sum (1 - b) scale),
// :38-54 (AffineConstFlow / ActNorm: z e^s + t, (z - t) e^-s) and nets/mlp.py:5-58 (Linear + LeakyReLU stack).
//
// lane = sample: the d <= 16 coordinates stay in registers for the whole stack, hidden activations (<= 64 wide) live in an
// LDS column per lane, weights are read with scalar loads (wave-uniform addresses).  The blob is built on the host
// (normflows_amd/core.py::_pack_realnvp):
//   [nrec, d, hmax, 0, off_0 .. off_{nrec-1}, records]     (all float32; small integers stored exactly)
//   record ActNorm      : [1, 0, 0, 0, s[d], t[d]]
//   record MaskedAffine : [2, has_s, has_t, 0, b[d], mlp_s?, mlp_t?]
//   mlp                 : [nlin, slope, n_0 .. n_nlin, then per linear: W (n_{k+1} x n_k row-major), bias (n_{k+1})]
#include "common.hpp"

namespace nf {

constexpr int RN_DMAX = 16;

// y[0..d) = MLP(x[0..d)) for one lane; `buf` = this lane's LDS columns (2 x hmax, stride bs); returns the record size.
__device__ __forceinline__ int rn_mlp(const float *__restrict__ rec, const float (&x)[RN_DMAX], float (&y)[RN_DMAX],
                                      float *buf, int bs, int hmax) {
    const int nlin = (int)rec[0];
    const float slope = rec[1];
    const float *sizes = rec + 2;
    const float *w = sizes + nlin + 1;
    float *cur = buf, *nxt = buf + (size_t)hmax * bs;
    for (int k = 0; k < nlin; ++k) {
        const int nin = (int)sizes[k], nout = (int)sizes[k + 1];
        const float *W = w, *bvec = w + (size_t)nout * nin;
        const bool last = k == nlin - 1;
        for (int j = 0; j < nout; ++j) {
            float a = bvec[j];
            if (k == 0) {
#pragma unroll
                for (int i = 0; i < RN_DMAX; ++i)
                    if (i < nin) a = fmaf(W[j * nin + i], x[i], a);
            } else {
                for (int i = 0; i < nin; ++i) a = fmaf(W[j * nin + i], cur[(size_t)i * bs], a);
            }
            if (!last) a = a > 0.0f ? a : a * slope;   // LeakyReLU(slope) behind every linear but the last (mlp.py:31-38)
            nxt[(size_t)j * bs] = a;
        }
        float *tmp = cur; cur = nxt; nxt = tmp;
        w = bvec + nout;
    }
    const int dout = (int)sizes[nlin];
#pragma unroll
    for (int i = 0; i < RN_DMAX; ++i) y[i] = i < dout ? cur[(size_t)i * bs] : 0.0f;
    return (int)(w - rec);
}

__global__ void __launch_bounds__(256)
realnvp_chain_kernel(const float *__restrict__ z, float *__restrict__ y, float *__restrict__ logdet,
                     const float *__restrict__ blob, int64_t B, int direction, int acc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float *buf = reinterpret_cast<float *>(smem_raw) + threadIdx.x;   // this lane's column; rows are blockDim.x apart
    const int bs = blockDim.x;
    const int nrec = (int)blob[0], d = (int)blob[1], hmax = (int)blob[2];
    const float *offs = blob + 4;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < B; r += (int64_t)gridDim.x * blockDim.x) {
        float v[RN_DMAX];
#pragma unroll
        for (int i = 0; i < RN_DMAX; ++i) v[i] = i < d ? z[r * d + i] : 0.0f;
        float ld = 0.0f;
        for (int q = 0; q < nrec; ++q) {
            const int ri = direction == 0 ? q : nrec - 1 - q;
            const float *rec = blob + (int)offs[ri];
            const int type = (int)rec[0];
            if (type == 1) {
                const float *s = rec + 4, *t = s + d;
#pragma unroll
                for (int i = 0; i < RN_DMAX; ++i) {
                    if (i < d) {
                        if (direction == 0) { v[i] = v[i] * M<float>::exp(s[i]) + t[i]; ld += s[i]; }
                        else { v[i] = (v[i] - t[i]) * M<float>::exp(-s[i]); ld -= s[i]; }
                    }
                }
            } else {
                const bool has_s = rec[1] != 0.0f, has_t = rec[2] != 0.0f;
                const float *b = rec + 4;
                const float *p = b + d;
                float zm[RN_DMAX], sc[RN_DMAX], tr[RN_DMAX];
#pragma unroll
                for (int i = 0; i < RN_DMAX; ++i) { zm[i] = i < d ? b[i] * v[i] : 0.0f; sc[i] = 0.0f; tr[i] = 0.0f; }
                if (has_s) p += rn_mlp(p, zm, sc, buf, bs, hmax);
                if (has_t) p += rn_mlp(p, zm, tr, buf, bs, hmax);
#pragma unroll
                for (int i = 0; i < RN_DMAX; ++i) {
                    if (i < d) {
                        const float s_ = M<float>::finite(sc[i]) ? sc[i] : M<float>::nan();   // coupling.py:213-214
                        const float t_ = M<float>::finite(tr[i]) ? tr[i] : M<float>::nan();
                        const float m = 1.0f - b[i];
                        if (direction == 0) { v[i] = zm[i] + m * (v[i] * M<float>::exp(s_) + t_); ld += m * s_; }
                        else { v[i] = zm[i] + m * (v[i] - t_) * M<float>::exp(-s_); ld -= m * s_; }
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < RN_DMAX; ++i)
            if (i < d) y[r * d + i] = v[i];
        ld_store(logdet + r, ld, acc);
    }
}

}  // namespace nf

extern "C" int nf_realnvp_chain(const void *z, void *y, void *logdet, const void *blob, int64_t B, int d, int hmax,
                                int direction, int acc, nf_stream_t stream) {
    if (B < 0 || d < 1 || d > nf::RN_DMAX || hmax < 1 || hmax > 64) return d > nf::RN_DMAX || hmax > 64 ? NF_ENOTSUP : NF_EINVAL;
    if ((direction != 0 && direction != 1) || acc < NF_LD_SUB || acc > NF_LD_ADD) return NF_EINVAL;
    if (B == 0) return NF_OK;
    if (!z || !y || !logdet || !blob) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    int block = 256;
    while (block > 64 && (size_t)2 * hmax * block * sizeof(float) > 64 * 1024) block >>= 1;
    const size_t lds = (size_t)2 * hmax * block * sizeof(float);
    const int grid = nf::grid_for(B, block);
    hipLaunchKernelGGL(nf::realnvp_chain_kernel, dim3(grid), dim3(block), lds, st, (const float *)z, (float *)y,
                       (float *)logdet, (const float *)blob, B, direction, acc);
    NF_CHECK_LAUNCH();
    return NF_OK;
}
