// train_reduce.hpp -- the fixed-order reductions of the training step's partial tiles as device routines, shared by the stand-alone
// reduction kernels (wgrad.hip: nf::wgrad_reduce_kernel; final_bwd.hip: nf::final_bwd_reduce_kernel) and by the ONE launch per
// layer that runs all of a coupling layer's reductions together (train_bwd.hip: nf::layer_reduce_kernel, round 6).
//
// Why one launch: a benchmark-shaped layer's backward left seven reduction launches of 4-10 us each behind its four heavy kernels
// (ring weight gradient, two residual-block launches with two problems each, the initial layer, the batch-shared spline
// parameters, the LU's two factors): each one a dependent kernel boundary and a grid too small to pull its partial tiles at more
// than ~2 TB/s.  Summation order per output element is unchanged (chunk lanes q, q + RL, ... then lane order): bit-identical
// results, still deterministic, still no atomics.
#pragma once
#include "rqs_bwd_common.hpp"

namespace nf {

#ifndef NF_RL
#define NF_RL 16
#endif
constexpr int RL = NF_RL;              // chunk lanes of a 64-element reduction group (block = 64 x RL threads)

// One 64-element group [e0, e0 + 64) of a partial-tile reduction: out (dW then db) (=|+=) sum over chunks of part[c][e] in a fixed
// order; e < nW goes to dW, the rest to db.  Called by ALL 64 * RL threads of the block (two block barriers inside).
//   skip_every > 1: every skip_every-th row of the M rows is padding and has no output row;
//   colmap: dW keeps the columns n with colmap[n] >= 0, compacted to Nout columns.
__device__ __forceinline__ void wgrad_reduce_group(const float *__restrict__ part, float *__restrict__ dW, float *__restrict__ db,
                                                   int64_t nW, int64_t n, int64_t stride, int chunks, int accumulate, int N,
                                                   int skip_every, const int *__restrict__ colmap, int Nout, int64_t e0,
                                                   float (*sm)[64]) {
    const int el = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int64_t e = e0 + el;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    if (e < n) {
        int c = q;
        for (; c + 3 * RL < chunks; c += 4 * RL) {
            s0 += part[(size_t)c * stride + e];
            s1 += part[(size_t)(c + RL) * stride + e];
            s2 += part[(size_t)(c + 2 * RL) * stride + e];
            s3 += part[(size_t)(c + 3 * RL) * stride + e];
        }
        for (; c < chunks; c += RL) s0 += part[(size_t)c * stride + e];
    }
    sm[q][el] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (q == 0 && e < n) {
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < RL; ++i) s += sm[i][el];
        float *o = e < nW ? dW + e : db + (e - nW);
        if (skip_every) {       // every skip_every-th row of dY is padding: not part of the (M - M / skip_every)-row outputs
            const int64_t m = e < nW ? e / N : e - nW, g = m / skip_every;
            o = (m - g * skip_every == skip_every - 1) ? nullptr : (e < nW ? dW + e - g * N : db + (m - g));
        }
        if (colmap && e < nW) {     // dW keeps the columns n with colmap[n] >= 0, compacted to Nout columns
            const int64_t m = e / N;
            const int c = colmap[e - m * N];
            o = c < 0 ? nullptr : dW + m * Nout + c;
        }
        if (o) *o = accumulate ? *o + s : s;
    }
    __syncthreads();
}

// The same reduction for a group of 256 elements [e0, e0 + 256) with 16-byte loads: lane el owns elements e0 + 4 el .. + 3, every
// one of them summed in exactly the order of wgrad_reduce_group (chunk lanes q, q + RL, ... in four interleaved accumulators, then
// lane order): bit-identical results, a quarter of the load instructions and four times the bytes in flight per thread -- the
// one-launch layer reduction reads 110 MB of freshly written partial tiles and was latency-bound at 3.2 TB/s with 4-byte loads.
// Requires n % 4 == 0, stride % 4 == 0 and a 16-byte aligned `part` (checked by the host).  sm4: [RL][64] float4.
__device__ __forceinline__ void wgrad_reduce_group4(const float *__restrict__ part, float *__restrict__ dW, float *__restrict__ db,
                                                    int64_t nW, int64_t n, int64_t stride, int chunks, int N, int skip_every,
                                                    const int *__restrict__ colmap, int Nout, int64_t e0, f32x4 (*sm4)[64]) {
    const int el = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int64_t e = e0 + 4 * el;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    if (e < n) {
        const float *p0 = part + e;
        int c = q;
        for (; c + 3 * RL < chunks; c += 4 * RL) {
            s0 += *reinterpret_cast<const f32x4 *>(p0 + (size_t)c * stride);
            s1 += *reinterpret_cast<const f32x4 *>(p0 + (size_t)(c + RL) * stride);
            s2 += *reinterpret_cast<const f32x4 *>(p0 + (size_t)(c + 2 * RL) * stride);
            s3 += *reinterpret_cast<const f32x4 *>(p0 + (size_t)(c + 3 * RL) * stride);
        }
        for (; c < chunks; c += RL) s0 += *reinterpret_cast<const f32x4 *>(p0 + (size_t)c * stride);
    }
    sm4[q][el] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (q == 0 && e < n) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < RL; ++i) s += sm4[i][el];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t ek = e + k;
            float *o = ek < nW ? dW + ek : db + (ek - nW);
            if (skip_every) {
                const int64_t m = ek < nW ? ek / N : ek - nW, g = m / skip_every;
                o = (m - g * skip_every == skip_every - 1) ? nullptr : (ek < nW ? dW + ek - g * N : db + (m - g));
            }
            if (colmap && ek < nW) {
                const int64_t m = ek / N;
                const int c = colmap[ek - m * N];
                o = c < 0 ? nullptr : dW + m * Nout + c;
            }
            if (o) *o = s[k];
        }
    }
    __syncthreads();
}

// Knot-space sums of nf_final_bwd -> gradients of the raw batch-shared parameters of feature j (unnormalized_widths / heights /
// derivatives of the unconditional transform, nsf/coupling.py:221-253 over utils/splines.py:100-157): threads 0..239 of the block
// sum the workgroups' partials in a fixed order (10 groups x 24 sums, stride 10), then per feature the chain
//   knot_k = lo + (hi - lo) (k min + scale C_k),  C_k = sum_{i < k} softmax(raw)_i
//   =>  d/d raw_i = (hi - lo) scale softmax_i sum_k G_k ([i < k] - C_k);   d_j = min_d + softplus(raw_j).
// Called by all threads of a block of >= 256 threads; sub: [10][24], sums: [24] floats of shared memory.
constexpr int FBR_PART = F_NI * 24;    // knot-space sums of a workgroup: [feature][7 w | 7 h | 7 d | 3 pad]
__device__ __forceinline__ void final_bwd_reduce_feature(const float *__restrict__ part, int nparts, const float *__restrict__ uw,
                                                         const float *__restrict__ uh, const float *__restrict__ ud,
                                                         float *__restrict__ guw, float *__restrict__ guh,
                                                         float *__restrict__ gud, const RqsParams<float> &p, int j,
                                                         float (*sub)[24], float *sums) {
    const int t = threadIdx.x;
    if (t < 240) {
        const int k = t % 24, grp = t / 24;
        float s = 0.0f;
        for (int w = grp; w < nparts; w += 10) s += part[(size_t)w * FBR_PART + j * 24 + k];
        sub[grp][k] = s;
    }
    __syncthreads();
    if (t < 24) {
        float s = 0.0f;
#pragma unroll
        for (int grp = 0; grp < 10; ++grp) s += sub[grp][t];
        sums[t] = s;
    }
    __syncthreads();
    if (t < 2) {                              // axis
        const int ax = t;
        const float *raw = (ax ? uh : uw) + j * F_K, *Gk = sums + 7 * ax;
        float m = raw[0];
        for (int k = 1; k < F_K; ++k) m = fmaxf(m, raw[k]);
        float e[F_K], tot = 0.0f;
        for (int k = 0; k < F_K; ++k) { e[k] = expf(raw[k] - m); tot += e[k]; }
        const float f = ax ? (p.top - p.bottom) * p.scale_h : (p.right - p.left) * p.scale_w;
        float C[F_K + 1];
        C[0] = 0.0f;
        for (int k = 0; k < F_K; ++k) C[k + 1] = C[k] + e[k] / tot;
        float base = 0.0f;                    // sum_k G_k C_k
        for (int k = 1; k < F_K; ++k) base += Gk[k - 1] * C[k];
        float tail = 0.0f;                    // sum_{k > i} G_k, built from the top
        float *out = (ax ? guh : guw) + j * F_K;
        for (int i = F_K - 1; i >= 0; --i) {
            out[i] = f * (e[i] / tot) * (tail - base);
            if (i >= 1) tail += Gk[i - 1];    // knot i joins the sum for parameter i - 1
        }
    }
    if (t >= 64 && t < 64 + (F_K - 1)) {
        const int k = t - 64;
        const float r = ud[j * (F_K - 1) + k];
        gud[j * (F_K - 1) + k] = sums[14 + k] * (r > 20.0f ? 1.0f : sigmoid(r));
    }
}

// ---- one launch for a list of reductions (train_bwd.hip) ---------------------------------------------------------------------
struct ReduceJob {
    const float *part;
    float *dW, *db;
    int64_t nW, n, stride;      // n = nW + (db ? M : 0); stride = floats per chunk
    const int *colmap;
    int chunks, N, skip_every, Nout;
    int block0;                 // first block of the job in the launch's grid (256-element groups, one per block)
};
constexpr int RJ_MAX = 14;
struct ReduceJobs {
    ReduceJob j[RJ_MAX];
    int nj;
    int nblocks;                // blocks of the wgrad jobs; the spline job's F_NI blocks follow
    // the batch-shared spline parameters (nf_final_bwd's knot-space partials); fb_part == nullptr: none
    const float *fb_part, *uw, *uh, *ud;
    float *guw, *guh, *gud;
    int fb_nparts;
    RqsParams<float> p;
    // optional: sum of a (B) vector (the log-det cotangent LULinearPermute's diagonal gradient needs) -> *vsum_out, by ONE extra
    // block in a fixed order (16-byte loads when aligned); vsum == nullptr: none
    const float *vsum;
    float *vsum_out;
    int64_t vsum_n;
};

}  // namespace nf
