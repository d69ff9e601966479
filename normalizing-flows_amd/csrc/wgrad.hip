// Weight / bias gradient of a Linear layer over a large batch: dW[M][N] = sum_b dY[b][m] * X[b][n], db[m] = sum_b dY[b][m]
// (the backward of the conditioner's nn.Linear layers, nets/resnet.py:37-50, 92-104, in the training step core.py:87-102).
//
// Shape: M, N <= a few hundred, K = batch = 65 536: a split-K problem.  Both operands of v_mfma_f32_32x32x2_f32 come
// straight from the row-major activations, no LDS, no transposes: for a k-pair (two batch rows) lane (i = lane & 31,
// h = lane >> 5) loads dY[b + h][64 p + 2 i .. +1] (8 bytes) and X[b + h][4 i .. +3] (16 bytes) and feeds 2 x 4 MFMAs:
// the wave's tiles are the INTERLEAVED row sets {64 p + 2 i + s} x column sets {4 i + t}, so one load per operand serves
// eight 32 x 32 tiles (128 accumulator registers).  One wave = one workgroup = 64 rows of M x 128 columns over one
// K-chunk; operands are fetched a buffer set (8 k-pairs, 64 MFMAs) ahead, two waves per SIMD.  Partial tiles go to a caller-owned scratch
// [chunk][M*N + M] and a second kernel sums the chunks in a fixed order: deterministic, unlike atomics.
// N not a multiple of 4 takes the dword path below (NT column tiles, lane i = column 32 t + i).
#include "common.hpp"
#include "fused_common.hpp"
#include "train_reduce.hpp"

namespace nf {

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---- vector path: N % 4 == 0, N <= 128 ------------------------------------------------------------------
// X_RELU: the second operand is relu(X) (the block's activation in front of the layer, applied when the value is consumed:
// the caller keeps only the pre-activation)
template <bool A_VEC, bool X_RELU>   // 8-byte dY loads need even M; template parameters keep the k loop a single basic block
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2)))
wgrad_partial_vec_kernel(const float *__restrict__ dY, const float *__restrict__ X, float *__restrict__ part, int64_t B,
                         int M, int N, int chunk_rows, int want_bias, int64_t zdY, int64_t zX, int64_t zpart) {
    // blockIdx.z = 1: the second problem of a pair launch (same shape; its tensors sit at these element offsets from the first's)
    dY += (int64_t)blockIdx.z * zdY;
    X += (int64_t)blockIdx.z * zX;
    part += (int64_t)blockIdx.z * zpart;
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.y * 64 + 2 * i;      // rows m0, m0 + 1 of dW
    const int n0 = 4 * i;                        // columns n0 .. n0 + 3
    const bool mv0 = m0 < M, mv1 = m0 + 1 < M;
    const bool nv = n0 < N;                      // N % 4 == 0: all four or none
    const int mc = mv1 ? m0 : 0, nc = nv ? n0 : 0;  // out-of-range lanes read valid addresses; their tiles are never stored
    const int64_t b0 = (int64_t)blockIdx.x * chunk_rows;
    int64_t b1 = b0 + chunk_rows;
    if (b1 > B) b1 = B;
    f32x16 acc[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[s][t] = f32x16{0};
    float bs0 = 0.0f, bs1 = 0.0f;
    // Loads are unconditional (row index clamped): a load under a lane-dependent branch forces a wait right behind it and
    // the two-trip prefetch collapses.  Rows past the chunk end are zeroed when consumed instead.
    struct Trip { f32x2 a[4]; f32x4 x[4]; };
    auto load = [&](int64_t b, Trip &tr) {
#ifdef NF_WG_ABL_NOLOAD
        if (B != 12345) return;
#endif
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int64_t r = b + 2 * j + h;
            r = r < B ? r : B - 1;
            if (A_VEC) tr.a[j] = *reinterpret_cast<const f32x2 *>(dY + r * M + mc);
            else tr.a[j] = f32x2{dY[r * M + (mv0 ? m0 : 0)], dY[r * M + mc + (mv1 ? 1 : 0)]};
            tr.x[j] = *reinterpret_cast<const f32x4 *>(X + r * N + nc);
        }
    };
    auto mma = [&](int64_t b, const Trip &tr) {
#ifdef NF_WG_ABL_NOMMA
        bs0 += tr.a[0][0] + tr.x[0][0] + tr.a[1][0] + tr.x[1][0] + tr.a[2][0] + tr.x[2][0] + tr.a[3][0] + tr.x[3][0];
        if (B != 12345) return;
#endif
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool live = b + 2 * j + h < b1;
            const float a0 = live ? tr.a[j][0] : 0.0f, a1 = live ? tr.a[j][1] : 0.0f;
            bs0 += a0;
            bs1 += a1;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float xv = X_RELU ? fmaxf(tr.x[j][t], 0.0f) : tr.x[j][t];
                acc[0][t] = MFMA32(a0, xv, acc[0][t]);
                acc[1][t] = MFMA32(a1, xv, acc[1][t]);
            }
        }
    };
    // Two buffer SETS of two trips each, used alternately: a set is refilled as soon as the loop starts consuming the
    // other one, so every load has 64 MFMAs (~2 us) to land before the (compiler-placed, conservative vmcnt(0)) wait
    // in front of its first use; no register copies.  sched_barrier keeps the machine scheduler from sinking the loads
    // to their uses.  The kernel is held to 256 registers so that two waves share a SIMD and cover each other's waits.
    Trip a0, a1, c0, c1;
    load(b0, a0);
    load(b0 + 8, a1);
    for (int64_t b = b0; b < b1; b += 32) {
        load(b + 16, c0);
        load(b + 24, c1);
        __builtin_amdgcn_sched_barrier(0);
        mma(b, a0);
        mma(b + 8, a1);
        __builtin_amdgcn_sched_barrier(0);
        load(b + 32, a0);
        load(b + 40, a1);
        __builtin_amdgcn_sched_barrier(0);
        mma(b + 16, c0);
        mma(b + 24, c1);
        __builtin_amdgcn_sched_barrier(0);
    }
    // C layout of tile (s, t): row index ri = (reg & 3) + 8 (reg >> 2) + 4 h  ->  m = 64 p + 2 ri + s;  col = lane & 31 -> n = 4 i + t
    float *out = part + (size_t)blockIdx.x * ((size_t)M * N + M);
    if (nv) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mm = blockIdx.y * 64 + 2 * ((r & 3) + 8 * (r >> 2) + 4 * h) + s;
                if (mm < M)
                    *reinterpret_cast<f32x4 *>(out + (size_t)mm * N + n0) =
                        f32x4{acc[s][0][r], acc[s][1][r], acc[s][2][r], acc[s][3][r]};
            }
        }
    }
    if (want_bias) {
        bs0 += __shfl_xor(bs0, 32);
        bs1 += __shfl_xor(bs1, 32);
        if (h == 0) {
            if (mv0) out[(size_t)M * N + m0] = bs0;
            if (mv1) out[(size_t)M * N + m0 + 1] = bs1;
        }
    }
}

// ---- general path: any N <= 128 -----------------------------------------------------------------------
constexpr int WG_MT = 4;  // m-tiles (waves) per workgroup

template <int NT>
__global__ void __launch_bounds__(64 * WG_MT)
wgrad_partial_kernel(const float *__restrict__ dY, const float *__restrict__ X, float *__restrict__ part, int64_t B,
                     int M, int N, int chunk_rows, int want_bias, int x_relu) {
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
                for (int t = 0; t < NT; ++t) {
                    x[j][t] = nval[t] ? pb[t][r * N] : 0.0f;
                    if (x_relu) x[j][t] = fmaxf(x[j][t], 0.0f);
                }
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
            for (int t = 0; t < NT; ++t) {
                float xv = (nval[t] && rv) ? pb[t][r * N] : 0.0f;
                if (x_relu) xv = fmaxf(xv, 0.0f);
                acc[t] = MFMA32(a, xv, acc[t]);
            }
        }
        // C layout: row = (reg & 3) + 8 (reg >> 2) + 4 h (m within the tile), col = lane & 31 (n within the tile)
        float *out = part + (size_t)blockIdx.x * ((size_t)M * N + M);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = t * 32 + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mm = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (mm < M && n < N) out[(size_t)mm * N + n] = acc[t][r];
            }
        }
        if (want_bias) {
            bsum += __shfl_xor(bsum, 32);
            if (h == 0 && mval) out[(size_t)M * N + m] = bsum;
        }
    }
}

// ---- workgroup-tiled path: N in [96, 128], N % 4 == 0 (the 128-wide layers of the conditioner) ------------------------------
// The wave-private kernels above take both MFMA operands straight from global memory: 1.5 KB per 8 MFMAs and wave, every
// m-group re-reading X from L2, and nothing but its own two buffer sets hides a wave's load latency -- 0.43 of the fp32 MFMA
// peak.  Here a 4-wave workgroup owns a 128 (M) x 128 (N) tile: a K-step of 32 batch rows of dY and X is loaded ONCE per
// workgroup (coalesced 16-byte loads, the next step's into registers while the current step's 64 MFMAs per wave run from
// LDS), each wave computes a 64 x 64 quadrant (4 accumulators) with ds_read_b32 operands -- half the operand traffic per MFMA
// and a workgroup-wide prefetch.  Same partial-tile format and reduce kernel.
constexpr int W2_KS = 32;     // batch rows per LDS tile
constexpr int W2_T = 128;     // tile edge (M and N)

__global__ void __launch_bounds__(256, 2)
wgrad_tile_kernel(const float *__restrict__ dY, const float *__restrict__ X, float *__restrict__ part, int64_t B, int M, int N,
                  int chunk_rows, int want_bias, int x_relu, int64_t zdY, int64_t zX, int64_t zpart) {
    dY += (int64_t)blockIdx.z * zdY;
    X += (int64_t)blockIdx.z * zX;
    part += (int64_t)blockIdx.z * zpart;
    __shared__ __attribute__((aligned(16))) float As[2][W2_KS][W2_T];
    __shared__ __attribute__((aligned(16))) float Bs[2][W2_KS][W2_T];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, i = lane & 31, h = lane >> 5;
    const int wm = wid >> 1, wn = wid & 1;
    const int m0 = blockIdx.y * W2_T;
    const int64_t b0 = (int64_t)blockIdx.x * chunk_rows;
    int64_t b1 = b0 + chunk_rows;
    if (b1 > B) b1 = B;
    // loader mapping: thread t covers columns [4 (t & 31), +4) of rows (t >> 5) + 8 j, j = 0..3
    const int lc = (tid & 31) * 4, lr = tid >> 5;
    const bool a_ok = m0 + lc < M, x_ok = lc < N;     // M, N multiples of 4
    f32x4 ra[4], rx[4];
    auto fetch = [&](int64_t b) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t r = b + lr + 8 * j;
            const int64_t rc = r < b1 ? r : b1 - 1;            // clamped address, zeroed when stored
            ra[j] = a_ok ? *reinterpret_cast<const f32x4 *>(dY + rc * M + m0 + lc) : f32x4{0.f, 0.f, 0.f, 0.f};
            rx[j] = x_ok ? *reinterpret_cast<const f32x4 *>(X + rc * N + lc) : f32x4{0.f, 0.f, 0.f, 0.f};
            if (r >= b1) { ra[j] = f32x4{0.f, 0.f, 0.f, 0.f}; rx[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 xv = rx[j];
            if (x_relu) { xv[0] = fmaxf(xv[0], 0.f); xv[1] = fmaxf(xv[1], 0.f); xv[2] = fmaxf(xv[2], 0.f); xv[3] = fmaxf(xv[3], 0.f); }
            *reinterpret_cast<f32x4 *>(&As[buf][lr + 8 * j][lc]) = ra[j];
            *reinterpret_cast<f32x4 *>(&Bs[buf][lr + 8 * j][lc]) = xv;
        }
    };
    f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
    float bs0 = 0.0f, bs1 = 0.0f;
    fetch(b0);
    commit(0);
    __syncthreads();
    int buf = 0;
    for (int64_t b = b0; b < b1; b += W2_KS) {
        const bool more = b + W2_KS < b1;
        if (more) fetch(b + W2_KS);
        const float *ap = &As[buf][h][wm * 64 + i], *bp = &Bs[buf][h][wn * 64 + i];
#pragma unroll
        for (int kp = 0; kp < W2_KS / 2; ++kp) {
            const float a0 = ap[kp * 2 * W2_T], a1 = ap[kp * 2 * W2_T + 32];
            const float x0 = bp[kp * 2 * W2_T], x1 = bp[kp * 2 * W2_T + 32];
            bs0 += a0;
            bs1 += a1;
            acc00 = MFMA32(a0, x0, acc00);
            acc01 = MFMA32(a0, x1, acc01);
            acc10 = MFMA32(a1, x0, acc10);
            acc11 = MFMA32(a1, x1, acc11);
        }
        if (more) commit(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    // C layout: row (m within the 32-subtile) = (reg & 3) + 8 (reg >> 2) + 4 h, col (n within the subtile) = lane & 31
    float *out = part + (size_t)blockIdx.x * ((size_t)M * N + M);
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
#pragma unroll
        for (int t_ = 0; t_ < 2; ++t_) {
            const f32x16 &acc = s_ == 0 ? (t_ == 0 ? acc00 : acc01) : (t_ == 0 ? acc10 : acc11);
            const int n = wn * 64 + 32 * t_ + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mm = m0 + wm * 64 + 32 * s_ + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (mm < M && n < N) out[(size_t)mm * N + n] = acc[r];
            }
        }
    }
    if (want_bias && wn == 0) {
        bs0 += __shfl_xor(bs0, 32);
        bs1 += __shfl_xor(bs1, 32);
        if (h == 0) {
            const int ma = m0 + wm * 64 + i, mb = ma + 32;
            if (ma < M) out[(size_t)M * N + ma] = bs0;
            if (mb < M) out[(size_t)M * N + mb] = bs1;
        }
    }
}

// ---- the same workgroup tile with both operands streamed by LDS-DMA through a 4-slot ring (N == 128, M % 128 == 0,
// B % 16 == 0) ----------------------------------------------------------------------------------------------------------
// The tile kernel above stages the next 32 rows through registers: one K-step (1.9 us of MFMAs) of look-ahead, which HBM
// latency under load exceeds, plus the ds_write pass and two barriers per step.  Here a K-step is 16 rows (8 KB of dY + 8 KB
// of X per workgroup = 4 DMA instructions per wave, no registers), NR slots = NR - 1 steps of look-ahead (0.95 us each), one
// barrier per step; `s_waitcnt vmcnt(4 (NR - 2))` = the younger steps already requested may stay outstanding (loads retire
// in order; one more is requested right after the barrier), drained in the tail.  ReLU of the second operand is applied on the LDS read.
// (Round 3, measured and dropped: all M / 128 tiles of a row chunk on ONE XCD, so that the chunk of X comes from HBM once instead
// of once per tile -- 147 -> 152 us at 768 x 128: the kernel is bound by its MFMA issue (busy 0.62), not by the 390 MB it reads;
// 768 instead of 512 workgroups: 143 us.)
#ifndef NF_W3_NR
#define NF_W3_NR 3      // measured 4 / 3 / 2 slots: 145 / 139 / 143 us at 768 x 128
#endif
#ifndef NF_W3_OCC
#define NF_W3_OCC 3
#endif
#ifndef NF_W3_HW
#define NF_W3_HW 4      // helper waves: they issue the ring's requests (a vector-memory instruction costs its wave 100-250 issue cycles:
#endif                  // 4 per step against the step's 32 MFMAs = 2048 cycles when the MFMA waves issued them)
constexpr int W3_KS = 16, W3_NR = NF_W3_NR, W3_HW = NF_W3_HW, W3_NT = 64 * (4 + W3_HW);
__global__ void __launch_bounds__(W3_NT, W3_HW ? 4 : NF_W3_OCC)      // second argument = waves per SIMD: two 8-wave workgroups per CU
wgrad_ring_kernel(const float *__restrict__ dY, const float *__restrict__ X, float *__restrict__ part, int64_t B, int M,
                  int chunk_rows, int want_bias, int x_relu, int64_t zdY, int64_t zX, int64_t zpart) {
    typedef __attribute__((address_space(3))) void *lds_ptr;
    constexpr int N = 128;
    dY += (int64_t)blockIdx.z * zdY;
    X += (int64_t)blockIdx.z * zX;
    part += (int64_t)blockIdx.z * zpart;
    __shared__ __attribute__((aligned(16))) float ring[W3_NR][2][W3_KS][W2_T];     // [slot][dY | X][row][128] = 64 KB
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int m0 = blockIdx.y * W2_T;
    const int64_t b0 = (int64_t)blockIdx.x * chunk_rows;
    int64_t b1 = b0 + chunk_rows;
    if (b1 > B) b1 = B;
    const int nsteps = (int)((b1 - b0) / W3_KS);       // chunk_rows and B are multiples of 16
    // a requesting wave's 4 DMA instructions of a step: rows 4 dw + {0,1} and + {2,3} of each operand (64 lanes x 16 B = two 128-float rows)
    const bool mfma_wave = wid < 4, dma_wave = W3_HW ? !mfma_wave : true;
    const int dw = W3_HW ? (wid - 4) : wid;
    static_assert(W3_HW == 0 || W3_HW == 4, "four requesting waves");
    const int lrow = lane >> 5, lcol = (lane & 31) * 4;
    auto issue = [&](int s) {
        const int64_t r0 = b0 + (int64_t)s * W3_KS + 4 * dw + lrow;
        float *slotA = &ring[s % W3_NR][0][4 * dw][0], *slotB = &ring[s % W3_NR][1][4 * dw][0];
        __builtin_amdgcn_global_load_lds(dY + r0 * M + m0 + lcol, (lds_ptr)slotA, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(dY + (r0 + 2) * M + m0 + lcol, (lds_ptr)(slotA + 2 * W2_T), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(X + r0 * N + lcol, (lds_ptr)slotB, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(X + (r0 + 2) * N + lcol, (lds_ptr)(slotB + 2 * W2_T), 16, 0, 0);
    };
    if (W3_HW && !mfma_wave) {
        // helper waves: requests and barriers only (the same barrier sequence as the MFMA waves below)
        for (int s = 0; s < W3_NR - 1 && s < nsteps; ++s) issue(s);
        for (int s = 0; s < nsteps; ++s) {
            if (s + W3_NR - 1 <= nsteps) NF_WAIT_VMCNT(4 * (W3_NR - 2));
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (s + W3_NR - 1 < nsteps) issue(s + W3_NR - 1);
        }
        return;
    }
    f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
    float bs0 = 0.0f, bs1 = 0.0f;
    if (!W3_HW)
        for (int s = 0; s < W3_NR - 1 && s < nsteps; ++s) issue(s);
    for (int s = 0; s < nsteps; ++s) {
        // step s landed (the requesting waves' pieces) once at most the younger steps' 4 (NR - 1) instructions are outstanding; in
        // the tail fewer are younger: drain
        if (!W3_HW) {
            if (s + W3_NR - 1 <= nsteps) NF_WAIT_VMCNT(4 * (W3_NR - 2));
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every piece of step s landed; every wave is done with slot (s - 1) % NR
        if (!W3_HW && s + W3_NR - 1 < nsteps) issue(s + W3_NR - 1);
        const float *ap = &ring[s % W3_NR][0][h][wm * 64 + i], *bp = &ring[s % W3_NR][1][h][wn * 64 + i];
        // (all 32 operand reads of the step hoisted in front of its 32 MFMAs -- the compiler reads one k-pair, waits, multiplies --
        // measured in round 3: 143 vs 141 us, no gain; helper waves 146 -> 141 us)
#pragma unroll
        for (int kp = 0; kp < W3_KS / 2; ++kp) {
            const float a0 = ap[kp * 2 * W2_T], a1 = ap[kp * 2 * W2_T + 32];
            float x0 = bp[kp * 2 * W2_T], x1 = bp[kp * 2 * W2_T + 32];
            if (x_relu) { x0 = fmaxf(x0, 0.0f); x1 = fmaxf(x1, 0.0f); }
            bs0 += a0;
            bs1 += a1;
            acc00 = MFMA32(a0, x0, acc00);
            acc01 = MFMA32(a0, x1, acc01);
            acc10 = MFMA32(a1, x0, acc10);
            acc11 = MFMA32(a1, x1, acc11);
        }
    }
    float *out = part + (size_t)blockIdx.x * ((size_t)M * N + M);
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
#pragma unroll
        for (int t_ = 0; t_ < 2; ++t_) {
            const f32x16 &acc = s_ == 0 ? (t_ == 0 ? acc00 : acc01) : (t_ == 0 ? acc10 : acc11);
            const int n = wn * 64 + 32 * t_ + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mm = m0 + wm * 64 + 32 * s_ + (r & 3) + 8 * (r >> 2) + 4 * h;
                out[(size_t)mm * N + n] = acc[r];
            }
        }
    }
    if (want_bias && wn == 0) {
        bs0 += __shfl_xor(bs0, 32);
        bs1 += __shfl_xor(bs1, 32);
        if (h == 0) {
            const int ma = m0 + wm * 64 + i;
            out[(size_t)M * N + ma] = bs0;
            out[(size_t)M * N + ma + 32] = bs1;
        }
    }
}

static int wgrad_tile_chunk_rows(int64_t B, int M) {
    const int mt = (M + W2_T - 1) / W2_T;
#ifndef NF_W2_SLOTS
#define NF_W2_SLOTS 512
#endif
    int64_t want = NF_W2_SLOTS / mt;               // at most 512 workgroups (two per CU): one more would cost a second round
    if (want < 1) want = 1;
    int64_t rows = (B + want - 1) / want;
    rows = (rows + W2_KS - 1) / W2_KS * W2_KS;
    if (rows < 4 * W2_KS) rows = 4 * W2_KS;
    return (int)rows;
}

// out (dW then db) (=|+=) sum over chunks of part[c][e] in a fixed order; e < nW goes to dW, the rest to db.
// Block = 64 elements x RL chunk lanes (lane q sums chunks q, q + RL, ... with four loads in flight), combined through
// LDS in lane order: deterministic, and short dependent chains (the kernel is pure load latency).  The group routine lives in
// train_reduce.hpp: the one-launch-per-layer reduction of the training step (train_bwd.hip) runs the same code.
__global__ void __launch_bounds__(64 * RL)
wgrad_reduce_kernel(const float *__restrict__ part, float *__restrict__ dW, float *__restrict__ db, int64_t nW, int64_t n,
                    int64_t stride, int chunks, int accumulate, int N, int skip_every, int64_t zpart, int64_t zdW,
                    int64_t zdb, const int *__restrict__ colmap, int Nout) {
    part += (int64_t)blockIdx.y * zpart;      // blockIdx.y = 1: the second problem of a pair launch
    dW += (int64_t)blockIdx.y * zdW;
    if (db) db += (int64_t)blockIdx.y * zdb;
    __shared__ float sm[RL][64];
    for (int64_t e0 = (int64_t)blockIdx.x * 64; e0 < n; e0 += (int64_t)gridDim.x * 64)
        wgrad_reduce_group(part, dW, db, nW, n, stride, chunks, accumulate, N, skip_every, colmap, Nout, e0, sm);
}

// The reduction as a host call for the kernels of other translation units (resblock_bwd.hip): part = [problem][chunk][nW + M].
int wgrad_reduce_launch(const float *part, float *dW, float *db, int64_t nW, int M, int chunks, int N, int np, int64_t zpart,
                        int64_t zdW, int64_t zdb, const int *colmap, int Nout, hipStream_t st) {
    const int64_t n = nW + (db ? M : 0);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid_for(n, 64), np), dim3(64 * RL), 0, st, part, dW, db, nW, n, nW + M, chunks, 0,
                       N, 0, zpart, zdW, zdb, colmap, Nout);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

static int wgrad_chunk_rows(int64_t B, int M, bool vec) {
    // vector path: ~2048 waves in flight (two per SIMD), chunk rows a multiple of its 32-row loop step;
    // general path: ~1024 waves.  At least 64 rows (fewer, longer chunks = less partial traffic).
    const int mgroups = vec ? (M + 63) / 64 : (M + 31) / 32;
    // small outputs: the partial-tile traffic (chunks x M x N) outweighs the second wave per SIMD
    int64_t want = (((vec && M > 256) ? 2048 : 1024) + mgroups - 1) / mgroups;
    int64_t rows = (B + want - 1) / want;
    rows = (rows + 31) / 32 * 32;
    if (rows < 64) rows = 64;
    return (int)rows;
}

}  // namespace nf

static inline bool wgrad_use_ring(int64_t B, int M, int N) {
#ifdef NF_WGRAD_NO_RING
    return false;
#else
#ifdef NF_WGRAD_RING_M128
    const int mmin = 128;
#else
    const int mmin = 256;
#endif
    return N == 128 && M % 128 == 0 && M >= mmin && B % 16 == 0;
#endif
}

static inline bool wgrad_use_tile(int M, int N) {
#ifdef NF_WGRAD_NO_TILE
    return false;
#else
    return N >= 96 && N <= 128 && N % 4 == 0 && M % 4 == 0 && M >= 256;   // M = 128: 39 us vs 35 us wave-private (measured)
#endif
}

extern "C" int64_t nf_linear_wgrad_scratch_floats(int64_t B, int M, int N) {
    if (B < 0 || M < 1 || N < 1) return NF_EINVAL;
    if (wgrad_use_tile(M, N) || wgrad_use_ring(B, M, N)) {
        const int rows_t = nf::wgrad_tile_chunk_rows(B, M);
        return ((B + rows_t - 1) / rows_t) * ((int64_t)M * N + M);
    }
    const int rows = nf::wgrad_chunk_rows(B, M, N % 4 == 0);
    const int64_t chunks = (B + rows - 1) / rows;
    return chunks * ((int64_t)M * N + M);
}

extern "C" int nf_linear_wgrad_skip(const void *dY, const void *X, void *dW, void *db, void *scratch, int64_t B, int M, int N,
                                    int accumulate, int relu_x, int skip_every, nf_stream_t stream);
extern "C" int nf_linear_wgrad_act(const void *dY, const void *X, void *dW, void *db, void *scratch, int64_t B, int M, int N,
                                   int accumulate, int relu_x, nf_stream_t stream) {
    return nf_linear_wgrad_skip(dY, X, dW, db, scratch, B, M, N, accumulate, relu_x, 0, stream);
}
extern "C" int nf_linear_wgrad(const void *dY, const void *X, void *dW, void *db, void *scratch, int64_t B, int M, int N,
                               int accumulate, nf_stream_t stream) {
    return nf_linear_wgrad_skip(dY, X, dW, db, scratch, B, M, N, accumulate, 0, 0, stream);
}

static int wgrad_impl(const void *dY, const void *X, void *dW, void *db, void *scratch, int64_t B, int M, int N, int accumulate,
                      int relu_x, int skip_every, nf_stream_t stream, const void *dY1, const void *X1, void *dW1, void *db1,
                      bool reduce = true, int want_bias_partials = -1);

// The partial launch of nf_linear_wgrad[_act] alone: scratch = [chunk][M * N + M] (nf_linear_wgrad_chunks(B, M, N) chunks), summed
// later by nf_wgrad_reduce_jobs / nf_coupling_train_bwd's one reduction launch per layer.
extern "C" int nf_linear_wgrad_chunks(int64_t B, int M, int N) {
    if (B < 1 || M < 1 || N < 1) return NF_EINVAL;
    const int64_t per = (int64_t)M * N + M, tot = nf_linear_wgrad_scratch_floats(B, M, N);
    return tot < 0 ? (int)tot : (int)(tot / per);
}
extern "C" int nf_linear_wgrad_partials(const void *dY, const void *X, void *scratch, int64_t B, int M, int N, int relu_x,
                                        int want_bias, nf_stream_t stream) {
    if (want_bias != 0 && want_bias != 1) return NF_EINVAL;
    return wgrad_impl(dY, X, scratch /* non-null stand-in, never written */, nullptr, scratch, B, M, N, 0, relu_x, 0, stream, nullptr,
                      nullptr, nullptr, nullptr, false, want_bias);
}

extern "C" int nf_linear_wgrad_skip(const void *dY, const void *X, void *dW, void *db, void *scratch, int64_t B, int M, int N,
                                    int accumulate, int relu_x, int skip_every, nf_stream_t stream) {
    return wgrad_impl(dY, X, dW, db, scratch, B, M, N, accumulate, relu_x, skip_every, stream, nullptr, nullptr, nullptr, nullptr);
}

// Two problems of the same shape in ONE partial launch and ONE reduction (a residual block's two weight gradients arrive
// together: half the launches, and the second problem's workgroups fill the first's ramp and tail).  scratch: 2 x
// nf_linear_wgrad_scratch_floats(B, M, N); db0 / db1 both given or both NULL.
extern "C" int nf_linear_wgrad_pair(const void *dY0, const void *X0, void *dW0, void *db0, const void *dY1, const void *X1,
                                    void *dW1, void *db1, void *scratch, int64_t B, int M, int N, int accumulate, int relu_x,
                                    nf_stream_t stream) {
    if (!dY1 || !X1 || !dW1 || ((db0 == nullptr) != (db1 == nullptr))) return NF_EFAULT;
    if ((((uintptr_t)dY0 ^ (uintptr_t)dY1) | ((uintptr_t)X0 ^ (uintptr_t)X1) | ((uintptr_t)dW0 ^ (uintptr_t)dW1) |
         ((uintptr_t)db0 ^ (uintptr_t)db1)) & 3) return NF_EINVAL;
    return wgrad_impl(dY0, X0, dW0, db0, scratch, B, M, N, accumulate, relu_x, 0, stream, dY1, X1, dW1, db1);
}

static int wgrad_impl(const void *dY, const void *X, void *dW, void *db, void *scratch, int64_t B, int M, int N, int accumulate,
                      int relu_x, int skip_every, nf_stream_t stream, const void *dY1, const void *X1, void *dW1, void *db1,
                      bool reduce, int want_bias_partials) {
    if (B < 1 || M < 1 || N < 1 || (accumulate != 0 && accumulate != 1) || (relu_x != 0 && relu_x != 1)) return NF_EINVAL;
    if (skip_every < 0 || skip_every == 1 || (skip_every && M % skip_every != 0)) return NF_EINVAL;
    if (N > 128) return NF_ENOTSUP;  // four 32-column tiles of accumulators per wave
    if (!dY || !X || !dW || !scratch) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    const bool vec = N % 4 == 0;
    const bool ring = wgrad_use_ring(B, M, N) && ((((uintptr_t)dY | (uintptr_t)X | (uintptr_t)dY1 | (uintptr_t)X1) & 15) == 0);
    const bool tile = ring || wgrad_use_tile(M, N);
    const int rows = tile ? nf::wgrad_tile_chunk_rows(B, M) : nf::wgrad_chunk_rows(B, M, vec);
    const int chunks = (int)((B + rows - 1) / rows);
    float *part = (float *)scratch;
    const int want_bias = want_bias_partials >= 0 ? want_bias_partials : (db ? 1 : 0);
    const int np = dY1 ? 2 : 1;
    if (np == 2 && !tile && !vec) return NF_ENOTSUP;
    const int64_t single = (int64_t)chunks * ((int64_t)M * N + M);      // = nf_linear_wgrad_scratch_floats(B, M, N)
    const int64_t zdY = np == 2 ? (const float *)dY1 - (const float *)dY : 0, zX = np == 2 ? (const float *)X1 - (const float *)X : 0;
    const int64_t zdW = np == 2 ? (float *)dW1 - (float *)dW : 0, zdb = (np == 2 && db) ? (float *)db1 - (float *)db : 0;
    const int64_t zpart = np == 2 ? single : 0;
    if (ring) {
        hipLaunchKernelGGL(nf::wgrad_ring_kernel, dim3(chunks, M / nf::W2_T, np), dim3(nf::W3_NT), 0, st, (const float *)dY,
                           (const float *)X, part, B, M, rows, want_bias, relu_x, zdY, zX, zpart);
    } else if (tile) {
        hipLaunchKernelGGL(nf::wgrad_tile_kernel, dim3(chunks, (M + nf::W2_T - 1) / nf::W2_T, np), dim3(256), 0, st,
                           (const float *)dY, (const float *)X, part, B, M, N, rows, want_bias, relu_x, zdY, zX, zpart);
    } else if (vec) {
#define NF_WGRAD_VEC(AV, XR)                                                                                          \
    hipLaunchKernelGGL((nf::wgrad_partial_vec_kernel<AV, XR>), dim3(chunks, (M + 63) / 64, np), dim3(64), 0, st,      \
                       (const float *)dY, (const float *)X, part, B, M, N, rows, want_bias, zdY, zX, zpart)
        if (M % 2 == 0) { if (relu_x) NF_WGRAD_VEC(true, true); else NF_WGRAD_VEC(true, false); }
        else { if (relu_x) NF_WGRAD_VEC(false, true); else NF_WGRAD_VEC(false, false); }
#undef NF_WGRAD_VEC
    } else {
        const int mtiles = (M + 31) / 32;
        dim3 grid(chunks, (mtiles + nf::WG_MT - 1) / nf::WG_MT);
        const int nt = (N + 31) / 32;
#define NF_WGRAD_LAUNCH(NT)                                                                                         \
    hipLaunchKernelGGL(nf::wgrad_partial_kernel<NT>, grid, dim3(64 * nf::WG_MT), 0, st, (const float *)dY,           \
                       (const float *)X, part, B, M, N, rows, want_bias, relu_x)
        if (nt == 1) NF_WGRAD_LAUNCH(1);
        else if (nt == 2) NF_WGRAD_LAUNCH(2);
        else if (nt == 3) NF_WGRAD_LAUNCH(3);
        else NF_WGRAD_LAUNCH(4);
#undef NF_WGRAD_LAUNCH
    }
    NF_CHECK_LAUNCH();
    if (!reduce) return NF_OK;
    const int64_t nW = (int64_t)M * N, n = nW + (db ? M : 0);
    hipLaunchKernelGGL(nf::wgrad_reduce_kernel, dim3(nf::grid_for(n, 64), np), dim3(64 * nf::RL), 0, st, part, (float *)dW,
                       (float *)db, nW, n, nW + M, chunks, accumulate, N, skip_every, zpart, zdW, zdb, (const int *)nullptr, 0);
    NF_CHECK_LAUNCH();
    return NF_OK;
}
