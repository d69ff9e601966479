// Backward of one residual block of the conditioner (nets/resnet.py:37-50: y = h + W2 relu(W1 relu(h) + b1) + b2, hidden = 128)
// over a large batch as ONE pass over the rows: both input-gradient products AND both weight gradients of the block, and
// (INIT) the initial Linear layer's input / weight gradients behind it (nets/resnet.py:92-104).
//
//   gt    = (gh W2) * [t > 0]                      t = the block's pre-activation (saved by the forward)
//   gh_in = gh + (gt W1) * [h_in > 0]              h_in = the block's input
//   dW2   = gh^T relu(t),   db2 = colsum(gh);      dW1 = gt^T relu(h_in),   db1 = colsum(gt)
//   INIT: gx += gh_in Wfull^T, dW0 = gh_in^T x,  db0 = colsum(gh_in)     (Wfull (D, H) = the initial weight transposed on full rows,
//                                                                         zero rows at the transformed features)
//
// Separate kernels read gh / t / h_in twice and round-trip gt through HBM (302 MB per block at B = 65 536); here each row is
// read once, gt never leaves the CU (134 MB), and 8.6 GFLOP sit behind one launch ramp instead of three.
//
// Workgroup = 4 waves, one per CU, persistent over 64-row tiles.  The tile's gh / t / h_in arrive by LDS-DMA (16 B per lane,
// no registers) into row-major LDS tiles of pitch 132 floats (one 16-byte pad slot per row, filled with a duplicate load):
//   * the input-gradient products contract over features: A[m = row][k = feature] is read with ds_read_b128 (lane = row,
//     4 consecutive features = 4 k-steps in a permuted contraction order; pitch 132 puts the 16 rows of a lane group on
//     16 distinct 4-bank slots), B = the weight slice W[:, 32 wave .. +32] resident in 64 registers per weight;
//   * the weight gradients contract over rows: A[m = feature][k = row] and B[k = row][n = feature] are row-wise ds_read_b32
//     (32 consecutive floats, conflict-free at any pitch); each wave owns a 64 x 64 quadrant of the 128 x 128 output
//     (64 accumulator registers per weight) for the whole launch and writes it once, as a partial tile summed over the
//     workgroups by nf::wgrad_reduce_kernel in a fixed order (deterministic).
// A tile's loads are requested half a tile (>= 6 us) before they are consumed -- gh / t of the next tile after the barrier
// that retires them, h_in after the last reader -- so every s_waitcnt vmcnt(0) finds its loads landed; the epilogue's global
// stores are issued AFTER the wait + barrier they would otherwise sit in.
#include "common.hpp"
#include "fused_common.hpp"

namespace nf {

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

int wgrad_reduce_launch(const float *part, float *dW, float *db, int64_t nW, int M, int chunks, int N, int np, int64_t zpart,
                        int64_t zdW, int64_t zdb, const int *colmap, int Nout, hipStream_t st);       // wgrad.hip

constexpr int BB_R = 64, BB_P = 132, BB_TILE = BB_R * BB_P, BB_H = 128, BB_D = 64;
constexpr int BB_SLOTS = BB_R * (BB_P / 4);          // 16-byte slots of a tile = 2112 = 33 DMA instructions of 64 lanes
constexpr int BB_NI = BB_SLOTS / 64;                 // 33

#ifdef NF_BB_TRACE
static unsigned long long *g_bb_trace = nullptr;
extern "C" void nf_resblock_bwd_debug_trace(void *buf) { g_bb_trace = (unsigned long long *)buf; }
#define BB_T(i) do { if (a.trace && blockIdx.x == 0 && threadIdx.x == 0) a.trace[tcount * 12 + (i)] = wall_clock64(); } while (0)
#else
#define BB_T(i) do {} while (0)
#endif

struct BlockBwdArgs {
#ifdef NF_BB_TRACE
    unsigned long long *trace;
#endif
    const float *gh, *t, *hin, *W1, *W2;
    float *gh_in;          // (B, 128); not written by the INIT variant
    float *part;           // [2][grid][128 * 128 + 128]: (dW2, db2) then (dW1, db1)
    const float *x;        // INIT: (B, 64)
    const float *wfull;    // INIT: (64, 128), the initial weight transposed
    float *gx;             // INIT: (B, 64), accumulated into
    float *part0;          // INIT: [grid][128 * 64 + 128]
    int64_t B;
};

#ifndef NF_BB_EPI
#define NF_BB_EPI 0      // 1: epilogues in the weight-gradient loops' shadow except in the INIT variant (registers); 2: everywhere; 0: nowhere
#endif
#define BB_BARRIER_ALL() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define BB_BARRIER_LDS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <bool INIT>
__global__ void __launch_bounds__(256, 1)
resblock_bwd_kernel(BlockBwdArgs a) {
    typedef __attribute__((address_space(3))) void *lds_ptr;
    extern __shared__ __attribute__((aligned(16))) float smem_bb[];
    float *Gt = smem_bb, *Tt = Gt + BB_TILE, *Ht = Tt + BB_TILE, *Dt = Ht + BB_TILE, *Xt = Dt + BB_TILE;
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, hh = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int om = wid >> 1, in_ = wid & 1;          // this wave's 64 x 64 quadrant of dW2 / dW1; its (row half, column block) of gx
    const int grid = gridDim.x;
    const int64_t ntiles = a.B / BB_R;
    constexpr bool EPI = NF_BB_EPI == 2 || (NF_BB_EPI == 1 && !INIT);

    // DMA slot map: instruction k of a tile fills LDS floats [256 k, 256 k + 256); lane's slot s = 64 k + lane is column
    // group c = s % 33 of row s / 33 (c = 32: the pad slot, loaded with the row's last group again)
    // The 9 per-lane element offsets are loop-invariant 32-bit registers; the tile's base pointer is uniform (SGPR pair): the
    // DMA takes them as saddr + voffset, no per-instruction address arithmetic.
    unsigned goff[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) {
        const int s = 64 * (wid + 4 * q) + lane, row = s / 33, c = s - 33 * row;
        goff[q] = (unsigned)((row < BB_R ? row : 0) * BB_H + 4 * (c < 32 ? c : 31));
    }
    auto issue = [&](const float *src, float *tile) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
            __builtin_amdgcn_global_load_lds(src + goff[q], (lds_ptr)(tile + 256 * (wid + 4 * q)), 16, 0, 0);
        if (wid == 0) __builtin_amdgcn_global_load_lds(src + goff[8], (lds_ptr)(tile + 256 * 32), 16, 0, 0);   // instruction 33
    };
    // One DMA instruction costs its wave ~300 cycles of issue (measured: the product behind a tile's 18 requests ran 2.3 us
    // longer); spread one per four MFMAs through the input-gradient products, that time sits in the MFMA pipe's shadow.
    auto issue_one = [&](const float *src, float *tile, int q) {
        if (q < 8) __builtin_amdgcn_global_load_lds(src + goff[q], (lds_ptr)(tile + 256 * (wid + 4 * q)), 16, 0, 0);
        else if (wid == 0) __builtin_amdgcn_global_load_lds(src + goff[8], (lds_ptr)(tile + 256 * 32), 16, 0, 0);
    };
    auto issue_x = [&](const float *src) {          // 64 rows x 64 floats, contiguous: 16 instructions
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_global_load_lds(src + 256 * (wid + 4 * q) + 4 * lane, (lds_ptr)(Xt + 256 * (wid + 4 * q)), 16, 0, 0);
    };

    int64_t tile = blockIdx.x;
    if (tile < ntiles) {
        issue(a.gh + tile * (BB_R * BB_H), Gt);
        issue(a.t + tile * (BB_R * BB_H), Tt);
        issue(a.hin + tile * (BB_R * BB_H), Ht);
        if (INIT) {
            issue_x(a.x + tile * (BB_R * BB_D));
        }
    }
    // weight slices: B operand of the input-gradient products, k = 8 Q + 4 hh + s (the order ds_read_b128 delivers A in)
    float W2r[64], W1r[64];
#pragma unroll
    for (int Q = 0; Q < 16; ++Q)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = 8 * Q + 4 * hh + s;
            W2r[4 * Q + s] = a.W2[k * BB_H + 32 * wid + i];
            W1r[4 * Q + s] = a.W1[k * BB_H + 32 * wid + i];
        }
    f32x16 acc2[4], acc1[4], acc0[2];
#pragma unroll
    for (int q = 0; q < 4; ++q) { acc2[q] = f32x16{0}; acc1[q] = f32x16{0}; }
    acc0[0] = f32x16{0}; acc0[1] = f32x16{0};
    float bs2a = 0.f, bs2b = 0.f, bs1a = 0.f, bs1b = 0.f, bs0 = 0.f;
    int tcount = 0;
    BB_T(7);
#ifdef NF_BB_TRACE
    if (a.trace && blockIdx.x == 0 && threadIdx.x == 0) a.trace[6] = clock64();
#endif
    BB_BARRIER_ALL();
    tcount = 1;

    for (; tile < ntiles; tile += grid, ++tcount) {
        const bool more = tile + grid < ntiles, first = tile == (int64_t)blockIdx.x;     // the first tile's h_in: requested above
        BB_T(0);
        // per-tile lane index: with the loop-invariant LDS / global addresses hoisted out of the tile loop (a hundred-odd
        // registers) the INIT variant spills; recomputing them costs a few VALU instructions per 512 MFMAs
        int l_ = lane;
        asm volatile("" : "+v"(l_));
        const int i = l_ & 31, hh = l_ >> 5;
        // ---- gt = (gh W2) [t > 0]: the products ----
        float outv[32];
        f32x16 C[2];
#pragma unroll
        for (int rh = 0; rh < 2; ++rh) {
            C[rh] = f32x16{0};
            const float *ap = Gt + (32 * rh + i) * BB_P + 4 * hh;
#pragma unroll
            for (int Q = 0; Q < 16; ++Q) {
                const f32x4 av = *reinterpret_cast<const f32x4 *>(ap + 8 * Q);
#pragma unroll
                for (int s = 0; s < 4; ++s) C[rh] = MFMA32(av[s], W2r[4 * Q + s], C[rh]);
                if (rh == 0 && Q < 9 && !first) issue_one(a.hin + tile * (BB_R * BB_H), Ht, Q);     // this tile's h_in (Ht: free since E)
            }
        }
        if (!EPI) {
#pragma unroll
            for (int kp = 0; kp < 32; ++kp) {
                const int rh = kp >> 4, r = kp & 15, idx = (32 * rh + 8 * (r >> 2) + 4 * hh + (r & 3)) * BB_P + 32 * wid + i;
                Dt[idx] = Tt[idx] > 0.0f ? C[rh][r] : 0.0f;
                outv[kp] = Gt[idx];
            }
        }
        BB_T(1);
        // ---- dW2 += gh^T relu(t), db2 += colsum(gh); (EPI) in the MFMAs' shadow, one element per k-pair, the epilogue of the
        //      product above: gt -> Dt (masked by t > 0), the residual term gh -> registers.  Operands are fetched one k-pair ahead.
#define BB_WGRAD_BODY(ACC, BSA, BSB, EPILOGUE)                                                                 \
            {                                                                                                    \
                float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;                                                \
                if (kp + 1 < BB_R / 2) {                                                                         \
                    na0 = ap[(kp + 1) * 2 * BB_P]; na1 = ap[(kp + 1) * 2 * BB_P + 32];                           \
                    nb0 = bp[(kp + 1) * 2 * BB_P]; nb1 = bp[(kp + 1) * 2 * BB_P + 32];                           \
                }                                                                                                \
                EPILOGUE                                                                                         \
                b0 = fmaxf(b0, 0.0f);                                                                            \
                b1 = fmaxf(b1, 0.0f);                                                                            \
                BSA += a0;                                                                                       \
                BSB += a1;                                                                                       \
                ACC[0] = MFMA32(a0, b0, ACC[0]);                                                                 \
                ACC[1] = MFMA32(a0, b1, ACC[1]);                                                                 \
                ACC[2] = MFMA32(a1, b0, ACC[2]);                                                                 \
                ACC[3] = MFMA32(a1, b1, ACC[3]);                                                                 \
                a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;                                                          \
            }
        {
            const float *ap = Gt + hh * BB_P + 64 * om + i, *bp = Tt + hh * BB_P + 64 * in_ + i;
            const int ebase = 4 * hh * BB_P + 32 * wid + i;
            float a0 = ap[0], a1 = ap[32], b0 = bp[0], b1 = bp[32];
            if (EPI) {
#pragma unroll
                for (int kp = 0; kp < BB_R / 2; ++kp) {
                    BB_WGRAD_BODY(acc2, bs2a, bs2b, {
                        const int rh = kp >> 4;
                        const int r = kp & 15;
                        const int idx = ebase + (32 * rh + 8 * (r >> 2) + (r & 3)) * BB_P;
                        Dt[idx] = Tt[idx] > 0.0f ? C[rh][r] : 0.0f;
                        outv[kp] = Gt[idx];
                    })
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll 8
                for (int kp = 0; kp < BB_R / 2; ++kp) BB_WGRAD_BODY(acc2, bs2a, bs2b, {})
            }
        }
        BB_T(2);
        BB_BARRIER_ALL();          // Dt complete, h_in (and x) landed; every wave is done with Gt and Tt
        BB_T(3);
        // ---- gh_in = gh + (gt W1) [h_in > 0]: the products; between them the next tile's gh / t requests (Gt, Tt: free since M) ----
#pragma unroll
        for (int rh = 0; rh < 2; ++rh) {
            C[rh] = f32x16{0};
            const float *ap = Dt + (32 * rh + i) * BB_P + 4 * hh;
#pragma unroll
            for (int Q = 0; Q < 16; ++Q) {
                const f32x4 av = *reinterpret_cast<const f32x4 *>(ap + 8 * Q);
#pragma unroll
                for (int s = 0; s < 4; ++s) C[rh] = MFMA32(av[s], W1r[4 * Q + s], C[rh]);
                if (Q < 9 && more) {
                    if (rh == 0) issue_one(a.gh + (tile + grid) * (BB_R * BB_H), Gt, Q);
                    else issue_one(a.t + (tile + grid) * (BB_R * BB_H), Tt, Q);
                }
            }
        }
        if (!EPI) {
#pragma unroll
            for (int kp = 0; kp < 32; ++kp) {
                const int rh = kp >> 4, r = kp & 15, idx = (32 * rh + 8 * (r >> 2) + 4 * hh + (r & 3)) * BB_P + 32 * wid + i;
                outv[kp] += Ht[idx] > 0.0f ? C[rh][r] : 0.0f;
            }
        }
        BB_T(4);
        // ---- dW1 += gt^T relu(h_in), db1 += colsum(gt); (EPI) in the shadow: gh_in = gh + product masked by h_in > 0 ----
        {
            const float *ap = Dt + hh * BB_P + 64 * om + i, *bp = Ht + hh * BB_P + 64 * in_ + i;
            const int ebase = 4 * hh * BB_P + 32 * wid + i;
            float a0 = ap[0], a1 = ap[32], b0 = bp[0], b1 = bp[32];
            if (EPI) {
#pragma unroll
                for (int kp = 0; kp < BB_R / 2; ++kp) {
                    BB_WGRAD_BODY(acc1, bs1a, bs1b, {
                        const int rh = kp >> 4;
                        const int r = kp & 15;
                        const int idx = ebase + (32 * rh + 8 * (r >> 2) + (r & 3)) * BB_P;
                        outv[kp] += Ht[idx] > 0.0f ? C[rh][r] : 0.0f;
                    })
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll 8
                for (int kp = 0; kp < BB_R / 2; ++kp) BB_WGRAD_BODY(acc1, bs1a, bs1b, {})
            }
        }
        BB_T(5);
        BB_BARRIER_ALL();          // next tile's gh / t landed; every wave is done with Ht and Dt
        BB_T(6);
        if (!INIT) {
#pragma unroll
            for (int rh = 0; rh < 2; ++rh)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 32 * rh + 8 * (r >> 2) + 4 * hh + (r & 3);
                    a.gh_in[(tile * BB_R + row) * BB_H + 32 * wid + i] = outv[16 * rh + r];
                }
        } else {
            // gh_in (= the initial layer's output gradient) -> Dt: A operand of its two products
#pragma unroll
            for (int rh = 0; rh < 2; ++rh)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    Dt[(32 * rh + 8 * (r >> 2) + 4 * hh + (r & 3)) * BB_P + 32 * wid + i] = outv[16 * rh + r];
            BB_T(8);
            BB_BARRIER_LDS();
            BB_T(9);
            // gx[rows 32 om .. +32][32 in_ .. +32] += gh_in Wfull.  The old values are requested here, BEFORE the weight-gradient
            // loop: behind the h_in DMAs just issued their latency is ~5 us, more than the 64 MFMAs of this product cover
            // (measured: requested right before it, 6.0 us for the product; fire-and-forget float adds were as slow).
            // its weight slice (64 registers) is fetched per tile (L2, 16-byte loads of the transposed image) instead of living
            // through the block's products: resident, the kernel spills, and a spill reload waits (vmcnt is in order) for every
            // DMA in flight
            f32x4 W0r[16];
#pragma unroll
            for (int Q = 0; Q < 16; ++Q)
                W0r[Q] = *reinterpret_cast<const f32x4 *>(a.wfull + (32 * in_ + i) * BB_H + 8 * Q + 4 * hh);
            float *gp = a.gx + (tile * BB_R + 32 * om + 4 * hh) * BB_D + 32 * in_ + i;
            float old[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) old[r] = gp[(8 * (r >> 2) + (r & 3)) * BB_D];
            // dW0[32 wid .. +32][0 .. 64] += gh_in^T x, db0 += colsum(gh_in)
            {
                const float *ap = Dt + hh * BB_P + 32 * wid + i, *bp = Xt + hh * BB_D + i;
                float a0 = ap[0], b0 = bp[0], b1 = bp[32];
#pragma unroll 8
                for (int kp = 0; kp < BB_R / 2; ++kp) {     // two MFMAs per k-pair: without the look-ahead the LDS latency shows
                    float na0 = 0.f, nb0 = 0.f, nb1 = 0.f;
                    if (kp + 1 < BB_R / 2) {
                        na0 = ap[(kp + 1) * 2 * BB_P];
                        nb0 = bp[(kp + 1) * 2 * BB_D];
                        nb1 = bp[(kp + 1) * 2 * BB_D + 32];
                    }
                    bs0 += a0;
                    acc0[0] = MFMA32(a0, b0, acc0[0]);
                    acc0[1] = MFMA32(a0, b1, acc0[1]);
                    a0 = na0; b0 = nb0; b1 = nb1;
                }
            }
            BB_T(10);
            {
                f32x16 C3 = {0};
                const float *ap = Dt + (32 * om + i) * BB_P + 4 * hh;
#pragma unroll
                for (int Q = 0; Q < 16; ++Q) {
                    const f32x4 av = *reinterpret_cast<const f32x4 *>(ap + 8 * Q);
#pragma unroll
                    for (int s = 0; s < 4; ++s) C3 = MFMA32(av[s], W0r[Q][s], C3);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) gp[(8 * (r >> 2) + (r & 3)) * BB_D] = old[r] + C3[r];
            }
            BB_T(11);
            BB_BARRIER_LDS();      // every wave is done with Dt and Xt
            if (more) {
                issue_x(a.x + (tile + grid) * (BB_R * BB_D));
            }
        }
    }

    BB_T(0);
#ifdef NF_BB_TRACE
    if (a.trace && blockIdx.x == 0 && threadIdx.x == 0) { a.trace[5] = clock64(); a.trace[4] = wall_clock64(); }
#endif
    // ---- partial tiles: [problem][workgroup][128 * 128 + 128] ----
    constexpr int64_t nW = BB_H * BB_H, stride = nW + BB_H;
    float *o2 = a.part + (int64_t)blockIdx.x * stride, *o1 = a.part + ((int64_t)grid + blockIdx.x) * stride;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int n = 64 * in_ + 32 * (q & 1) + i;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mm = 64 * om + 32 * (q >> 1) + (r & 3) + 8 * (r >> 2) + 4 * hh;
            o2[mm * BB_H + n] = acc2[q][r];
            o1[mm * BB_H + n] = acc1[q][r];
        }
    }
    bs2a += __shfl_xor(bs2a, 32); bs2b += __shfl_xor(bs2b, 32);
    bs1a += __shfl_xor(bs1a, 32); bs1b += __shfl_xor(bs1b, 32);
    if (in_ == 0 && hh == 0) {
        o2[nW + 64 * om + i] = bs2a; o2[nW + 64 * om + 32 + i] = bs2b;
        o1[nW + 64 * om + i] = bs1a; o1[nW + 64 * om + 32 + i] = bs1b;
    }
    if (INIT) {
        constexpr int64_t nW0 = BB_H * BB_D;
        float *o0 = a.part0 + (int64_t)blockIdx.x * (nW0 + BB_H);
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                o0[(32 * wid + (r & 3) + 8 * (r >> 2) + 4 * hh) * BB_D + 32 * q + i] = acc0[q][r];
        bs0 += __shfl_xor(bs0, 32);
        if (hh == 0) o0[nW0 + 32 * wid + i] = bs0;
    }
}

static int bb_grid(int64_t B) {
    const int64_t nt = B / BB_R;
    return (int)(nt < 256 ? nt : 256);
}

}  // namespace nf

extern "C" int64_t nf_resblock_bwd_scratch_floats(int64_t B, int with_init) {
    using namespace nf;
    if (B < BB_R || B % BB_R) return NF_EINVAL;
    const int64_t g = bb_grid(B);
    return 2 * g * ((int64_t)BB_H * BB_H + BB_H) + (with_init ? g * ((int64_t)BB_H * BB_D + BB_H) : 0);
}

extern "C" int nf_resblock_bwd(const void *gh, const void *t, const void *h_in, const void *W1, const void *W2, void *gh_in,
                               void *dW1, void *db1, void *dW2, void *db2, const void *x, const void *wfull, void *gx,
                               void *dW0, void *db0, const void *col_map, int n_cols, void *scratch, int64_t B, int H, int D,
                               nf_stream_t stream) {
    using namespace nf;
    if (H != BB_H || B < BB_R || B % BB_R) return NF_ENOTSUP;
    if (!gh || !t || !h_in || !W1 || !W2 || !dW1 || !db1 || !dW2 || !db2 || !scratch) return NF_EFAULT;
    const bool init = x != nullptr;
    if (init && D != BB_D) return NF_ENOTSUP;
    if (init && (!wfull || !gx || !dW0 || !db0)) return NF_EFAULT;
    if (col_map && (n_cols < 1 || n_cols > BB_D)) return NF_EINVAL;
    if (!init && !gh_in) return NF_EFAULT;
    if (((uintptr_t)gh | (uintptr_t)t | (uintptr_t)h_in | (uintptr_t)x | (uintptr_t)gh_in | (uintptr_t)gx) & 15) return NF_EINVAL;
    if ((((uintptr_t)dW1 ^ (uintptr_t)dW2) | ((uintptr_t)db1 ^ (uintptr_t)db2)) & 3) return NF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int grid = bb_grid(B);
    BlockBwdArgs a;
    a.gh = (const float *)gh; a.t = (const float *)t; a.hin = (const float *)h_in;
    a.W1 = (const float *)W1; a.W2 = (const float *)W2; a.gh_in = (float *)gh_in;
    a.part = (float *)scratch;
    a.x = (const float *)x; a.wfull = (const float *)wfull; a.gx = (float *)gx;
    const int64_t stride = (int64_t)BB_H * BB_H + BB_H;
    a.part0 = (float *)scratch + 2 * grid * stride;
    a.B = B;
#ifdef NF_BB_TRACE
    a.trace = g_bb_trace;
#endif
    const size_t lds = (size_t)(4 * BB_TILE + (init ? BB_R * BB_D : 0)) * sizeof(float);
    static LdsOptIn opt0 = {}, opt1 = {};
    if (init) {
        if (opt_in_lds(reinterpret_cast<const void *>(&resblock_bwd_kernel<true>), lds, opt1) != NF_OK) return NF_ENOTSUP;
        hipLaunchKernelGGL(resblock_bwd_kernel<true>, dim3(grid), dim3(256), lds, st, a);
    } else {
        if (opt_in_lds(reinterpret_cast<const void *>(&resblock_bwd_kernel<false>), lds, opt0) != NF_OK) return NF_ENOTSUP;
        hipLaunchKernelGGL(resblock_bwd_kernel<false>, dim3(grid), dim3(256), lds, st, a);
    }
    NF_CHECK_LAUNCH();
    int rc = wgrad_reduce_launch(a.part, (float *)dW2, (float *)db2, (int64_t)BB_H * BB_H, BB_H, grid, BB_H, 2, grid * stride,
                                 (float *)dW1 - (float *)dW2, (float *)db1 - (float *)db2, nullptr, 0, st);
    if (rc != NF_OK) return rc;
    if (init)
        rc = wgrad_reduce_launch(a.part0, (float *)dW0, (float *)db0, (int64_t)BB_H * BB_D, BB_H, grid, BB_D, 1, 0, 0, 0,
                                 (const int *)col_map, n_cols, st);
    return rc;
}
