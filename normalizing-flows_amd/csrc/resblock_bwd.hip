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
// Workgroup = 8 waves (two per SIMD), one per CU, persistent over 64-row tiles.  The tile's gh / t / h_in arrive by LDS-DMA (16 B
// per lane, no registers) into row-major LDS tiles of pitch 132 floats (one 16-byte pad slot per row, filled with a duplicate load).
// Everything multiplies on v_mfma_f32_16x16x4_f32 (round 3; round 2 used 32x32x2 tiles with 4 waves: 64-register weight slices and
// 64-register gradient quadrants per weight = 439-481 registers, one wave per SIMD, MFMA busy 0.53 -- every LDS round trip, barrier
// and epilogue of the lone wave was exposed; with 16-wide tiles the per-wave state halves and two waves share each SIMD):
//   * the input-gradient products contract over features: wave w owns output columns [16 w, 16 w + 16); A[m = row][k] is read
//     with ds_read_b128 (lane = (row c, k-quarter q4): 4 consecutive features = the lane's k-entry of 4 consecutive MFMAs, i.e. a
//     permuted contraction order k = 16 Q + 4 q4 + j), B = the weight slice W[k][16 w + c] resident in 32 registers per weight;
//     four 16-row blocks = four accumulators, cycled so that no MFMA waits for its predecessor (40-cycle dependent latency);
//   * the weight gradients contract over rows: A[m = feature][k = row] and B[k = row][n = feature] are row-wise ds_read_b32
//     (16 consecutive floats of 4 rows); wave w owns the 32 x 64 block (out rows 32 (w >> 1), in columns 64 (w & 1)) of the
//     128 x 128 output = 8 accumulators of 4 registers per weight for the whole launch, written once as a partial tile summed
//     over the workgroups by nf::wgrad_reduce_kernel in a fixed order (deterministic).
// A tile's loads are requested half a tile before they are consumed -- gh / t of the next tile after the barrier that retires
// them, h_in after the last reader -- so every s_waitcnt vmcnt(0) finds its loads landed.
#include "common.hpp"
#include "fused_common.hpp"

namespace nf {

typedef float f32x4b __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

int wgrad_reduce_launch(const float *part, float *dW, float *db, int64_t nW, int M, int chunks, int N, int np, int64_t zpart,
                        int64_t zdW, int64_t zdb, const int *colmap, int Nout, hipStream_t st);       // wgrad.hip

constexpr int BB_R = 64, BB_P = 132, BB_TILE = BB_R * BB_P, BB_H = 128, BB_D = 64, BB_NW = 8;
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

#define BB_BARRIER_ALL() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define BB_BARRIER_LDS() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <bool INIT>
__global__ void __launch_bounds__(64 * BB_NW, 1)
resblock_bwd_kernel(BlockBwdArgs a) {
    typedef __attribute__((address_space(3))) void *lds_ptr;
    extern __shared__ __attribute__((aligned(16))) float smem_bb[];
    float *Gt = smem_bb, *Tt = Gt + BB_TILE, *Ht = Tt + BB_TILE, *Dt = Ht + BB_TILE, *Xt = Dt + BB_TILE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int om = wid >> 1, in_ = wid & 1;          // this wave's 32 x 64 block of dW2 / dW1: out rows 32 om.., in columns 64 in_..
    const int grid = gridDim.x;
    const int64_t ntiles = a.B / BB_R;

    // DMA slot map: instruction k of a tile fills LDS floats [256 k, 256 k + 256); lane's slot s = 64 k + lane is column
    // group c = s % 33 of row s / 33 (c = 32: the pad slot, loaded with the row's last group again).  Wave w issues instructions
    // k = w + 8 q.  The per-lane element offsets are loop-invariant 32-bit registers; the tile's base pointer is uniform (SGPR
    // pair): the DMA takes them as saddr + voffset, no per-instruction address arithmetic.
    unsigned goff[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int s = 64 * (wid + BB_NW * q) + lane, row = s / 33, c = s - 33 * row;
        goff[q] = (unsigned)((row < BB_R ? row : 0) * BB_H + 4 * (c < 32 ? c : 31));
    }
    // One DMA instruction costs its wave a few hundred cycles of issue: spread through the products' MFMA streams
    // (round 6: the request is inline asm.  As a builtin the compiler knows an LDS-DMA is pending and answers the next LDS read's
    // wait with lgkmcnt(0) -- which defeated the hand-made look-ahead of the products below on every step that follows a request.
    // Landing is waited for by hand anyway: BB_BARRIER_ALL.  m0 is reserved: saved and restored around the instruction.)
    auto dma16 = [&](const float *base, uint32_t byte_off, float *dst) {
        uint32_t m0_;
        const uint32_t ldsa = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)dst);
    // (the "s" constraint alone does not make the pointer scalar: a base the compiler cannot prove wave-uniform came out as a VGPR
    // pair in the instruction -- both halves through v_readfirstlane)
    const uint64_t b64 = (uint64_t)(uintptr_t)base;
    base = reinterpret_cast<const float *>((uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b64 >> 32)) << 32) |
                                                       (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b64)));
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                     : "=&s"(m0_) : "s"(ldsa), "v"(byte_off), "s"(base) : "memory");
    };
    auto issue_one = [&](const float *src, float *tile, int q) {
        if (wid + BB_NW * q < BB_NI) dma16(src, goff[q] * 4u, tile + 256 * (wid + BB_NW * q));
    };
    auto issue = [&](const float *src, float *tile) {
#pragma unroll
        for (int q = 0; q < 5; ++q) issue_one(src, tile, q);
    };
    auto issue_x = [&](const float *src) {          // 64 rows x 64 floats, contiguous: 16 instructions
#pragma unroll
        for (int q = 0; q < 2; ++q) dma16(src, (uint32_t)(256 * (wid + BB_NW * q) + 4 * lane) * 4u, Xt + 256 * (wid + BB_NW * q));
    };

    int64_t tile = blockIdx.x;
    if (tile < ntiles) {
        issue(a.gh + tile * (BB_R * BB_H), Gt);
        issue(a.t + tile * (BB_R * BB_H), Tt);
        issue(a.hin + tile * (BB_R * BB_H), Ht);
        if (INIT) issue_x(a.x + tile * (BB_R * BB_D));
    }
    // weight slices: B operand of the input-gradient products, lane (column c, k-quarter q4), k = 16 Q + 4 q4 + j (the order
    // ds_read_b128 delivers A in)
    float W2r[32], W1r[32];
    {
        const int c = lane & 15, q4 = lane >> 4;
#pragma unroll
        for (int Q = 0; Q < 8; ++Q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 16 * Q + 4 * q4 + j;
                W2r[4 * Q + j] = a.W2[k * BB_H + 16 * wid + c];
                W1r[4 * Q + j] = a.W1[k * BB_H + 16 * wid + c];
            }
    }
    f32x4b acc2[2][4], acc1[2][4], acc0[4];
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) { acc2[ob][ib] = f32x4b{0.f, 0.f, 0.f, 0.f}; acc1[ob][ib] = f32x4b{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int ib = 0; ib < 4; ++ib) acc0[ib] = f32x4b{0.f, 0.f, 0.f, 0.f};
    float bs2[2] = {0.f, 0.f}, bs1[2] = {0.f, 0.f}, bs0 = 0.f;
    int tcount = 0;
    BB_T(7);
#ifdef NF_BB_TRACE
    if (a.trace && blockIdx.x == 0 && threadIdx.x == 0) a.trace[6] = clock64();
#endif
    BB_BARRIER_ALL();
    tcount = 1;

    // out[row][16 wid + c] = sum_k A[row][k] W[k][16 wid + c] for the 64 rows of an LDS tile: C[mb][r] = row 16 mb + 4 q4 + r.
    // Round 6: software-pipelined by hand.  hipcc emitted "4 ds_read_b128 -> s_waitcnt lgkmcnt(0) -> 16 MFMAs" per Q: every one of
    // the 8 Q steps of a product paid an LDS round trip with the matrix pipe idle unless the SIMD's other wave happened to be in its
    // MFMAs.  Here a step is HALF a Q (two 16-row blocks: 2 reads, 8 MFMAs alternating between two accumulators, so no MFMA waits
    // for its predecessor's 40-cycle latency) and step p + 1's operands are requested before step p's MFMAs are issued: the same 16
    // operand registers as before, now double-buffered; the scheduling barriers keep the compiler from undoing the order.
#ifndef NF_BB_NOPIPE
    auto product = [&](const float *At, const float (&Wr)[32], f32x4b (&C)[4], int c, int q4, auto &&between) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) C[mb] = f32x4b{0.f, 0.f, 0.f, 0.f};
        const float *ap = At + c * BB_P + 4 * q4;
        f32x4 cur0 = *reinterpret_cast<const f32x4 *>(ap), cur1 = *reinterpret_cast<const f32x4 *>(ap + 16 * BB_P);
#pragma unroll
        for (int p_ = 0; p_ < 16; ++p_) {
            const int Q = p_ >> 1, mp = p_ & 1;
            f32x4 nx0 = cur0, nx1 = cur1;
            if (p_ + 1 < 16) {
                const int Qn = (p_ + 1) >> 1, mn = (p_ + 1) & 1;
                nx0 = *reinterpret_cast<const f32x4 *>(ap + 16 * (2 * mn) * BB_P + 16 * Qn);
                nx1 = *reinterpret_cast<const f32x4 *>(ap + 16 * (2 * mn + 1) * BB_P + 16 * Qn);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                C[2 * mp] = MFMA16(cur0[j], Wr[4 * Q + j], C[2 * mp]);
                C[2 * mp + 1] = MFMA16(cur1[j], Wr[4 * Q + j], C[2 * mp + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (mp == 1) between(Q);
            cur0 = nx0;
            cur1 = nx1;
        }
    };
    // acc[ob][ib] += A^T B over the tile's rows: A columns [32 om + 16 ob ..], B columns [64 in_ + 16 ib ..] (ReLU on B); the next
    // four rows' six operands are requested before this step's eight MFMAs
    auto wgrad = [&](const float *At, const float *Bt, f32x4b (&acc)[2][4], float (&bs)[2], int c, int q4) {
        const float *ap = At + q4 * BB_P + 32 * om + c, *bp = Bt + q4 * BB_P + 64 * in_ + c;
        float av[2], bv[4];
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) av[ob] = ap[16 * ob];
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) bv[ib] = bp[16 * ib];
#pragma unroll 4
        for (int kb = 0; kb < BB_R / 4; ++kb) {
            float an[2], bn[4];
            const int kn = kb + 1 < BB_R / 4 ? kb + 1 : kb;        // (the last step re-reads its own operands: no branch in the loop)
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) an[ob] = ap[kn * 4 * BB_P + 16 * ob];
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) bn[ib] = bp[kn * 4 * BB_P + 16 * ib];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) bv[ib] = fmaxf(bv[ib], 0.0f);
            bs[0] += av[0];
            bs[1] += av[1];
#pragma unroll
            for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                for (int ib = 0; ib < 4; ++ib) acc[ob][ib] = MFMA16(av[ob], bv[ib], acc[ob][ib]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) av[ob] = an[ob];
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) bv[ib] = bn[ib];
        }
    };
#else
    auto product = [&](const float *At, const float (&Wr)[32], f32x4b (&C)[4], int c, int q4, auto &&between) {
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) C[mb] = f32x4b{0.f, 0.f, 0.f, 0.f};
        const float *ap = At + c * BB_P + 4 * q4;
#pragma unroll
        for (int Q = 0; Q < 8; ++Q) {
            f32x4 av[4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) av[mb] = *reinterpret_cast<const f32x4 *>(ap + 16 * mb * BB_P + 16 * Q);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) C[mb] = MFMA16(av[mb][j], Wr[4 * Q + j], C[mb]);
            between(Q);
        }
    };
    auto wgrad = [&](const float *At, const float *Bt, f32x4b (&acc)[2][4], float (&bs)[2], int c, int q4) {
        const float *ap = At + q4 * BB_P + 32 * om + c, *bp = Bt + q4 * BB_P + 64 * in_ + c;
#pragma unroll 4
        for (int kb = 0; kb < BB_R / 4; ++kb) {
            float av[2], bv[4];
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) av[ob] = ap[kb * 4 * BB_P + 16 * ob];
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) bv[ib] = fmaxf(bp[kb * 4 * BB_P + 16 * ib], 0.0f);
            bs[0] += av[0];
            bs[1] += av[1];
#pragma unroll
            for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                for (int ib = 0; ib < 4; ++ib) acc[ob][ib] = MFMA16(av[ob], bv[ib], acc[ob][ib]);
        }
    };
#endif

    for (; tile < ntiles; tile += grid, ++tcount) {
        const bool more = tile + grid < ntiles, first = tile == (int64_t)blockIdx.x;     // the first tile's h_in: requested above
        BB_T(0);
        // per-tile lane index (loop-invariant LDS / global addresses hoisted out of the tile loop cost registers)
        int l_ = lane;
        asm volatile("" : "+v"(l_));
        const int c = l_ & 15, q4 = l_ >> 4;
        float outv[16];
        f32x4b C[4];
        // ---- gt = (gh W2) [t > 0]; behind the MFMAs this tile's h_in requests (Ht: free since the last barrier) ----
        product(Gt, W2r, C, c, q4, [&](int Q) { if (Q < 5 && !first) issue_one(a.hin + tile * (BB_R * BB_H), Ht, Q); });
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int idx = (16 * mb + 4 * q4 + r) * BB_P + 16 * wid + c;
                Dt[idx] = Tt[idx] > 0.0f ? C[mb][r] : 0.0f;
                outv[4 * mb + r] = Gt[idx];
            }
        BB_T(1);
        // ---- dW2 += gh^T relu(t), db2 += colsum(gh) ----
        wgrad(Gt, Tt, acc2, bs2, c, q4);
        BB_T(2);
        BB_BARRIER_ALL();          // Dt complete, h_in (and x) landed; every wave is done with Gt and Tt
        BB_T(3);
        // ---- gh_in = gh + (gt W1) [h_in > 0]; behind the MFMAs the next tile's gh / t requests (Gt, Tt: free) ----
        product(Dt, W1r, C, c, q4, [&](int Q) {
            if (more) {
                if (Q < 5) issue_one(a.gh + (tile + grid) * (BB_R * BB_H), Gt, Q);
                else if (Q == 5) { issue_one(a.t + (tile + grid) * (BB_R * BB_H), Tt, 0); issue_one(a.t + (tile + grid) * (BB_R * BB_H), Tt, 1); }
                else if (Q == 6) { issue_one(a.t + (tile + grid) * (BB_R * BB_H), Tt, 2); issue_one(a.t + (tile + grid) * (BB_R * BB_H), Tt, 3); }
                else issue_one(a.t + (tile + grid) * (BB_R * BB_H), Tt, 4);
            }
        });
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int idx = (16 * mb + 4 * q4 + r) * BB_P + 16 * wid + c;
                outv[4 * mb + r] += Ht[idx] > 0.0f ? C[mb][r] : 0.0f;
            }
        BB_T(4);
        // ---- dW1 += gt^T relu(h_in), db1 += colsum(gt) ----
        wgrad(Dt, Ht, acc1, bs1, c, q4);
        BB_T(5);
        BB_BARRIER_ALL();          // next tile's gh / t landed; every wave is done with Ht and Dt
        BB_T(6);
        // gh_in -> Dt: written back as whole rows (full 128-byte lines), or (INIT) the A operand of the initial layer's products
#pragma unroll
        for (int mb = 0; mb < 4; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) Dt[(16 * mb + 4 * q4 + r) * BB_P + 16 * wid + c] = outv[4 * mb + r];
        BB_T(8);
        BB_BARRIER_LDS();
        BB_T(9);
        if (!INIT) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int idx = tid + 64 * BB_NW * it, row = idx >> 5, c4 = idx & 31;
                *reinterpret_cast<f32x4 *>(a.gh_in + (tile * BB_R + row) * BB_H + 4 * c4) = *reinterpret_cast<const f32x4 *>(Dt + row * BB_P + 4 * c4);
            }
            BB_BARRIER_LDS();      // (the next tile's first epilogue writes Dt before its first barrier: every wave must have read its rows)
        } else {
            // gx[rows 16 mb.., columns 16 cb..] += gh_in Wfull: wave w owns column block cb = w & 3 and row blocks 2 (w >> 2), +1.
            // The old values are requested BEFORE the weight-gradient loop (behind the h_in DMAs just issued their latency is
            // ~5 us); the weight slice (32 registers) is fetched per tile (L2) instead of living through the block's products.
            const int cb = wid & 3, mb0 = 2 * (wid >> 2);
            float W0r[32];
#pragma unroll
            for (int Q = 0; Q < 8; ++Q) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(a.wfull + (16 * cb + c) * BB_H + 16 * Q + 4 * q4);
#pragma unroll
                for (int j = 0; j < 4; ++j) W0r[4 * Q + j] = v[j];
            }
            float *gp = a.gx + (tile * BB_R + 16 * mb0 + 4 * q4) * BB_D + 16 * cb + c;
            float old[8];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) old[4 * m + r] = gp[(16 * m + r) * BB_D];
            // dW0[16 w .. +16][0 .. 64] += gh_in^T x, db0 += colsum(gh_in)
            {
                const float *ap = Dt + q4 * BB_P + 16 * wid + c, *bp = Xt + q4 * BB_D + c;
#pragma unroll 4
                for (int kb = 0; kb < BB_R / 4; ++kb) {
                    const float av = ap[kb * 4 * BB_P];
                    float bv[4];
#pragma unroll
                    for (int ib = 0; ib < 4; ++ib) bv[ib] = bp[kb * 4 * BB_D + 16 * ib];
                    bs0 += av;
#pragma unroll
                    for (int ib = 0; ib < 4; ++ib) acc0[ib] = MFMA16(av, bv[ib], acc0[ib]);
                }
            }
            BB_T(10);
            {
                f32x4b C3[2] = {f32x4b{0.f, 0.f, 0.f, 0.f}, f32x4b{0.f, 0.f, 0.f, 0.f}};
                const float *ap = Dt + (16 * mb0 + c) * BB_P + 4 * q4;
#pragma unroll
                for (int Q = 0; Q < 8; ++Q) {
                    const f32x4 a0 = *reinterpret_cast<const f32x4 *>(ap + 16 * Q), a1 = *reinterpret_cast<const f32x4 *>(ap + 16 * BB_P + 16 * Q);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        C3[0] = MFMA16(a0[j], W0r[4 * Q + j], C3[0]);
                        C3[1] = MFMA16(a1[j], W0r[4 * Q + j], C3[1]);
                    }
                }
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) gp[(16 * m + r) * BB_D] = old[4 * m + r] + C3[m][r];
            }
            BB_T(11);
            BB_BARRIER_LDS();      // every wave is done with Dt and Xt
            if (more) issue_x(a.x + (tile + grid) * (BB_R * BB_D));
        }
    }

    BB_T(0);
#ifdef NF_BB_TRACE
    if (a.trace && blockIdx.x == 0 && threadIdx.x == 0) { a.trace[5] = clock64(); a.trace[4] = wall_clock64(); }
#endif
    // ---- partial tiles: [problem][workgroup][128 * 128 + 128] ----
    constexpr int64_t nW = BB_H * BB_H, stride = nW + BB_H;
    float *o2 = a.part + (int64_t)blockIdx.x * stride, *o1 = a.part + ((int64_t)grid + blockIdx.x) * stride;
    const int c = lane & 15, q4 = lane >> 4;
#pragma unroll
    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mm = 32 * om + 16 * ob + 4 * q4 + r, n = 64 * in_ + 16 * ib + c;
                o2[mm * BB_H + n] = acc2[ob][ib][r];
                o1[mm * BB_H + n] = acc1[ob][ib][r];
            }
#pragma unroll
    for (int ob = 0; ob < 2; ++ob) {
        bs2[ob] += __shfl_xor(bs2[ob], 16); bs2[ob] += __shfl_xor(bs2[ob], 32);
        bs1[ob] += __shfl_xor(bs1[ob], 16); bs1[ob] += __shfl_xor(bs1[ob], 32);
    }
    if (in_ == 0 && q4 == 0) {
#pragma unroll
        for (int ob = 0; ob < 2; ++ob) {
            o2[nW + 32 * om + 16 * ob + c] = bs2[ob];
            o1[nW + 32 * om + 16 * ob + c] = bs1[ob];
        }
    }
    if (INIT) {
        constexpr int64_t nW0 = BB_H * BB_D;
        float *o0 = a.part0 + (int64_t)blockIdx.x * (nW0 + BB_H);
#pragma unroll
        for (int ib = 0; ib < 4; ++ib)
#pragma unroll
            for (int r = 0; r < 4; ++r) o0[(16 * wid + 4 * q4 + r) * BB_D + 16 * ib + c] = acc0[ib][r];
        bs0 += __shfl_xor(bs0, 16); bs0 += __shfl_xor(bs0, 32);
        if (q4 == 0) o0[nW0 + 16 * wid + c] = bs0;
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

// Workgroups (= partial tiles per problem) of a launch over B rows.
extern "C" int nf_resblock_bwd_grid(int64_t B) {
    using namespace nf;
    if (B < BB_R || B % BB_R) return NF_EINVAL;
    return bb_grid(B);
}

// The pass over the rows alone: partial tiles [2][grid][128 * 128 + 128] ((dW2, db2) then (dW1, db1)) and, with x, the initial
// layer's [grid][128 * 64 + 128] behind them in `scratch` (nf_resblock_bwd_scratch_floats), summed later by the caller
// (nf_resblock_bwd: three nf::wgrad_reduce_kernel problems; nf_coupling_train_bwd: the layer's one reduction launch).
extern "C" int nf_resblock_bwd_partials(const void *gh, const void *t, const void *h_in, const void *W1, const void *W2, void *gh_in,
                                        const void *x, const void *wfull, void *gx, void *scratch, int64_t B, int H, int D,
                                        nf_stream_t stream) {
    using namespace nf;
    if (H != BB_H || B < BB_R || B % BB_R) return NF_ENOTSUP;
    if (!gh || !t || !h_in || !W1 || !W2 || !scratch) return NF_EFAULT;
    const bool init = x != nullptr;
    if (init && D != BB_D) return NF_ENOTSUP;
    if (init && (!wfull || !gx)) return NF_EFAULT;
    if (!init && !gh_in) return NF_EFAULT;
    if (((uintptr_t)gh | (uintptr_t)t | (uintptr_t)h_in | (uintptr_t)x | (uintptr_t)gh_in | (uintptr_t)gx) & 15) return NF_EINVAL;
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
        hipLaunchKernelGGL(resblock_bwd_kernel<true>, dim3(grid), dim3(64 * BB_NW), lds, st, a);
    } else {
        if (opt_in_lds(reinterpret_cast<const void *>(&resblock_bwd_kernel<false>), lds, opt0) != NF_OK) return NF_ENOTSUP;
        hipLaunchKernelGGL(resblock_bwd_kernel<false>, dim3(grid), dim3(64 * BB_NW), lds, st, a);
    }
    NF_CHECK_LAUNCH();
    return NF_OK;
}

extern "C" int nf_resblock_bwd(const void *gh, const void *t, const void *h_in, const void *W1, const void *W2, void *gh_in,
                               void *dW1, void *db1, void *dW2, void *db2, const void *x, const void *wfull, void *gx,
                               void *dW0, void *db0, const void *col_map, int n_cols, void *scratch, int64_t B, int H, int D,
                               nf_stream_t stream) {
    using namespace nf;
    if (H != BB_H || B < BB_R || B % BB_R) return NF_ENOTSUP;
    if (!gh || !t || !h_in || !W1 || !W2 || !dW1 || !db1 || !dW2 || !db2 || !scratch) return NF_EFAULT;
    const bool init = x != nullptr;
    if (init && (!wfull || !gx || !dW0 || !db0)) return NF_EFAULT;
    if (col_map && (n_cols < 1 || n_cols > BB_D)) return NF_EINVAL;
    if ((((uintptr_t)dW1 ^ (uintptr_t)dW2) | ((uintptr_t)db1 ^ (uintptr_t)db2)) & 3) return NF_EINVAL;
    int rc = nf_resblock_bwd_partials(gh, t, h_in, W1, W2, gh_in, x, wfull, gx, scratch, B, H, D, stream);
    if (rc != NF_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int grid = bb_grid(B);
    const int64_t stride = (int64_t)BB_H * BB_H + BB_H;
    const float *part = (const float *)scratch, *part0 = part + 2 * grid * stride;
    rc = wgrad_reduce_launch(part, (float *)dW2, (float *)db2, (int64_t)BB_H * BB_H, BB_H, grid, BB_H, 2, grid * stride,
                             (float *)dW1 - (float *)dW2, (float *)db1 - (float *)db2, nullptr, 0, st);
    if (rc != NF_OK) return rc;
    if (init)
        rc = wgrad_reduce_launch(part0, (float *)dW0, (float *)db0, (int64_t)BB_H * BB_D, BB_H, grid, BB_D, 1, 0, 0, 0,
                                 (const int *)col_map, n_cols, st);
    return rc;
}
