// made_bwd.hip -- the backward pass of MADE (nets/made.py:296-304 under core.py:87-102 + loss.backward(): what torch autograd does
// with F.linear(x, weight * mask, bias) per MaskedLinear, :80-81) as three launches per MADE: the input-gradient chain, every weight
// / bias gradient of the network, and their fixed-order reduction.  Operands: what nf_made_forward_train (made_fwd.hip EPI 3) left:
// save[l][Bp][Hp] pre-activations and the ReLU sign bits.
//
//   g_p (B, mult D)  ->  g_h = Wf^T g_p
//   block b = NB-1 .. 0:   g_t = (W2^T g_h) . [t_b > 0];   g_h += (W1^T g_t) . [h_b > 0]
//   g_x = W0^T g_h
//   dWf = g_p (x) h_NB;  dW2_b = g_h(b+1) (x) relu(t_b);  dW1_b = g_t (x) relu(h_b);  dW0 = g_h(0) (x) x;  db = column sums
//
// 1. made_bwd_kernel<NSB>: the chain on the 64-row-tile engine of the forward pass (mlp_tile.hpp): the same kernel shape with the
//    TRANSPOSED masked weights -- with the hidden units sorted by degree a row-block of W^T needs a SUFFIX of k-groups [kg0, kg0 +
//    nkg) (item = [nkg, rb, kg0]); gradients live in accumulator registers, the next product's B operand is published to LDS, the
//    ReLU masks are the forward's sign bits (one dword per lane and item); g_p enters in chunks of Hp columns (mult D = 2944 for the
//    autoregressive spline layer does not fit the LDS at once).  Every g_h / g_t is also written row-major to G[l][Bp][Hp] for 2.
// 2. made_wgrad_kernel: dW = dY^T X over the batch for ALL masked linears of the network in one launch: 128 x 128 output tiles that
//    hold a mask non-zero (host table), both operands streamed by LDS-DMA through a 3-slot ring by four helper waves while four waves
//    issue the MFMAs (the design of wgrad.hip's wgrad_ring_kernel, with row strides and per-problem operands); split over row chunks.
// 3. made_wgrad_reduce_kernel: the chunks' partial tiles summed in a fixed order (deterministic) and scattered from slot space to
//    the parameters' own layout, masked entries left zero.
#include "mlp_tile.hpp"

namespace nf {

// ---- 1. input-gradient chain ---------------------------------------------------------------------------------------------------------
template <int NSB, int TR = MF_ROWS>
__global__ void __launch_bounds__(64 * MF_NW, 1)
made_bwd_kernel(const float *__restrict__ gp, const unsigned *__restrict__ bits, float *__restrict__ gx, float *__restrict__ G,
                const float *__restrict__ blob, const int *__restrict__ table, int64_t B, int64_t Bp) {
    static_assert(TR == 64 || (TR == 128 && NSB == 1), "128-row tiles: 256-slot networks only (mlp_tile.hpp mf_tr128)");
    constexpr int NSH = TR / 64;             // sample blocks per half of a tile
    constexpr int NS = NSB * NSH;
    constexpr int HRB = 8 * NSB;
    constexpr int HP = 256 * NSB;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *acts = lds;                                  // [HP / 8 k-groups][2][TR][4]
    float *xreg = lds + (size_t)HRB * 4 * 8 * TR;       // [Dp / 8][2][TR][4]: the g_x tile on its way out
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, hh = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = table[0], NB = table[5], NC = table[7], nitems = table[10];
    const int MD = table[12] ? table[12] : table[6] * D;       // row length of g_params
    const int ldg = table[14] ? table[14] : MD, ldgx = table[15] ? table[15] : D;      // row strides of g_params / g_x (conv path: padded)
    const int plain = table[13];                               // plain MLP (made_fwd.hip): g_t = (Wf^T g_p).[t>0]; g_h = (W1^T g_t).[h>0]
    const int nfin = table[8] > 0 ? table[8] : 1;              // rounds of the last product: 4 feature row-blocks x 2 sample blocks each
    const int *items = table + MF_HDR + w * nitems * 4;       // [nitems][nkg, rb, kg0, -]
    const float *stream = blob + table[16 + w];
    const int rbs[2] = {w, HRB - 1 - w};
    const int sb0s[2] = {0, NSB == 2 ? 0 : NSH};
    const int lane_b = (TR * hh + n) * 4;
    constexpr int KG = 8 * TR;               // floats of one k-group of activations
    const int64_t ntiles = (B + TR - 1) / TR;
    MfRing ring;
    mf_ring_start(ring, stream, lane);

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row0 = tile * TR;
        const int nrows = (int)((B - row0) < TR ? (B - row0) : TR);
        ring.ap = stream + lane * 4;
        const unsigned *btile = bits + ((size_t)tile * 2 * NB * 2) * 512 + tid;
        float *gtile = G + (size_t)row0 * HP;
        f32x16 gh[2][NS], u[2][NS];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int ss = 0; ss < NS; ++ss)
#pragma unroll
                for (int r = 0; r < 16; ++r) gh[s][ss][r] = 0.0f;
        // ---- g_h = Wf^T g_p, g_p in chunks of HP columns (B-operand order; rows beyond the batch and columns beyond mult D zero) ----
        for (int c = 0; c < NC; ++c) {
            MF_BARRIER();
            {
                const int r = tid & (TR - 1), cg = tid / TR;
                const float *gr = gp + (row0 + r) * ldg;
                for (int c4 = cg; c4 < HP / 4; c4 += 64 * MF_NW / TR) {
                    const int col = c * HP + 4 * c4;
                    f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (r < nrows && col < MD) {
                        if ((MD & 3) == 0) v = *reinterpret_cast<const f32x4 *>(gr + col);
                        else
#pragma unroll
                            for (int i = 0; i < 4; ++i) if (col + i < MD) v[i] = gr[col + i];
                    }
                    *reinterpret_cast<f32x4 *>(acts + ((size_t)c4 * TR + r) * 4) = v;
                }
            }
            MF_BARRIER();
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int *it = items + 4 * (2 * c + s);
                mf_item<NS, true, TR>(ring, it[0], acts + lane_b + 128 * sb0s[s] + it[2] * KG, gh[s]);
            }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) if (G) mf_save_rows<NS, false>(gtile + (size_t)(2 * NB) * Bp * HP, HP, nrows, rbs[s], sb0s[s], hh, n, gh[s]);
        // ---- residual blocks, last first ------------------------------------------------------------------------------------------------
        if (plain) {
            const int *itb = items + 4 * (2 * NC);
            unsigned bt[2], bh[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bt[s] = btile[((size_t)1 * 2 + s) * 512];
                bh[s] = btile[((size_t)0 * 2 + s) * 512];
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                mf_masked<NS, false>(gh[s], gh[s], bt[s]);               // g_t (G[2] above holds the unmasked product: unused)
                if (G) mf_save_rows<NS, false>(gtile + (size_t)1 * Bp * HP, HP, nrows, rbs[s], sb0s[s], hh, n, gh[s]);
            }
            MF_BARRIER();
#pragma unroll
            for (int s = 0; s < 2; ++s) mf_publish<NS, false, TR>(acts, rbs[s], sb0s[s], hh, n, gh[s]);
            MF_BARRIER();
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int *it = itb + 4 * (2 + s);
                mf_item<NS, false, TR>(ring, it[0], acts + lane_b + 128 * sb0s[s] + it[2] * KG, u[s]);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                mf_masked<NS, false>(gh[s], u[s], bh[s]);                // g_h
                if (G) mf_save_rows<NS, false>(gtile, HP, nrows, rbs[s], sb0s[s], hh, n, gh[s]);
            }
        }
        for (int b = plain ? -1 : NB - 1; b >= 0; --b) {
            const int *itb = items + 4 * (2 * NC + 4 * (NB - 1 - b));
            unsigned bt[2], bh[2];
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                bt[s] = btile[((size_t)(2 * b + 1) * 2 + s) * 512];
                bh[s] = btile[((size_t)(2 * b) * 2 + s) * 512];
            }
            MF_BARRIER();
#pragma unroll
            for (int s = 0; s < 2; ++s) mf_publish<NS, false, TR>(acts, rbs[s], sb0s[s], hh, n, gh[s]);
            MF_BARRIER();
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int *it = itb + 4 * s;
                mf_item<NS, false, TR>(ring, it[0], acts + lane_b + 128 * sb0s[s] + it[2] * KG, u[s]);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                mf_masked<NS, false>(u[s], u[s], bt[s]);                 // g_t
                if (G) mf_save_rows<NS, false>(gtile + (size_t)(2 * b + 1) * Bp * HP, HP, nrows, rbs[s], sb0s[s], hh, n, u[s]);
            }
            MF_BARRIER();
#pragma unroll
            for (int s = 0; s < 2; ++s) mf_publish<NS, false, TR>(acts, rbs[s], sb0s[s], hh, n, u[s]);
            MF_BARRIER();
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int *it = itb + 4 * (2 + s);
                mf_item<NS, false, TR>(ring, it[0], acts + lane_b + 128 * sb0s[s] + it[2] * KG, u[s]);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                mf_masked<NS, true>(gh[s], u[s], bh[s]);                 // g_h of the block's input
                if (G) mf_save_rows<NS, false>(gtile + (size_t)(2 * b) * Bp * HP, HP, nrows, rbs[s], sb0s[s], hh, n, gh[s]);
            }
        }
        // ---- g_x = W0^T g_h: one 32-feature row-block and sample block per wave ------------------------------------------------------
        MF_BARRIER();
#pragma unroll
        for (int s = 0; s < 2; ++s) mf_publish<NS, false, TR>(acts, rbs[s], sb0s[s], hh, n, gh[s]);
        MF_BARRIER();
        for (int rd = 0; rd < nfin; ++rd) {
            const int *it = items + 4 * (2 * NC + 4 * NB + rd);
            const int rb = it[1], sf = w >> 2;
            if (rb >= 0) {
                f32x16 o[NSH];
                mf_item<NSH, false, TR>(ring, it[0], acts + lane_b + 128 * NSH * sf + it[2] * KG, o);
                mf_publish<NSH, false, TR>(xreg, rb, NSH * sf, hh, n, o);
            }
        }
        MF_BARRIER();
        {
            const int r = tid & (TR - 1), cg = tid / TR;
            float *xr = gx + (row0 + r) * ldgx;
            if (r < nrows)
                for (int c = cg; 4 * c < D; c += 64 * MF_NW / TR) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(xreg + ((size_t)c * TR + r) * 4);
                    if ((D & 3) == 0) *reinterpret_cast<f32x4 *>(xr + 4 * c) = v;
                    else
#pragma unroll
                        for (int i = 0; i < 4; ++i) if (4 * c + i < D) xr[4 * c + i] = v[i];
                }
        }
        MF_BARRIER();                              // the next tile overwrites the activations and the g_x tile
    }
}

template <int NSB, int TR = MF_ROWS>
static int made_bwd_launch(const void *gp, const void *bits, void *gx, void *G, const void *blob, const int32_t *table, int64_t B,
                           hipStream_t st, int dp) {
    const int64_t ntiles = (B + TR - 1) / TR;
    const int grid = (int)(ntiles < 256 ? ntiles : 256);
    const size_t lds = sizeof(float) * ((size_t)8 * NSB * 4 * 8 * TR + (dp > 128 ? 2 * MF_XFLOATS : MF_XFLOATS));
    static LdsOptIn opted;
    if (opt_in_lds(reinterpret_cast<const void *>(&made_bwd_kernel<NSB, TR>), lds, opted) != NF_OK) return NF_ENOTSUP;
    hipLaunchKernelGGL((made_bwd_kernel<NSB, TR>), dim3((unsigned)grid), dim3(64 * MF_NW), lds, st, (const float *)gp,
                       (const unsigned *)bits, (float *)gx, (float *)G, (const float *)blob, (const int *)table, B,
                       (B + MF_ROWS - 1) / MF_ROWS * MF_ROWS);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// ---- 2. weight gradients --------------------------------------------------------------------------------------------------------------
// wt (int32): hdr[16] = [ntiles, nproblems, ...]; problems at wt + 16: 8 ints [dY base, dY matrix index, ldY, X base, X matrix index,
// ldX, relu(X), -] (base 0 = g_p padded, 1 = x padded, 2 = G, 3 = save; matrix index l = the l-th (Bp, ld) matrix behind the base);
// tiles behind them: 8 ints [problem, m0, n0, want_bias, ...].  part: [chunk][tile][128 * 128 + 128].
constexpr int MW_T = 128, MW_KS = 16, MW_NR = 3, MW_NT = 64 * 8, MW_PART = MW_T * MW_T + MW_T;
#define MW_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__global__ void __launch_bounds__(MW_NT, 4)
made_wgrad_kernel(const float *__restrict__ b0p, const float *__restrict__ b1p, const float *__restrict__ b2p,
                  const float *__restrict__ b3p, float *__restrict__ part, const int *__restrict__ wt, int chunk_rows, int64_t Bp,
                  int tile_major) {
    typedef __attribute__((address_space(3))) void *lds_ptr;
    __shared__ __attribute__((aligned(16))) float ring[MW_NR][2][MW_KS][MW_T];     // [slot][dY | X][row][128] = 48 KB
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntl = wt[0], npr = wt[1];
    const int bt = tile_major ? blockIdx.x : blockIdx.y, bc = tile_major ? blockIdx.y : blockIdx.x;
    const int *tl = wt + 16 + 8 * npr + 8 * bt;
    const int *pr = wt + 16 + 8 * tl[0];
    const int m0 = tl[1], n0 = tl[2], want_bias = tl[3];
    auto base = [&](int k) { return k == 0 ? b0p : k == 1 ? b1p : k == 2 ? b2p : b3p; };
    const int ldY = pr[2], ldX = pr[5], x_relu = pr[6];
    const float *dY = base(pr[0]) + (size_t)pr[1] * Bp * ldY + m0;
    const float *X = base(pr[3]) + (size_t)pr[4] * Bp * ldX + n0;
    const int64_t r_begin = (int64_t)bc * chunk_rows;
    int64_t r_end = r_begin + chunk_rows;
    if (r_end > Bp) r_end = Bp;
    const int nsteps = (int)((r_end - r_begin) / MW_KS);         // chunk_rows and Bp are multiples of 16
    const bool mfma_wave = wid < 4;
    const int dw = wid - 4;
    const int lrow = lane >> 5, lcol = (lane & 31) * 4;
    auto issue = [&](int s) {
        const int64_t r0 = r_begin + (int64_t)s * MW_KS + 4 * dw + lrow;
        float *slotA = &ring[s % MW_NR][0][4 * dw][0], *slotB = &ring[s % MW_NR][1][4 * dw][0];
        __builtin_amdgcn_global_load_lds(dY + r0 * ldY + lcol, (lds_ptr)slotA, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(dY + (r0 + 2) * ldY + lcol, (lds_ptr)(slotA + 2 * MW_T), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(X + r0 * ldX + lcol, (lds_ptr)slotB, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(X + (r0 + 2) * ldX + lcol, (lds_ptr)(slotB + 2 * MW_T), 16, 0, 0);
    };
    if (!mfma_wave) {
        // helper waves: requests and barriers only (the same barrier sequence as the MFMA waves below)
        for (int s = 0; s < MW_NR - 1 && s < nsteps; ++s) issue(s);
        for (int s = 0; s < nsteps; ++s) {
            if (s + MW_NR - 1 <= nsteps) NF_WAIT_VMCNT(4 * (MW_NR - 2));
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (s + MW_NR - 1 < nsteps) issue(s + MW_NR - 1);
        }
        return;
    }
    const int wm = wid >> 1, wn = wid & 1;
    f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
    float bs0 = 0.0f, bs1 = 0.0f;
    for (int s = 0; s < nsteps; ++s) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // step s landed; every wave is done with slot (s - 1) % NR
        const float *ap = &ring[s % MW_NR][0][h][wm * 64 + i], *bp = &ring[s % MW_NR][1][h][wn * 64 + i];
#pragma unroll
        for (int kp = 0; kp < MW_KS / 2; ++kp) {
            const float a0 = ap[kp * 2 * MW_T], a1 = ap[kp * 2 * MW_T + 32];
            float x0 = bp[kp * 2 * MW_T], x1 = bp[kp * 2 * MW_T + 32];
            if (x_relu) { x0 = fmaxf(x0, 0.0f); x1 = fmaxf(x1, 0.0f); }
            bs0 += a0;
            bs1 += a1;
            acc00 = MW_MFMA(a0, x0, acc00);
            acc01 = MW_MFMA(a0, x1, acc01);
            acc10 = MW_MFMA(a1, x0, acc10);
            acc11 = MW_MFMA(a1, x1, acc11);
        }
    }
    float *out = part + ((size_t)bc * ntl + bt) * MW_PART;
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
#pragma unroll
        for (int t_ = 0; t_ < 2; ++t_) {
            const f32x16 &acc = s_ == 0 ? (t_ == 0 ? acc00 : acc01) : (t_ == 0 ? acc10 : acc11);
            const int nn = wn * 64 + 32 * t_ + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mm = wm * 64 + 32 * s_ + (r & 3) + 8 * (r >> 2) + 4 * h;
                out[(size_t)mm * MW_T + nn] = acc[r];
            }
        }
    }
    if (want_bias && wn == 0) {
        bs0 += __shfl_xor(bs0, 32);
        bs1 += __shfl_xor(bs1, 32);
        if (h == 0) {
            out[MW_T * MW_T + wm * 64 + i] = bs0;
            out[MW_T * MW_T + wm * 64 + i + 32] = bs1;
        }
    }
}

// ---- 2b. weight gradients with operands in the one-pass kernels' scratch order (round 6) ------------------------------------------------
// MAF's density direction under autograd (autograd.MafInverseFn): the hidden layers' output gradients ARE the transposed solve's
// activation scratch and the linears' inputs ARE the inverse pass's (csrc/maf_inverse_h.hip), both [32-row wave tile][NL layers]
// [position / 8][2 halves][32 samples][4] over padded POSITIONS.  Round 5 rearranged both into row-major tensors for the kernel above
// (nf_maf_scratch_rows: 2 x 245 us per config-5 layer, 2.7 GB of traffic); here the contraction reads the scratches as they are and
// works in position space -- the problems' row / column maps and the mask-non-zero tile list are built over positions
// (flows/maf_pack.position_wgrad_tables), the reduction above scatters as before.
//   * flags (problem int 7): 1 = dY in scratch order (matrix index = scratch layer, ld unused), 2 = X likewise, 4 = the product is
//     negated (the solve runs on the cotangent g_p(v, g_ld), the weight gradients belong to g_p(-v, -g_ld)).
//   * a step is still 16 rows x 128 slots per operand = 8 LDS-DMA requests of 1 KB: request j = k-groups 2 j, 2 j + 1 of the tile for
//     the step's 16 samples (four 256-byte runs), written to LDS as [k-group][half][16][4] with 16 bytes between requests -- so the
//     MFMA lanes' 16-byte reads (lane i = position quad i of the tile, its row = 2 kp + h) hit every bank group exactly four times:
//     the minimum for 32 lanes x 16 bytes.  A lane's quad holds four POSITIONS of one sample where the row-major layout gives one
//     position of four samples, so the wave (wm, wn) takes components 2 wm, 2 wm + 1 [2 wn, 2 wn + 1]: MFMA row [column] i of
//     accumulator (s, t) is position 4 i + 2 wm + s [4 i + 2 wn + t] of the tile.
// Rows: B a multiple of 64 (the scratch has no rows beyond the batch to read zeros from); slots: the scratch's position count a
// multiple of 128.  Anything else keeps the rearrangement.
constexpr int MWP_REQ = 2 * 2 * 16 * 4 + 4;          // floats of one request + 16 bytes
constexpr int MWP_SLOT = 8 * MWP_REQ;                // >= MW_KS * MW_T (a row-major operand uses the first 2048 floats)
static_assert(MWP_SLOT >= MW_KS * MW_T, "slot holds either layout");

__global__ void __launch_bounds__(MW_NT, 4)
made_wgrad_pos_kernel(const float *__restrict__ b0p, const float *__restrict__ b1p, const float *__restrict__ b2p,
                      const float *__restrict__ b3p, float *__restrict__ part, const int *__restrict__ wt, int chunk_rows, int64_t Bp,
                      int tile_major, int NL, int Hs) {
    typedef __attribute__((address_space(3))) void *lds_ptr;
    __shared__ __attribute__((aligned(16))) float ring[MW_NR][2][MWP_SLOT];
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 31, h = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntl = wt[0], npr = wt[1];
    const int bt = tile_major ? blockIdx.x : blockIdx.y, bc = tile_major ? blockIdx.y : blockIdx.x;
    const int *tl = wt + 16 + 8 * npr + 8 * bt;
    const int *pr = wt + 16 + 8 * tl[0];
    const int m0 = tl[1], n0 = tl[2], want_bias = tl[3];
    auto base = [&](int k) { return k == 0 ? b0p : k == 1 ? b1p : k == 2 ? b2p : b3p; };
    const int ldY = pr[2], ldX = pr[5], x_relu = pr[6], flags = pr[7];
    const bool sy = flags & 1, sx = flags & 2;
    const float sgn = (flags & 4) ? -1.0f : 1.0f;
    // row-major: the tile's first column of row 0; scratch order: the tile's first k-group of wave tile 0, layer pr[1] / pr[4]
    const float *dY = sy ? base(pr[0]) + (size_t)pr[1] * Hs * 32 + (size_t)(m0 >> 3) * 256 : base(pr[0]) + (size_t)pr[1] * Bp * ldY + m0;
    const float *X = sx ? base(pr[3]) + (size_t)pr[4] * Hs * 32 + (size_t)(n0 >> 3) * 256 : base(pr[3]) + (size_t)pr[4] * Bp * ldX + n0;
    const size_t wtile = (size_t)NL * Hs * 32;              // floats of one 32-row wave tile of a scratch
    const int64_t r_begin = (int64_t)bc * chunk_rows;
    int64_t r_end = r_begin + chunk_rows;
    if (r_end > Bp) r_end = Bp;
    const int nsteps = (int)((r_end - r_begin) / MW_KS);
    const bool mfma_wave = wid < 4;
    const int dw = wid - 4;
    const int lrow = lane >> 5, lcol = (lane & 31) * 4;
    const int s_off = (lane >> 5) * 256 + ((lane >> 4) & 1) * 128 + (lane & 15) * 4;     // (k-group of the request, half, sample) of the lane
    auto issue_op = [&](const float *P, bool scr, int ld, float *slot, int s) {
        const int64_t row0 = r_begin + (int64_t)s * MW_KS;
        if (scr) {
            const float *src = P + (size_t)(row0 >> 5) * wtile + ((row0 >> 4) & 1) * 64 + s_off;
            __builtin_amdgcn_global_load_lds(src + (size_t)(2 * dw) * 512, (lds_ptr)(slot + (2 * dw) * MWP_REQ), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(src + (size_t)(2 * dw + 1) * 512, (lds_ptr)(slot + (2 * dw + 1) * MWP_REQ), 16, 0, 0);
        } else {
            const int64_t r0 = row0 + 4 * dw + lrow;
            __builtin_amdgcn_global_load_lds(P + r0 * ld + lcol, (lds_ptr)(slot + 4 * dw * MW_T), 16, 0, 0);
            __builtin_amdgcn_global_load_lds(P + (r0 + 2) * ld + lcol, (lds_ptr)(slot + (4 * dw + 2) * MW_T), 16, 0, 0);
        }
    };
    auto issue = [&](int s) {
        issue_op(dY, sy, ldY, &ring[s % MW_NR][0][0], s);
        issue_op(X, sx, ldX, &ring[s % MW_NR][1][0], s);
    };
    if (!mfma_wave) {
        for (int s = 0; s < MW_NR - 1 && s < nsteps; ++s) issue(s);
        for (int s = 0; s < nsteps; ++s) {
            if (s + MW_NR - 1 <= nsteps) NF_WAIT_VMCNT(4 * (MW_NR - 2));
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            if (s + MW_NR - 1 < nsteps) issue(s + MW_NR - 1);
        }
        return;
    }
    const int wm = wid >> 1, wn = wid & 1;
    f32x16 acc00 = {0}, acc01 = {0}, acc10 = {0}, acc11 = {0};
    float bs0 = 0.0f, bs1 = 0.0f;
    const int q_off = (i >> 2) * MWP_REQ + (i & 3) * 64 + h * 4;      // the lane's position quad, row h of a pair
    for (int s = 0; s < nsteps; ++s) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const float *sa = &ring[s % MW_NR][0][0], *sb = &ring[s % MW_NR][1][0];
#pragma unroll
        for (int kp = 0; kp < MW_KS / 2; ++kp) {
            float a0, a1, x0, x1;
            if (sy) {           // (the wave's two components of the quad are adjacent: one 8-byte read)
                const f32x2e v = *reinterpret_cast<const f32x2e *>(sa + q_off + 2 * wm + kp * 8);
                a0 = v[0];
                a1 = v[1];
            } else {
                const float *ap = sa + (2 * kp + h) * MW_T + wm * 64 + i;
                a0 = ap[0];
                a1 = ap[32];
            }
            if (sx) {
                const f32x2e v = *reinterpret_cast<const f32x2e *>(sb + q_off + 2 * wn + kp * 8);
                x0 = v[0];
                x1 = v[1];
            } else {
                const float *bp = sb + (2 * kp + h) * MW_T + wn * 64 + i;
                x0 = bp[0];
                x1 = bp[32];
            }
            if (x_relu) { x0 = fmaxf(x0, 0.0f); x1 = fmaxf(x1, 0.0f); }
            bs0 += a0;
            bs1 += a1;
            acc00 = MW_MFMA(a0, x0, acc00);
            acc01 = MW_MFMA(a0, x1, acc01);
            acc10 = MW_MFMA(a1, x0, acc10);
            acc11 = MW_MFMA(a1, x1, acc11);
        }
    }
    float *out = part + ((size_t)bc * ntl + bt) * MW_PART;
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
#pragma unroll
        for (int t_ = 0; t_ < 2; ++t_) {
            const f32x16 &acc = s_ == 0 ? (t_ == 0 ? acc00 : acc01) : (t_ == 0 ? acc10 : acc11);
            const int nn = sx ? 4 * i + 2 * wn + t_ : wn * 64 + 32 * t_ + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rho = (r & 3) + 8 * (r >> 2) + 4 * h;
                const int mm = sy ? 4 * rho + 2 * wm + s_ : wm * 64 + 32 * s_ + rho;
                out[(size_t)mm * MW_T + nn] = sgn * acc[r];
            }
        }
    }
    if (want_bias && wn == 0) {
        bs0 += __shfl_xor(bs0, 32);
        bs1 += __shfl_xor(bs1, 32);
        if (h == 0) {
            out[MW_T * MW_T + (sy ? 4 * i + 2 * wm : wm * 64 + i)] = sgn * bs0;
            out[MW_T * MW_T + (sy ? 4 * i + 2 * wm + 1 : wm * 64 + i + 32)] = sgn * bs1;
        }
    }
}

// ---- 3. reduction + scatter -----------------------------------------------------------------------------------------------------------
// sc (int32): per problem 8 ints [weight offset in `grads`, ld of the weight, bias offset, row-map offset, column-map offset, bias-map
// offset (0: the row map), ...] (maps: slot -> parameter row / column, -1 = padding; offsets into sc itself; destination = weight
// offset + row ld + column, so a map may also carry pre-multiplied positions with ld = 1: the conv weights' own (o, c, ky, kx) layout).  Element e of tile t: sum over the chunks in order;
// written only where the parameter's mask is non-zero (`grads` is zero-filled by the host: the masked entries' gradient).
__global__ void __launch_bounds__(256)
made_wgrad_reduce_kernel(const float *__restrict__ part, float *__restrict__ grads, const unsigned char *__restrict__ mask,
                         const int *__restrict__ wt, const int *__restrict__ sc, int chunks) {
    const int ntl = wt[0], npr = wt[1];
    const int t = blockIdx.y;
    const int *tl = wt + 16 + 8 * npr + 8 * t;
    const int *ps = sc + 8 * tl[0];
    const int m0 = tl[1], n0 = tl[2], want_bias = tl[3];
    const int *rowmap = sc + ps[3], *colmap = sc + ps[4];
    const int nel = MW_T * MW_T + (want_bias ? MW_T : 0);
    // Round 6 (late): 64 elements x 4 chunk lanes per block (lane q sums chunks q, q + 4, ... in four interleaved accumulators, the
    // four lanes are combined in lane order through LDS: a fixed order, deterministic) -- one thread per element walked all chunks as
    // one chain of dependent-latency loads (24 us per call in config 4's training step, 16 blocks per tile).
    __shared__ float sm[4][64];
    const int el = threadIdx.x & 63, q = threadIdx.x >> 6;
    for (int e0 = blockIdx.x * 64; e0 < nel; e0 += gridDim.x * 64) {
        const int e = e0 + el;
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
        if (e < nel) {
            const float *p = part + (size_t)t * MW_PART + e;
            const size_t stride = (size_t)ntl * MW_PART;
            int c = q;
            for (; c + 12 < chunks; c += 16) {
                s0 += p[(size_t)c * stride];
                s1 += p[(size_t)(c + 4) * stride];
                s2 += p[(size_t)(c + 8) * stride];
                s3 += p[(size_t)(c + 12) * stride];
            }
            for (; c < chunks; c += 4) s0 += p[(size_t)c * stride];
        }
        sm[q][el] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (q == 0 && e < nel) {
            const float s = (sm[0][el] + sm[1][el]) + (sm[2][el] + sm[3][el]);
            if (e < MW_T * MW_T) {
                const int row = rowmap[m0 + (e >> 7)], col = colmap[n0 + (e & 127)];
                if (row >= 0 && col >= 0) {
                    const size_t dst = (size_t)ps[0] + (size_t)row * ps[1] + col;
                    if (mask[dst]) grads[dst] = s;
                }
            } else {
                const int *bmap = ps[5] ? sc + ps[5] : rowmap;
                const int row = bmap[m0 + (e - MW_T * MW_T)];
                if (row >= 0) grads[(size_t)ps[2] + row] = s;
            }
        }
        __syncthreads();
    }
}

// ---- 4. the streams of a training step from the parameters as they are NOW ---------------------------------------------------------------
// out[i] = flat[src[i]]: `flat` = [0, the network's parameters flattened one after the other], `src` = the host packer's stream with
// parameter POSITIONS in place of values (made_pack.train_structure; entry 0 = the zero of padded / masked slots).  Under autograd the
// parameters change every step, so the value-independent part (tables, src) is built once and the streams by this gather.
__global__ void __launch_bounds__(256)
pack_gather_kernel(const float *__restrict__ flat, const int *__restrict__ src, float *__restrict__ out, int64_t n) {
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * blockDim.x * 4) {
        if (i + 3 < n) {
            const int4 k = *reinterpret_cast<const int4 *>(src + i);
            *reinterpret_cast<f32x4 *>(out + i) = f32x4{flat[k.x], flat[k.y], flat[k.z], flat[k.w]};
        } else {
            for (int64_t j = i; j < n; ++j) out[j] = flat[src[j]];
        }
    }
}

}  // namespace nf

static int made_bwd_check(int64_t B, int D, int hidden_padded, int mult) {
    if (B < 0 || D < 2 || D > (hidden_padded == 256 ? 256 : 128) || mult < 1) return NF_EINVAL;   // (a 64 KB g_x tile fits next to 256 slots)
    if (hidden_padded != 256 && hidden_padded != 512) return NF_ENOTSUP;
    return NF_OK;
}

// The input-gradient chain of MADE: g_params (B, mult D) -> g_x (B, D), and every layer's output gradient to G ((2 num_blocks + 1) x
// Bp x hidden_padded, Bp = B rounded up to 64; NULL: not stored).  bits: from nf_made_forward_train; blob / table:
// flows/made_pack.pack_made_backward.
extern "C" int nf_made_backward(const void *g_params, const void *bits, void *g_x, void *G, const void *blob, const int32_t *table,
                                int64_t B, int D, int hidden_padded, int mult, nf_stream_t stream) {
    const int rc = made_bwd_check(B, D, hidden_padded, mult);
    if (rc != NF_OK) return rc;
    if (B == 0) return NF_OK;
    if (!g_params || !bits || !g_x || !blob || !table) return NF_EFAULT;       // (G may be NULL: only g_x is wanted)
    hipStream_t st = (hipStream_t)stream;
    const int dp = (D + 31) / 32 * 32;
    if (nf::mf_tr128(B, hidden_padded, dp)) return nf::made_bwd_launch<1, 128>(g_params, bits, g_x, G, blob, table, B, st, dp);    // (as the forward)
    if (hidden_padded == 256) return nf::made_bwd_launch<1>(g_params, bits, g_x, G, blob, table, B, st, dp);
    return nf::made_bwd_launch<2>(g_params, bits, g_x, G, blob, table, B, st, dp);
}

static int made_wgrad_chunk_rows(int64_t Bp, int ntiles) {
    // measured at config 5's layer (50 tiles, B = 65 536; ablation builds with -DNF_MW_SLOTS / -DNF_MW_TILE_MAJOR): 768 / 1024 / 1536 / 2048 / 3072 workgroups
    // = 1.19 / 1.12 / 0.98 / 1.07 / 0.99 ms for this launch + 18 / 24 / 33 / 40 / 55 us for the reduction: three rounds of 512
#ifndef NF_MW_SLOTS
#define NF_MW_SLOTS 1536
#endif
    int64_t want = (NF_MW_SLOTS + ntiles - 1) / ntiles;
    if (want < 1) want = 1;
    int64_t rows = (Bp + want - 1) / want;
    rows = (rows + 63) / 64 * 64;
    if (rows < 1024) {       // few tiles: short chunks multiply the partial-tile traffic and the ring's ramp -- but keep >= ~512 workgroups
        rows = (Bp * ntiles / 512 + 63) / 64 * 64;
        rows = rows > 1024 ? 1024 : (rows < 256 ? 256 : rows);
    }
    if (rows > 8192) rows = 8192;      // (very large batches: bound the length of one float32 accumulation chain; more partial tiles instead)
    return (int)rows;
}

// floats of `part` for nf_made_wgrad (Bp = B rounded up to 64)
extern "C" int64_t nf_made_wgrad_scratch_floats(int64_t B, int ntiles) {
    if (B < 0 || ntiles < 1) return NF_EINVAL;
    const int64_t Bp = (B + 63) / 64 * 64;
    const int rows = made_wgrad_chunk_rows(Bp, ntiles);
    return ((Bp + rows - 1) / rows) * (int64_t)ntiles * nf::MW_PART;
}

// Every weight and bias gradient of the MADE in one launch + one fixed-order reduction: grads (flat, zero-filled by the caller; the
// layout of flows/made_pack.pack_made_backward) receives the sums where `mask` (bytes, the same layout) is non-zero.  gp_pad / x_pad:
// g_params / x with Bp rows and the row length rounded up to 128 (zero padding; the tensors themselves when they already have it).
extern "C" int nf_made_wgrad(const void *gp_pad, const void *x_pad, const void *G, const void *save, void *grads, const void *mask,
                             void *part, const int32_t *wtable, const int32_t *stable, int ntiles, int64_t B, nf_stream_t stream) {
    if (B < 0 || ntiles < 1) return NF_EINVAL;
    if (B == 0) return NF_OK;
    if (!gp_pad || !x_pad || !G || !save || !grads || !mask || !part || !wtable || !stable) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    const int64_t Bp = (B + 63) / 64 * 64;
    const int rows = made_wgrad_chunk_rows(Bp, ntiles);
    const int chunks = (int)((Bp + rows - 1) / rows);
    // tile-major: the workgroups in flight together work on the same row chunk (its operands are shared through L2 / MALL)
#ifndef NF_MW_TILE_MAJOR
#define NF_MW_TILE_MAJOR 1
#endif
    const int tile_major = NF_MW_TILE_MAJOR;
    const dim3 grid = tile_major ? dim3((unsigned)ntiles, (unsigned)chunks) : dim3((unsigned)chunks, (unsigned)ntiles);
    hipLaunchKernelGGL(nf::made_wgrad_kernel, grid, dim3(nf::MW_NT), 0, st, (const float *)gp_pad, (const float *)x_pad,
                       (const float *)G, (const float *)save, (float *)part, (const int *)wtable, rows, Bp, tile_major);
    NF_CHECK_LAUNCH();
    hipLaunchKernelGGL(nf::made_wgrad_reduce_kernel, dim3(64, (unsigned)ntiles), dim3(256), 0, st, (const float *)part, (float *)grads,
                       (const unsigned char *)mask, (const int *)wtable, (const int *)stable, chunks);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// nf_made_wgrad with the hidden operands in the one-pass kernels' scratch order (made_wgrad_pos_kernel): gscratch = the activation
// scratch nf_maf_solve_t left, fscratch = the one nf_maf_inverse_h_[tri_]bits left (both num_layers x positions per row, wave tiles of
// 32 rows); wtable / stable: flows/maf_pack.position_wgrad_tables; grads, mask, part as for nf_made_wgrad.  B a multiple of 64,
// positions a multiple of 128 (-ENOTSUP otherwise: the caller rearranges with nf_maf_scratch_rows and takes nf_made_wgrad).
extern "C" int nf_made_wgrad_pos(const void *gp_pad, const void *x_pad, const void *gscratch, const void *fscratch, void *grads,
                                 const void *mask, void *part, const int32_t *wtable, const int32_t *stable, int ntiles, int64_t B,
                                 int num_layers, int positions, nf_stream_t stream) {
    if (B < 0 || ntiles < 1 || num_layers < 1 || positions < 128) return NF_EINVAL;
    if (B % 64 || positions % 128) return NF_ENOTSUP;
    if (B == 0) return NF_OK;
    if (!gp_pad || !x_pad || !gscratch || !fscratch || !grads || !mask || !part || !wtable || !stable) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    const int rows = made_wgrad_chunk_rows(B, ntiles);
    const int chunks = (int)((B + rows - 1) / rows);
    const int tile_major = NF_MW_TILE_MAJOR;
    const dim3 grid = tile_major ? dim3((unsigned)ntiles, (unsigned)chunks) : dim3((unsigned)chunks, (unsigned)ntiles);
    hipLaunchKernelGGL(nf::made_wgrad_pos_kernel, grid, dim3(nf::MW_NT), 0, st, (const float *)gp_pad, (const float *)x_pad,
                       (const float *)gscratch, (const float *)fscratch, (float *)part, (const int *)wtable, rows, B, tile_major,
                       num_layers, positions);
    NF_CHECK_LAUNCH();
    hipLaunchKernelGGL(nf::made_wgrad_reduce_kernel, dim3(64, (unsigned)ntiles), dim3(256), 0, st, (const float *)part, (float *)grads,
                       (const unsigned char *)mask, (const int *)wtable, (const int *)stable, chunks);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// out (n) = flat[src (n)]: the packed weight streams of nf_made_forward_train / nf_made_backward from the current parameters.
// The same gather from up to 16 parameter tensors in place (round 6; 16 since the last session: a MADE of two or three blocks has 12 / 16): position k of the virtual `flat` = [0, p_0 ..., p_1 ..., ...]
// is found by its segment (first[j] <= k < first[j + 1]; position 0 is the zero) -- no torch.cat of the parameters per module and step.
namespace nf {
struct GatherSegs {
    const float *p[16];
    int first[17];           // first[0] = 1; first[j + 1] = first[j] + numel(p_j)
    int np;
};
__global__ void __launch_bounds__(256)
pack_gather_multi_kernel(GatherSegs g, const int *__restrict__ src, float *__restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = src[i];
        float v = 0.0f;
        if (k > 0) {
            int j = 0;
#pragma unroll
            for (int q = 1; q < 16; ++q)
                if (q < g.np && k >= g.first[q]) j = q;
            v = g.p[j][k - g.first[j]];
        }
        out[i] = v;
    }
}
}  // namespace nf

extern "C" int nf_pack_gather_multi(const void *const *params, const int64_t *numels, int n_params, const int32_t *src, void *out,
                                    int64_t n, nf_stream_t stream) {
    if (n < 0 || n_params < 1 || n_params > 16) return NF_EINVAL;
    if (n == 0) return NF_OK;
    if (!params || !numels || !src || !out) return NF_EFAULT;
    nf::GatherSegs g;
    int64_t first = 1;
    for (int j = 0; j < 16; ++j) { g.p[j] = nullptr; g.first[j] = 0; }
    for (int j = 0; j < n_params; ++j) {
        if (!params[j] || numels[j] < 0 || first + numels[j] >= (1ll << 31)) return params[j] ? NF_EINVAL : NF_EFAULT;
        g.p[j] = (const float *)params[j];
        g.first[j] = (int)first;
        first += numels[j];
    }
    g.first[n_params] = (int)first;
    g.np = n_params;
    hipLaunchKernelGGL(nf::pack_gather_multi_kernel, dim3(nf::grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, g, (const int *)src,
                       (float *)out, n);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// The same gather for up to 32 MODULES of one structure per launch (round 6, last session; blockIdx.y = module): the conditioners of a
// Glow level's K blocks share `src`, so their packed streams for a training step are one launch instead of K (96 launches of ~7 us per
// step of BASELINE configs[3]).  Up to 8 parameter tensors per module.
namespace nf {
constexpr int PGB_MOD = 32, PGB_NP = 8;
struct GatherBatch {
    const float *p[PGB_MOD][PGB_NP];
    float *out[PGB_MOD];
    int first[PGB_NP + 1];
    int np;
};
__global__ void __launch_bounds__(256)
pack_gather_batch_kernel(GatherBatch g, const int *__restrict__ src, int64_t n) {
    const int m = blockIdx.y;
    float *__restrict__ out = g.out[m];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = src[i];
        float v = 0.0f;
        if (k > 0) {
            int j = 0;
#pragma unroll
            for (int q = 1; q < PGB_NP; ++q)
                if (q < g.np && k >= g.first[q]) j = q;
            v = g.p[m][j][k - g.first[j]];
        }
        out[i] = v;
    }
}
}  // namespace nf

// params: HOST array of n_modules x n_params device pointers (module-major); numels: the n_params element counts (the same for every
// module); outs: HOST array of n_modules device pointers to n floats each.
extern "C" int nf_pack_gather_batch(const void *const *params, const int64_t *numels, int n_params, const int32_t *src, void *const *outs,
                                    int64_t n, int n_modules, nf_stream_t stream) {
    if (n < 0 || n_modules < 0 || n_params < 1 || n_params > nf::PGB_NP) return NF_EINVAL;
    if (n == 0 || n_modules == 0) return NF_OK;
    if (!params || !numels || !src || !outs) return NF_EFAULT;
    nf::GatherBatch g = {};
    int64_t first = 1;
    for (int j = 0; j < n_params; ++j) {
        if (numels[j] < 0 || first + numels[j] >= (1ll << 31)) return NF_EINVAL;
        g.first[j] = (int)first;
        first += numels[j];
    }
    g.first[n_params] = (int)first;
    g.np = n_params;
    for (int m0 = 0; m0 < n_modules; m0 += nf::PGB_MOD) {
        const int mm = n_modules - m0 < nf::PGB_MOD ? n_modules - m0 : nf::PGB_MOD;
        for (int m = 0; m < mm; ++m) {
            if (!outs[m0 + m]) return NF_EFAULT;
            g.out[m] = (float *)outs[m0 + m];
            for (int j = 0; j < n_params; ++j) {
                if (!params[(size_t)(m0 + m) * n_params + j]) return NF_EFAULT;
                g.p[m][j] = (const float *)params[(size_t)(m0 + m) * n_params + j];
            }
        }
        hipLaunchKernelGGL(nf::pack_gather_batch_kernel, dim3(nf::grid_for(n, 256), (unsigned)mm), dim3(256), 0, (hipStream_t)stream, g,
                           (const int *)src, n);
        NF_CHECK_LAUNCH();
    }
    return NF_OK;
}

extern "C" int nf_pack_gather(const void *flat, const int32_t *src, void *out, int64_t n, nf_stream_t stream) {
    if (n < 0) return NF_EINVAL;
    if (n == 0) return NF_OK;
    if (!flat || !src || !out) return NF_EFAULT;
    if ((((uintptr_t)src) | ((uintptr_t)out)) & 15) return NF_EINVAL;
    hipLaunchKernelGGL(nf::pack_gather_kernel, dim3(nf::grid_for((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, (const float *)flat,
                       (const int *)src, (float *)out, n);
    NF_CHECK_LAUNCH();
    return NF_OK;
}
