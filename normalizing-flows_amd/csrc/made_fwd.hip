// made_fwd.hip -- ONE pass of MADE (nets/made.py:296-304: initial MaskedLinear, residual blocks :196-214, final MaskedLinear; every
// linear is F.linear(x, weight * mask, bias), :80-81) as ONE launch, with the element-wise affine transform of
// MaskedAffineAutoregressive.forward (flows/affine/autoregressive.py:24-27 -> :101-110) as its epilogue: the single-pass direction
// of MAF (BASELINE configs[4]: 10 layers, d = 128, hidden 512), which round 3 still ran as six library GEMMs + element-wise passes
// per layer.  Exact-fp32 MFMA; bound: the fp32 MFMA rate on the MASKED work (53 % of the dense blocks at config 5).
//
// Geometry (host packer: flows/made_pack.py)
//   * a workgroup of 8 waves owns 64 rows for the WHOLE network; hidden slots = units sorted by degree, Hp = 256 NSB (NSB = 1, 2);
//     every GEMM is transposed, Out^T[units x 64 rows] = W . Act^T, on v_mfma_f32_32x32x2_f32: lane (hh = lane >> 5, n = lane & 31)
//     holds sample n of a 32-sample block; its 16 accumulator registers are output rows 8 q + 4 hh + i of a 32-row block.
//   * the pre-activations live in ACCUMULATOR registers for the whole network (h: the residual stream, t: a block's inner layer:
//     2 x 32 registers per wave and tensor); what the NEXT layer contracts over -- relu(h), relu(t), raw h for the final layer -- is
//     published to LDS in B-operand order act[k / 8][hh][64 samples][4] (one ds_write_b128 per register quad: the natural order of
//     the accumulators IS that order; one ds_read_b128 = the B values of four k-steps), 128 KB at Hp = 512.
//   * work items: NSB = 2: wave w owns hidden row-blocks {w, 15 - w} for both sample blocks; NSB = 1 and the final layer: row-blocks
//     {w, 7 - w}, the first for sample block 0, the second for sample block 1.  With the units sorted by degree row-block rb needs
//     the k-groups [0, nkg(rb)), nkg ~ 4 (rb + 1): the pairing gives every wave the same number of MFMAs (68 k-groups per hidden
//     layer at config 5), so the waves meet at the layer boundaries without waiting.
//   * weights: every WAVE walks one contiguous stream for the whole network (its items in consumption order; per item a bias group
//     of 4 KB, then 1 KB per k-group, lane = its own 16 bytes) through an 8-entry register ring of plain global loads that is
//     requested 8 entries (8 KB) ahead of the MFMAs -- across item, layer and TILE boundaries
//     (the stream ends with a copy of its first 8 entries; the workgroups are persistent over the 64-row tiles): no LDS, no barrier
//     and no start-up bubble for the weights anywhere.  No LDS-DMA in this kernel, so the compiler's own counted vmcnt waits are
//     exact.
//   * two LDS-only barriers per layer boundary (all reads of the old activations done | new ones published).
//   * x tile: 64 rows x Dp features in the same B-operand order (32 KB): B operand of the initial layer and the x of the affine
//     epilogue, which overwrites it in place with z = scale x + shift; the tile leaves with 16-byte stores.
// Algorithmic work per row: 2 (D H + 2 NB H^2 + H mult D) FLOP dense (2.49 MFLOP at config 5), 1.33 MFLOP masked; HBM: 4 D in,
// 4 D + 4 out (affine) resp. 4 mult D out (raw parameters).
#include "mlp_tile.hpp"

namespace nf {

// EPI 0: z = scale x + shift, logdet = sum log scale (autoregressive.py:101-110, :124-128: rows 2 f = unconstrained scale, 2 f + 1
// = shift);  EPI 1: the raw MADE output (B, mult D) for the callers that apply another element-wise transform;  EPI 2: the
// autoregressive rational-quadratic spline of neural_spline/autoregressive.py:94-134 (density direction: one MADE pass, then
// utils/splines.py:16-219 element-wise with 8 bins and linear tails) on the final layer's accumulators -- the final layer runs in
// groups of four features whose rows are packed so that a lane holds the 2 x 24 parameters of two features (the layout of
// nsf_wide.hip; mlp_tile.hpp mf_final_item), the spline runs in registers (rqs_regs), x is read from the tile and y written into it.
// EPI 3: EPI 1 under autograd (core.py:87-102 through a MADE): every layer's pre-activations are also written row-major to
// save[l][Bp][Hp] (l = 0: the initial layer's h; 2 b + 1: block b's inner t; 2 b + 2: its output h) and the signs of what a ReLU
// follows to bits[tile][2 b | 2 b + 1][item][512 lanes] -- the operands of made_bwd.hip.
template <int NSB, int EPI, int TR = MF_ROWS>
__global__ void __launch_bounds__(64 * MF_NW, 1)
made_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, float *__restrict__ logdet, const float *__restrict__ blob,
                const int *__restrict__ table, int64_t B, int acc_mode, RqsParams<float> p, float *__restrict__ save,
                unsigned *__restrict__ bits, int64_t Bp) {
    static_assert(TR == 64 || (TR == 128 && NSB == 1 && EPI == 3), "128-row tiles: the 256-slot training forward only (mf_tr128)");
    constexpr int NSH = TR / 64;             // sample blocks per HALF of a tile (a 256-slot item covers one half, a final-layer item too)
    constexpr int NS = NSB * NSH;            // sample blocks per hidden work item
    constexpr int HP = 256 * NSB;
    constexpr int HRB = 8 * NSB;             // hidden row-blocks
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *acts = lds;                                  // [HRB * 4 k-groups][2][TR][4]
    float *xreg = lds + (size_t)HRB * 4 * 8 * TR;       // [Dp / 8][2][TR][4]
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, hh = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = table[0], Dp = table[1], NB = table[5], NFB = table[7], nrounds = table[8], nitems = table[10];
    const int MD = table[12] ? table[12] : table[6] * D;       // raw output row length: mult D for a MADE, out_features for a ResidualNet
    const int ldx = table[14] ? table[14] : D;                 // row stride of x (the conv path hands over 128-padded rows)
    const int plain = table[13];     // 1: a plain MLP  x -> W0 -> relu -> W1 -> relu -> Wf  (NB = 1 without the block's second linear and
                                     // its residual; EPI 1 / 3 only): the 3x3 -> 1x1 -> 3x3 conv conditioner over pixel rows (conv_rows.hip)
    const int *items = table + MF_HDR + w * nitems * 2;       // [nitems][nkg, rb]
    const float *stream = blob + table[16 + w];
    const int rbs[2] = {w, HRB - 1 - w};                      // (the packer's wave_items: the hidden row-blocks of this wave)
    const int sb0s[2] = {0, NSB == 2 ? 0 : NSH};
    const int lane_b = (TR * hh + n) * 4;       // the lane's offset inside a k-group of activations (sample block 0)
    const int64_t ntiles = (B + TR - 1) / TR;
    MfRing ring;
    mf_ring_start(ring, stream, lane);

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row0 = tile * TR;
        const int nrows = (int)((B - row0) < TR ? (B - row0) : TR);
        ring.ap = stream + lane * 4;            // (the ring already holds the stream's first entries: the wrap-around copy)
        int tq = tid;                           // per-tile address arithmetic from an index the compiler cannot hoist out of the tile loop
        asm volatile("" : "+v"(tq));            // (round 6, as in nsf_wide.hip: hoisted, those values stayed live across the products)
        asm volatile("" : "+v"(ring.ap));       // (likewise the restarted stream's first request addresses: four 64-bit pairs)
        // ---- x tile -> LDS (B-operand order; rows beyond the batch and features beyond D are zero) ---------------------------------
        {
            const int r = tq & (TR - 1), cg = tq / TR;
            const float *xr = x + (row0 + r) * ldx;
#pragma unroll 1
            for (int c = cg; c < Dp / 4; c += 64 * MF_NW / TR) {
                f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
                if (r < nrows && 4 * c < D) {           // (Dp rounds D up to 32: the last chunks may lie wholly beyond the row)
                    if ((D & 3) == 0) v = *reinterpret_cast<const f32x4 *>(xr + 4 * c);
                    else
#pragma unroll
                        for (int i = 0; i < 4; ++i) if (4 * c + i < D) v[i] = xr[4 * c + i];
                }
                *reinterpret_cast<f32x4 *>(xreg + ((size_t)c * TR + r) * 4) = v;
            }
        }
        f32x16 h[2][NS], t[2][NS];
        MF_BARRIER();
        // ---- initial layer: h = b0 + W0 x ----------------------------------------------------------------------------------------------
#pragma unroll
        for (int s = 0; s < 2; ++s) mf_item<NS, false, TR>(ring, items[2 * s], xreg + lane_b + 128 * sb0s[s], h[s]);
        float *stile = nullptr;
        unsigned *btile = nullptr;
        if constexpr (EPI == 3) {
            stile = save + (size_t)row0 * HP;
            btile = bits + ((size_t)tile * 2 * NB * 2) * 512 + tq;
#pragma unroll
            for (int s = 0; s < 2; ++s) mf_save_rows<NS, true>(stile, HP, nrows, rbs[s], sb0s[s], hh, n, h[s]);
        }
        // ---- residual blocks (nets/made.py:196-214): t = b1 + W1 relu(h);  h += b2 + W2 relu(t) -------------------------------
        for (int b = 0; b < NB; ++b) {
            MF_BARRIER();        // (b > 0: every wave has finished reading relu(t) of the previous block)
#pragma unroll
            for (int s = 0; s < 2; ++s) mf_publish<NS, true, TR>(acts, rbs[s], sb0s[s], hh, n, h[s]);
            if constexpr (EPI == 3)
#pragma unroll
                for (int s = 0; s < 2; ++s) btile[((size_t)(2 * b) * 2 + s) * 512] = mf_sign_bits<NS>(h[s]);
            MF_BARRIER();
#pragma unroll
            for (int s = 0; s < 2; ++s) mf_item<NS, false, TR>(ring, items[2 * (2 + 4 * b + s)], acts + lane_b + 128 * sb0s[s], t[s]);
            MF_BARRIER();
#pragma unroll
            for (int s = 0; s < 2; ++s) mf_publish<NS, true, TR>(acts, rbs[s], sb0s[s], hh, n, t[s]);
            if constexpr (EPI == 3)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    btile[((size_t)(2 * b + 1) * 2 + s) * 512] = mf_sign_bits<NS>(t[s]);
                    mf_save_rows<NS, true>(stile + (size_t)(2 * b + 1) * Bp * HP, HP, nrows, rbs[s], sb0s[s], hh, n, t[s]);
                }
            MF_BARRIER();
            if (plain) break;            // (relu(t) is published: the final layer's input)
#pragma unroll
            for (int s = 0; s < 2; ++s) mf_item<NS, true, TR>(ring, items[2 * (4 + 4 * b + s)], acts + lane_b + 128 * sb0s[s], h[s]);
            if constexpr (EPI == 3)
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    mf_save_rows<NS, true>(stile + (size_t)(2 * b + 2) * Bp * HP, HP, nrows, rbs[s], sb0s[s], hh, n, h[s]);
        }
        // ---- final layer on the RAW block output (:303-304) + epilogue -------------------------------------------------------------
        if (!plain) {
            MF_BARRIER();
#pragma unroll
            for (int s = 0; s < 2; ++s) mf_publish<NS, false, TR>(acts, rbs[s], sb0s[s], hh, n, h[s]);
            MF_BARRIER();
        }
        if constexpr (EPI == 2) {
            const int G = NFB, nfi = nrounds;                 // (the header slots of the block variants: groups, final items per wave)
            const int *fit = items + 2 * (2 + 4 * NB);
            float ldt[4][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}};
#pragma nounroll
            for (int j = 0; j < nfi; ++j) {
                const int g = fit[2 * j + 1];
                if (g < 0) continue;
                f32x16 o[3][2];
                mf_final_item<MF_ROWS>(ring, fit[2 * j], acts + lane_b, o);
                float lsum[2] = {0.0f, 0.0f};
#pragma unroll
                for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                    for (int f = 0; f < 2; ++f) {
                        float prm[24];
#pragma unroll
                        for (int v = 0; v < 24; ++v) prm[v] = o[(24 * f + v) >> 4][sb][(24 * f + v) & 15];
                        const int tf = 4 * g + 2 * hh + f;
                        const bool valid = tf < D;
                        const int col = valid ? tf : 0;
                        float *xp = xreg + ((size_t)(col >> 2) * 64 + 32 * sb + n) * 4 + (col & 3);
                        float yv, lad;
                        rqs_regs_h<false>(p, *xp, prm, yv, lad);      // (round 5: binary bin descent; round 6: its first level before the knots exist)
                        if (valid) {
                            *xp = yv;
                            lsum[sb] += lad;
                        }
                        __builtin_amdgcn_sched_barrier(0);      // one evaluation at a time: interleaved, the four cost 7 spilled VGPRs at Hp = 512
                    }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    ldt[q][0] = j == q ? lsum[0] : ldt[q][0];
                    ldt[q][1] = j == q ? lsum[1] : ldt[q][1];
                }
            }
            MF_BARRIER();                      // every wave is done with the activations: their region now holds the partial sums
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j >= nfi) break;
                const int g = fit[2 * j + 1];
                if (g >= 0) {
#pragma unroll
                    for (int sb = 0; sb < 2; ++sb) {
                        const float v = ldt[j][sb] + __shfl_xor(ldt[j][sb], 32);
                        if (hh == 0) acts[g * 64 + 32 * sb + n] = v;
                    }
                }
            }
            MF_BARRIER();
            if (tq < nrows) {
                float v = 0.0f;
                for (int g = 0; g < G; ++g) v += acts[g * 64 + tq];      // fixed order: deterministic
                ld_store(logdet + row0 + tq, v, acc_mode);
            }
            const int r = tq & 63, cg = tq >> 6;
            float *yr = y + (row0 + r) * D;
            if (r < nrows)
                for (int c = cg; 4 * c < D; c += MF_NW) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(xreg + ((size_t)c * 64 + r) * 4);
                    if ((D & 3) == 0) *reinterpret_cast<f32x4 *>(yr + 4 * c) = v;
                    else
#pragma unroll
                        for (int i = 0; i < 4; ++i) if (4 * c + i < D) yr[4 * c + i] = v[i];
                }
            MF_BARRIER();                      // the next tile overwrites the x tile and the activations
            continue;
        }
        float ldsum[2] = {0.0f, 0.0f};
        for (int rd = 0; rd < nrounds; ++rd) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int *it = items + 2 * (2 + 4 * NB + 2 * rd + s);
                const int fb = it[1];                         // sample half s (TR = 64: sample block s)
                if (fb >= 0) {
                    f32x16 o[NSH];
                    mf_item<NSH, false, TR>(ring, it[0], acts + lane_b + 128 * NSH * s, o);
                    if constexpr (EPI == 0) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int f0 = 16 * fb + 4 * q + 2 * hh;
                            float *xp = xreg + ((size_t)((2 * fb + (q >> 1)) * 2 + (q & 1)) * 64 + 32 * s + n) * 4 + 2 * hh;
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                if (f0 + e < D) {
                                    const float scale = 1.0f / (1.0f + __expf(-(o[0][4 * q + 2 * e] + 2.0f))) + 1e-3f;
                                    xp[e] = scale * xp[e] + o[0][4 * q + 2 * e + 1];
                                    ldsum[s] += __logf(scale);
                                }
                            }
                        }
                    } else {
#pragma unroll
                        for (int ss = 0; ss < NSH; ++ss) {
                            const int rl = 32 * (NSH * s + ss) + n;
                            const int64_t r = row0 + rl;
                            if (rl < nrows) {
                                float *yp = y + r * (int64_t)MD + 32 * fb + 4 * hh;
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    const int c = 32 * fb + 8 * q + 4 * hh;
                                    if ((MD & 3) == 0) {
                                        if (c < MD) *reinterpret_cast<f32x4 *>(yp + 8 * q) = f32x4{o[ss][4 * q], o[ss][4 * q + 1], o[ss][4 * q + 2], o[ss][4 * q + 3]};
                                    } else {
#pragma unroll
                                        for (int i = 0; i < 4; ++i) if (c + i < MD) yp[8 * q + i] = o[ss][4 * q + i];
                                    }
                                }
                            }
                        }
                    }
                }
            }
        }
        MF_BARRIER();                          // every wave is done with the activations (and, EPI 0, has written its z values)
        if constexpr (EPI == 0) {
            // per-sample log-det: the lane-halves' sums, then the 8 row-blocks' in a FIXED order (deterministic)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const float v = ldsum[s] + __shfl_xor(ldsum[s], 32);
                const int fb = (s == 0 ? w : 7 - w);
                if (hh == 0) acts[fb * 64 + 32 * s + n] = v;
            }
            MF_BARRIER();
            if (tq < nrows) {
                float v = 0.0f;
                for (int fb = 0; fb < 8; ++fb) v += acts[fb * 64 + tq];
                ld_store(logdet + row0 + tq, v, acc_mode);
            }
            const int r = tq & 63, cg = tq >> 6;
            float *yr = y + (row0 + r) * D;
            if (r < nrows)
                for (int c = cg; 4 * c < D; c += MF_NW) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(xreg + ((size_t)c * 64 + r) * 4);
                    if ((D & 3) == 0) *reinterpret_cast<f32x4 *>(yr + 4 * c) = v;
                    else
#pragma unroll
                        for (int i = 0; i < 4; ++i) if (4 * c + i < D) yr[4 * c + i] = v[i];
                }
            MF_BARRIER();                      // the next tile overwrites the x tile and the activations
        }
    }
}

template <int NSB, int EPI, int TR = MF_ROWS>
static int made_fwd_launch(const void *x, void *y, void *logdet, const void *blob, const int32_t *table, int64_t B, int acc, hipStream_t st,
                           const RqsParams<float> &p = RqsParams<float>(), void *save = nullptr, void *bits = nullptr, int table_dp = 128) {
    const int64_t ntiles = (B + TR - 1) / TR;
    const int grid = (int)(ntiles < 256 ? ntiles : 256);        // persistent: one workgroup per CU (160 KB of LDS at Hp = 512)
    const int xfloats = table_dp > 128 ? 2 * MF_XFLOATS : MF_XFLOATS;       // x tile: 64 rows x Dp (<= 256 next to 256 hidden slots)
    const size_t lds = sizeof(float) * ((size_t)8 * NSB * 4 * 8 * TR + xfloats);       // (TR = 128: 128 rows x Dp <= 64 = the same 32 KB)
    static LdsOptIn opted;
    if (opt_in_lds(reinterpret_cast<const void *>(&made_fwd_kernel<NSB, EPI, TR>), lds, opted) != NF_OK) return NF_ENOTSUP;
    hipLaunchKernelGGL((made_fwd_kernel<NSB, EPI, TR>), dim3((unsigned)grid), dim3(64 * MF_NW), lds, st, (const float *)x, (float *)y,
                       (float *)logdet, (const float *)blob, (const int *)table, B, acc, p, (float *)save, (unsigned *)bits,
                       (B + MF_ROWS - 1) / MF_ROWS * MF_ROWS);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

}  // namespace nf

static int made_fwd_check(int64_t B, int D, int hidden_padded, int mult, int dmax = 128) {
    if (B < 0 || D < 2 || D > dmax || mult < 1) return NF_EINVAL;
    if (hidden_padded != 256 && hidden_padded != 512) return NF_ENOTSUP;
    return NF_OK;
}

// MaskedAffineAutoregressive.forward in one launch (affine/autoregressive.py:24-27, :101-110 over nets/made.py:296-304).
extern "C" int nf_made_forward_affine(const void *x, void *y, void *logdet, const void *blob, const int32_t *table, int64_t B, int D,
                                      int hidden_padded, int acc, nf_stream_t stream) {
    const int rc = made_fwd_check(B, D, hidden_padded, 2);
    if (rc != NF_OK) return rc;
    if (acc < NF_LD_SUB || acc > NF_LD_ADD) return NF_EINVAL;
    if (B == 0) return NF_OK;
    if (!x || !y || !logdet || !blob || !table) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    if (hidden_padded == 256) return nf::made_fwd_launch<1, 0>(x, y, logdet, blob, table, B, acc, st);
    return nf::made_fwd_launch<2, 0>(x, y, logdet, blob, table, B, acc, st);
}

// MADE.forward (nets/made.py:296-304) in one launch: params (B, mult D), rows mult f + p as the reference's final layer orders them.
extern "C" int nf_made_forward(const void *x, void *params, const void *blob, const int32_t *table, int64_t B, int D,
                               int hidden_padded, int mult, nf_stream_t stream) {
    const int rc = made_fwd_check(B, D, hidden_padded, mult);
    if (rc != NF_OK) return rc;
    if (B == 0) return NF_OK;
    if (!x || !params || !blob || !table) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    if (hidden_padded == 256) return nf::made_fwd_launch<1, 1>(x, params, nullptr, blob, table, B, NF_LD_WRITE, st);
    return nf::made_fwd_launch<2, 1>(x, params, nullptr, blob, table, B, NF_LD_WRITE, st);
}

// MaskedPiecewiseRationalQuadraticAutoregressive.forward (neural_spline/autoregressive.py:94-134, density direction of the
// autoregressive spline layer; wrapper.py:241-245) in one launch: MADE + the element-wise spline with 8 bins and linear tails.
// blob / table: flows/made_pack.pack_made_forward(made, 23, spline=True).
extern "C" int nf_made_forward_spline(const void *x, void *y, void *logdet, const void *blob, const int32_t *table, int64_t B, int D,
                                      int hidden_padded, int acc, double tail_bound, double min_bin_width, double min_bin_height,
                                      double min_derivative, nf_stream_t stream) {
    const int rc = made_fwd_check(B, D, hidden_padded, 23);
    if (rc != NF_OK) return rc;
    if (acc < NF_LD_SUB || acc > NF_LD_ADD) return NF_EINVAL;
    if (min_bin_width * nf::F_K > 1.0 || min_bin_height * nf::F_K > 1.0) return NF_EINVAL;
    if (B == 0) return NF_OK;
    if (!x || !y || !logdet || !blob || !table) return NF_EFAULT;
    auto p = nf::make_rqs_params<float>(nf::F_K, NF_TAILS_LINEAR, tail_bound, 0, 1, 0, 1, min_bin_width, min_bin_height, min_derivative, 1.0);
    hipStream_t st = (hipStream_t)stream;
    if (hidden_padded == 256) return nf::made_fwd_launch<1, 2>(x, y, logdet, blob, table, B, acc, st, p);
    return nf::made_fwd_launch<2, 2>(x, y, logdet, blob, table, B, acc, st, p);
}

// 128-row tiles of the 256-slot training kernels on (1, default) / off (0); returns the previous setting.  A forward and its backward
// must run under the same setting (the ReLU-sign words are indexed by tile).
extern "C" int nf_config_made_tr128(int on) {
    const int prev = nf::mf_tr128_switch();
    nf::mf_tr128_switch() = on ? 1 : 0;
    return prev;
}

// MADE.forward under autograd: nf_made_forward + the operands of nf_made_backward / nf_made_wgrad (csrc/made_bwd.hip).  Bp = B rounded
// up to 64; save: (2 num_blocks + 1) x Bp x hidden_padded floats; bits: (Bp / 64) x 2 num_blocks x 2 x 512 dwords.
extern "C" int nf_made_forward_train(const void *x, void *params, void *save, void *bits, const void *blob, const int32_t *table,
                                     int64_t B, int D, int hidden_padded, int mult, nf_stream_t stream) {
    const int rc = made_fwd_check(B, D, hidden_padded, mult, hidden_padded == 256 ? 256 : 128);   // (a 64 KB x tile fits next to 256 slots)
    if (rc != NF_OK) return rc;
    if (B == 0) return NF_OK;
    if (!x || !params || !save || !bits || !blob || !table) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    const int dp = (D + 31) / 32 * 32;
    if (nf::mf_tr128(B, hidden_padded, dp))      // (256 slots, <= 64 features, whole 128-row tiles: nf_made_backward decides the same way)
        return nf::made_fwd_launch<1, 3, 128>(x, params, nullptr, blob, table, B, NF_LD_WRITE, st, nf::RqsParams<float>(), save, bits, dp);
    if (hidden_padded == 256) return nf::made_fwd_launch<1, 3>(x, params, nullptr, blob, table, B, NF_LD_WRITE, st, nf::RqsParams<float>(), save, bits, dp);
    return nf::made_fwd_launch<2, 3>(x, params, nullptr, blob, table, B, NF_LD_WRITE, st, nf::RqsParams<float>(), save, bits, dp);
}
