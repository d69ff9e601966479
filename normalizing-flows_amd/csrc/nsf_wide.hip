// nsf_wide.hip -- CoupledRationalQuadraticSpline (normflows/flows/neural_spline/wrapper.py:14-85 -> nsf/coupling.py:71-128, 150-164,
// 221-253, 329-362 over nets/resnet.py:37-50, 92-104 and utils/splines.py:16-219) as ONE launch for the shapes beyond the benchmark
// kernel's (rqs_fused.hip: D <= 64, hidden <= 128): up to 128 features and 512 hidden units, 4 / 8 / 16 bins, linear tails, float32.  Round 3
// ran these as library GEMMs + nf_rqs_coupling on a materialised conditioner output (2944 B per row and 32 transform features).
//
// The engine is mlp_tile.hpp's (made_fwd.hip describes it): 8 waves own 64 rows for the whole layer, pre-activations in accumulator
// registers, the next layer's B operand in LDS in MFMA order, one contiguous weight stream per wave through a register ring,
// persistent over the tiles.  What is specific here (host packer: flows/nsf_wide_pack.py):
//   * the x tile is held with its columns SORTED (B-operand order over POSITIONS: identity feature i at position i < PI, transform
//     feature j at PI + j; PI = the identity count rounded up to 32, zeros at the padding positions): the initial layer contracts
//     over the first PI positions only -- the conditioner sees the identity features alone (nsf/coupling.py:83-84), so a NaN / inf in
//     a transform column passes through its own element (utils/splines.py:40-41) without reaching the conditioner through a zero
//     weight; the LU matrix is packed in position order;
//   * the final layer runs in GROUPS of four transform features = 3 row-blocks for both sample blocks (6 accumulators); the packed
//     row order makes a lane's 48 accumulator values of a sample block the 2 x 24 parameter lists (8 widths, 8 heights, 7
//     derivatives, pad) of features 4 g + 2 hh + {0, 1}: the spline runs on them in registers (fused_common.hpp rqs_regs: branch-
//     free, hardware transcendentals, log2(e) / sqrt(hidden) folded into the packed rows), reads x from the tile and writes y into it;
//   * the identity features go through the batch-shared spline (nsf/coupling.py:221-253) from knot tables staged in the activation
//     region while it is free: AFTER the network in the density direction (the conditioner sees the raw features, :83-92), BEFORE
//     it in the sampling direction (:112-118);
//   * per-row log-det: the lane-halves' sums, the groups' and the identity columns' partials through LDS in a fixed order.
// Bound: fp32 MFMA.  FLOP per row: 2 (nI H + 4 H^2 + 23 nT H) algorithmic (executed: full-width initial layer, padded hidden
// units, 24 rows per feature); HBM: 8 D + 4 bytes per row.
#include "mlp_tile.hpp"

namespace nf {

constexpr int nw_tabw(int KB) { return 3 * (KB + 1); }          // floats per identity feature: cumw[K + 1] | cumh[K + 1] | deriv[K + 1]
constexpr int nw_tab_floats(int KB) { return KB == 16 ? 3328 : 2048; }   // table region at the start of the activation region (64 features)

// knot tables of the batch-shared spline, once per parameter version (same arithmetic as rqs_fused.hip's pack_tables_kernel)
__global__ void nsf_wide_tables_kernel(const float *__restrict__ uw, const float *__restrict__ uh, const float *__restrict__ ud,
                                       float *__restrict__ tab, int nI, RqsParams<float> p) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nI) return;
    const int K = p.K;
    const float *wj = uw + j * K, *hj = uh + j * K, *dj = ud + j * (K - 1);
    auto wacc = [=](int k) { return wj[k]; };
    auto hacc = [=](int k) { return hj[k]; };
    auto dacc = [=](int k) { return dj[k]; };
    rqs_build_table<float>(p, wacc, hacc, dacc, tab + j * 3 * (K + 1));
}

// float index of POSITION `pos` of row `row` (0 .. TR - 1) of the tile in B-operand order [pos / 4][row][4]
template <int TR>
__device__ __forceinline__ int nw_xidx(int pos, int row) { return ((pos >> 2) * TR + row) * 4 + (pos & 3); }

// batch-shared spline on the identity columns of the tile, in place; thread = (row n = tid % TR, feature residue tid / TR)
template <bool INV, int TR, int KB>
__device__ __forceinline__ float nw_identity(float *xreg, const float *tabs, const RqsParams<float> &p, int nI, int tid) {
    const int n = tid % TR;
    float ld = 0.0f;
#pragma unroll 1
    for (int i = tid / TR; i < nI; i += 64 * MF_NW / TR) {
        float *xp = xreg + nw_xidx<TR>(i, n);                 // identity feature i sits at position i
        float y, lad;
        rqs_table_fast<INV, KB>(p, *xp, tabs + i * nw_tabw(KB), y, lad);
        *xp = y;
        ld += lad;
    }
    return ld;
}

// The adjacent LULinearPermute (mixing.py:535-563) as one dense product on the tile, in place: every wave computes its 32 output
// columns of one sample block from the tile (B operand as it stands), all waves meet, then the columns are written back in B-operand
// order (the accumulator quads ARE 16-byte groups of that order).
template <int TR>
__device__ __forceinline__ void nw_lu_stage(MfRing &ring, const int *it, float *xreg, int lane_b, int hh, int n) {
    constexpr int NSL = TR / 64;           // sample blocks per LU item: 4 row-blocks x TR / 32 sample blocks over 8 waves
    const int rb = it[1];
    f32x16 o1[NSL];
    if (rb >= 0) mf_item<NSL, false, TR>(ring, it[0], xreg + lane_b + 128 * it[2], o1);
    MF_BARRIER();
    if (rb >= 0) mf_publish<NSL, false, TR>(xreg, rb, it[2], hh, n, o1);
    MF_BARRIER();
}

// NHI hidden items per wave with NS sample blocks each, TR rows per tile: (1, 2, 128) Hp = 128 -- the activations of a 128-wide
// network leave room for 128-row tiles: half the barriers, tile prologues and weight-stream traffic per row --, (1, 2, 64) Hp = 256,
// (2, 2, 64) Hp = 512.
// DIR 0: density direction = prqct.forward (nsf/coupling.py:71-98); DIR 1: sampling direction = prqct.inverse (:100-128).
// LU: with the adjacent LULinearPermute -- applied BEFORE the coupling layer in the density direction (core.py:193-195 walks the
// flows backwards: the LU layer behind a coupling layer comes first), AFTER it in the sampling direction (core.py:177-179).
// KB bins (round 5: 4 and 16 beside 8): a lane-half's 48 accumulator values per sample block are FPL = 16 / KB whole parameter lists of
// MP = 3 KB slots, so a final-layer group holds FPG = 2 FPL transform features (8 / 4 / 2).
template <int NHI, int NS, int DIR, bool LU, int TR, int KB = F_K>
__global__ void __launch_bounds__(64 * MF_NW, 1)
nsf_wide_kernel(const float *__restrict__ x, float *__restrict__ y, float *__restrict__ logdet, const float *__restrict__ blob,
                const int *__restrict__ table, const float *__restrict__ tabs, const float *__restrict__ lu_lad, int64_t B,
                int acc_mode, RqsParams<float> p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, hh = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int D = table[0], Dp = table[1], Hp = table[3], NB = table[4], nI = table[5], nT = table[6], par_i = table[7],
              par_t = table[8], G = table[9], nfi = table[10], PI = table[15];
    constexpr int KGS = 8 * TR, NIG = 64 * MF_NW / TR;       // floats per k-group of activations; identity-feature residues
    constexpr int MP = 3 * KB, FPL = 16 / KB, FPG = 2 * FPL, NW_TABW = nw_tabw(KB);
    constexpr int NFI = KB == 16 ? 8 : 4;                    // final items a wave may own (flows/nsf_wide_pack.bins_geometry)
    float *acts = lds;                                       // [Hp / 8 k-groups][2][TR][4]
    float *xreg = lds + (size_t)(Hp / 8) * KGS;              // [Dp / 8][2][TR][4]
    float *ldp = acts + nw_tab_floats(KB);                   // log-det partials [G + NIG][TR], behind the staged tables
    const int nitems = (1 + 2 * NB) * NHI + nfi + (LU ? 1 : 0);
    const int *items_all = table + MF_HDR + w * nitems * 3;  // [nitems][nkg, rb | g, sb0]
    const int *items = items_all + ((LU && DIR == 0) ? 3 : 0);   // the network's items (the density direction's LU entry comes first)
    const float lu_ld = LU ? (DIR == 0 ? lu_lad[0] : -lu_lad[0]) : 0.0f;
    const float *stream = blob + table[16 + w];
    const int lane_b = (TR * hh + n) * 4;
    const int64_t ntiles = (B + TR - 1) / TR;
    MfRing ring;
    mf_ring_start(ring, stream, lane);

    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row0 = tile * TR;
        const int nrows = (int)((B - row0) < TR ? (B - row0) : TR);
        ring.ap = stream + lane * 4;
        // the tile's prologue / epilogue address arithmetic from a thread index the compiler cannot hoist out of the tile loop: as loop
        // invariants those values stayed live across the products (the last 1-4 spilled registers of the 64-row instantiations)
        int tq = tid;
        asm volatile("" : "+v"(tq));
        {   // x tile -> LDS, columns sorted into positions (rows beyond the batch and the padding positions are zero)
            const int r = tq % TR, cg = tq / TR;
            const float *xr = x + (row0 + r) * D;
#pragma unroll 1        // (runtime trip counts: the unroller's remainder bookkeeping stayed live across the whole tile -- 1-8 spilled registers)
            for (int c = cg; 4 * c < D; c += NIG) {
                f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
                if (r < nrows) {
                    if ((D & 3) == 0) v = *reinterpret_cast<const f32x4 *>(xr + 4 * c);
                    else
#pragma unroll
                        for (int i = 0; i < 4; ++i) if (4 * c + i < D) v[i] = xr[4 * c + i];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int col = 4 * c + i;
                    if (col < D) xreg[nw_xidx<TR>(((col ^ par_i) & 1) ? PI + (col >> 1) : (col >> 1), r)] = v[i];
                }
            }
#pragma unroll 1
            for (int ps = nI + cg; ps < PI; ps += NIG) xreg[nw_xidx<TR>(ps, r)] = 0.0f;
#pragma unroll 1
            for (int ps = PI + nT + cg; ps < Dp; ps += NIG) xreg[nw_xidx<TR>(ps, r)] = 0.0f;
        }
        float ld_ident = 0.0f;
        if constexpr (LU && DIR == 0) {
            MF_BARRIER();
            nw_lu_stage<TR>(ring, items_all, xreg, lane_b, hh, n);
        }
        if constexpr (DIR == 1) {                            // sampling: the identity half's inverse spline comes first (:112-114)
#pragma unroll 1
            for (int i = tq; i < nI * NW_TABW; i += 64 * MF_NW) acts[i] = tabs[i];
            MF_BARRIER();
            ld_ident = nw_identity<true, TR, KB>(xreg, acts, p, nI, tq);
        }
        f32x16 h[NHI][NS], t[NHI][NS];
        MF_BARRIER();
        // ---- initial layer: h = b0 + W0 x over the identity positions [0, PI) of the tile ----------------------------------------------
#pragma unroll
        for (int s = 0; s < NHI; ++s) mf_item<NS, false, TR>(ring, items[3 * s], xreg + lane_b + 128 * items[3 * s + 2], h[s]);
        // ---- residual blocks (nets/resnet.py:37-50): t = b1 + W1 relu(h);  h += b2 + W2 relu(t) ----------------------------------
        for (int b = 0; b < NB; ++b) {
            MF_BARRIER();
#pragma unroll
            for (int s = 0; s < NHI; ++s) mf_publish<NS, true, TR>(acts, items[3 * s + 1], items[3 * s + 2], hh, n, h[s]);
            MF_BARRIER();
#pragma unroll
            for (int s = 0; s < NHI; ++s) {
                const int *it = items + 3 * ((1 + 2 * b) * NHI + s);
                mf_item<NS, false, TR>(ring, it[0], acts + lane_b + 128 * it[2], t[s]);
            }
            MF_BARRIER();
#pragma unroll
            for (int s = 0; s < NHI; ++s) mf_publish<NS, true, TR>(acts, items[3 * s + 1], items[3 * s + 2], hh, n, t[s]);
            MF_BARRIER();
#pragma unroll
            for (int s = 0; s < NHI; ++s) {
                const int *it = items + 3 * ((2 + 2 * b) * NHI + s);
                mf_item<NS, true, TR>(ring, it[0], acts + lane_b + 128 * it[2], h[s]);
            }
        }
        // ---- final layer on the raw block output (:104) in groups of four transform features + the spline ----------------------------
        MF_BARRIER();
#pragma unroll
        for (int s = 0; s < NHI; ++s) mf_publish<NS, false, TR>(acts, items[3 * s + 1], items[3 * s + 2], hh, n, h[s]);
        MF_BARRIER();
        float ldt[NFI][2];                                   // [final item][sample block] (nfi <= NFI)
#pragma unroll
        for (int q = 0; q < NFI; ++q) ldt[q][0] = ldt[q][1] = 0.0f;
#pragma nounroll
        for (int j = 0; j < nfi; ++j) {                       // (rolled: one copy of the item's code; the sums go to their slot by selects)
            const int *it = items + 3 * ((1 + 2 * NB) * NHI + j);
            const int g = it[1], sbo = it[2];                 // group of four transform features, first of its two sample blocks
            if (g < 0) continue;
            f32x16 o[3][2];
            mf_final_item<TR>(ring, it[0], acts + lane_b + 128 * sbo, o);
            float lsum[2] = {0.0f, 0.0f};
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                for (int f = 0; f < FPL; ++f) {
                    float prm[MP];
#pragma unroll
                    for (int v = 0; v < MP; ++v) prm[v] = o[(MP * f + v) >> 4][sb][(MP * f + v) & 15];
                    const int tf = FPG * g + FPL * hh + f;
                    const bool valid = tf < nT;
                    float *xp = xreg + nw_xidx<TR>(PI + (valid ? tf : 0), 32 * (sbo + sb) + n);
                    float yv, lad;
                    // round 5: binary bin descent (rqs_regs_t; the packed PAIR version of the benchmark kernel spilled 11-23 registers here)
#ifdef NF_EPI_SCALAR
                    rqs_regs<DIR == 1, KB>(p, *xp, prm, yv, lad);
#elif defined(NF_EPI_FULL_KNOTS)
                    rqs_regs_t<DIR == 1, KB>(p, *xp, prm, yv, lad);
#else           // round 6: first descent level before the knots exist (fused_common.hpp rqs_regs_h): half the live arrays
                    rqs_regs_h<DIR == 1, KB>(p, *xp, prm, yv, lad);
#endif
                    if (valid) {
                        *xp = yv;
                        lsum[sb] += lad;
                    }
                }
#pragma unroll
            for (int q = 0; q < NFI; ++q) {
                ldt[q][0] = j == q ? lsum[0] : ldt[q][0];
                ldt[q][1] = j == q ? lsum[1] : ldt[q][1];
            }
        }
        MF_BARRIER();                                        // every wave is done with the activations
        if constexpr (DIR == 0) {
#pragma unroll 1
            for (int i = tq; i < nI * NW_TABW; i += 64 * MF_NW) acts[i] = tabs[i];
        }
#pragma unroll
        for (int j = 0; j < NFI; ++j) {
            if (j >= nfi) break;
            const int *it = items + 3 * ((1 + 2 * NB) * NHI + j);
            const int g = it[1], sbo = it[2];
            if (g >= 0) {
#pragma unroll
                for (int sb = 0; sb < 2; ++sb) {
                    const float v = ldt[j][sb] + __shfl_xor(ldt[j][sb], 32);
                    if (hh == 0) ldp[g * TR + 32 * (sbo + sb) + n] = v;
                }
            }
        }
        if constexpr (DIR == 0) {                            // density: the identity half's spline after the conditioner (:88-92)
            MF_BARRIER();
            ld_ident = nw_identity<false, TR, KB>(xreg, acts, p, nI, tq);
        }
        ldp[(G + tq / TR) * TR + tq % TR] = ld_ident;
        MF_BARRIER();
        if constexpr (LU && DIR == 1) nw_lu_stage<TR>(ring, items_all + 3 * (nitems - 1), xreg, lane_b, hh, n);
        if (tq < nrows) {
            float v = lu_ld;
#pragma unroll 1
            for (int s = 0; s < G + NIG; ++s) v += ldp[s * TR + tq];      // fixed order: deterministic
            ld_store(logdet + row0 + tq, v, acc_mode);
        }
        {
            const int r = tq % TR, cg = tq / TR;
            float *yr = y + (row0 + r) * D;
            if (r < nrows) {
#pragma unroll 1
                for (int c = cg; 4 * c < D; c += NIG) {
                    f32x4 v;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int col = 4 * c + i < D ? 4 * c + i : D - 1;
                        v[i] = xreg[nw_xidx<TR>(((col ^ par_i) & 1) ? PI + (col >> 1) : (col >> 1), r)];
                    }
                    if ((D & 3) == 0) *reinterpret_cast<f32x4 *>(yr + 4 * c) = v;
                    else
#pragma unroll
                        for (int i = 0; i < 4; ++i) if (4 * c + i < D) yr[4 * c + i] = v[i];
                }
            }
        }
        MF_BARRIER();                                        // the next tile overwrites the x tile and the activations
    }
}

template <int NHI, int NS, int DIR, bool LU, int TR, int KB>
static int nsf_wide_launch(const void *x, void *y, void *logdet, const void *blob, const int32_t *table, const void *tabs,
                           const void *lu_lad, int64_t B, int Hp, int acc, const RqsParams<float> &p, hipStream_t st) {
    const int64_t ntiles = (B + TR - 1) / TR;
    const int grid = (int)(ntiles < 256 ? ntiles : 256);
    const size_t act_floats = (size_t)(Hp / 8) * 8 * TR;       // (>= the staged tables + the log-det partials: 2048 + 24 TR floats)
    const size_t lds = sizeof(float) * (act_floats + (size_t)16 * 8 * TR);
    static LdsOptIn opted;
    if (opt_in_lds(reinterpret_cast<const void *>(&nsf_wide_kernel<NHI, NS, DIR, LU, TR, KB>), lds, opted) != NF_OK) return NF_ENOTSUP;
    hipLaunchKernelGGL((nsf_wide_kernel<NHI, NS, DIR, LU, TR, KB>), dim3((unsigned)grid), dim3(64 * MF_NW), lds, st, (const float *)x, (float *)y,
                       (float *)logdet, (const float *)blob, (const int *)table, (const float *)tabs, (const float *)lu_lad, B, acc, p);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

}  // namespace nf

// Knot tables (cumulative widths | cumulative heights | derivatives, 27 floats per identity feature) of the batch-shared spline
// PiecewiseRationalQuadraticCDF (nsf/coupling.py:170-259), once per parameter version.
extern "C" int nf_nsf_wide_tables(const void *uw, const void *uh, const void *ud, void *tabs, int n_identity, int K, double tail_bound,
                                  double min_bin_width, double min_bin_height, double min_derivative, nf_stream_t stream) {
    if (K != 4 && K != 8 && K != 16) return NF_ENOTSUP;
    if (n_identity < 1 || n_identity > 64 || min_bin_width * K > 1.0 || min_bin_height * K > 1.0) return NF_EINVAL;
    if (!uw || !uh || !ud || !tabs) return NF_EFAULT;
    auto p = nf::make_rqs_params<float>(K, NF_TAILS_LINEAR, tail_bound, 0, 1, 0, 1, min_bin_width, min_bin_height, min_derivative, 1.0);
    hipLaunchKernelGGL(nf::nsf_wide_tables_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float *)uw, (const float *)uh,
                       (const float *)ud, (float *)tabs, n_identity, p);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

template <int DIR, bool LU, int KB>
static int nsf_wide_dispatch_k(const void *x, void *y, void *logdet, const void *blob, const int32_t *table, const void *tabs,
                               const void *lu_lad, int64_t B, int Hp, int acc, const nf::RqsParams<float> &p, hipStream_t st) {
    if (Hp == 128) return nf::nsf_wide_launch<1, 2, DIR, LU, 128, KB>(x, y, logdet, blob, table, tabs, lu_lad, B, Hp, acc, p, st);
    if (Hp == 256) return nf::nsf_wide_launch<1, 2, DIR, LU, 64, KB>(x, y, logdet, blob, table, tabs, lu_lad, B, Hp, acc, p, st);
    return nf::nsf_wide_launch<2, 2, DIR, LU, 64, KB>(x, y, logdet, blob, table, tabs, lu_lad, B, Hp, acc, p, st);
}

template <int DIR, bool LU>
static int nsf_wide_dispatch(const void *x, void *y, void *logdet, const void *blob, const int32_t *table, const void *tabs,
                             const void *lu_lad, int64_t B, int Hp, int acc, const nf::RqsParams<float> &p, hipStream_t st) {
    if (p.K == 4) return nsf_wide_dispatch_k<DIR, LU, 4>(x, y, logdet, blob, table, tabs, lu_lad, B, Hp, acc, p, st);
    if (p.K == 16) return nsf_wide_dispatch_k<DIR, LU, 16>(x, y, logdet, blob, table, tabs, lu_lad, B, Hp, acc, p, st);
    return nsf_wide_dispatch_k<DIR, LU, 8>(x, y, logdet, blob, table, tabs, lu_lad, B, Hp, acc, p, st);
}

// The coupling layer in one launch; blob / table: flows/nsf_wide_pack.pack_nsf_wide (packed for THIS direction when it carries the
// adjacent LULinearPermute); tabs: nf_nsf_wide_tables; lu_logdet: device scalar log|det| of the LU layer, or NULL (no LU in the pack).
// ... with K bins (4 | 8 | 16; pack and tables built for the same K: flows/nsf_wide_pack.pack_nsf_wide writes it to table[24])
extern "C" int nf_nsf_wide_k(const void *x, void *y, void *logdet, const void *blob, const int32_t *table, const void *tabs,
                             const void *lu_logdet, int64_t B, int D, int hidden_padded, int K, int direction, int acc, double tail_bound,
                             double min_bin_width, double min_bin_height, double min_derivative, nf_stream_t stream) {
    if (B < 0 || D < 2 || D > 128 || direction < 0 || direction > 1) return NF_EINVAL;
    if (acc < NF_LD_SUB || acc > NF_LD_ADD) return NF_EINVAL;
    if (hidden_padded != 128 && hidden_padded != 256 && hidden_padded != 512) return NF_ENOTSUP;
    if (K != 4 && K != 8 && K != 16) return NF_ENOTSUP;
    if (min_bin_width * K > 1.0 || min_bin_height * K > 1.0) return NF_EINVAL;
    if (B == 0) return NF_OK;
    if (!x || !y || !logdet || !blob || !table || !tabs) return NF_EFAULT;
    auto p = nf::make_rqs_params<float>(K, NF_TAILS_LINEAR, tail_bound, 0, 1, 0, 1, min_bin_width, min_bin_height, min_derivative, 1.0);
    hipStream_t st = (hipStream_t)stream;
    const int Hp = hidden_padded;
    if (direction == 0) {
        if (lu_logdet) return nsf_wide_dispatch<0, true>(x, y, logdet, blob, table, tabs, lu_logdet, B, Hp, acc, p, st);
        return nsf_wide_dispatch<0, false>(x, y, logdet, blob, table, tabs, nullptr, B, Hp, acc, p, st);
    }
    if (lu_logdet) return nsf_wide_dispatch<1, true>(x, y, logdet, blob, table, tabs, lu_logdet, B, Hp, acc, p, st);
    return nsf_wide_dispatch<1, false>(x, y, logdet, blob, table, tabs, nullptr, B, Hp, acc, p, st);
}

extern "C" int nf_nsf_wide(const void *x, void *y, void *logdet, const void *blob, const int32_t *table, const void *tabs,
                           const void *lu_logdet, int64_t B, int D, int hidden_padded, int direction, int acc, double tail_bound,
                           double min_bin_width, double min_bin_height, double min_derivative, nf_stream_t stream) {
    return nf_nsf_wide_k(x, y, logdet, blob, table, tabs, lu_logdet, B, D, hidden_padded, nf::F_K, direction, acc, tail_bound, min_bin_width,
                         min_bin_height, min_derivative, stream);
}
