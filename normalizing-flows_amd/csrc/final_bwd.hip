// final_bwd.hip -- backward of a benchmark-shaped NSF coupling layer's LAST stage in one pass over the rows: the coupling
// transform's backward (nsf/coupling.py:83-98 + utils/splines.py:16-219 under `loss.backward()`, core.py:87-102) AND the
// input gradient of the conditioner's final Linear (nets/resnet.py:104), gh = g_cond W_final, as the mirror image of the fused
// forward (rqs_fused.hip): the 736-wide gradient rows are produced group by group on the vector ALU in exactly the register
// layout in which the forward's MFMAs produced the parameters, and go straight into the MFMAs of gh as their B operand.
//
// Replaces (round 2): nf_rqs_coupling_bwd_p24 (125 us, HBM-bound) + a library GEMM (`Cijk_*`, 104 us) that re-read the 201 MB of
// gradient rows.  Here they are written once (for the final layer's weight gradient, nf_linear_wgrad_skip) and never read back.
//
// Mapping (4 waves x 32 rows per workgroup tile, persistent over tiles, ONE wave per SIMD = 512 registers):
//   lane l: row (l & 31) of the wave, lane-half hh = l >> 5 owns columns [16 Q + 8 hh, +8), Q = 0..3, of its row = "slots"
//   c = 8 Q + column-in-chunk (as in the forward); x / grad_y rows sit in a per-wave LDS stash [slot][65], lane position
//   33 hh + row: conflict-free for lane = row AND for lane = feature accesses.
//   1. identity half (batch-shared spline, nsf/coupling.py:88-92): lane = identity feature (x 2 row parities); the feature's
//      knot table (27 floats, from the packed blob) lives in registers for the whole launch; per row only the bin search and
//      the closed-form bin evaluation's partials -- the gradient is accumulated in KNOT space (7 + 7 + 7 register sums by
//      select-accumulate) and taken through the softmax / cumulative-sum / softplus chain ONCE per launch by
//      nf_final_bwd_reduce (linear in the knot gradients).  Round 2 repeated that chain per row: 44 us of a 157 us kernel.
//      Deterministic: per-workgroup partial sums added in a fixed order, no atomics.
//   2. transform half, per final-layer group g (4 features x 24 rows of W_final): the lane reads its two features'
//      24-float parameter rows (192 contiguous bytes; next group's prefetched under this group's MFMAs), runs the register
//      spline backward (rqs_regs_bwd) twice and stores the two gradient rows; gradient value v = 16 rb + reg of the lane is
//      the B operand of MFMA step (rb, reg): gh^T[unit, row] += W_t[unit][final row (g, rb, 8 (reg >> 2) + 4 hh + (reg & 3))]
//      g[...]; the A operand streams through a 2-slot LDS ring from the transposed stage image the per-step pack leaves
//      (24 stages of 16 KB: [q][unit block mb][lane][4 steps]).  192 MFMAs per group into 4 x 16 accumulators.
//   3. gh rows leave through a per-wave transpose tile (full 128-byte pieces), gx rows from the stash.
// Per row: reads 256 + 256 + 4 + 3072 B, writes 256 + 3072 + 512 B; 2 x 768 x 128 FLOP on fp32 MFMA + the spline backward
// on the same vector ALU (bound: MFMA + VALU time, ~80 + ~60 us at B = 65 536).
#include "rqs_bwd_common.hpp"

namespace nf {

#define FB_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

constexpr int FB_NW = 4;
constexpr int FB_THREADS = 64 * FB_NW;
constexpr int FB_ROWS = 32 * FB_NW;
constexpr int FB_P = 65;                          // floats per stash slot (33 hh + row < 65)
constexpr int FB_PLANE = 32 * FB_P;               // one plane (x or grad_y / grad_x) of a wave: 32 slots
constexpr int FB_WAVE = 2 * FB_PLANE + 32 + 64 * 48;   // + the rows' grad_logdet + the piece buffer of the parameter / gradient rows
constexpr int FB_NST = 24;                        // stages of the transposed final weight
constexpr int FB_PART = F_NI * 24;                // knot-space sums of a workgroup: [feature][7 w | 7 h | 7 d | 3 pad]

#ifdef FB_TRACE      // phase trace (tools/final_bwd_probe.py --trace): shader-clock cycles per phase, summed over the tiles of workgroup 0 / wave 0
static unsigned long long *g_fb_trace = nullptr;
extern "C" void nf_final_bwd_debug_trace(void *buf) { g_fb_trace = (unsigned long long *)buf; }
#define FB_T(i) do { const unsigned long long t1_ = clock64(); T_[i] += t1_ - t0_; t0_ = t1_; } while (0)
#else
#define FB_T(i) do {} while (0)
#endif

struct FinalBwdArgs {
#ifdef FB_TRACE
    unsigned long long *trace;
#endif
    const float *x, *gy, *gld, *cond, *wt, *tables;
    float *gx, *gcond, *gh, *part;
    int64_t B;
    int parity;
    RqsParams<float> p;
};

__global__ void __launch_bounds__(FB_THREADS, 1)
final_bwd_kernel(FinalBwdArgs a) {
    typedef __attribute__((address_space(3))) void *lds_ptr;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *ring = smem;                                    // 2 x F_STAGE
    const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5, srow = lane & 31;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *X = ring + 2 * F_STAGE + wid * FB_WAVE, *G = X + FB_PLANE, *GL = G + FB_PLANE, *CS = GL + 32;
    const RqsParams<float> p = a.p;
    const int par_t = a.parity ? 0 : 1, par_i = par_t ^ 1;
    const int soff = 33 * hh + srow;                       // lane = row: position inside a slot
    // lane = identity feature fj (slot 8 Q + 2 r + par_i of lane-half fhh), rows of parity sp
    const int fj = lane & 31, sp = lane >> 5;
    const int foff = (8 * (fj >> 3) + 2 * (fj & 3) + par_i) * FB_P + 33 * ((fj >> 2) & 1);
    float kw[F_K + 1], kh[F_K + 1], kd[F_K + 1];
#pragma unroll
    for (int k = 0; k <= F_K; ++k) {
        kw[k] = a.tables[fj * F_TABW + k];
        kh[k] = a.tables[fj * F_TABW + (F_K + 1) + k];
        kd[k] = a.tables[fj * F_TABW + 2 * (F_K + 1) + k];
    }
    float Gw[F_K - 1], Gh[F_K - 1], Gd[F_K - 1];
#pragma unroll
    for (int k = 0; k < F_K - 1; ++k) Gw[k] = Gh[k] = Gd[k] = 0.0f;
    const float sc = 1.44269504088896340736f / p.wh_div, inv_div = 1.0f / p.wh_div;

    const int64_t ntiles = (a.B + FB_ROWS - 1) / FB_ROWS;
    const int my_tiles = (int)((ntiles - blockIdx.x + gridDim.x - 1) / gridDim.x);
    const int total_stages = FB_NST * my_tiles;
    int stage = 0;
    auto issue = [&](int gs) {
        const int s = gs % FB_NST;
        const float *src = a.wt + (size_t)s * F_STAGE + wid * 1024 + lane * 4;
        float *dst = ring + (gs & 1) * F_STAGE + wid * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds(src + i * 256, (lds_ptr)(dst + i * 256), 16, 0, 0);
    };
    // 2-slot ring: stage s has landed for every wave; stage s + 1 is requested into the slot stage s - 1 just left.  `after`:
    // vector-memory operations this wave issued AFTER the requests of stage s (they retire in order, so they may stay in flight)
    auto acquire = [&](int after) -> const float * {
        if (after == 26) asm volatile("s_waitcnt vmcnt(26)" ::: "memory");
        else if (after == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (stage + 1 < total_stages) issue(stage + 1);
        const float *buf = ring + (stage & 1) * F_STAGE;
        ++stage;
        return buf;
    };
    if (total_stages > 0) issue(0);

    // Row traffic (the 24-float parameter rows in, the gradient rows out): a lane's two features of a group are 192 contiguous
    // bytes of ITS row -- as per-lane 16-byte accesses every instruction touches 64 different lines (first version: the row
    // operations cost 80 us of a 264 us launch, nothing of it hidden).  Instead the wave moves its 64 pieces (32 rows x 2
    // lane-halves) through a 12 KB LDS buffer `CS` with 13 coalesced instructions each way: instruction i, lane l handles 16-byte
    // chunk l % 12 of piece 5 i + l / 12 (5 pieces = 10 lines per instruction); in LDS the pieces are contiguous (48 floats each),
    // which is exactly the LDS-DMA's lane order.  Piece 2 row + hh belongs to lane (row, hh).
    const int l12 = lane / 12, lq = lane - 12 * l12;
    unsigned poff[13];       // float offset of this lane's chunk of instruction i inside the wave's 32 rows (group 0)
#pragma unroll
    for (int i = 0; i < 13; ++i) {
        const int pi = 5 * i + l12;
        poff[i] = (unsigned)((pi >> 1) * (F_NI * 24) + (pi & 1) * 96 + 4 * lq);
    }
    auto piece_ok = [&](int i, int64_t row0_) -> bool {
        const int pi = 5 * i + l12;
        return lane < 60 && pi < 64 && row0_ + (pi >> 1) < a.B;
    };
    auto group_off = [&](int g, int64_t row0_) -> int64_t { return row0_ * (int64_t)(F_NI * 24) + (8 * (g >> 1) + 2 * (g & 1)) * 24; };
    auto request_rows = [&](int g, int64_t row0_) {
        const float *base = a.cond + group_off(g, row0_);
#pragma unroll
        for (int i = 0; i < 13; ++i)
            if (piece_ok(i, row0_)) __builtin_amdgcn_global_load_lds(base + poff[i], (lds_ptr)(CS + 240 * i), 16, 0, 0);
    };

#ifdef FB_TRACE
    unsigned long long T_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t0_ = clock64();
#endif
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row0 = tile * FB_ROWS + wid * 32;
        const int64_t row = row0 + srow;
        const bool valid = row < a.B;
        const bool full = __builtin_amdgcn_readfirstlane((tile + 1) * FB_ROWS <= a.B ? 1 : 0) != 0;   // every row of the tile exists
        request_rows(0, row0);
        // ---- rows -> stash (rows beyond the batch: x outside the interval, zero cotangents: every gradient is zero) ----
#pragma unroll
        for (int Q = 0; Q < 4; ++Q) {
            f32x4 xa = {1e30f, 1e30f, 1e30f, 1e30f}, xb = xa, ga = {0.f, 0.f, 0.f, 0.f}, gb = ga;
            if (valid) {
                const float *xs = a.x + row * F_D + 16 * Q + 8 * hh, *gs = a.gy + row * F_D + 16 * Q + 8 * hh;
                xa = *reinterpret_cast<const f32x4 *>(xs);
                xb = *reinterpret_cast<const f32x4 *>(xs + 4);
                ga = *reinterpret_cast<const f32x4 *>(gs);
                gb = *reinterpret_cast<const f32x4 *>(gs + 4);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                X[(8 * Q + c) * FB_P + soff] = xa[c];
                X[(8 * Q + 4 + c) * FB_P + soff] = xb[c];
                G[(8 * Q + c) * FB_P + soff] = ga[c];
                G[(8 * Q + 4 + c) * FB_P + soff] = gb[c];
            }
        }
        const float gl = valid ? a.gld[row] : 0.0f;
        if (hh == 0) GL[srow] = gl;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();

        FB_T(0);      // tile prologue: requests, x / grad_y rows into the stash
        // ---- identity half: lane = feature, the wave's 32 rows two at a time ----
#ifndef FB_ABL_NOIDENT
#pragma nounroll
        for (int it = 0; it < 16; ++it) {
            const int s = 2 * it + sp;
            const float xv = X[foff + s], gyv = G[foff + s], glv = GL[s];
            const bool inside = xv >= p.left && xv <= p.right;
            int bin = 0;
            float xlo = kw[0], xhi = kw[1], ylo = kh[0], yhi = kh[1], d0 = kd[0], d1 = kd[1];
#pragma unroll
            for (int k = 1; k < F_K; ++k) {
                const bool ge = xv >= kw[k];
                bin = ge ? k : bin;
                xlo = ge ? kw[k] : xlo; xhi = ge ? kw[k + 1] : xhi;
                ylo = ge ? kh[k] : ylo; yhi = ge ? kh[k + 1] : yhi;
                d0 = ge ? kd[k] : d0; d1 = ge ? kd[k + 1] : d1;
            }
            float gv[7];
            rqs_eval_bin_vjp(xv, xlo, xhi, ylo, yhi, d0, d1, gyv, glv, gv);
#pragma unroll
            for (int i = 0; i < 7; ++i) gv[i] = inside ? gv[i] : 0.0f;
            G[foff + s] = inside ? gv[0] : gyv;
#pragma unroll
            for (int k = 1; k < F_K; ++k) {       // knot k is the bin's lower knot when bin == k, its upper knot when bin == k - 1
                const bool lo = bin == k, hi = bin == k - 1;
                Gw[k - 1] += (lo ? gv[1] : 0.0f) + (hi ? gv[2] : 0.0f);
                Gh[k - 1] += (lo ? gv[3] : 0.0f) + (hi ? gv[4] : 0.0f);
                Gd[k - 1] += (lo ? gv[5] : 0.0f) + (hi ? gv[6] : 0.0f);
            }
        }
#endif

        FB_T(1);      // identity half
        // ---- transform half + gh = g W_final, group by group ----
        f32x16 acc0, acc1, acc2, acc3;
#pragma unroll
        for (int c = 0; c < 16; ++c) acc0[c] = acc1[c] = acc2[c] = acc3[c] = 0.0f;
        float *mine = CS + (2 * srow + hh) * 48;      // this lane's piece
#pragma nounroll
        for (int g = 0; g < F_K; ++g) {
            // the group's parameter rows have landed (requested a whole group ago; everything younger may stay in flight: nothing is)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            FB_T(2);  // wait for the group's parameter rows
            float gcat[48];
            {
                float prm[2][24], gf[2][24], xt[2], gyt[2], gxt[2];
                int so[2];
#pragma unroll
                for (int f = 0; f < 2; ++f) {
#pragma unroll
                    for (int q = 0; q < 6; ++q) {
                        const f32x4 v = *reinterpret_cast<const f32x4 *>(mine + 24 * f + 4 * q);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int k = 4 * q + r;
                            prm[f][k] = k < 2 * F_K ? v[r] * sc : (k < F_M ? v[r] : 0.0f);
                        }
                    }
                    so[f] = (8 * (g >> 1) + 2 * (2 * (g & 1) + f) + par_t) * FB_P + soff;
                    xt[f] = X[so[f]];
                    gyt[f] = G[so[f]];
                }
#ifdef FB_ABL_NOSPLINE
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    gxt[f] = xt[f] + gyt[f];
#pragma unroll
                    for (int k = 0; k < 24; ++k) gf[f][k] = prm[f][k] * gyt[f];
                }
#else
                rqs_regs_bwd_pair(p, xt, prm, gyt, gl, gf, inv_div, gxt);
#endif
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    G[so[f]] = gxt[f];
#pragma unroll
                    for (int k = 0; k < 24; ++k) gcat[24 * f + k] = gf[f][k];
                }
            }
            // gradient rows -> the lane's piece (over the parameters it has just read)
#pragma unroll
            for (int q = 0; q < 12; ++q)
                *reinterpret_cast<f32x4 *>(mine + 4 * q) = f32x4{gcat[4 * q], gcat[4 * q + 1], gcat[4 * q + 2], gcat[4 * q + 3]};
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            FB_T(3);  // two spline backward elements + the piece write
            // Row traffic: one burst per group right behind stage 0's acquire (13 coalesced gradient-row stores, 13 requests of the
            // next group's parameter rows); they retire in order behind stage 1's requests, so they may stay in flight at its acquire.
            // (Measured and dropped: the same 26 instructions spread one per four-MFMA slot over stages 0 and 1 -- the burst's 53 k
            // cycles disappeared and 63 k came back inside the MFMA streams and at the stage barriers: with one wave per SIMD every
            // vector-memory instruction costs the wave ~100-250 issue cycles wherever it stands.)
            const bool more = g + 1 < F_K;
#pragma unroll
            for (int rb = 0; rb < 3; ++rb) {
                const float *buf = acquire((rb == 1 && full) ? (more ? 26 : 13) : 0);
                FB_T(4);  // stage wait + barrier + next stage's requests
                if (rb == 0) {
#ifndef FB_ABL_NOROWS
                    float *gbase = a.gcond + group_off(g, row0);
#pragma unroll
                    for (int i = 0; i < 13; ++i) {
                        const f32x4 v = *reinterpret_cast<const f32x4 *>(CS + 240 * i + 4 * lane);
                        if (piece_ok(i, row0)) *reinterpret_cast<f32x4 *>(gbase + poff[i]) = v;
                    }
                    // (the reads above are complete before the next group's parameters may overwrite the buffer)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                    if (more) request_rows(g + 1, row0);
#endif
                    FB_T(5);  // gradient-row stores + the next group's requests
                }
#ifndef FB_ABL_NOMFMA
                f32x4 w4[16];        // the stage's A operands up front: one wave per SIMD has nobody to hide LDS latency behind
#pragma unroll
                for (int i = 0; i < 16; ++i) w4[i] = *reinterpret_cast<const f32x4 *>(buf + (i * 64 + lane) * 4);
                __builtin_amdgcn_sched_barrier(0);      // (left alone the scheduler sinks every read to its first use and waits for it there)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb) {
                        f32x16 &acc = mb == 0 ? acc0 : (mb == 1 ? acc1 : (mb == 2 ? acc2 : acc3));
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc = FB_MFMA(w4[q * 4 + mb][r], gcat[16 * rb + 4 * q + r], acc);
                    }
                }
#else
                (void)buf;
#pragma unroll
                for (int v = 0; v < 16; ++v) acc0[v] += gcat[16 * rb + v];
#endif
                asm volatile("" ::"v"(acc0[0]), "v"(acc1[0]), "v"(acc2[0]), "v"(acc3[0]));
                FB_T(6);  // 16 A-operand reads + 64 MFMAs (issue time: the last MFMAs may still be executing)
            }
        }

        // ---- gh rows through the wave's transpose tile (the x plane is dead), gx rows from the stash ----
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        {
            float *twr = X + srow * 36 + 4 * hh;
            const int rl = lane >> 3, cl = lane & 7;
            const float *tww = X + rl * 36 + 4 * cl;
            float *dst = a.gh + (row0 + rl) * F_H + 4 * cl;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const f32x16 &V = mb == 0 ? acc0 : (mb == 1 ? acc1 : (mb == 2 ? acc2 : acc3));
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<f32x4 *>(twr + 8 * q) = f32x4{V[4 * q], V[4 * q + 1], V[4 * q + 2], V[4 * q + 3]};
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int it = 0; it < 4; ++it)
                    if (row0 + rl + 8 * it < a.B)
                        *reinterpret_cast<f32x4 *>(dst + (size_t)(8 * it) * F_H + 32 * mb) = *reinterpret_cast<const f32x4 *>(tww + 8 * it * 36);
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (valid) {
#pragma unroll
            for (int Q = 0; Q < 4; ++Q) {
                f32x4 ga, gb;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    ga[c] = G[(8 * Q + c) * FB_P + soff];
                    gb[c] = G[(8 * Q + 4 + c) * FB_P + soff];
                }
                float *dst = a.gx + row * F_D + 16 * Q + 8 * hh;
                *reinterpret_cast<f32x4 *>(dst) = ga;
                *reinterpret_cast<f32x4 *>(dst + 4) = gb;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        FB_T(7);      // gh / gx rows out
    }
#ifdef FB_TRACE
    if (a.trace && blockIdx.x == 0 && tid == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a.trace[i] = T_[i];
    }
#endif

    // ---- the workgroup's knot-space sums of the batch-shared parameters, fixed order ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {
        float *red = ring + (wid * 2 + sp) * FB_PART + fj * 24;
#pragma unroll
        for (int k = 0; k < F_K - 1; ++k) {
            red[k] = Gw[k];
            red[7 + k] = Gh[k];
            red[14 + k] = Gd[k];
        }
        red[21] = red[22] = red[23] = 0.0f;
    }
    __syncthreads();
    for (int i = tid; i < FB_PART; i += FB_THREADS) {
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < 2 * FB_NW; ++w) s += ring[w * FB_PART + i];
        a.part[(size_t)blockIdx.x * FB_PART + i] = s;
    }
}

// Knot-space sums -> gradients of the raw batch-shared parameters (unnormalized_widths / heights / derivatives of the
// unconditional transform, nsf/coupling.py:221-253 over utils/splines.py:100-157): one thread per (feature, parameter) sums the
// workgroups' partials in a fixed order, then per feature the chain  knot_k = lo + (hi - lo) (k min + scale C_k),
// C_k = sum_{i < k} softmax(raw)_i  =>  d/d raw_i = (hi - lo) scale softmax_i sum_k G_k ([i < k] - C_k);  d_j = min_d + softplus(raw_j).
__global__ void __launch_bounds__(256)
final_bwd_reduce_kernel(const float *__restrict__ part, int nparts, const float *__restrict__ uw, const float *__restrict__ uh,
                        const float *__restrict__ ud, float *__restrict__ guw, float *__restrict__ guh, float *__restrict__ gud,
                        RqsParams<float> p) {
    // one workgroup per feature: 10 thread groups x 24 sums walk the partial rows with stride 10, their results are added in
    // group order (fixed: bit-reproducible); a single workgroup reading all 0.8 MB took 120 us
    __shared__ float sub[10][24];
    __shared__ float sums[24];
    const int j = blockIdx.x, t = threadIdx.x;
    if (t < 240) {
        const int k = t % 24, grp = t / 24;
        float s = 0.0f;
        for (int w = grp; w < nparts; w += 10) s += part[(size_t)w * FB_PART + j * 24 + k];
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

}  // namespace nf

using namespace nf;

static int final_bwd_grid(int64_t B) {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int64_t ntiles = (B + FB_ROWS - 1) / FB_ROWS;
    return (int)(ntiles < cus ? ntiles : cus);
}

// Rows of `partials` nf_final_bwd writes for a batch of B rows (one per resident workgroup: one per CU, fewer for small batches).
extern "C" int nf_final_bwd_partials(int64_t B) {
    if (B < 0) return NF_EINVAL;
    return B == 0 ? 0 : final_bwd_grid(B);
}

extern "C" int nf_final_bwd(const void *x, const void *grad_y, const void *grad_logdet, const void *cond24, const void *w_t,
                            const void *wpack, void *grad_x, void *grad_cond24, void *grad_h, void *partials, int mask_parity,
                            int64_t B, int D, int hidden, int num_blocks, int K, double tail_bound, double min_bin_width,
                            double min_bin_height, double min_derivative, nf_stream_t stream) {
    if (D != F_D || hidden != F_H || K != F_K || num_blocks < 0 || num_blocks > 16) return NF_ENOTSUP;
    if (B < 0 || (mask_parity != 0 && mask_parity != 1)) return NF_EINVAL;
    if (min_bin_width * K > 1.0 || min_bin_height * K > 1.0) return NF_EINVAL;
    if (B == 0) return NF_OK;
    if (!x || !grad_y || !grad_logdet || !cond24 || !w_t || !wpack || !grad_x || !grad_cond24 || !grad_h || !partials) return NF_EFAULT;
    FusedLayout lay;
    lay.nblk = num_blocks;
    FinalBwdArgs a;
#ifdef FB_TRACE
    a.trace = g_fb_trace;
#endif
    a.x = (const float *)x; a.gy = (const float *)grad_y; a.gld = (const float *)grad_logdet; a.cond = (const float *)cond24;
    a.wt = (const float *)w_t; a.tables = (const float *)wpack + F_HDR + lay.off_tables();
    a.gx = (float *)grad_x; a.gcond = (float *)grad_cond24; a.gh = (float *)grad_h; a.part = (float *)partials;
    a.B = B; a.parity = mask_parity;
    a.p = make_rqs_params<float>(K, NF_TAILS_LINEAR, tail_bound, 0, 1, 0, 1, min_bin_width, min_bin_height, min_derivative,
                                 sqrt((double)hidden));
    const size_t lds = (size_t)(2 * F_STAGE + FB_NW * FB_WAVE) * sizeof(float);
    static LdsOptIn opted = {};
    if (opt_in_lds(reinterpret_cast<const void *>(&final_bwd_kernel), lds, opted) != NF_OK) return NF_ENOTSUP;
    hipLaunchKernelGGL(final_bwd_kernel, dim3(final_bwd_grid(B)), dim3(FB_THREADS), lds, (hipStream_t)stream, a);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

extern "C" int nf_final_bwd_reduce(const void *partials, int n_partials, const void *uw, const void *uh, const void *ud,
                                   void *grad_uw, void *grad_uh, void *grad_ud, int K, double tail_bound, double min_bin_width,
                                   double min_bin_height, double min_derivative, nf_stream_t stream) {
    if (K != F_K) return NF_ENOTSUP;
    if (n_partials < 0) return NF_EINVAL;
    if (!uw || !uh || !ud || !grad_uw || !grad_uh || !grad_ud || (n_partials > 0 && !partials)) return NF_EFAULT;
    auto p = make_rqs_params<float>(K, NF_TAILS_LINEAR, tail_bound, 0, 1, 0, 1, min_bin_width, min_bin_height, min_derivative, 1.0);
    hipLaunchKernelGGL(final_bwd_reduce_kernel, dim3(F_NI), dim3(256), 0, (hipStream_t)stream, (const float *)partials, n_partials,
                       (const float *)uw, (const float *)uh, (const float *)ud, (float *)grad_uw, (float *)grad_uh,
                       (float *)grad_ud, p);
    NF_CHECK_LAUNCH();
    return NF_OK;
}
