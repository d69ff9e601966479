// final_bwd.hip -- backward of a benchmark-shaped NSF coupling layer's LAST stage in one pass over the rows: the coupling
// transform's backward (nsf/coupling.py:83-98 + utils/splines.py:16-219 under `loss.backward()`, core.py:87-102) AND the
// input gradient of the conditioner's final Linear (nets/resnet.py:104), gh = g_cond W_final: the 736-wide gradient rows are
// produced group by group on the vector ALU and go straight from the registers they were computed in into the MFMAs of gh.
//
// Replaces (round 2): nf_rqs_coupling_bwd_p24 (125 us, HBM-bound) + a library GEMM (`Cijk_*`, 104 us) that re-read the 201 MB of
// gradient rows.  Here they are written once (for the final layer's weight gradient, nf_linear_wgrad_skip) and never read back.
//
// Mapping (second version).  The first version mirrored the forward kernel -- 32 rows per wave on v_mfma_f32_32x32x2_f32, a lane
// owning TWO spline elements per group -- and needed 400+ registers: one wave per SIMD, and with nobody to share the SIMD with
// every LDS round trip, every vector-memory issue (~100-250 cycles each) and every dependent transcendental was exposed: the
// phase trace (tools/final_bwd_probe.py --trace) showed the spline arithmetic at ~7 cycles per instruction and the MFMAs at 0.78
// of their issue rate, 250-265 us per launch (ablations were additive to the microsecond: nothing overlapped).  This version is
// built for TWO waves per SIMD:
//   * a wave owns 16 rows and multiplies on v_mfma_f32_16x16x4_f32: lane l = (row n = l & 15, quarter hq = l >> 4); the MFMA's
//     four k-entries are the four transform features of a final-layer group, so a lane owns exactly ONE spline element per group
//     (feature tf = 8 (g >> 1) + 4 (hq >> 1) + 2 (g & 1) + (hq & 1), its 24-float parameter row = 96 contiguous bytes) and its 24
//     gradient values are the B operands of the group's 24 k-steps; 8 unit blocks x 4 accumulator registers hold gh of the row.
//     32 + 24 + the spline routine's temporaries fit in 256 registers.
//   * 8 waves per workgroup tile of 128 rows, persistent over tiles; the A operand (final weight, transposed roles) streams through
//     the forward's 2-slot LDS ring from the image the per-step pack leaves: 24 stages [k-step vv][unit-block quad][lane][4].
//   * x / grad_y rows sit row-major in a per-wave LDS stash (coalesced 16-byte loads, pitch 68); parameter rows come in and
//     gradient rows leave through a per-wave 6 KB piece buffer with 7 coalesced instructions each way (a lane's own 96 bytes
//     would be 64 different lines per instruction).
//   * identity half (batch-shared spline, nsf/coupling.py:88-92): lane = identity feature (x 2 row parities) over the wave's 16
//     rows; the feature's knot table comes from the packed blob; per row only the bin search and the reverse-mode VJP of the
//     closed-form bin evaluation -- the gradient is accumulated in KNOT space (7 + 7 + 7 register sums by select-accumulate) and
//     taken through the softmax / cumulative-sum / softplus chain ONCE per launch by nf_final_bwd_reduce (it is linear in the knot
//     gradients).  Deterministic: per-workgroup partial sums added in a fixed order, no atomics.
// Per row: reads 256 + 256 + 4 + 3072 B, writes 256 + 3072 + 512 B; 2 x 768 x 128 FLOP on fp32 MFMA + the spline backward
// on the same vector ALU (bound: MFMA + VALU time).
#include "rqs_bwd_common.hpp"
#include "train_reduce.hpp"
#include <type_traits>

namespace nf {

typedef float f32x4a __attribute__((ext_vector_type(4)));
#define FB_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// Round 6: NF_FB_NW = 4 -> 64-row tiles, TWO workgroups per CU (the 8-wave workgroup spent a third of its time at the ring's stage
// barriers with every wave of the CU in the same phase: two independent workgroups drift apart and fill each other's waits); the ring
// then streams half stages (NF_FB_KS = 4 k-steps, 8 KB per slot) so that both workgroups' LDS fits: 2 x (16 + 4 x 14.9) KB.
#ifndef NF_FB_NW
#define NF_FB_NW 8
#endif
#ifndef NF_FB_KS
#define NF_FB_KS 8
#endif
constexpr int FB_NW = NF_FB_NW;                   // waves per workgroup
constexpr int FB_KS = NF_FB_KS;                   // k-steps of the final weight per ring stage (8: 16 KB stages; 4: 8 KB)
constexpr int FB_STG = FB_KS * 512;               // floats per ring stage
constexpr int FB_SPG = 24 / FB_KS;                // stages per group
static_assert((FB_NW == 8 || FB_NW == 4) && (FB_KS == 8 || FB_KS == 4) && FB_STG % (256 * FB_NW) == 0, "ring geometry");
constexpr int FB_THREADS = 64 * FB_NW;
constexpr int FB_WR = 16;                         // rows per wave
constexpr int FB_ROWS = FB_WR * FB_NW;            // 128 rows per tile
constexpr int FB_P = 68;                          // stash row pitch (floats): 16-byte aligned rows
constexpr int FB_PLANE = FB_WR * FB_P;            // one plane (x or grad_y / grad_x) of a wave
constexpr int FB_CS = FB_WR * 4 * 24;             // piece buffer: [row][quarter][24] = the group's parameter / gradient rows of the wave
constexpr int FB_WAVE = 2 * FB_PLANE + FB_WR + FB_CS;
constexpr int FB_NST = 8 * FB_SPG;                // stages of the transposed final weight (8 groups)
constexpr int FB_PART = FBR_PART;                 // knot-space sums of a workgroup: [feature][7 w | 7 h | 7 d | 3 pad]
constexpr int FB_NI = 7;                          // coalesced instructions per direction for the 32 runs of 192 bytes of a group

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

__global__ void __launch_bounds__(FB_THREADS, FB_NW == 4 ? 2 : 1)
final_bwd_kernel(FinalBwdArgs a) {
    typedef __attribute__((address_space(3))) void *lds_ptr;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *ring = smem;                                    // 2 x FB_STG
    const int tid = threadIdx.x, lane = tid & 63, hq = lane >> 4, n = lane & 15;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *X = ring + 2 * FB_STG + wid * FB_WAVE, *G = X + FB_PLANE, *GL = G + FB_PLANE, *CS = GL + FB_WR;
    const RqsParams<float> p = a.p;
    const int par_t = a.parity ? 0 : 1, par_i = par_t ^ 1;
    // lane = identity feature fj (column 2 fj + par_i), rows of parity sp
    const int fj = lane & 31, sp = lane >> 5;
    const int fcol = 2 * fj + par_i;
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
        constexpr int PW = FB_STG / FB_NW;     // floats of a stage per requesting wave
        const float *src = a.wt + (size_t)s * FB_STG + wid * PW + lane * 4;
        float *dst = ring + (gs & 1) * FB_STG + wid * PW;
#pragma unroll
#ifdef NF_BUILTIN_DMA
        for (int i = 0; i < PW / 256; ++i) __builtin_amdgcn_global_load_lds(src + i * 256, (lds_ptr)(dst + i * 256), 16, 0, 0);
#else       // round 6: requests as inline asm (common.hpp NF_DMA16): the products' operand look-ahead survives compilation
        for (int i = 0; i < PW / 256; ++i) NF_DMA16(src - lane * 4 + i * 256, lane * 16, dst + i * 256);
#endif
    };
    // 2-slot ring: stage s has landed for every wave; stage s + 1 is requested into the slot stage s - 1 just left.  `after`:
    // vector-memory operations this wave issued AFTER the requests of stage s (they retire in order, so they may stay in flight)
    auto acquire = [&](int after) -> const float * {
        if (after == 2 * FB_NI + 9) NF_WAIT_VMCNT(23);
        else if (after == 2 * FB_NI) NF_WAIT_VMCNT(14);
        else if (after == FB_NI) NF_WAIT_VMCNT(7);
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // the raw barrier: __syncthreads() is a fence, and with an LDS-DMA possibly pending the compiler implements it as
        // s_waitcnt vmcnt(0) -- the counted waits above never took effect before this was found (round 3, in the Glow kernels)
#ifndef FB_ABL_NOBAR      // (timing-only ablation: how much of the launch is waves waiting for each other at the ring's stage barriers)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
        if (stage + 1 < total_stages) issue(stage + 1);
        const float *buf = ring + (stage & 1) * FB_STG;
        ++stage;
        return buf;
    };
    if (total_stages > 0) issue(0);

    // Row traffic: the wave's 32 runs of 192 bytes per group (16 rows x 2 feature pairs) with 7 coalesced instructions each way:
    // instruction i, lane l handles 16-byte chunk l % 12 of run 5 i + l / 12; in LDS the runs are contiguous (48 floats each:
    // [row][quarter][24] is exactly that order), which is the LDS-DMA's lane order.
    const int l12 = lane / 12, lq = lane - 12 * l12;
    unsigned poff[FB_NI];    // float offset of this lane's chunk of instruction i inside the wave's 16 rows (group 0)
#pragma unroll
    for (int i = 0; i < FB_NI; ++i) {
        const int ri = 5 * i + l12;
        poff[i] = (unsigned)((ri >> 1) * (F_NI * 24) + (ri & 1) * 96 + 4 * lq);
    }
    auto piece_ok = [&](int i, int64_t row0_) -> bool {
        const int ri = 5 * i + l12;
        return lane < 60 && ri < 2 * FB_WR && row0_ + (ri >> 1) < a.B;
    };
    auto group_off = [&](int g, int64_t row0_) -> int64_t { return row0_ * (int64_t)(F_NI * 24) + (8 * (g >> 1) + 2 * (g & 1)) * 24; };
    auto request_rows = [&](int g, int64_t row0_) {
        const float *base = a.cond + group_off(g, row0_);
#pragma unroll
        for (int i = 0; i < FB_NI; ++i)
#ifdef NF_BUILTIN_DMA
            if (piece_ok(i, row0_)) __builtin_amdgcn_global_load_lds(base + poff[i], (lds_ptr)(CS + 240 * i), 16, 0, 0);
#else
            if (piece_ok(i, row0_)) NF_DMA16(base, poff[i] * 4u, CS + 240 * i);
#endif
    };

#ifdef FB_TRACE
    unsigned long long T_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t0_ = clock64();
#endif
    // the NEXT tile's x / grad_y rows (and its first group's parameter rows) are requested under the last group's MFMA stages of
    // the current one, where few registers are live: a tile used to start with every wave of the chip waiting ~10 us for 8 KB
    f32x4 nx[4], ng[4];
    float ngl = 0.0f;
    auto fetch_rows = [&](int64_t row0_) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = lane + 64 * i, r = idx >> 4, c4 = idx & 15;
            nx[i] = f32x4{1e30f, 1e30f, 1e30f, 1e30f};
            ng[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (row0_ + r < a.B) {
                nx[i] = *reinterpret_cast<const f32x4 *>(a.x + (row0_ + r) * F_D + 4 * c4);
                ng[i] = *reinterpret_cast<const f32x4 *>(a.gy + (row0_ + r) * F_D + 4 * c4);
            }
        }
        ngl = row0_ + n < a.B ? a.gld[row0_ + n] : 0.0f;
    };
    if (blockIdx.x < ntiles) {
        request_rows(0, (int64_t)blockIdx.x * FB_ROWS + wid * FB_WR);
        fetch_rows((int64_t)blockIdx.x * FB_ROWS + wid * FB_WR);
    }
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row0 = tile * FB_ROWS + wid * FB_WR;
        const int64_t next_tile = tile + gridDim.x;
        const bool has_next = next_tile < ntiles;
        const int64_t next_row0 = next_tile * FB_ROWS + wid * FB_WR;
        // every row of this tile exists (and of the next one, whose prefetch rides in the counted waits)
        const bool full = __builtin_amdgcn_readfirstlane(((tile + 1) * FB_ROWS <= a.B && (!has_next || (next_tile + 1) * FB_ROWS <= a.B)) ? 1 : 0) != 0;
        // ---- the wave's 16 rows of x / grad_y -> stash, row-major (rows beyond the batch: x outside the interval, zero cotangents:
        // every gradient is zero) ----
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = lane + 64 * i, r = idx >> 4, c4 = idx & 15;
            *reinterpret_cast<f32x4 *>(X + r * FB_P + 4 * c4) = nx[i];
            *reinterpret_cast<f32x4 *>(G + r * FB_P + 4 * c4) = ng[i];
        }
        const float gl = ngl;
        if (hq == 0) GL[n] = gl;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        FB_T(0);      // tile prologue: requests, x / grad_y rows into the stash

        // ---- identity half: lane = feature, the wave's 16 rows two at a time ----
#ifndef FB_ABL_NOIDENT
        {
            float kw[F_K + 1], kh[F_K + 1], kd[F_K + 1];
#pragma unroll
            for (int k = 0; k <= F_K; ++k) {
                kw[k] = a.tables[fj * F_TABW + k];
                kh[k] = a.tables[fj * F_TABW + (F_K + 1) + k];
                kd[k] = a.tables[fj * F_TABW + 2 * (F_K + 1) + k];
            }
#pragma nounroll
            for (int it = 0; it < FB_WR / 2; ++it) {
                const int s = 2 * it + sp;
                const float xv = X[s * FB_P + fcol], gyv = G[s * FB_P + fcol], glv = GL[s];
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
                G[s * FB_P + fcol] = inside ? gv[0] : gyv;
#pragma unroll
                for (int k = 1; k < F_K; ++k) {       // knot k is the bin's lower knot when bin == k, its upper knot when bin == k - 1
                    const bool lo = bin == k, hi = bin == k - 1;
                    Gw[k - 1] += (lo ? gv[1] : 0.0f) + (hi ? gv[2] : 0.0f);
                    Gh[k - 1] += (lo ? gv[3] : 0.0f) + (hi ? gv[4] : 0.0f);
                    Gd[k - 1] += (lo ? gv[5] : 0.0f) + (hi ? gv[6] : 0.0f);
                }
            }
        }
#endif
        FB_T(1);      // identity half

        // ---- transform half + gh = g W_final, group by group ----
        f32x4a acc[8];
#pragma unroll
        for (int ub = 0; ub < 8; ++ub) acc[ub] = f32x4a{0.f, 0.f, 0.f, 0.f};
        float *mine = CS + (4 * n + hq) * 24;      // this lane's piece: the parameter row of its feature, then its gradient row
        // (the last group is peeled: only there the next tile's rows are requested into registers, which then are live from that
        // point to the next prologue instead of through every group's spline arithmetic)
        auto group = [&](int g, auto last_c) __attribute__((always_inline)) {
            constexpr bool LAST = decltype(last_c)::value;
            // the group's parameter rows have landed (requested at least two stages ago)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            FB_T(2);  // wait for the group's parameter rows
            float gq[24];
            {
                float prm[24];
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(mine + 4 * q);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int k = 4 * q + r;
                        prm[k] = k < 2 * F_K ? v[r] * sc : (k < F_M ? v[r] : 0.0f);
                    }
                }
                const int so = n * FB_P + 2 * (8 * (g >> 1) + 4 * (hq >> 1) + 2 * (g & 1) + (hq & 1)) + par_t;
                const float xt = X[so], gyt = G[so];
#ifdef FB_ABL_NOSPLINE
                G[so] = xt + gyt;
#pragma unroll
                for (int k = 0; k < 24; ++k) gq[k] = prm[k] * gyt;
#else
                G[so] = rqs_regs_bwd<false>(p, xt, prm, gyt, gl, gq, inv_div);
#endif
            }
            // gradient row -> the lane's piece (over the parameters it has just read)
#pragma unroll
            for (int q = 0; q < 6; ++q)
                *reinterpret_cast<f32x4 *>(mine + 4 * q) = f32x4{gq[4 * q], gq[4 * q + 1], gq[4 * q + 2], gq[4 * q + 3]};
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            FB_T(3);  // the spline backward element + the piece write
            // Row traffic: one burst per group right behind stage 0's acquire (7 coalesced gradient-row stores, 7 requests of the
            // next group's parameter rows); they retire in order behind stage 1's requests, so they may stay in flight at its acquire.
            constexpr bool more = !LAST;
#pragma unroll
            for (int rb = 0; rb < FB_SPG; ++rb) {
                const float *buf = acquire((rb == 1 && full) ? (more ? 2 * FB_NI : (has_next ? 2 * FB_NI + 9 : FB_NI)) : 0);
                FB_T(4);  // stage wait + barrier + next stage's requests
                if (rb == 0) {
#ifndef FB_ABL_NOROWS
                    float *gbase = a.gcond + group_off(g, row0);
#pragma unroll
                    for (int i = 0; i < FB_NI; ++i) {
                        if (piece_ok(i, row0))
                            *reinterpret_cast<f32x4 *>(gbase + poff[i]) = *reinterpret_cast<const f32x4 *>(CS + 240 * i + 4 * lane);
                    }
                    // (the reads above are complete before the next group's parameters may overwrite the buffer)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_wave_barrier();
                    if constexpr (more) request_rows(g + 1, row0);
                    else if (has_next) {
                        request_rows(0, next_row0);
                        fetch_rows(next_row0);
                    }
#endif
                    FB_T(5);  // gradient-row stores + the next group's requests
                }
#ifndef FB_ABL_NOMFMA
                // stage rb: k-steps v = KS rb .. KS rb + KS - 1 (B operand = the lane's gradient value v), 8 unit blocks each
#ifdef NF_NO_PREFETCH
#pragma unroll
                for (int vv = 0; vv < FB_KS; ++vv) {
#pragma unroll
                    for (int uq = 0; uq < 2; ++uq) {
                        const f32x4 w4 = *reinterpret_cast<const f32x4 *>(buf + ((vv * 2 + uq) * 64 + lane) * 4);
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[4 * uq + j] = FB_MFMA(w4[j], gq[FB_KS * rb + vv], acc[4 * uq + j]);
                    }
                }
#else
                {   // the A operands two reads (8 MFMAs) ahead of their use -- real since the requests are inline asm (NF_DMA16)
                    f32x4 w0 = *reinterpret_cast<const f32x4 *>(buf + lane * 4), w1 = *reinterpret_cast<const f32x4 *>(buf + (64 + lane) * 4);
#pragma unroll
                    for (int t = 0; t < 2 * FB_KS; ++t) {
                        const f32x4 w4 = w0;
                        w0 = w1;
                        if (t + 2 < 2 * FB_KS) w1 = *reinterpret_cast<const f32x4 *>(buf + ((t + 2) * 64 + lane) * 4);
                        __builtin_amdgcn_sched_barrier(0);   // keeps the request in FRONT of the k-group's MFMAs (without it: +0.5 % lost again)
                        const int vv = t >> 1, uq = t & 1;
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[4 * uq + j] = FB_MFMA(w4[j], gq[FB_KS * rb + vv], acc[4 * uq + j]);
                    }
                }
#endif
#else
                (void)buf;
#pragma unroll
                for (int v = 0; v < FB_KS; ++v) acc[v][0] += gq[FB_KS * rb + v];
#endif
                asm volatile("" ::"v"(acc[0][0]), "v"(acc[7][0]));
                FB_T(6);  // 16 A-operand reads + 64 MFMAs (issue time: the last MFMAs may still be executing)
            }
        };
#pragma nounroll
        for (int g = 0; g < F_K - 1; ++g) group(g, std::false_type{});
        group(F_K - 1, std::true_type{});

        // ---- gx rows from the stash, then gh rows through the wave's transpose tile (both planes are dead by then) ----
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = lane + 64 * i, r = idx >> 4, c4 = idx & 15;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(G + r * FB_P + 4 * c4);
            if (row0 + r < a.B) *reinterpret_cast<f32x4 *>(a.gx + (row0 + r) * F_D + 4 * c4) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        {
            float *T = X;                          // [16 rows][132]: 2112 floats <= the two planes (2176)
#pragma unroll
            for (int ub = 0; ub < 8; ++ub)
                *reinterpret_cast<f32x4 *>(T + n * 132 + 16 * ub + 4 * hq) = f32x4{acc[ub][0], acc[ub][1], acc[ub][2], acc[ub][3]};
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int idx = lane + 64 * i, r = idx >> 5, c4 = idx & 31;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(T + r * 132 + 4 * c4);
                if (row0 + r < a.B) *reinterpret_cast<f32x4 *>(a.gh + (row0 + r) * F_H + 4 * c4) = v;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        FB_T(7);      // gx / gh rows out
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
        float *red = ring + (wid * 2 + sp) * FB_PART + fj * 24;       // 16 x 768 floats: the ring and the first waves' stashes
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
// unconditional transform, nsf/coupling.py:221-253 over utils/splines.py:100-157): one workgroup per feature
// (train_reduce.hpp::final_bwd_reduce_feature: 10 thread groups x 24 sums walk the partial rows with stride 10, their results are
// added in group order -- fixed: bit-reproducible; a single workgroup reading all 0.8 MB took 120 us).
__global__ void __launch_bounds__(256)
final_bwd_reduce_kernel(const float *__restrict__ part, int nparts, const float *__restrict__ uw, const float *__restrict__ uh,
                        const float *__restrict__ ud, float *__restrict__ guw, float *__restrict__ guh, float *__restrict__ gud,
                        RqsParams<float> p) {
    __shared__ float sub[10][24];
    __shared__ float sums[24];
    final_bwd_reduce_feature(part, nparts, uw, uh, ud, guw, guh, gud, p, blockIdx.x, sub, sums);
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
    const int slots = cus * (FB_NW == 4 ? 2 : 1);       // resident workgroups
    return (int)(ntiles < slots ? ntiles : slots);
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
    const size_t lds = (size_t)(2 * FB_STG + FB_NW * FB_WAVE) * sizeof(float);
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
