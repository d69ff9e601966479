// mlp_tile.hpp -- the engine shared by the one-launch network kernels on 64-row tiles (made_fwd.hip: MADE; nsf_wide.hip: the NSF
// coupling layer's ResidualNet for hidden widths / feature counts beyond rqs_fused.hip): a workgroup of 8 waves owns 64 rows for the
// whole network, pre-activations live in accumulator registers, the next layer's B operand in LDS in MFMA order, and every wave
// walks ONE contiguous weight stream through an 8-entry register ring (see made_fwd.hip for the full description).
#pragma once
#include <cstdlib>
#include "common.hpp"
#include "fused_common.hpp"

namespace nf {

constexpr int MF_ROWS = 64;       // rows per workgroup and tile
constexpr int MF_NW = 8;          // waves per workgroup: two per SIMD
constexpr int MF_HDR = 32;
constexpr int MF_XFLOATS = 16 * 2 * 64 * 4;      // x tile: Dp <= 128 features

// 128-ROW tiles for the training kernels of a 256-slot network with <= 64 input features (round 6, last session; made_fwd.hip EPI 3,
// made_bwd.hip): a work item then spans TWO sample blocks (NS = 2) like the 512-slot kernels -- every weight fragment feeds eight MFMAs
// instead of four, half as many layer barriers and weight requests per row -- and the activations + the x tile fill the LDS exactly
// (128 + 32 KB).  Both kernels of a training step must take the same tile size (the ReLU-sign words are indexed by tile): this predicate
// is the one place that decides.  NF_MADE_TR128=0 in the environment or nf_config_made_tr128(0) switches it off (A/B runs, tests).
inline int &mf_tr128_switch() {       // (one instance for the library: nf_config_made_tr128 flips it -- tests, A/B runs)
    static int on = [] { const char *e = getenv("NF_MADE_TR128"); return (e && e[0] == '0') ? 0 : 1; }();
    return on;
}
inline bool mf_tr128(int64_t B, int hidden_padded, int dp) {
    if (!mf_tr128_switch() || hidden_padded != 256 || dp > 64 || B <= 0 || B % 128) return false;
    // The kernels are persistent over min(tiles, 256) workgroups: a launch lasts ceil(tiles / 256) rounds, a 128-row tile about 15/8 of
    // a 64-row tile (measured at 65 536 rows: -6.5 %).  128-row tiles only where the rounds come out shorter -- 32 768 rows: 1 round
    // against 2, 65 536: 2 against 4; NOT 49 152: 2 long rounds against 3 short ones.
    const int64_t r128 = (B / 128 + 255) / 256, r64 = (B / 64 + 255) / 256;
    return 15 * r128 < 8 * r64;
}

#define MF_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
// LDS-only barrier: the weight ring's global loads stay in flight across it (a __syncthreads() fence would drain them)
#define MF_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

// The wave's weight stream: ring entry j holds stream entry (position of a[0]) + j at every item start; an entry is consumed from
// its register and re-requested 8 entries ahead.  `ap` = the current item's first entry + 4 lane.
struct MfRing {
    f32x4 a[8];
    const float *ap;
};

__device__ __forceinline__ void mf_ring_start(MfRing &r, const float *stream, int lane) {
    r.ap = stream + lane * 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) r.a[j] = *reinterpret_cast<const f32x4 *>(r.ap + j * 256);
}

// four k-groups: ring entries HALF .. HALF + 3 = stream entries e0 .. e0 + 3 of the item; acc[sb] += W[32 x 32] . act[32 x 32 samples].
// MF_SCHED (ablation switch, tools/build_variant.py; measured in ONE gpurun call at config 5's layer, B = 65 536):
//   0 = the compiler's own schedule (default, 0.721 ms): hipcc sinks the eight re-requests of a loop iteration to its end and waits
//       for the first of them at the top of the next one -- which turns the two waves of a SIMD into a ping-pong: one issues its 64
//       MFMAs (4096 cycles) while the other's eight requests (one L2 round trip) are in flight;
//   1 = every entry's re-request pinned right behind the MFMAs that consumed it (__builtin_amdgcn_sched_barrier): 0.772 ms -- the
//       scheduling barriers also pin each k-group's LDS reads right in front of its MFMAs;
//   2 = 1 + the B operand software-pipelined one k-group ahead (b = this k-group's values on entry, the next group's on exit;
//       bp = the NEXT k-group's address, clamped to `blast`): 0.745 ms.
#ifndef MF_SCHED
#define MF_SCHED 0
#endif
// TR = rows per tile (64; 128 for the narrow NSF networks whose activations leave room for it): a k-group of activations is
// [2 hh][TR rows][4] = 8 TR floats.
template <int NS, int HALF, int TR = 64>
__device__ __forceinline__ void mf_group(MfRing &r, int e0, const float *&bp, const float *blast, f32x4 (&b)[2], f32x16 (&acc)[NS]) {
    constexpr int KGS = 8 * TR;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x4 bn[2];
        bn[0] = *reinterpret_cast<const f32x4 *>(bp);
        if constexpr (NS == 2) bn[1] = *reinterpret_cast<const f32x4 *>(bp + 128);
        if constexpr (MF_SCHED == 2) {
            bp = bp + KGS < blast ? bp + KGS : blast;
            __builtin_amdgcn_sched_barrier(0);       // (the reads go out IN FRONT of this k-group's MFMAs, not behind them)
        } else {
            bp += KGS;
            b[0] = bn[0];
            if constexpr (NS == 2) b[1] = bn[1];
        }
        const f32x4 av = r.a[HALF + j];
        if constexpr (MF_SCHED == 0) r.a[HALF + j] = *reinterpret_cast<const f32x4 *>(r.ap + (size_t)(e0 + j + 8) * 256);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[0] = MF_MFMA(av[i], b[0][i], acc[0]);
            if constexpr (NS == 2) acc[1] = MF_MFMA(av[i], b[1][i], acc[1]);
        }
        if constexpr (MF_SCHED != 0) {
            r.a[HALF + j] = *reinterpret_cast<const f32x4 *>(r.ap + (size_t)(e0 + j + 8) * 256);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (MF_SCHED == 2) {
            b[0] = bn[0];
            if constexpr (NS == 2) b[1] = bn[1];
        }
    }
}

// One work item: acc = (ADD ? acc : 0) + bias + W[32 rows][8 nkg] . act[8 nkg][32-sample blocks]; Bl = LDS activations + the lane's
// offset 4 (64 hh + n [+ 32 sb]).  nkg is a multiple of 4 (packer).
template <int NS, bool ADD, int TR = 64>
__device__ __forceinline__ void mf_item(MfRing &r, int nkg, const float *Bl, f32x16 (&acc)[NS]) {
    constexpr int KGS = 8 * TR;
    f32x4 b[2];
    const float *blast = Bl + (size_t)(nkg > 0 ? nkg - 1 : 0) * KGS;
    const float *bp = Bl;
    if constexpr (MF_SCHED == 2) {
        b[0] = *reinterpret_cast<const f32x4 *>(Bl);
        if constexpr (NS == 2) b[1] = *reinterpret_cast<const f32x4 *>(Bl + 128);
        bp = nkg > 1 ? Bl + KGS : Bl;
    }
    // bias group = ring entries 0..3: entry q holds bias[8 q + 4 hh + 0..3] = accumulator registers 4 q + i
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 bq = r.a[q];
        r.a[q] = *reinterpret_cast<const f32x4 *>(r.ap + (size_t)(q + 8) * 256);
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[s][4 * q + i] = ADD ? acc[s][4 * q + i] + bq[i] : bq[i];
    }
    if constexpr (MF_SCHED != 0) __builtin_amdgcn_sched_barrier(0);
    int kg = 0;
    for (; kg + 8 <= nkg; kg += 8) {
        mf_group<NS, 4, TR>(r, 4 + kg, bp, blast, b, acc);
        mf_group<NS, 0, TR>(r, 8 + kg, bp, blast, b, acc);
    }
    if (kg < nkg) {          // an odd number of A groups: bias + A groups is even, the next item starts in ring half 0
        mf_group<NS, 4, TR>(r, 4 + kg, bp, blast, b, acc);
    } else {                 // the next item's first entries sit in ring half 1: swap the halves
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 tmp = r.a[j];
            r.a[j] = r.a[4 + j];
            r.a[4 + j] = tmp;
        }
    }
    r.ap += (size_t)(4 + nkg) * 256;
}

// publish the row-block's values (ReLU'd or raw) as the next layer's B operand: act[(4 rb + q)][hh][32 sb + n][4]
template <int NS, bool RELU, int TR = 64>
__device__ __forceinline__ void mf_publish(float *acts, int rb, int sb0, int hh, int n, const f32x16 (&v)[NS]) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = RELU ? fmaxf(v[s][4 * q + i], 0.0f) : v[s][4 * q + i];
            *reinterpret_cast<f32x4 *>(acts + ((size_t)((4 * rb + q) * 2 + hh) * TR + 32 * (sb0 + s) + n) * 4) = o;
        }
}


// ---- final layer of a spline conditioner in GROUPS of four features = 3 row-blocks, both sample blocks of a pair ------------------
// (nsf_wide.hip, made_fwd.hip EPI 2): accumulator register `reg` of row-block r3 in lane-half hh is slot v = 16 r3 + reg of the lane's
// parameter list, feature 4 g + 2 hh + v / 24, parameter v % 24 (8 widths | 8 heights | 7 derivatives | pad): the order the register
// spline routine (fused_common.hpp rqs_regs) reads.  Stream of the item: 12 bias entries (row-block r3, quad q), then per k-group one
// A fragment per row-block.

// ring entry e of the current item (idx = e mod 8; a compile-time constant after unrolling at every call site): consume it,
// re-request it 8 entries ahead
__device__ __forceinline__ f32x4 mf_take(MfRing &r, int idx, int e) {
    const f32x4 v = r.a[idx];
    r.a[idx] = *reinterpret_cast<const f32x4 *>(r.ap + (size_t)(e + 8) * 256);
    return v;
}

// NK (8 or 4) k-groups: 3 NK ring entries starting at ring phase K0
template <int K0, int NK, int TR>
__device__ __forceinline__ void mf_final_kgs(MfRing &r, int e0, const float *bp, f32x16 (&o)[3][2]) {
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const f32x4 b0 = *reinterpret_cast<const f32x4 *>(bp + k * (8 * TR));
        const f32x4 b1 = *reinterpret_cast<const f32x4 *>(bp + k * (8 * TR) + 128);
#pragma unroll
        for (int r3 = 0; r3 < 3; ++r3) {
            const f32x4 av = mf_take(r, (K0 + 3 * k + r3) & 7, e0 + 3 * k + r3);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                o[r3][0] = MF_MFMA(av[i], b0[i], o[r3][0]);
                o[r3][1] = MF_MFMA(av[i], b1[i], o[r3][1]);
            }
        }
    }
}

// one group: o[r3][sb] = bias + W[32 rows] . act[., 32 samples]; nkg a multiple of 4 (packer); leaves the ring in phase 0
template <int TR>
__device__ __forceinline__ void mf_final_item(MfRing &r, int nkg, const float *Bl, f32x16 (&o)[3][2]) {
    // 12 bias entries: the item starts in ring phase 0; entries 8..11 come from the re-requested a[0..3]
#pragma unroll
    for (int e = 0; e < 12; ++e) {
        const f32x4 bq = mf_take(r, e & 7, e);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[e >> 2][0][4 * (e & 3) + i] = bq[i];
            o[e >> 2][1][4 * (e & 3) + i] = bq[i];
        }
    }
    // A entries start at ring phase 4 (12 mod 8); 8 k-groups = 24 entries = three revolutions keep the phase
    int kg = 0;
    for (; kg + 8 <= nkg; kg += 8) mf_final_kgs<4, 8, TR>(r, 12 + 3 * kg, Bl + (size_t)kg * (8 * TR), o);
    if (kg < nkg) {          // four more k-groups = 12 entries: 4 + 12 = 0 (mod 8), the next item starts in ring half 0
        mf_final_kgs<4, 4, TR>(r, 12 + 3 * kg, Bl + (size_t)kg * (8 * TR), o);
    } else {                 // 12 + 3 nkg = 4 (mod 8): the next item's first entries sit in ring half 1
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 tmp = r.a[j];
            r.a[j] = r.a[4 + j];
            r.a[4 + j] = tmp;
        }
    }
    r.ap += (size_t)(12 + 3 * nkg) * 256;
}

// ---- training (made_fwd.hip EPI 3 -> made_bwd.hip): what the backward pass needs of a row-block's pre-activations -------------------
// the values themselves, row-major [row][ld] (the weight-gradient kernel's operand order; rows beyond the batch are written as zeros
// so that the padded rows of a tile contribute nothing), and their signs as one dword per lane and item (bit 16 s + r = register r of
// sample block s: the ReLU mask of the input-gradient chain).
template <int NS, bool GUARD>
__device__ __forceinline__ void mf_save_rows(float *S, int ld, int nrows, int rb, int sb0, int hh, int n, const f32x16 (&v)[NS]) {
    float *base = S + 32 * rb;                                          // wave-uniform
    const unsigned lane_off = (unsigned)((32 * sb0 + n) * ld + 4 * hh);   // floats; < 64 ld
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const bool live = !GUARD || 32 * (sb0 + s) + n < nrows;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = live ? v[s][4 * q + i] : 0.0f;
            *reinterpret_cast<f32x4 *>(base + (lane_off + (unsigned)(32 * s * ld + 8 * q))) = o;
        }
    }
}

template <int NS>
__device__ __forceinline__ unsigned mf_sign_bits(const f32x16 (&v)[NS]) {
    unsigned bits = 0;
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) bits |= (v[s][r] > 0.0f ? 1u : 0u) << (16 * s + r);
    return bits;
}

// ADD: dst += src where the bit is set;  !ADD: dst = src where the bit is set, else 0
template <int NS, bool ADD>
__device__ __forceinline__ void mf_masked(f32x16 (&dst)[NS], const f32x16 (&src)[NS], unsigned bits) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float m = (bits >> (16 * s + r)) & 1u ? src[s][r] : 0.0f;
            dst[s][r] = ADD ? dst[s][r] + m : m;
        }
}

}  // namespace nf
