// maf_inverse_h.hip -- the incremental inverse of a masked autoregressive AFFINE layer (MAF sampling direction, BASELINE
// configs[4]; normflows/flows/affine/autoregressive.py:29-38, :114-128 over nets/made.py:217-304), second mapping: 32 samples per
// wave, the two lane-halves SHARE each sample's hidden units -- two waves per SIMD.
//
// The schedule, the packed blob and the table are those of maf_inverse.hip / flows/maf_pack.py (hidden units sorted by degree,
// tiles of <= 32 units holding whole degrees, per tile a block part on MFMA over all earlier tiles and a sequential part over its
// degrees); what changes is who holds what.  Round 1/2 (maf_inverse.hip, still the AR-NSF path): one wave = 64 samples, lane =
// sample, six 32-float register vectors per lane + operand stages = 482 registers = ONE wave per SIMD: MFMA issue (0.60 ms),
// the sequential vector-ALU part (0.46 ms), 96 product start-ups and every scratch / LDS round trip added up (1.8 ms per layer,
// MFMA busy 0.30).  Here:
//   * lane l = (sample n = l & 31, half hh = l >> 5).  A 32-unit tile's unit u belongs to half (u >> 2) & 1, register
//     (u & 3) + 4 (u >> 3) -- exactly where v_mfma_f32_32x32x2_f32 leaves row u of a [32 units x 32 samples] product: the block
//     part's accumulators ARE the per-lane vectors (no LDS transpose), 16 registers per layer instead of 32.
//   * sequential part: a unit's dot product over the tile's 32 source units = each half's 16 + one cross-half add (ds_bpermute);
//     the owner's pre-activation rides in the same sum, so one exchange gives both halves the total and the owner keeps it.  Half
//     the multiply-adds per lane; the diagonal blocks are staged into LDS with their columns in (half, register) order.
//   * activation scratch in B-operand order [k / 8][half][32 samples][4] as before: a tile is published with four 16-byte stores
//     per layer straight from the register quads, read back through the per-wave LDS-DMA ring (one 1 KB request per k-block).
// ~230 registers: 8 waves of 32 samples per workgroup = one workgroup per CU at B = 65 536, two waves per SIMD.
#include "common.hpp"
#include "fused_common.hpp"
#include <type_traits>

namespace nf {

constexpr int HT = 32;    // units per tile (flows/maf_pack.py TILE)
constexpr int HS = 16;    // degrees per tile
#ifndef NF_MAF_HNW
#define NF_MAF_HNW 4
#endif
constexpr int HNW = NF_MAF_HNW;    // waves per workgroup (ablation, round 4, config 5 in one gpurun call: 4 -> 13.9 ms, 8 = one 8-wave
                                   // workgroup per CU -> 14.5 ms, 2 -> 24.5 ms); two workgroups per CU (their tile phases drift apart: one's sequential part overlaps the other's block part)
constexpr int H_HDR = 8, H_ENT = 24;
// floats of a tile record after the A operands, NL = 1 + 2 num_blocks hidden layers: bias[NL][32] | biasF | W0d[32][16] | Wd[NL-1] | WFd
constexpr int h_seq(int NL) { return NL * HT + HT + HT * HS + (NL - 1) * HT * HT + HT * HT; }
#ifndef NF_MAF_LB
#define NF_MAF_LB 4     // round 5, same gpurun call: 8 activation slots 11.74 ms, 4 slots 11.14 ms (config 5; tools/maf_ablate5.py)
#endif
constexpr int h_lb(bool fast) { return fast ? NF_MAF_LB : 4; }     // activation slots of a wave's ring; + 4 + 4 weight slots of 1 KB

#define HMFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// c0[r] (+)= sum_k A[u(r, hh)][k] act[k][n] over K (a multiple of 32) features / units; PAIR: the NEXT tile's products over the
// same operands go to `stash` raw; INIT: c0 starts from the stash (second tile of a pair).  A, A2: [K/8][2][32][4] (L2),
// Sl: the wave's scratch [K/8][2][32][4].  ALL three streams arrive by LDS-DMA in the wave's own ring: LB slots (1 KB each) for the
// activation k-blocks -- requested LB steps ahead -- and 4 + 4 for the A [, A2] weight k-blocks, requested 4 steps ahead.
// TAIL (round 5): after the K features / units streamed from the scratch, 32 more whose activations are still IN REGISTERS -- the tile
// this wave has just finished: `breg` = its accumulator vector of the source layer, whose quads ARE the B operands of the four
// k-blocks -- contracted against the next four weight k-blocks of the same A [, A2] streams.  The newest tile then never makes the
// round trip store -> wait -> DMA (ablation: without the tile's 20 stores 10.3 instead of 11.2 ms; most of that was the wait for
// them in front of the next tile's first request), and an odd tile of a pair (K = its partner's 32 units) streams no activations.
template <bool PAIR, bool INIT, int LB, bool TAIL = false>
__device__ __forceinline__ void h_block(const float *__restrict__ A, const float *__restrict__ A2, const float *Sl, int K, int lane,
                                        float *stash, float *ring, f32x16 &c0, const f32x16 *breg = nullptr) {
    static_assert(LB == 4 || LB == 8, "activation slots: a power of two >= the 4 weight slots");
    f32x16 c2 = {0};
#pragma unroll
    for (int r = 0; r < 16; ++r) c0[r] = 0.0f;
    if constexpr (INIT) {
        const f32x4 *ps = reinterpret_cast<const f32x4 *>(stash) + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = ps[q * 64];
#pragma unroll
            for (int i = 0; i < 4; ++i) c0[4 * q + i] = v[i];
        }
    }
    const int nkm = K >> 3;                       // k-blocks whose activations come from the scratch
    const int nkb = nkm + (TAIL ? 4 : 0);         // + the register tail: weight k-blocks nkm .. nkm + 3
    if (nkb > 0) {
        // The requests are INLINE ASM and there is no ordinary load in the loop, so the compiler inserts no vector-memory wait of
        // its own and the hand-counted ones are exact: vector-memory operations retire in order; step kb needs A(kb) [, A2(kb)],
        // requested in (pseudo-)step kb - 4 AFTER that step's activation request, and B(kb), requested LB steps ago, i.e. earlier
        // still.  Younger than A2(kb) are exactly the requests of steps kb - 3 .. kb - 1: per step one activation block while
        // block (step + LB) exists and LPS weight blocks while block (step + 4) exists.
        // History (round 3): (1) the builtin + A operands as ordinary register loads: the compiler waited vmcnt(0) at every first
        // use of a loaded register while an LDS-DMA might be pending -- the ring drained every fourth k-block (2.9 TB/s, 16.1 ms);
        // (2) activation requests as asm, A still in registers (14.5 ms): the compiler's counted waits for A (it counts only its
        // own loads) stand in front of the interleaved requests it does not know of, so ~2.3 steps were in flight instead of 4;
        // (3) everything through the ring, 4 + 4 + 4 slots (13.9 ms, round 3/4); (4) round 5: LB = 8 activation slots for the
        // regular-tile kernel (the activation stream is the one that comes from HBM; its window was 4 KB per wave).
        constexpr int LPS = PAIR ? 2 : 1;
        const uint32_t ring_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)ring);
        // SGPR base + 32-bit lane offset (round 5): the three streams' bases are wave-uniform, so no 64-bit per-lane address is
        // computed or kept alive per stream (the VGPR-address form cost ~30 spilled registers once p stayed live across tiles)
        auto dma1 = [&](uint32_t dst, const float *base, uint32_t byte_off) {
#ifdef NF_MAF_ABL_NO_DMA
            return;
#endif
            // (m0 is a reserved register: the compiler does not honour it as a clobber, so it is saved and restored here)
            uint32_t m0_;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(m0_) : "s"(dst), "v"(byte_off), "s"(base) : "memory");
        };
        const uint32_t lane16 = (uint32_t)lane * 16u;
        auto req_b = [&](int kb) {
            if (kb < nkm) dma1(ring_lds + (uint32_t)(kb & (LB - 1)) * 1024u, Sl, (uint32_t)kb * 1024u + lane16);
        };
        auto req_a = [&](int kb, int slot) {
            if (kb < nkb) {
                const uint32_t off = (uint32_t)kb * 1024u + lane16;
                dma1(ring_lds + (uint32_t)(LB + slot) * 1024u, A, off);
                if constexpr (PAIR) dma1(ring_lds + (uint32_t)(LB + 4 + slot) * 1024u, A2, off);
            }
        };
        auto wait_for = [&](int kb) {
            const int ca = min(nkb - 1 - kb, 3);                         // steps kb-3 .. kb-1 that requested weight blocks
            const int cb = max(min(nkm - LB - kb + 3, 3), 0);            // ... and activation blocks
            switch (cb + LPS * ca) {
                case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                case 1: NF_WAIT_VMCNT(1); break;
                case 2: NF_WAIT_VMCNT(2); break;
                case 3: NF_WAIT_VMCNT(3); break;
                case 4: NF_WAIT_VMCNT(4); break;
                case 5: NF_WAIT_VMCNT(5); break;
                case 6: NF_WAIT_VMCNT(6); break;
                case 7: NF_WAIT_VMCNT(7); break;
                case 8: NF_WAIT_VMCNT(8); break;
                default: NF_WAIT_VMCNT(9); break;
            }
        };
        auto step = [&](int kb, int slot) {
            wait_for(kb);
            const f32x4 b = *reinterpret_cast<const f32x4 *>(ring + (kb & (LB - 1)) * 256 + lane * 4);
            const f32x4 a = *reinterpret_cast<const f32x4 *>(ring + (LB + slot) * 256 + lane * 4);
            f32x4 a2 = a;
            if constexpr (PAIR) a2 = *reinterpret_cast<const f32x4 *>(ring + (LB + 4 + slot) * 256 + lane * 4);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the slots' reads have returned: they may be requested again
            req_b(kb + LB);
            req_a(kb + 4, slot);
#ifndef NF_MAF_ABL_NO_MFMA       // (ablation builds, tools/maf_ablate5.py: timing only, results are garbage)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                c0 = HMFMA(a[i], b[i], c0);
                if constexpr (PAIR) c2 = HMFMA(a2[i], b[i], c2);
            }
#else
            c0[0] += a[0] * b[0] + a2[1];
#endif
        };
        // prologue: the first LB - 4 activation blocks, then pseudo-steps -4 .. -1 in the steps' own order (nkb is a multiple of 4)
#pragma unroll
        for (int e = 0; e < LB - 4; ++e) req_b(e);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            req_b(j + LB - 4);
            req_a(j, j);
        }
        for (int kb = 0; kb < nkm; kb += 4) {
            step(kb, 0);
            step(kb + 1, 1);
            step(kb + 2, 2);
            step(kb + 3, 3);
        }
        if constexpr (TAIL) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {         // weight k-block nkm + q (ring slot q: nkm is a multiple of 4) x register quad q
                wait_for(nkm + q);
                const f32x4 a = *reinterpret_cast<const f32x4 *>(ring + (LB + q) * 256 + lane * 4);
                f32x4 a2 = a;
                if constexpr (PAIR) a2 = *reinterpret_cast<const f32x4 *>(ring + (LB + 4 + q) * 256 + lane * 4);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#ifndef NF_MAF_ABL_NO_MFMA
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    c0 = HMFMA(a[i], (*breg)[4 * q + i], c0);
                    if constexpr (PAIR) c2 = HMFMA(a2[i], (*breg)[4 * q + i], c2);
                }
#endif
            }
        }
    }
    if constexpr (PAIR) {
        f32x4 *ps = reinterpret_cast<f32x4 *>(stash) + lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) ps[q * 64] = f32x4{c2[4 * q], c2[4 * q + 1], c2[4 * q + 2], c2[4 * q + 3]};
    }
}

// (Round 3, measured and dropped: the NL products of a tile as ONE stream -- requests and A look-ahead running through the
// product boundaries, one start-up per tile instead of one per product: 14.9 ms against 14.5-14.8, 18 spilled registers.  The
// start-ups are not what is left either.)
template <int LB, bool TAIL = false>
__device__ __forceinline__ void h_block_mode(int mode, const float *__restrict__ A, const float *__restrict__ A2, const float *Sl,
                                             int K, int lane, float *stash, float *ring, f32x16 &out, const f32x16 *breg = nullptr) {
    if (mode == 1) h_block<true, false, LB, TAIL>(A, A2, Sl, K, lane, stash, ring, out, breg);
    else if (mode == 2) h_block<false, true, LB, TAIL>(A, nullptr, Sl, K, lane, stash, ring, out, breg);
    else h_block<false, false, LB, TAIL>(A, nullptr, Sl, K, lane, nullptr, ring, out, breg);
}

__device__ __forceinline__ void h_finish(float us, float sh, float zf, float &xn, float &ld) {
    // v_rcp_f32 (1 ulp) instead of the IEEE division sequence (~10 instructions each, twice per feature on the sequential chain)
    const float scale = __builtin_amdgcn_rcpf(1.0f + __expf(-(us + 2.0f))) + 1e-3f;
    xn = (zf - sh) * __builtin_amdgcn_rcpf(scale);
    ld -= __logf(scale);
}

// ---- round 5: REGULAR tiles (flows/maf_pack.py format 1: <= 8 degrees of <= 4 units) -- the sequential part statically unrolled ----
// Unit i of step g sits at the row whose accumulator is register 2 g + (i & 1) of lane-half i >> 1: after step g registers
// 0 .. 2 g + 1 of both halves are final, a step's four targets are two per half (every lane owns what it finishes: no selects),
// and a dot product contracts only over the registers that can be non-zero under the mask (degree <= the step's): 4 (g + 1) packed
// multiply-adds per layer and step instead of 4 x 16 plain ones, weights as float4 = two targets x one register pair straight from
// the record (triangular, already in reading order), one v_permlane32_swap per target PAIR instead of one LDS permute per target.
// Measured on config 5 (B = 65 536, 10 layers): see DESIGN.md section 5c.
constexpr int HF_STEPS = 8;
constexpr int hf_w0(int g) { int o = 0; for (int i = 0; i < g; ++i) o += i / 2 + 1; return o; }   // float4 units within a half
constexpr int HF_W0 = hf_w0(HF_STEPS), HF_WD = HF_STEPS * (HF_STEPS + 1), HF_WF = HF_STEPS * (HF_STEPS + 1) / 2;
constexpr int hf_half4(int NL) { return HF_W0 + (NL - 1) * HF_WD + HF_WF; }
constexpr int hf_seq(int NL) { return (NL + 1) * HT + 8 * hf_half4(NL); }    // floats of a regular tile's record after the A operands
// tile kind 2 ("regular with extras", flows/maf_pack.py): <= 7 degrees, the first m <= 4 own a FIFTH unit that sits in the otherwise
// empty slot 7 (lane-half g & 1, register 14 + (g >> 1)); behind the two regular halves an extra region per half (float4 units):
// X1 [NL-1][7][2] registers 14, 15 -> the four regular targets | X2 [NL-1][4][3] the extra target over pairs 0..g and (14, 15) |
// XW [4] its window weights | XF [7] registers 14, 15 -> the scale / shift rows.  BASELINE configs[4]'s tile 0 is such a tile.
constexpr int HX_STEPS = 7, HX_MAX = 4;
constexpr int hx_half4(int NL) { return (NL - 1) * (HX_STEPS * 2 + HX_MAX * 3) + HX_MAX + HX_STEPS; }
constexpr int hfx_seq(int NL) { return hf_seq(NL) + 8 * hx_half4(NL); }

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 hf_fma(float w0, float w1, f32x2 s, f32x2 a) { return __builtin_elementwise_fma(f32x2{w0, w1}, s, a); }
// lanes 0-31 receive the wave-wide (both halves') total of `lo`, lanes 32-63 that of `hi`
__device__ __forceinline__ float hf_xchg(float lo, float hi) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// one hidden -> hidden product of step G: the four targets' partial sums over THIS half's registers 0 .. 2 G + 1 of `src`, the two
// exchanges; returns the totals of the half's own targets (registers 2 G, 2 G + 1)
// (the weights of a product are read HF_LA source pairs ahead of their multiply-adds and no further: left to itself the scheduler
// hoists all 2 (G + 1) 16-byte reads of a product -- up to 64 registers -- in front of the first multiply-add, and the kernel spills)
constexpr int HF_LA = 2;
// X: kind-2 tile -- registers 14, 15 (the extras) are sources too (wx1: two float4), and with `dox` (G < m) the step's own extra is
// a fifth target (wx2: its weights over pairs 0..G and (14, 15), three float4); its wave-wide total comes back in tx on every lane.
template <int G, bool X = false>
__device__ __forceinline__ void hf_product(const f32x16 &src, const f32x4 *w, float &ta, float &tb, const f32x4 *wx1 = nullptr,
                                           const f32x4 *wx2 = nullptr, bool dox = false, float *tx = nullptr) {
    f32x2 a0 = {0.0f, 0.0f}, a1 = a0, a2 = a0, a3 = a0;
    f32x4 wq[G + 1][2];
#pragma unroll
    for (int q = 0; q < HF_LA && q <= G; ++q) { wq[q][0] = w[2 * q]; wq[q][1] = w[2 * q + 1]; }
#pragma unroll
    for (int q = 0; q <= G; ++q) {
        if (q + HF_LA <= G) { wq[q + HF_LA][0] = w[2 * (q + HF_LA)]; wq[q + HF_LA][1] = w[2 * (q + HF_LA) + 1]; }
        const f32x4 w0 = wq[q][0], w1 = wq[q][1];
        const f32x2 s = {src[2 * q], src[2 * q + 1]};
        a0 = hf_fma(w0[0], w0[1], s, a0);
        a1 = hf_fma(w0[2], w0[3], s, a1);
        a2 = hf_fma(w1[0], w1[1], s, a2);
        a3 = hf_fma(w1[2], w1[3], s, a3);
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (X) {
        const f32x2 s7 = {src[14], src[15]};
        const f32x4 w0 = wx1[0], w1 = wx1[1];
        a0 = hf_fma(w0[0], w0[1], s7, a0);
        a1 = hf_fma(w0[2], w0[3], s7, a1);
        a2 = hf_fma(w1[0], w1[1], s7, a2);
        a3 = hf_fma(w1[2], w1[3], s7, a3);
        if constexpr (G < HX_MAX) {
            if (dox) {
                const f32x4 f0 = wx2[0], f1 = wx2[1], f2 = wx2[2];
                const float f[12] = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3], f2[0], f2[1], f2[2], f2[3]};
                f32x2 ax = {0.0f, 0.0f};
#pragma unroll
                for (int q = 0; q <= G; ++q) ax = hf_fma(f[2 * q], f[2 * q + 1], f32x2{src[2 * q], src[2 * q + 1]}, ax);
                ax = hf_fma(f[2 * (G + 1)], f[2 * (G + 1) + 1], s7, ax);
                const float sx = ax[0] + ax[1];
                *tx = hf_xchg(sx, sx);
            }
        }
    }
    ta = hf_xchg(a0[0] + a0[1], a2[0] + a2[1]);
    tb = hf_xchg(a1[0] + a1[1], a3[0] + a3[1]);
}

template <int NB, int G, bool X = false>
__device__ __forceinline__ void hf_step(f32x16 (&p)[1 + 2 * NB], const f32x16 &pF, f32x16 &xg, const f32x4 *wh, float zf, float &ld,
                                        float &xn, float &prm_out, const f32x4 *wx = nullptr, int m = 0, int hh = 0) {
    constexpr int NL = 1 + 2 * NB;
    // kind-2 tile: the step's extra unit = register RX of lane-half G & 1 (the other half's RX belongs to another step's extra)
    constexpr int RX = 14 + ((G < HX_MAX ? G : 0) >> 1);
    constexpr int OX2 = (NL - 1) * HX_STEPS * 2, OXW = OX2 + (NL - 1) * HX_MAX * 3, OXF = OXW + HX_MAX;
    const bool dox = X && G < HX_MAX && G < m;
    const bool ownx = dox && hh == (G & 1);
    {   // initial layer: h0 = pre + W0[window] . x for the half's own two targets; h0 folds into block 1's second pre-activation
        const f32x4 *w = wh + hf_w0(G);
        f32x2 a0 = {p[0][2 * G], 0.0f}, a1 = {p[0][2 * G + 1], 0.0f};
#pragma unroll
        for (int q = 0; q <= G / 2; ++q) {
            const f32x4 wq = w[q];
            const f32x2 x2 = {xg[2 * q], xg[2 * q + 1]};
            a0 = hf_fma(wq[0], wq[1], x2, a0);
            a1 = hf_fma(wq[2], wq[3], x2, a1);
        }
        const float ha = a0[0] + a0[1], hb = a1[0] + a1[1];
        p[2][2 * G] += ha;
        p[2][2 * G + 1] += hb;
        p[0][2 * G] = fmaxf(ha, 0.0f);
        p[0][2 * G + 1] = fmaxf(hb, 0.0f);
        if constexpr (X && G < HX_MAX) {
            if (dox) {          // the extra's initial layer: window pairs 0, 1 (G <= 3), weights on its own half only (zeros on the other)
                const f32x4 wq = wx[OXW + G];
                const float hx = p[0][RX] + (wq[0] * xg[0] + wq[1] * xg[1]) + (wq[2] * xg[2] + wq[3] * xg[3]);
                p[2][RX] = ownx ? p[2][RX] + hx : p[2][RX];
                p[0][RX] = ownx ? fmaxf(hx, 0.0f) : p[0][RX];
            }
        }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        float ta, tb, tx = 0.0f;
        hf_product<G, X>(p[2 * b], wh + HF_W0 + (2 * b) * HF_WD + G * (G + 1), ta, tb, wx + ((2 * b) * HX_STEPS + G) * 2,
                         wx + OX2 + ((2 * b) * HX_MAX + (G < HX_MAX ? G : 0)) * 3, dox, &tx);                  // t_b = L0_b(relu(h_b))
        p[2 * b + 1][2 * G] = fmaxf(p[2 * b + 1][2 * G] + ta, 0.0f);
        p[2 * b + 1][2 * G + 1] = fmaxf(p[2 * b + 1][2 * G + 1] + tb, 0.0f);
        if constexpr (X && G < HX_MAX) p[2 * b + 1][RX] = ownx ? fmaxf(p[2 * b + 1][RX] + tx, 0.0f) : p[2 * b + 1][RX];
        hf_product<G, X>(p[2 * b + 1], wh + HF_W0 + (2 * b + 1) * HF_WD + G * (G + 1), ta, tb, wx + ((2 * b + 1) * HX_STEPS + G) * 2,
                         wx + OX2 + ((2 * b + 1) * HX_MAX + (G < HX_MAX ? G : 0)) * 3, dox, &tx);              // h_{b+1} = h_b + L1_b(relu(t_b))
        const float ha = p[2 * b + 2][2 * G] + ta, hb = p[2 * b + 2][2 * G + 1] + tb;
        if constexpr (X && G < HX_MAX) {
            const float hx = p[2 * b + 2][RX] + tx;
            if (b + 1 < NB) {
                const int nx = 2 * b + 4 < NL ? 2 * b + 4 : 0;
                p[nx][RX] = ownx ? p[nx][RX] + hx : p[nx][RX];
                p[2 * b + 2][RX] = ownx ? fmaxf(hx, 0.0f) : p[2 * b + 2][RX];
            } else {
                p[2 * b + 2][RX] = ownx ? hx : p[2 * b + 2][RX];
            }
        }
        if (b + 1 < NB) {
            const int nx = 2 * b + 4 < NL ? 2 * b + 4 : 0;
            p[nx][2 * G] += ha;
            p[nx][2 * G + 1] += hb;
            p[2 * b + 2][2 * G] = fmaxf(ha, 0.0f);
            p[2 * b + 2][2 * G + 1] = fmaxf(hb, 0.0f);
        } else {
            p[2 * b + 2][2 * G] = ha;          // the final layer's input (made.py:304: no activation before it)
            p[2 * b + 2][2 * G + 1] = hb;
        }
    }
    {   // the step's feature: (unconstrained scale, shift) = final rows over the registers 0 .. 2 G + 1 of both halves
        const f32x4 *w = wh + HF_W0 + (NL - 1) * HF_WD + G * (G + 1) / 2;
        f32x2 au = {0.0f, 0.0f}, as = au;
        f32x4 wq[G + 1];
#pragma unroll
        for (int q = 0; q < 2 * HF_LA && q <= G; ++q) wq[q] = w[q];
#pragma unroll
        for (int q = 0; q <= G; ++q) {
            if (q + 2 * HF_LA <= G) wq[q + 2 * HF_LA] = w[q + 2 * HF_LA];
            const f32x2 s = {p[NL - 1][2 * q], p[NL - 1][2 * q + 1]};
            au = hf_fma(wq[q][0], wq[q][1], s, au);
            as = hf_fma(wq[q][2], wq[q][3], s, as);
            if (q & 1) __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (X) {
            const f32x4 wf = wx[OXF + G];
            const f32x2 s7 = {p[NL - 1][14], p[NL - 1][15]};
            au = hf_fma(wf[0], wf[1], s7, au);
            as = hf_fma(wf[2], wf[3], s7, as);
        }
        // lower half: scale (its block part + bias sit in register G of half 0), upper half: shift (register G of half 1)
        const float v = hf_xchg(au[0] + au[1], as[0] + as[1]) + pF[G];
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
        prm_out = v;       // (the lane-half's own MADE output of the feature -- scale below, shift above: the training forward stores it)
        h_finish(__uint_as_float(r[0]), __uint_as_float(r[1]), zf, xn, ld);
    }
    xg[G + 1] = xn;
}

template <int I, int N, class F>
__device__ __forceinline__ void h_static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        h_static_for<I + 1, N>(f);
    }
}

// NB residual blocks (nets/made.py:140-214): NL = 1 + 2 NB hidden layers whose activations later tiles contract over
// (S_0 = relu(h_0); per block b: S_{2b+1} = relu(t_b), S_{2b+2} = relu(h_{b+1}), the last one raw = the final layer's input).
// FAST: the launch covers REGULAR tiles only (format 1) and runs their statically unrolled sequential part; otherwise the tiles
// [t_beg, t_end) carry the format-0 record.  A layer whose tiles are of both kinds runs as consecutive launches (host:
// maf_h_launch): the per-sample state between tiles -- the feature row, the activation scratch, the pair stash -- is in memory
// anyway, the carry is the last feature written to y, and the log-determinant is accumulated launch by launch.  (One kernel holding
// both sequential parts needed more than 256 registers: 56 spilled.)
// TRAIN (round 6): the instantiation that also stores every feature's MADE outputs (`prm`); the inference instantiations carry none of
// it -- the config-5 kernel sits at exactly 256 registers, and three more live values spilled.
template <int NB, bool FAST, bool TRAIN = false>
__global__ void __launch_bounds__(64 * HNW, (NB <= 2 && HNW <= 4) ? 8 / HNW : 1)
maf_inverse_h_kernel(const float *__restrict__ z, float *__restrict__ y, float *__restrict__ logdet, const float *__restrict__ blob,
                     const int *__restrict__ table, float *S, float *Xs, float *Ps, int64_t B, int acc, int t_beg, int t_end,
                     unsigned int *__restrict__ bits, float *__restrict__ prm) {
    constexpr int NL = 1 + 2 * NB, H_SEQ = FAST ? hfx_seq(NL) : h_seq(NL), LB = h_lb(FAST);
    __shared__ __attribute__((aligned(16))) float seqw[H_SEQ];     // the tile's biases and diagonal blocks, shared by the workgroup's waves
    extern __shared__ __attribute__((aligned(16))) float dyn[];      // the waves' rings
    const int lane = threadIdx.x & 63, n = lane & 31, hh = lane >> 5;
    // wave-uniform BY CONSTRUCTION for the compiler too: every pointer and the tile mode below derive from it -- left in a vector
    // register the whole product dispatch ran under exec masks with vector loop counters
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *ringw = dyn + wid * ((LB + 8) * 256);
    const int64_t wt = (int64_t)blockIdx.x * HNW + wid;
    const bool active = wt * 32 < B;   // idle waves of the last workgroup still take part in the staging barriers
    const int D = table[0], Dp = table[1], Hp = table[3], T = table[4];
    const int64_t sample = wt * 32 + n;
    const bool valid = sample < B;
    const int64_t wts = active ? wt : 0;
    const float *zr = z + (valid ? sample : B - 1) * D;
    float *Sw = S + wts * ((int64_t)NL * Hp * 32);   // [layer][Hp/8][2][32][4]
    float *Xw = Xs + wts * ((int64_t)Dp * 32);       // [Dp/8][2][32][4]
    float *Pw = Ps ? Ps + wts * ((int64_t)NL * HT * 32) : nullptr;  // pair stash: [product][4][64][4] raw accumulators
    float ld = 0.0f, xcarry;
    if (t_beg == 0) {
        h_finish(blob[0], blob[1], zr[0], xcarry, ld);
        if (active && hh == 0) Xw[n * 4] = xcarry;
        if (valid && hh == 0) y[sample * D] = xcarry;
        // prm (round 6, the training forward): MADE's output at the solution, (B, 2 D) as nets/made.py:296-304 returns it -- every
        // feature's (unconstrained scale, shift) exists here when the feature is finished; lane-half 0 stores the first, 1 the second
        if constexpr (TRAIN) {
            if (valid) prm[sample * 2 * D + hh] = blob[hh];
        }
    } else {
        xcarry = y[(valid ? sample : B - 1) * D + table[H_HDR + H_ENT * t_beg] - 1];     // the last feature of the previous launch
    }
    const int t_stop = min(t_end, T);
    f32x16 p[NL], pF;       // p: the hidden layers' accumulators / activations of the CURRENT tile; FAST: kept across the tile boundary
#pragma unroll              // (the next tile's block part contracts over them from the registers)
    for (int l = 0; l < NL; ++l)
#pragma unroll
        for (int r = 0; r < 16; ++r) p[l][r] = 0.0f;
    if constexpr (FAST) {
        if (t_beg > 0 && active) {     // the predecessor tile ran in another launch: its published activations back into the registers
#pragma unroll
            for (int l = 0; l < NL; ++l)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 a = *reinterpret_cast<const f32x4 *>(Sw + (size_t)l * Hp * 32 + ((size_t)((4 * (t_beg - 1) + q) * 2 + hh) * 32 + n) * 4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) p[l][4 * q + i] = a[i];
                }
        }
    }
    for (int t = t_beg; t < t_stop; ++t) {
        const int *te = table + H_HDR + H_ENT * t;
        const int dlo = te[0], ns = te[1], K0 = te[2];
        const int Kh = HT * t;
        const float *rec = blob + te[3];
        const float *A0 = rec;
        const float *Ah = A0 + K0 * HT;             // A1..A_{NL-1}, AF: Kh * 32 floats each
        // stage the sequential part's weights, one copy per workgroup
#ifndef NF_MAF_ABL_NO_BARRIER
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // LDS-only barrier: a fence would wait for the activation stores in flight
#endif
        if constexpr (FAST) {       // regular tile: triangular record in reading order, copied as it is
            const f32x4 *src = reinterpret_cast<const f32x4 *>(Ah + (size_t)NL * Kh * HT);
#ifdef NF_MAF_ABL_NO_STAGE
            if (t == t_beg)
#endif
            const int n4 = (te[20] == 2 ? hfx_seq(NL) : hf_seq(NL)) / 4;
            for (int i = threadIdx.x; i < n4; i += 64 * HNW) reinterpret_cast<f32x4 *>(seqw)[i] = src[i];
        } else {                    // the five 32 x 32 diagonal blocks with their columns in (half, register) order: source quad
                                    // (u, v / 4 = 2 q + h) -> destination quad (u, 4 h + q)
            const f32x4 *src = reinterpret_cast<const f32x4 *>(Ah + (size_t)NL * Kh * HT);
            constexpr int HEAD4 = (NL * HT + HT + HT * HS) / 4;      // biases, final biases, window weights: copied as they are
            for (int i = threadIdx.x; i < H_SEQ / 4; i += 64 * HNW) {
                int d = i;
                if (i >= HEAD4) {
                    const int j = i - HEAD4, vq = j & 7;
                    d = HEAD4 + (j & ~7) + 4 * (vq & 1) + (vq >> 1);
                }
                reinterpret_cast<f32x4 *>(seqw)[d] = src[i];
            }
        }
#ifndef NF_MAF_ABL_NO_BARRIER
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // LDS-only barrier: a fence would wait for the activation stores in flight
#endif
        if (!active) continue;
        const float *bias = seqw;
        const float *biasF = bias + NL * HT;

        f32x16 zin;
#pragma unroll
        for (int j = 0; j < (FAST ? HF_STEPS : HS); ++j) zin[j] = (j < ns) ? zr[dlo + j] : 0.0f;

        {
            // tile pairing: even tile = its own products + the next tile's over the same operands (raw accumulators to the
            // stash); odd tile = the stash + the 32 units of its partner
#ifdef NF_MAF_ABL_NO_PAIR
            const int mode = 0;
#else
            const int mode = !Pw ? 0 : ((t & 1) ? 2 : (t + 1 < T ? 1 : 0));
#endif
            const float *Ah2 = Ah;
            int Kh2 = 0;
            if (mode == 1) {
                const int *te2 = te + H_ENT;
                Ah2 = blob + te2[3] + (size_t)te2[2] * HT;
                Kh2 = Kh + HT;
            }
            const int koff = mode == 2 ? Kh - HT : 0;         // the partner's units: the last 32 of this tile's K range
            if constexpr (FAST) {
                // The previous tile's activations are still in p: the last 32 units of every product's K range come from the
                // registers (h_block TAIL), the earlier ones from the scratch.  Products in DESCENDING layer order: product l
                // reads p[l] (the previous tile's layer-l activations) and yields the new p[l + 1], whose old value product l + 1 has
                // already consumed.  No fence: what is streamed was published at least one tile ago and every
                // product since ended on vmcnt(0); the feature scratch (written by the previous tile's steps) is read LAST.
                // (The first tile of a launch found its predecessor's activations loaded from the scratch, see the kernel's prologue;
                // tile 0 has no predecessor: its hidden products are empty.)
                if (t > 0) {
                    const int Km = mode == 2 ? 0 : Kh - HT;
#pragma unroll
                    for (int l = NL - 1; l >= 0; --l)      // (straight into p[l + 1]: its old value died with product l + 1)
                        h_block_mode<LB, true>(mode, Ah + (size_t)l * Kh * HT + (size_t)koff * HT, Ah2 + (size_t)l * Kh2 * HT,
                                               Sw + (size_t)l * Hp * 32, Km, lane, Pw + (size_t)l * HT * 32, ringw,
                                               l + 1 < NL ? p[l + 1 < NL ? l + 1 : 0] : pF, &p[l]);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        pF[r] = 0.0f;
#pragma unroll
                        for (int l = 1; l < NL; ++l) p[l][r] = 0.0f;
                    }
                    if (Pw && T > 1) {       // tile 0 leads the first pair: its partner's products over no operands = zeros in the stash
                        // (round 6: from a lane offset the compiler cannot hoist -- the 4 NL store addresses used to be computed once per
                        // wave and kept as twelve 64-bit pairs across the whole tile loop: 24 spilled registers in the config-5 kernel)
                        unsigned zo = (unsigned)lane * 16u;
                        asm volatile("" : "+v"(zo));
#pragma unroll
                        for (int l = 0; l < NL; ++l)
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                *reinterpret_cast<f32x4 *>(reinterpret_cast<char *>(Pw + (size_t)l * HT * 32 + q * 256) + zo) =
                                    f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                    }
                }
                h_block<false, false, LB>(A0, nullptr, Xw, K0, lane, nullptr, ringw, p[0]);
            } else {
                __threadfence_block();  // the activation scratch written by the other lanes of this wave is read below
                const int Kb = mode == 2 ? HT : Kh;
                h_block<false, false, LB>(A0, nullptr, Xw, K0, lane, nullptr, ringw, p[0]);
#pragma unroll
                for (int l = 0; l < NL; ++l)
                    h_block_mode<LB>(mode, Ah + (size_t)l * Kh * HT + (size_t)koff * HT, Ah2 + (size_t)l * Kh2 * HT,
                                     Sw + (size_t)l * Hp * 32 + (size_t)koff * 32, Kb, lane, Pw + (size_t)l * HT * 32, ringw,
                                     l + 1 < NL ? p[l + 1 < NL ? l + 1 : 0] : pF);
            }
        }
#ifndef NF_MAF_ABL_NO_BIAS
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int u = (r & 3) + 8 * (r >> 2) + 4 * hh;
#pragma unroll
            for (int l = 0; l < NL; ++l) p[l][r] += bias[l * HT + u];
            pF[r] += biasF[u];
        }
#endif
        f32x16 xg = {0};   // window features dlo-1 .. dlo+14 (0-based): xg[0] is the carry, xg[s+1] the output of step s
        xg[0] = xcarry;

        if constexpr (FAST) {
            const f32x4 *wh = reinterpret_cast<const f32x4 *>(seqw + (NL + 1) * HT) + hh * hf_half4(NL);
            const f32x4 *wx = reinterpret_cast<const f32x4 *>(seqw + hf_seq(NL)) + hh * hx_half4(NL);
            const bool xtile = te[20] == 2;
            const int mx = te[21];
            h_static_for<0, HF_STEPS>([&](auto G_) {
                constexpr int G = decltype(G_)::value;
                if (G < ns) {
                    float xn, pv = 0.0f;
#ifndef NF_MAF_ABL_NO_SEQ
                    if (G < HX_STEPS && xtile) hf_step<NB, (G < HX_STEPS ? G : 0), true>(p, pF, xg, wh, zin[G], ld, xn, pv, wx, mx, hh);
                    else hf_step<NB, G>(p, pF, xg, wh, zin[G], ld, xn, pv);
#else
                    xn = zin[G] + p[0][G] + pF[G] + wh[G][0];
#endif
                    xcarry = xn;
                    const int f = dlo + G;
                    if constexpr (TRAIN) {      // wave-uniform base + 32-bit lane offset: no 64-bit per-lane address kept across the steps
                        if (valid) (prm + wts * 64 * D)[(unsigned)(n * 2 * D + hh) + (unsigned)(2 * f)] = pv;
                    }
                    if (hh == 0) {
                        Xw[((size_t)((f >> 3) * 2 + ((f >> 2) & 1)) * 32 + n) * 4 + (f & 3)] = xn;
                        if (valid) y[sample * D + f] = xn;
                    }
                }
            });
        } else {
        const float *W0d = biasF + HT;
        const float *Wd = W0d + HT * HS;
        const float *WFd = Wd + (NL - 1) * HT * HT;
        auto xsum = [](float v) { return v + __shfl_xor(v, 32, 64); };

        // partial dot product of row u of a diagonal block with the 16 source registers of this lane-half
#define MAFH_DOT(WBASE, SRC)                                                                     \
            const f32x4 *w_ = reinterpret_cast<const f32x4 *>(WBASE) + u * (HT / 4) + 4 * hh;    \
            float a0 = 0.0f, a1 = 0.0f;                                                          \
            _Pragma("unroll") for (int q = 0; q < 4; q += 2) {                                   \
                const f32x4 wa = w_[q], wb = w_[q + 1];                                          \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                  \
                    a0 = fmaf(wa[i], SRC[4 * q + i], a0);                                        \
                    a1 = fmaf(wb[i], SRC[4 * q + 4 + i], a1);                                    \
                }                                                                                \
            }
        for (int s = 0; s < ns; ++s) {
            const unsigned m = (unsigned)te[4 + s];
            // initial layer: h0 = pre + W0[window] . x ; the residual h0 is folded into the pre-activation of block 1's
            // second linear (p[2]), p[0] keeps relu(h0) = input of block 1's first linear
            for (unsigned mm = m; mm; mm &= mm - 1) {
                const int u = __builtin_amdgcn_readfirstlane(__builtin_ctz(mm));
                const int ru = (u & 3) + 4 * (u >> 3);
                const bool own = hh == ((u >> 2) & 1);
                const f32x4 *w_ = reinterpret_cast<const f32x4 *>(W0d) + u * (HS / 4);
                float a = p[0][ru];
#pragma unroll
                for (int f = 0; f < HS; f += 4) {
                    const f32x4 w = w_[f / 4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) a = fmaf(w[i], xg[f + i], a);
                }
                p[2][ru] = own ? p[2][ru] + a : p[2][ru];
                p[0][ru] = own ? fmaxf(a, 0.0f) : p[0][ru];
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                // t_b = L0_b(relu(h_b)): every unit of the degree finishes a layer before the next one starts (same-degree
                // units see each other across layers, made.py:63-81)
                for (unsigned mm = m; mm; mm &= mm - 1) {
                    const int u = __builtin_amdgcn_readfirstlane(__builtin_ctz(mm));
                    const int ru = (u & 3) + 4 * (u >> 3);
                    const bool own = hh == ((u >> 2) & 1);
                    MAFH_DOT(Wd + (2 * b) * HT * HT, p[2 * b])
                    const float tot = xsum((own ? p[2 * b + 1][ru] : 0.0f) + (a0 + a1));
                    p[2 * b + 1][ru] = own ? fmaxf(tot, 0.0f) : p[2 * b + 1][ru];
                }
                // h_{b+1} = h_b + L1_b(relu(t_b)); h_b sits in the pre-activation already
                for (unsigned mm = m; mm; mm &= mm - 1) {
                    const int u = __builtin_amdgcn_readfirstlane(__builtin_ctz(mm));
                    const int ru = (u & 3) + 4 * (u >> 3);
                    const bool own = hh == ((u >> 2) & 1);
                    MAFH_DOT(Wd + (2 * b + 1) * HT * HT, p[2 * b + 1])
                    const float hn = xsum((own ? p[2 * b + 2][ru] : 0.0f) + (a0 + a1));
                    if constexpr (true) {
                        if (b + 1 < NB) {
                            p[2 * b + 4 < NL ? 2 * b + 4 : 0][ru] = own ? p[2 * b + 4 < NL ? 2 * b + 4 : 0][ru] + hn : p[2 * b + 4 < NL ? 2 * b + 4 : 0][ru];
                            p[2 * b + 2][ru] = own ? fmaxf(hn, 0.0f) : p[2 * b + 2][ru];
                        } else {
                            p[2 * b + 2][ru] = own ? hn : p[2 * b + 2][ru];   // the final layer's input (made.py:304: no activation before it)
                        }
                    }
                }
            }
            {
                float us, sh, xn;
                {
                    const int u = 2 * s, ru = (u & 3) + 4 * (u >> 3);
                    const bool own = hh == ((u >> 2) & 1);
                    MAFH_DOT(WFd, p[NL - 1])
                    us = xsum((own ? pF[ru] : 0.0f) + (a0 + a1));
                }
                {
                    const int u = 2 * s + 1, ru = (u & 3) + 4 * (u >> 3);
                    const bool own = hh == ((u >> 2) & 1);
                    MAFH_DOT(WFd, p[NL - 1])
                    sh = xsum((own ? pF[ru] : 0.0f) + (a0 + a1));
                }
                h_finish(us, sh, zin[s], xn, ld);
                if constexpr (TRAIN) {
                    if (valid) (prm + wts * 64 * D)[(unsigned)(n * 2 * D + hh) + (unsigned)(2 * (dlo + s))] = hh ? sh : us;
                }
                // a STATICALLY indexed write (16 selects): with `xg[s + 1] = xn` the compiler's dynamically indexed register write
                // went out of the vector's registers in the round-3 build of the block part (it overwrote request addresses that
                // live across the tile loop: a memory fault on layers with 10-16 degrees per tile; bisected with debug builds)
#pragma unroll
                for (int j_ = 1; j_ < HS; ++j_) xg[j_] = (j_ == s + 1) ? xn : xg[j_];
                xcarry = xn;
                const int f = dlo + s;
                if (hh == 0) {
                    Xw[((size_t)((f >> 3) * 2 + ((f >> 2) & 1)) * 32 + n) * 4 + (f & 3)] = xn;
                    if (valid) y[sample * D + f] = xn;
                }
            }
        }
#undef MAFH_DOT
        }
        // ---- publish the tile: the register quads ARE the B-operand entries (k-block 4 t + q, half hh) ----
        // Not the tiles nobody streams back: the layer's last tile, and on the fast path (tile t + 1 contracts tile t from the
        // registers) the one before it when this launch runs through the end of the layer (a later LAUNCH would reload it).
#ifdef NF_MAF_ABL_PUBLISH_ALL
        const bool pub = true;
#else
        const bool pub = bits != nullptr || (FAST ? (t + 2 < T || (t + 1 < T && t_stop < T)) : t + 1 < T);   // (bits: the training
        // forward -- nf_maf_scratch_rows reads the whole scratch back as the weight-gradient launch's activations)
#endif
#ifdef NF_MAF_ABL_NO_PUBLISH
        if (p[0][0] == 1.2345f)
#else
        if (pub)
#endif
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const size_t o = ((size_t)((4 * t + q) * 2 + hh) * 32 + n) * 4;
#pragma unroll
            for (int l = 0; l < NL; ++l)
                *reinterpret_cast<f32x4 *>(Sw + (size_t)l * Hp * 32 + o) = f32x4{p[l][4 * q], p[l][4 * q + 1], p[l][4 * q + 2], p[l][4 * q + 3]};
        }
        // ... and, for the implicit backward (maf_solve_t_kernel below), the ReLU masks of the tile: per lane NB words, bit
        // (l & 1) 16 + r of word l >> 1 = [activation of layer l in register r > 0] (S_l = relu(pre_l) for l < NL - 1)
        if (bits) {      // (positions = this pack's: the transposed pack of the solve mirrors the same plan and format)
            unsigned int *bw = bits + (((size_t)wts * T + t) * 64 + lane) * NB;
#pragma unroll
            for (int w = 0; w < NB; ++w) {
                unsigned int word = 0;
#pragma unroll
                for (int b = 0; b < 32; ++b) word |= (p[2 * w + (b >> 4)][b & 15] > 0.0f ? 1u : 0u) << b;
                bw[w] = word;
            }
        }
    }
    if (valid && hh == 0) ld_store(logdet + sample, ld, acc);
}

// ---- round 5: the implicit backward of the inverse in ONE pass (autograd.MafInverseFn; flows/maf_pack.pack_made_transposed) ----
// Solves  v s + J^T g_p(v, g_ld) = g_x  (J = dMADE/dx at the solution x, g_p = the affine transform's parameter cotangent) by
// back-substitution: with virtual feature f' = D - 1 - m and virtual degree d' = D - d the input-gradient chain of MADE has exactly
// the structure the incremental inverse walks, for a virtual network whose initial layer is Wf^T (two inputs per feature: the
// cotangents of the unconstrained scale and the shift), whose hidden layers are the transposed hidden layers in reverse order, whose
// final layer is W0^T (one row per feature) and whose "activations" are the forward pass's ReLU masks (`bits`, as the format-0
// inverse kernel above leaves them: the tiles are the forward plan's in reverse order with the forward positions).  Same mapping,
// block part, ring, tile pairing and generic sequential part as maf_inverse_h_kernel<NB, false>; replaces the 15-25 sweeps of
// nf_made_backward (with a host read-back every other sweep) the round-4 implicit backward iterated.
constexpr int t_seq(int NL) { return HT * 2 * HS + (NL - 1) * HT * HT + HT * HT; }      // W0d'[32][32] | Wd'[NL-1] | WFd'

__device__ __forceinline__ void t_finish(float gxm, float xf, float us, float gxf, float gl, float &vv, float &ga) {
    const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-(us + 2.0f)));
    const float rs = __builtin_amdgcn_rcpf(sg + 1e-3f);
    vv = (gxf - gxm) * rs;
    ga = (vv * xf + gl * rs) * (sg * (1.0f - sg));
}

// Round 6 (late): REGULAR-8 tiles (table entry [21] = 1: the forward tile is a format-1 regular tile of exactly 8 degrees x 4 units, so
// virtual step S finalises forward step G = 7 - S: registers 2 G, 2 G + 1 of both lane-halves) run a statically unrolled sequential
// part: a step's four targets, their rows in the staged blocks and their mask bits are compile-time constants (the generic part walks
// a position bitmask with readfirstlane and writes its targets through 16-way selects), and the dot products skip the register quads
// that only hold units of LATER steps -- their weights are exact zeros under the masks, so skipping them leaves every sum as it is:
// the same accumulators in the same order, BIT-IDENTICAL to the generic part on finite data.  FAST instantiation: launched over runs of
// such tiles by nf_maf_solve_t_tri (host copy of the table), the other tiles by the generic instantiation; between launches the state
// is in memory as for the inverse kernel (feature scratch = the carry, activation scratch, pair stash).
template <int NB, int S_>
__device__ __forceinline__ void tf_step(f32x16 (&p)[1 + 2 * NB], const f32x16 &pF, f32x16 &xa, f32x16 &xb, const float *W0d, const float *Wd,
                                        const float *WFd, const unsigned int (&bw)[NB], int hh, float zx, float zu, float zg, float gl,
                                        float &vv, float &ga) {
    constexpr int NL = 1 + 2 * NB, G = 7 - S_, Q0 = G >> 1;       // the half's registers >= 2 G: quads Q0 .. 3
    auto xsum = [](float a) { return a + __shfl_xor(a, 32, 64); };
    // the two accumulators of the generic dot product (a0: quads 0, 2; a1: quads 1, 3) over the quads >= Q0
    auto dot = [&](const float *WBASE, int u, const f32x16 &src) {
        const f32x4 *w_ = reinterpret_cast<const f32x4 *>(WBASE) + u * (HT / 4) + 4 * hh;
        float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q < Q0) continue;
            const f32x4 w = w_[q];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (q & 1) a1 = fmaf(w[i], src[4 * q + i], a1);
                else a0 = fmaf(w[i], src[4 * q + i], a0);
            }
        }
        return a0 + a1;
    };
    // virtual layer 0: the block part + the window pairs 0 .. S (the later pairs are zeros)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        constexpr int dummy = 0;
        (void)dummy;
        const int ru = 2 * G + (i & 1), half = i >> 1, u = (ru & 3) + 8 * (ru >> 2) + 4 * half;
        const f32x4 *w_ = reinterpret_cast<const f32x4 *>(W0d) + u * (2 * HS / 4);
        float a = p[0][ru];
#pragma unroll
        for (int f = 0; f <= S_; f += 2) {
            const f32x4 w = w_[f / 2];
            a = fmaf(w[0], xa[f], a);
            a = fmaf(w[1], xb[f], a);
            a = fmaf(w[2], xa[f + 1], a);
            a = fmaf(w[3], xb[f + 1], a);
        }
        p[0][ru] = hh == half ? a : p[0][ru];
    }
#pragma unroll
    for (int k = 1; k < NL; ++k) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ru = 2 * G + (i & 1), half = i >> 1, u = (ru & 3) + 8 * (ru >> 2) + 4 * half;
            const bool own = hh == half;
            const float d = dot(Wd + (k - 1) * HT * HT, u, p[k - 1]);
            const float tot = xsum((own ? p[k][ru] : 0.0f) + d);
            const int lf = 2 * NB - k;
            const bool on = (bw[lf >> 1] >> ((lf & 1) * 16 + ru)) & 1u;
            const float val = on ? tot : 0.0f;
            p[k][ru] = own ? ((k & 1) ? val : p[k >= 2 ? k - 2 : 0][ru] + val) : p[k][ru];
        }
    }
    {
        constexpr int u = S_, ru = (u & 3) + 4 * (u >> 3);
        const bool own = hh == ((u >> 2) & 1);
        const float d = dot(WFd, u, p[NL - 1]);
        const float gxm = xsum((own ? pF[ru] : 0.0f) + d);
        t_finish(gxm, zx, zu, zg, gl, vv, ga);
    }
    xa[S_ + 1] = ga;
    xb[S_ + 1] = vv;
}

template <int NB, bool FAST = false>
__global__ void __launch_bounds__(64 * HNW, (NB <= 2 && HNW <= 4) ? 8 / HNW : 1)
maf_solve_t_kernel(const float *__restrict__ x, const float *__restrict__ prm, const float *__restrict__ gx,
                   const float *__restrict__ gld, const unsigned int *__restrict__ bits, float *__restrict__ v,
                   const float *__restrict__ blob, const int *__restrict__ table, float *S, float *Xs, float *Ps, int64_t B,
                   int t_beg, int t_end) {
    constexpr int NL = 1 + 2 * NB, H_SEQ = t_seq(NL), LB = 4;
    __shared__ __attribute__((aligned(16))) float seqw[H_SEQ];
    extern __shared__ __attribute__((aligned(16))) float dyn[];
    const int lane = threadIdx.x & 63, n = lane & 31, hh = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *ringw = dyn + wid * ((LB + 8) * 256);
    const int64_t wt = (int64_t)blockIdx.x * HNW + wid;
    const bool active = wt * 32 < B;
    const int D = table[0], Dq = table[1], Hp = table[3], T = table[4];
    const int64_t sample = wt * 32 + n;
    const bool valid = sample < B;
    const int64_t wts = active ? wt : 0;
    const int64_t row = valid ? sample : B - 1;
    const float *xr = x + row * D, *pr = prm + row * 2 * D, *gxr = gx + row * D;
    const float gl = gld ? gld[row] : 0.0f;
    float *Sw = S + wts * ((int64_t)NL * Hp * 32);
    float *Xw = Xs + wts * ((int64_t)Dq * 32);
    float *Pw = Ps ? Ps + wts * ((int64_t)NL * HT * 32) : nullptr;
    const unsigned int *bwp = bits + ((size_t)wts * T * 64 + lane) * NB;
    auto xpos = [&](int q) { return ((size_t)((q >> 3) * 2 + ((q >> 2) & 1)) * 32 + n) * 4 + (q & 3); };
    float ca, cb;                                   // the carry: (g_us, g_sh) of the last feature produced
    if (t_beg == 0) {
        float vv;
        t_finish(0.0f, xr[D - 1], pr[2 * (D - 1)], gxr[D - 1], gl, vv, ca);
        cb = vv;
        if (active && hh == 0) { Xw[xpos(0)] = ca; Xw[xpos(1)] = cb; }
        if (valid && hh == 0) v[sample * D + D - 1] = vv;
    } else {                                        // a later launch of the layer: the previous launch's last feature, from the scratch
        const int fq = table[H_HDR + H_ENT * t_beg] - 1;
        ca = Xw[xpos(2 * fq)];
        cb = Xw[xpos(2 * fq + 1)];
    }
    auto xsum = [](float a) { return a + __shfl_xor(a, 32, 64); };
    const int t_stop = min(t_end, T);

    for (int t = t_beg; t < t_stop; ++t) {
        const int *te = table + H_HDR + H_ENT * t;
        const int dlo = te[0], ns = te[1], K0 = te[2], tf = te[20];
        const int Kh = HT * t;
        const float *rec = blob + te[3];
        const float *A0 = rec;
        const float *Ah = A0 + K0 * HT;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        {
            const f32x4 *src = reinterpret_cast<const f32x4 *>(Ah + (size_t)NL * Kh * HT);
            constexpr int HEAD4 = (HT * 2 * HS) / 4;       // the window weights: copied as they are; the diagonal blocks with their
            for (int i = threadIdx.x; i < H_SEQ / 4; i += 64 * HNW) {      // columns in (half, register) order
                int d = i;
                if (i >= HEAD4) {
                    const int j = i - HEAD4, vq = j & 7;
                    d = HEAD4 + (j & ~7) + 4 * (vq & 1) + (vq >> 1);
                }
                reinterpret_cast<f32x4 *>(seqw)[d] = src[i];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (!active) continue;
        const float *W0d = seqw;
        const float *Wd = W0d + HT * 2 * HS;
        const float *WFd = Wd + (NL - 1) * HT * HT;

        // step s produces real feature D - 1 - (dlo + s): its x, unconstrained scale and cotangent
        f32x16 zx, zu, zg;
#pragma unroll
        for (int j = 0; j < (FAST ? 8 : HS); ++j) {         // (FAST: eight steps -- the upper halves of the vectors are never touched)
            const int f = D - 1 - (dlo + j);
            zx[j] = (j < ns) ? xr[f] : 0.0f;
            zu[j] = (j < ns) ? pr[2 * f] : 0.0f;
            zg[j] = (j < ns) ? gxr[f] : 0.0f;
        }
        unsigned int bw[NB];
#pragma unroll
        for (int w = 0; w < NB; ++w) bw[w] = bwp[(size_t)tf * 64 * NB + w];

        __threadfence_block();
        f32x16 p[NL], pF;
        h_block<false, false, LB>(A0, nullptr, Xw, K0, lane, nullptr, ringw, p[0]);
        {
            const int mode = !Pw ? 0 : ((t & 1) ? 2 : (t + 1 < T ? 1 : 0));
            const float *Ah2 = Ah;
            int Kh2 = 0;
            if (mode == 1) {
                const int *te2 = te + H_ENT;
                Ah2 = blob + te2[3] + (size_t)te2[2] * HT;
                Kh2 = Kh + HT;
            }
            const int koff = mode == 2 ? Kh - HT : 0;
            const int Kb = mode == 2 ? HT : Kh;
#pragma unroll
            for (int l = 0; l < NL; ++l)
                h_block_mode<LB>(mode, Ah + (size_t)l * Kh * HT + (size_t)koff * HT, Ah2 + (size_t)l * Kh2 * HT,
                                 Sw + (size_t)l * Hp * 32 + (size_t)koff * 32, Kb, lane, Pw + (size_t)l * HT * 32, ringw,
                                 l + 1 < NL ? p[l + 1 < NL ? l + 1 : 0] : pF);
        }
        // window inputs: pair j = (g_us, g_sh) of virtual feature dlo - 1 + j; pair 0 is the carry
        f32x16 xa = {0}, xb = {0};
        xa[0] = ca;
        xb[0] = cb;
#define MAFT_DOT(WBASE, SRC)                                                                     \
            const f32x4 *w_ = reinterpret_cast<const f32x4 *>(WBASE) + u * (HT / 4) + 4 * hh;    \
            float a0 = 0.0f, a1 = 0.0f;                                                          \
            _Pragma("unroll") for (int q = 0; q < 4; q += 2) {                                   \
                const f32x4 wa = w_[q], wb = w_[q + 1];                                          \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                  \
                    a0 = fmaf(wa[i], SRC[4 * q + i], a0);                                        \
                    a1 = fmaf(wb[i], SRC[4 * q + 4 + i], a1);                                    \
                }                                                                                \
            }
        if constexpr (FAST) {
            h_static_for<0, 8>([&](auto S_) {
                constexpr int SS = decltype(S_)::value;
                float vv, ga;
                tf_step<NB, SS>(p, pF, xa, xb, W0d, Wd, WFd, bw, hh, zx[SS], zu[SS], zg[SS], gl, vv, ga);
                ca = ga;
                cb = vv;
                const int fq = dlo + SS;
                if (hh == 0) {
                    Xw[xpos(2 * fq)] = ga;
                    Xw[xpos(2 * fq + 1)] = vv;
                    if (valid) v[sample * D + (D - 1 - fq)] = vv;
                }
            });
        } else
        for (int s = 0; s < ns; ++s) {
            const unsigned m = (unsigned)te[4 + s];
            // virtual layer 0: G_top = Wf^T g_p -- the block part + the window pairs; raw (no mask)
            for (unsigned mm = m; mm; mm &= mm - 1) {
                const int u = __builtin_amdgcn_readfirstlane(__builtin_ctz(mm));
                const int ru = (u & 3) + 4 * (u >> 3);
                const bool own = hh == ((u >> 2) & 1);
                const f32x4 *w_ = reinterpret_cast<const f32x4 *>(W0d) + u * (2 * HS / 4);
                float a = p[0][ru];
#pragma unroll
                for (int f = 0; f < HS; f += 2) {
                    const f32x4 w = w_[f / 2];      // (us, sh) weights of window pairs f, f + 1
                    a = fmaf(w[0], xa[f], a);
                    a = fmaf(w[1], xb[f], a);
                    a = fmaf(w[2], xa[f + 1], a);
                    a = fmaf(w[3], xb[f + 1], a);
                }
                p[0][ru] = own ? a : p[0][ru];
            }
            // virtual layers k = 1 .. 2 NB: odd k: G = mask (V^T G_prev); even k: G = G_{k-2} + mask (V^T G_prev); the mask of
            // virtual layer k is the forward pass's sign bit of layer 2 NB - k at the same unit
#pragma unroll
            for (int k = 1; k < NL; ++k) {
                for (unsigned mm = m; mm; mm &= mm - 1) {
                    const int u = __builtin_amdgcn_readfirstlane(__builtin_ctz(mm));
                    const int ru = (u & 3) + 4 * (u >> 3);
                    const bool own = hh == ((u >> 2) & 1);
                    MAFT_DOT(Wd + (k - 1) * HT * HT, p[k - 1])
                    const float tot = xsum((own ? p[k][ru] : 0.0f) + (a0 + a1));
                    const int lf = 2 * NB - k;        // (constant after unrolling)
                    const bool on = (bw[lf >> 1] >> ((lf & 1) * 16 + ru)) & 1u;
                    const float val = on ? tot : 0.0f;
                    p[k][ru] = own ? ((k & 1) ? val : p[k >= 2 ? k - 2 : 0][ru] + val) : p[k][ru];
                }
            }
            {
                float gxm;
                {
                    const int u = s, ru = (u & 3) + 4 * (u >> 3);
                    const bool own = hh == ((u >> 2) & 1);
                    MAFT_DOT(WFd, p[NL - 1])
                    gxm = xsum((own ? pF[ru] : 0.0f) + (a0 + a1));
                }
                float vv, ga;
                t_finish(gxm, zx[s], zu[s], zg[s], gl, vv, ga);
#pragma unroll
                for (int j_ = 1; j_ < HS; ++j_) {        // statically indexed writes (see maf_inverse_h_kernel)
                    xa[j_] = (j_ == s + 1) ? ga : xa[j_];
                    xb[j_] = (j_ == s + 1) ? vv : xb[j_];
                }
                ca = ga;
                cb = vv;
                const int fq = dlo + s;
                if (hh == 0) {
                    Xw[xpos(2 * fq)] = ga;
                    Xw[xpos(2 * fq + 1)] = vv;
                    if (valid) v[sample * D + (D - 1 - fq)] = vv;
                }
            }
        }
#undef MAFT_DOT
#pragma unroll          // (every tile, the last one too: nf_maf_scratch_rows reads the whole scratch back as MADE's hidden gradients)
        for (int q = 0; q < 4; ++q) {
            const size_t o = ((size_t)((4 * t + q) * 2 + hh) * 32 + n) * 4;
#pragma unroll
            for (int l = 0; l < NL; ++l)
                *reinterpret_cast<f32x4 *>(Sw + (size_t)l * Hp * 32 + o) = f32x4{p[l][4 * q], p[l][4 * q + 1], p[l][4 * q + 2], p[l][4 * q + 3]};
        }
    }
}

}  // namespace nf

// Scratch of nf_maf_inverse_h: per sample NL hidden_padded activations + the padded feature row + the tile-pair stash.
extern "C" int64_t nf_maf_inverse_h_scratch_floats(int64_t B, int D, int hidden_padded, int num_blocks) {
    if (B < 0 || D < 1 || hidden_padded < 0 || num_blocks < 1 || num_blocks > 3) return NF_EINVAL;
    const int64_t nwt = (B + 31) / 32, Dp = (D + 31) / 32 * 32, NL = 1 + 2 * num_blocks;
    return nwt * 32 * (NL * (int64_t)hidden_padded + Dp + NL * nf::HT);
}

template <int NB, bool FAST>
static int maf_h_run(const void *z, void *y, void *logdet, const void *blob, const int32_t *table, float *S, float *Xs, float *Ps,
                     int64_t nwt, int64_t B, int acc, int t_beg, int t_end, hipStream_t st, unsigned int *bits = nullptr,
                     float *prm = nullptr) {
    using namespace nf;
    constexpr int NL = 1 + 2 * NB;
    const int grid = (int)((nwt + HNW - 1) / HNW);
    const size_t lds_ring = (size_t)HNW * (h_lb(FAST) + 8) * 256 * sizeof(float);
    if (prm) {
        static LdsOptIn opted_t;
        if (opt_in_lds(reinterpret_cast<const void *>(&maf_inverse_h_kernel<NB, FAST, true>),
                       lds_ring + sizeof(float) * (FAST ? hfx_seq(NL) : h_seq(NL)), opted_t) != NF_OK)
            return NF_ENOTSUP;
        hipLaunchKernelGGL((maf_inverse_h_kernel<NB, FAST, true>), dim3(grid), dim3(64 * HNW), lds_ring, st, (const float *)z, (float *)y,
                           (float *)logdet, (const float *)blob, (const int *)table, S, Xs, Ps, B, acc, t_beg, t_end, bits, prm);
        NF_CHECK_LAUNCH();
        return NF_OK;
    }
    static LdsOptIn opted;
    if (opt_in_lds(reinterpret_cast<const void *>(&maf_inverse_h_kernel<NB, FAST>),
                   lds_ring + sizeof(float) * (FAST ? hfx_seq(NL) : h_seq(NL)), opted) != NF_OK)
        return NF_ENOTSUP;
    hipLaunchKernelGGL((maf_inverse_h_kernel<NB, FAST>), dim3(grid), dim3(64 * HNW), lds_ring, st, (const float *)z, (float *)y,
                       (float *)logdet, (const float *)blob, (const int *)table, S, Xs, Ps, B, acc, t_beg, t_end, bits, prm);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// table_host: the host copy of `table` (format 1: the launcher needs the kinds of the tiles) or NULL (format 0: one launch)
template <int NB>
static int maf_h_launch(const void *z, void *y, void *logdet, const void *blob, const int32_t *table, const int32_t *table_host,
                        void *scratch, int64_t B, int D, int hidden_padded, int acc, hipStream_t st, unsigned int *bits = nullptr,
                        float *prm = nullptr) {
    using namespace nf;
    constexpr int NL = 1 + 2 * NB;
    const int64_t nwt = (B + 31) / 32;
    const int64_t Dp = (D + 31) / 32 * 32;
    float *S = (float *)scratch;
    float *Xs = S + nwt * 32 * (int64_t)NL * hidden_padded;
    // the feature scratch is read with zero weights before it is written (K0 is padded to 32): it must hold finite values
    if (hipMemsetAsync(Xs, 0, (size_t)nwt * 32 * Dp * sizeof(float), st) != hipSuccess) return NF_EIO;
    float *Ps = Xs + nwt * 32 * Dp;
    if (!table_host) return maf_h_run<NB, false>(z, y, logdet, blob, table, S, Xs, Ps, nwt, B, acc, 0, 1 << 30, st, bits, prm);
    // maximal runs of tiles of one kind, one launch each; the first accumulates as the caller says, the others on top of it
    const int T = table_host[4];
    for (int t0 = 0; t0 < T;) {
        const bool fast = table_host[H_HDR + H_ENT * t0 + 20] != 0;       // kind 1 (regular) and 2 (regular with extras)
        int t1 = t0 + 1;
        while (t1 < T && (table_host[H_HDR + H_ENT * t1 + 20] != 0) == fast) ++t1;
        const int a = t0 == 0 ? acc : (acc == NF_LD_SUB ? NF_LD_SUB : NF_LD_ADD);
        const int rc = fast ? maf_h_run<NB, true>(z, y, logdet, blob, table, S, Xs, Ps, nwt, B, a, t0, t1, st, bits, prm)
                            : maf_h_run<NB, false>(z, y, logdet, blob, table, S, Xs, Ps, nwt, B, a, t0, t1, st, bits, prm);
        if (rc != NF_OK) return rc;
        t0 = t1;
    }
    return NF_OK;
}

static int maf_h_entry(const void *z, void *y, void *logdet, const void *blob, const int32_t *table, const int32_t *table_host,
                       void *scratch, int64_t B, int D, int hidden_padded, int num_blocks, int acc, nf_stream_t stream,
                       unsigned int *bits = nullptr, float *prm = nullptr) {
    if (B < 0 || D < 2 || hidden_padded < 32 || hidden_padded % 32) return NF_EINVAL;
    if (acc < NF_LD_SUB || acc > NF_LD_ADD) return NF_EINVAL;
    if (num_blocks < 1 || num_blocks > 3) return NF_ENOTSUP;
    if (table_host && (table_host[0] != D || table_host[3] != hidden_padded || table_host[6] != num_blocks || table_host[7] != 1 ||
                       table_host[4] < 1 || table_host[4] * nf::HT != hidden_padded))
        return NF_EINVAL;
    if (B == 0) return NF_OK;
    if (!z || !y || !logdet || !blob || !table || !scratch) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    if (num_blocks == 1) return maf_h_launch<1>(z, y, logdet, blob, table, table_host, scratch, B, D, hidden_padded, acc, st, bits, prm);
    if (num_blocks == 2) return maf_h_launch<2>(z, y, logdet, blob, table, table_host, scratch, B, D, hidden_padded, acc, st, bits, prm);
    return maf_h_launch<3>(z, y, logdet, blob, table, table_host, scratch, B, D, hidden_padded, acc, st, bits, prm);
}

// nf_maf_inverse on the half-sharing mapping: same blob / table format and semantics as nf_maf_inverse (maf_inverse.hip), for
// MADE conditioners of 1, 2 or 3 residual blocks (table[6]; flows/maf_pack.py).  Format-0 packs only (table[7] == 0).
extern "C" int nf_maf_inverse_h(const void *z, void *y, void *logdet, const void *blob, const int32_t *table, void *scratch,
                                int64_t B, int D, int hidden_padded, int num_blocks, int acc, nf_stream_t stream) {
    return maf_h_entry(z, y, logdet, blob, table, nullptr, scratch, B, D, hidden_padded, num_blocks, acc, stream);
}

// The same for format-1 packs (flows/maf_pack.pack_made(..., tri=True)): regular tiles -- at most 8 degrees of at most 4 hidden units,
// 15 of the 16 tiles of BASELINE configs[4] -- run the triangular, statically unrolled sequential part and the 8-deep activation ring.
// table_host = the HOST copy of the same table (the launcher splits the tiles into runs of one kind; nothing is read back from the
// device); -EINVAL when it does not describe this call (D, hidden_padded, num_blocks, format).
extern "C" int nf_maf_inverse_h_tri(const void *z, void *y, void *logdet, const void *blob, const int32_t *table,
                                    const int32_t *table_host, void *scratch, int64_t B, int D, int hidden_padded, int num_blocks,
                                    int acc, nf_stream_t stream) {
    if (!table_host) return NF_EFAULT;
    return maf_h_entry(z, y, logdet, blob, table, table_host, scratch, B, D, hidden_padded, num_blocks, acc, stream);
}

// nf_maf_inverse_h that also leaves the ReLU masks of the pass for nf_maf_solve_t: bits = ceil(B / 32) * T * 64 * num_blocks uint32
// (T = table[4] tiles; per 32-row wave, tile and lane num_blocks words: bit (l & 1) 16 + r of word l >> 1 = [activation of hidden layer
// l at the unit in accumulator register r of that lane > 0]).  Format-0 packs only.
extern "C" int nf_maf_inverse_h_bits(const void *z, void *y, void *logdet, const void *blob, const int32_t *table, void *scratch,
                                     void *bits, int64_t B, int D, int hidden_padded, int num_blocks, int acc, nf_stream_t stream) {
    if (!bits && B > 0) return NF_EFAULT;
    return maf_h_entry(z, y, logdet, blob, table, nullptr, scratch, B, D, hidden_padded, num_blocks, acc, stream, (unsigned int *)bits);
}

// ... and on a format-1 pack (table_host as in nf_maf_inverse_h_tri): the masks in THAT pack's positions, for a transposed pack built
// with the same option (maf_pack.pack_made_transposed(tri=True)).
extern "C" int nf_maf_inverse_h_tri_bits(const void *z, void *y, void *logdet, const void *blob, const int32_t *table,
                                         const int32_t *table_host, void *scratch, void *bits, int64_t B, int D, int hidden_padded,
                                         int num_blocks, int acc, nf_stream_t stream) {
    if (!table_host || (!bits && B > 0)) return NF_EFAULT;
    return maf_h_entry(z, y, logdet, blob, table, table_host, scratch, B, D, hidden_padded, num_blocks, acc, stream, (unsigned int *)bits);
}

// The training forward in one entry point (round 6): nf_maf_inverse_h_bits (table_host NULL: format-0 pack) / nf_maf_inverse_h_tri_bits
// (format 1) that ALSO writes MADE's output at the solution, prm (B, 2 D) float32 = (unconstrained scale, shift) per feature in the
// reference's order (nets/made.py:296-304 as affine/autoregressive.py:98-128 reads it) -- each pair exists in registers when its
// feature is finished.  The implicit backward needs it (nf_maf_solve_t, nf_maf_affine_bwd); round 5 recomputed it from the pass's last
// hidden tensor with one rearrangement and one library product per layer.
extern "C" int nf_maf_inverse_h_train(const void *z, void *y, void *logdet, const void *blob, const int32_t *table,
                                      const int32_t *table_host, void *scratch, void *bits, void *prm, int64_t B, int D,
                                      int hidden_padded, int num_blocks, int acc, nf_stream_t stream) {
    if ((!bits || !prm) && B > 0) return NF_EFAULT;
    return maf_h_entry(z, y, logdet, blob, table, table_host, scratch, B, D, hidden_padded, num_blocks, acc, stream, (unsigned int *)bits,
                       (float *)prm);
}

// Scratch of nf_maf_solve_t: per row NL hidden_padded cotangents + the padded (g_us, g_sh) row + the tile-pair stash.
namespace nf {

// ---- the one-pass kernels' activation scratch as row-major hidden tensors (round 5) ---------------------------------------------------
// S: [wave tile of 32 rows][NL layers][Hp / 8][2][32][4] (B-operand order over padded POSITIONS, as maf_inverse_h_kernel /
// maf_solve_t_kernel publish it) -> out[layer'][Bp rows][ldo]: column c of a row = sign * S[position pos_of_col[c]] (pos_of_col[c] < 0:
// zero), layer' = NL - 1 - layer with `reverse`; rows >= B are zero.  For the transposed solve this IS MADE's input-gradient chain at
// the solution (G[l] of nf_made_backward in the training kernels' column order: flows/maf_pack.solve_t_gradient_columns) -- the
// weight-gradient launch of the implicit backward reads it instead of running the chain again.  Workgroup = (32 rows, layer): the
// Hp x 32 block goes through LDS as [row][Hp + 4] (16-byte writes, row stride = 4 banks mod 32), rows leave as full lines.
__global__ void __launch_bounds__(256)
maf_scratch_rows_kernel(const float *__restrict__ S, const int *__restrict__ pos_of_col, float *__restrict__ out, int64_t B, int64_t Bp,
                        int NL, int Hp, int ldo, float sign, int reverse, int l0) {
    extern __shared__ __attribute__((aligned(16))) float buf[];
    const int64_t wt = blockIdx.x;
    const int l = blockIdx.y + l0, pitch = Hp + 4;      // (l0 > 0: nf_maf_scratch_layer -- one layer, written as out[0])
    const bool have = wt * 32 < B;
    if (have) {
        const f32x4 *src = reinterpret_cast<const f32x4 *>(S + ((size_t)wt * NL + l) * (size_t)Hp * 32);
#pragma unroll 8
        for (int i = threadIdx.x; i < Hp * 8; i += 256) {          // f32x4 index i = (position group g = i >> 5, row n = i & 31)
            const int g = i >> 5, n = i & 31;
            *reinterpret_cast<f32x4 *>(buf + n * pitch + 4 * g) = __builtin_nontemporal_load(src + i);
        }
    }
    __syncthreads();
    const int lo = l0 > 0 || gridDim.y == 1 ? (int)blockIdx.y : (reverse ? NL - 1 - l : l);
    const int c4n = ldo >> 2;
    if (256 % c4n == 0) {      // (ldo 256 / 512 / 1024: a thread keeps its four columns -- their positions are loaded once, not per row)
        const int c4 = threadIdx.x % c4n, dn = 256 / c4n;
        int k4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) k4[i] = pos_of_col[4 * c4 + i];
        float *dst = out + ((size_t)lo * Bp + wt * 32) * ldo + 4 * c4;
#pragma unroll 4
        for (int n = threadIdx.x / c4n; n < 32; n += dn) {
            const int64_t row = wt * 32 + n;
            if (row >= Bp) break;
            f32x4 o = {0.0f, 0.0f, 0.0f, 0.0f};
            if (have && row < B) {
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = k4[i] >= 0 ? sign * buf[n * pitch + k4[i]] : 0.0f;
            }
            *reinterpret_cast<f32x4 *>(dst + (size_t)n * ldo) = o;
        }
        return;
    }
    for (int e = threadIdx.x; e < 32 * c4n; e += 256) {
        const int n = e / c4n, c4 = e - n * c4n;
        const int64_t row = wt * 32 + n;
        if (row >= Bp) continue;
        f32x4 o = {0.0f, 0.0f, 0.0f, 0.0f};
        if (have && row < B) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = pos_of_col[4 * c4 + i];
                if (k >= 0) o[i] = sign * buf[n * pitch + k];
            }
        }
        *reinterpret_cast<f32x4 *>(out + ((size_t)lo * Bp + row) * ldo + 4 * c4) = o;
    }
}

}  // namespace nf

// The activation scratch of nf_maf_solve_t (or of the inverse kernels) as row-major hidden tensors out (NL, Bp, ldo) float32, Bp = B
// rounded up to 64: see maf_scratch_rows_kernel.  pos_of_col: ldo int32 (device).  hidden_padded = the scratch's position count (the
// pack's table[3]); scratch = the buffer handed to the kernel that filled it (its activation region comes first).
extern "C" int nf_maf_scratch_rows(const void *scratch, const int32_t *pos_of_col, void *out, int64_t B, int num_blocks, int hidden_padded,
                                   int ldo, double sign, int reverse_layers, nf_stream_t stream) {
    if (B < 0 || num_blocks < 1 || num_blocks > 3 || hidden_padded < 32 || hidden_padded % 32 || ldo < 4 || ldo % 4) return NF_EINVAL;
    if (B == 0) return NF_OK;
    if (!scratch || !pos_of_col || !out) return NF_EFAULT;
    const int NL = 1 + 2 * num_blocks;
    const int64_t Bp = (B + 63) / 64 * 64;
    const size_t lds = (size_t)32 * (hidden_padded + 4) * sizeof(float);
    static nf::LdsOptIn opted = {};
    if (nf::opt_in_lds(reinterpret_cast<const void *>(&nf::maf_scratch_rows_kernel), lds, opted) != NF_OK) return NF_ENOTSUP;
    hipLaunchKernelGGL(nf::maf_scratch_rows_kernel, dim3((unsigned)(Bp / 32), NL), dim3(256), lds, (hipStream_t)stream, (const float *)scratch,
                       (const int *)pos_of_col, (float *)out, B, Bp, NL, hidden_padded, ldo, (float)sign, reverse_layers, 0);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// ONE layer of such a scratch as a row-major tensor out (Bp, ldo) (round 6: with nf_made_wgrad_pos reading the scratches in place, the
// implicit backward only needs the inverse pass's last hidden tensor in rows -- the input of the library product that gives MADE's
// output at the solution).
extern "C" int nf_maf_scratch_layer(const void *scratch, const int32_t *pos_of_col, void *out, int64_t B, int num_blocks, int hidden_padded,
                                    int ldo, int layer, nf_stream_t stream) {
    if (B < 0 || num_blocks < 1 || num_blocks > 3 || hidden_padded < 32 || hidden_padded % 32 || ldo < 4 || ldo % 4) return NF_EINVAL;
    const int NL = 1 + 2 * num_blocks;
    if (layer < 0 || layer >= NL) return NF_EINVAL;
    if (B == 0) return NF_OK;
    if (!scratch || !pos_of_col || !out) return NF_EFAULT;
    const int64_t Bp = (B + 63) / 64 * 64;
    const size_t lds = (size_t)32 * (hidden_padded + 4) * sizeof(float);
    static nf::LdsOptIn opted = {};
    if (nf::opt_in_lds(reinterpret_cast<const void *>(&nf::maf_scratch_rows_kernel), lds, opted) != NF_OK) return NF_ENOTSUP;
    hipLaunchKernelGGL(nf::maf_scratch_rows_kernel, dim3((unsigned)(Bp / 32), 1), dim3(256), lds, (hipStream_t)stream, (const float *)scratch,
                       (const int *)pos_of_col, (float *)out, B, Bp, NL, hidden_padded, ldo, 1.0f, 0, layer);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

extern "C" int64_t nf_maf_solve_t_scratch_floats(int64_t B, int D, int hidden_padded, int num_blocks) {
    if (B < 0 || D < 1 || hidden_padded < 0 || num_blocks < 1 || num_blocks > 3) return NF_EINVAL;
    const int64_t nwt = (B + 31) / 32, Dq = (2 * (int64_t)D + 31) / 32 * 32, NL = 1 + 2 * num_blocks;
    return nwt * 32 * (NL * (int64_t)hidden_padded + Dq + NL * nf::HT);
}

template <int NB, bool FAST>
static int maf_t_run(const void *x, const void *prm, const void *gx, const void *gld, const void *bits, void *v, const void *blob,
                     const int32_t *table, float *S, float *Xs, float *Ps, int64_t nwt, int64_t B, int t_beg, int t_end, hipStream_t st) {
    using namespace nf;
    constexpr int NL = 1 + 2 * NB;
    const int grid = (int)((nwt + HNW - 1) / HNW);
    const size_t lds_ring = (size_t)HNW * 12 * 256 * sizeof(float);
    static LdsOptIn opted;
    if (opt_in_lds(reinterpret_cast<const void *>(&maf_solve_t_kernel<NB, FAST>), lds_ring + sizeof(float) * t_seq(NL), opted) != NF_OK)
        return NF_ENOTSUP;
    hipLaunchKernelGGL((maf_solve_t_kernel<NB, FAST>), dim3(grid), dim3(64 * HNW), lds_ring, st, (const float *)x, (const float *)prm,
                       (const float *)gx, (const float *)gld, (const unsigned int *)bits, (float *)v, (const float *)blob,
                       (const int *)table, S, Xs, Ps, B, t_beg, t_end);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// table_host: the host copy of `table` (the launcher splits the tiles into runs of regular-8 tiles and others) or NULL (one generic launch)
template <int NB>
static int maf_t_launch(const void *x, const void *prm, const void *gx, const void *gld, const void *bits, void *v, const void *blob,
                        const int32_t *table, const int32_t *table_host, void *scratch, int64_t B, int D, int hidden_padded,
                        hipStream_t st) {
    using namespace nf;
    constexpr int NL = 1 + 2 * NB;
    const int64_t nwt = (B + 31) / 32;
    const int64_t Dq = (2 * (int64_t)D + 31) / 32 * 32;
    float *S = (float *)scratch;
    float *Xs = S + nwt * 32 * (int64_t)NL * hidden_padded;
    if (hipMemsetAsync(Xs, 0, (size_t)nwt * 32 * Dq * sizeof(float), st) != hipSuccess) return NF_EIO;
    float *Ps = Xs + nwt * 32 * Dq;
    if (!table_host) return maf_t_run<NB, false>(x, prm, gx, gld, bits, v, blob, table, S, Xs, Ps, nwt, B, 0, 1 << 30, st);
    const int T = table_host[4];
    for (int t0 = 0; t0 < T;) {
        const bool fast = table_host[H_HDR + H_ENT * t0 + 21] != 0;
        int t1 = t0 + 1;
        while (t1 < T && (table_host[H_HDR + H_ENT * t1 + 21] != 0) == fast) ++t1;
        const int rc = fast ? maf_t_run<NB, true>(x, prm, gx, gld, bits, v, blob, table, S, Xs, Ps, nwt, B, t0, t1, st)
                            : maf_t_run<NB, false>(x, prm, gx, gld, bits, v, blob, table, S, Xs, Ps, nwt, B, t0, t1, st);
        if (rc != NF_OK) return rc;
        t0 = t1;
    }
    return NF_OK;
}

// The implicit backward's linear solve in ONE pass (module comment above maf_solve_t_kernel): v (B, D) with
// v s + J^T g_p(v, g_ld) = g_x at the solution x of the inverse; prm (B, 2 D) = MADE(x); bits from nf_maf_inverse_h_bits of the pass
// that produced x; blob / table: flows/maf_pack.pack_made_transposed (format 2); gld may be NULL (= 0).
extern "C" int nf_maf_solve_t(const void *x, const void *prm, const void *gx, const void *gld, const void *bits, void *v,
                              const void *blob, const int32_t *table, void *scratch, int64_t B, int D, int hidden_padded,
                              int num_blocks, nf_stream_t stream) {
    if (B < 0 || D < 2 || hidden_padded < 32 || hidden_padded % 32) return NF_EINVAL;
    if (num_blocks < 1 || num_blocks > 3) return NF_ENOTSUP;
    if (B == 0) return NF_OK;
    if (!x || !prm || !gx || !bits || !v || !blob || !table || !scratch) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    if (num_blocks == 1) return maf_t_launch<1>(x, prm, gx, gld, bits, v, blob, table, nullptr, scratch, B, D, hidden_padded, st);
    if (num_blocks == 2) return maf_t_launch<2>(x, prm, gx, gld, bits, v, blob, table, nullptr, scratch, B, D, hidden_padded, st);
    return maf_t_launch<3>(x, prm, gx, gld, bits, v, blob, table, nullptr, scratch, B, D, hidden_padded, st);
}

// nf_maf_solve_t on a transposed pack built over the FORMAT-1 forward positions (maf_pack.pack_made_transposed(tri=True)): the tiles
// its table marks regular-8 (entry [21]) run the statically unrolled sequential part (tf_step), launched per maximal run of tiles of
// one kind from the HOST copy of the table; -EINVAL when table_host does not describe this call.  Same results, bit for bit, as
// nf_maf_solve_t on the same pack (finite data).
extern "C" int nf_maf_solve_t_tri(const void *x, const void *prm, const void *gx, const void *gld, const void *bits, void *v,
                                  const void *blob, const int32_t *table, const int32_t *table_host, void *scratch, int64_t B, int D,
                                  int hidden_padded, int num_blocks, nf_stream_t stream) {
    if (B < 0 || D < 2 || hidden_padded < 32 || hidden_padded % 32) return NF_EINVAL;
    if (num_blocks < 1 || num_blocks > 3) return NF_ENOTSUP;
    if (!table_host) return NF_EFAULT;
    if (table_host[0] != D || table_host[3] != hidden_padded || table_host[6] != num_blocks || table_host[4] < 1 ||
        table_host[4] * nf::HT != hidden_padded)
        return NF_EINVAL;
    if (B == 0) return NF_OK;
    if (!x || !prm || !gx || !bits || !v || !blob || !table || !scratch) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    if (num_blocks == 1) return maf_t_launch<1>(x, prm, gx, gld, bits, v, blob, table, table_host, scratch, B, D, hidden_padded, st);
    if (num_blocks == 2) return maf_t_launch<2>(x, prm, gx, gld, bits, v, blob, table, table_host, scratch, B, D, hidden_padded, st);
    return maf_t_launch<3>(x, prm, gx, gld, bits, v, blob, table, table_host, scratch, B, D, hidden_padded, st);
}
