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
constexpr int HRB = 12;   // k-blocks of the per-wave activation ring (look-ahead HRB - 1)

#define HMFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// c0[r] (+)= sum_k A[u(r, hh)][k] act[k][n] over K (a multiple of 32) features / units; PAIR: the NEXT tile's products over the
// same operands go to `stash` raw; INIT: c0 starts from the stash (second tile of a pair).  A, A2: [K/8][2][32][4] (L2),
// Sl: the wave's scratch [K/8][2][32][4], streamed through `ring` HRB - 1 k-blocks ahead (in-order retirement: while requests
// are being issued exactly HRB - 1 DMA instructions are younger than the k-block about to be consumed).
template <bool PAIR, bool INIT>
__device__ __forceinline__ void h_block(const float *__restrict__ A, const float *__restrict__ A2, const float *Sl, int K, int lane,
                                        float *stash, float *ring, f32x16 &c0) {
    typedef __attribute__((address_space(3))) void *lds_ptr;
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
    const int nkb = K >> 3;
    if (nkb > 0) {
        // ALL three streams of the product -- the activation k-blocks (HBM / L2) and the A [, A2] weight k-blocks (L2) -- arrive by
        // LDS-DMA in the wave's own ring: 4 slots each (12 KB), slot = k-block mod 4, a slot is requested again (k-block + 4) as
        // soon as its three reads have returned.  The requests are INLINE ASM and there is no ordinary load in the loop, so the
        // compiler inserts no vector-memory wait of its own and the hand-counted ones are exact: vector-memory operations retire
        // in order, block kb's requests went out 4 steps ago, younger than them are the 3 x (1 + LPS) requests of the blocks
        // kb + 1 .. kb + 3 -- four steps' worth of requests are in flight under every step's MFMAs.
        // History (round 3): (1) the builtin + A operands as ordinary register loads: the compiler waited vmcnt(0) at every first
        // use of a loaded register while an LDS-DMA might be pending -- the ring drained every fourth k-block (2.9 TB/s, 16.1 ms);
        // (2) activation requests as asm, A still in registers (14.5 ms): the compiler's counted waits for A (it counts only its
        // own loads) stand in front of the interleaved requests it does not know of, so ~2.3 steps were in flight instead of 4.
        constexpr int LPS = PAIR ? 2 : 1, OPS = 1 + LPS;
        static_assert(HRB == 12, "4 slots per stream");
        const uint32_t ring_lds = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)ring);
        auto dma1 = [&](uint32_t dst, const float *src) {
            // (m0 is a reserved register: the compiler does not honour it as a clobber, so it is saved and restored here)
            uint32_t m0_;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(m0_) : "s"(dst), "v"(src) : "memory");
        };
        auto req = [&](int kb, int slot) {
            const size_t off = (size_t)kb * 256 + lane * 4;
            dma1(ring_lds + (uint32_t)slot * 1024u, Sl + off);
            dma1(ring_lds + 4096u + (uint32_t)slot * 1024u, A + off);
            if constexpr (PAIR) dma1(ring_lds + 8192u + (uint32_t)slot * 1024u, A2 + off);
        };
        auto step = [&](int kb, int slot) {
            const int rem = nkb - 1 - kb;       // blocks requested after this one
            if (rem >= 3) NF_WAIT_VMCNT(3 * OPS);
            else if (rem == 2) NF_WAIT_VMCNT(2 * OPS);
            else if (rem == 1) NF_WAIT_VMCNT(OPS);
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const f32x4 b = *reinterpret_cast<const f32x4 *>(ring + slot * 256 + lane * 4);
            const f32x4 a = *reinterpret_cast<const f32x4 *>(ring + 1024 + slot * 256 + lane * 4);
            f32x4 a2 = a;
            if constexpr (PAIR) a2 = *reinterpret_cast<const f32x4 *>(ring + 2048 + slot * 256 + lane * 4);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the slot's reads have returned: it may be requested again
            if (kb + 4 < nkb) req(kb + 4, slot);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                c0 = HMFMA(a[i], b[i], c0);
                if constexpr (PAIR) c2 = HMFMA(a2[i], b[i], c2);
            }
        };
        req(0, 0); req(1, 1); req(2, 2); req(3, 3);      // nkb is a multiple of 4
        for (int kb = 0; kb < nkb; kb += 4) {
            step(kb, 0);
            step(kb + 1, 1);
            step(kb + 2, 2);
            step(kb + 3, 3);
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
__device__ __forceinline__ void h_block_mode(int mode, const float *__restrict__ A, const float *__restrict__ A2, const float *Sl,
                                             int K, int lane, float *stash, float *ring, f32x16 &out) {
    if (mode == 1) h_block<true, false>(A, A2, Sl, K, lane, stash, ring, out);
    else if (mode == 2) h_block<false, true>(A, nullptr, Sl, K, lane, stash, ring, out);
    else h_block<false, false>(A, nullptr, Sl, K, lane, nullptr, ring, out);
}

__device__ __forceinline__ void h_finish(float us, float sh, float zf, float &xn, float &ld) {
    const float scale = 1.0f / (1.0f + __expf(-(us + 2.0f))) + 1e-3f;
    xn = (zf - sh) / scale;
    ld -= __logf(scale);
}

// NB residual blocks (nets/made.py:140-214): NL = 1 + 2 NB hidden layers whose activations later tiles contract over
// (S_0 = relu(h_0); per block b: S_{2b+1} = relu(t_b), S_{2b+2} = relu(h_{b+1}), the last one raw = the final layer's input).
template <int NB>
__global__ void __launch_bounds__(64 * HNW, (NB <= 2 && HNW <= 4) ? 2 : 1)
maf_inverse_h_kernel(const float *__restrict__ z, float *__restrict__ y, float *__restrict__ logdet, const float *__restrict__ blob,
                     const int *__restrict__ table, float *S, float *Xs, float *Ps, int64_t B, int acc) {
    constexpr int NL = 1 + 2 * NB, H_SEQ = h_seq(NL);
    __shared__ __attribute__((aligned(16))) float seqw[H_SEQ];     // the tile's biases and diagonal blocks, shared by the workgroup's waves
    extern __shared__ __attribute__((aligned(16))) float dyn[];      // the waves' activation rings
    const int lane = threadIdx.x & 63, n = lane & 31, hh = lane >> 5;
    // wave-uniform BY CONSTRUCTION for the compiler too: every pointer and the tile mode below derive from it -- left in a vector
    // register the whole product dispatch ran under exec masks with vector loop counters
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *ringw = dyn + wid * (HRB * 256);
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
    h_finish(blob[0], blob[1], zr[0], xcarry, ld);
    if (active && hh == 0) Xw[n * 4] = xcarry;
    if (valid && hh == 0) y[sample * D] = xcarry;
    auto xsum = [](float v) { return v + __shfl_xor(v, 32, 64); };

    for (int t = 0; t < T; ++t) {
        const int *te = table + H_HDR + H_ENT * t;
        const int dlo = te[0], ns = te[1], K0 = te[2];
        const int Kh = HT * t;
        const float *rec = blob + te[3];
        const float *A0 = rec;
        const float *Ah = A0 + K0 * HT;             // A1..A_{NL-1}, AF: Kh * 32 floats each
        // stage the sequential part's weights, one copy per workgroup; the five 32 x 32 diagonal blocks with their columns
        // in (half, register) order: source quad (u, v / 4 = 2 q + h) -> destination quad (u, 4 h + q)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // LDS-only barrier: a fence would wait for the activation stores in flight
        {
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
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // LDS-only barrier: a fence would wait for the activation stores in flight
        if (!active) continue;
        const float *bias = seqw;
        const float *biasF = bias + NL * HT;
        const float *W0d = biasF + HT;
        const float *Wd = W0d + HT * HS;
        const float *WFd = Wd + (NL - 1) * HT * HT;

        f32x16 zin;
#pragma unroll
        for (int j = 0; j < HS; ++j) zin[j] = (j < ns) ? zr[dlo + j] : 0.0f;

        __threadfence_block();  // the activation scratch written by the other lanes of this wave is read below
        f32x16 p[NL], pF;
        h_block<false, false>(A0, nullptr, Xw, K0, lane, nullptr, ringw, p[0]);
        {
            // tile pairing: even tile = its own products + the next tile's over the same operands (raw accumulators to the
            // stash); odd tile = the stash + the 32 units of its partner
            const int mode = !Pw ? 0 : ((t & 1) ? 2 : (t + 1 < T ? 1 : 0));
            const float *Ah2 = Ah;
            int Kh2 = 0;
            if (mode == 1) {
                const int *te2 = te + H_ENT;
                Ah2 = blob + te2[3] + (size_t)te2[2] * HT;
                Kh2 = Kh + HT;
            }
            const int koff = mode == 2 ? Kh - HT : 0;         // the partner's units: the last 32 of this tile's K range
            const int Kb = mode == 2 ? HT : Kh;
#pragma unroll
            for (int l = 0; l < NL; ++l)
                h_block_mode(mode, Ah + (size_t)l * Kh * HT + (size_t)koff * HT, Ah2 + (size_t)l * Kh2 * HT,
                             Sw + (size_t)l * Hp * 32 + (size_t)koff * 32, Kb, lane, Pw + (size_t)l * HT * 32, ringw,
                             l + 1 < NL ? p[l + 1 < NL ? l + 1 : 0] : pF);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int u = (r & 3) + 8 * (r >> 2) + 4 * hh;
#pragma unroll
            for (int l = 0; l < NL; ++l) p[l][r] += bias[l * HT + u];
            pF[r] += biasF[u];
        }
        f32x16 xg = {0};   // window features dlo-1 .. dlo+14 (0-based): xg[0] is the carry, xg[s+1] the output of step s
        xg[0] = xcarry;

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
        // ---- publish the tile: the register quads ARE the B-operand entries (k-block 4 t + q, half hh) ----
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const size_t o = ((size_t)((4 * t + q) * 2 + hh) * 32 + n) * 4;
#pragma unroll
            for (int l = 0; l < NL; ++l)
                *reinterpret_cast<f32x4 *>(Sw + (size_t)l * Hp * 32 + o) = f32x4{p[l][4 * q], p[l][4 * q + 1], p[l][4 * q + 2], p[l][4 * q + 3]};
        }
    }
    if (valid && hh == 0) ld_store(logdet + sample, ld, acc);
}

}  // namespace nf

// Scratch of nf_maf_inverse_h: per sample NL hidden_padded activations + the padded feature row + the tile-pair stash.
extern "C" int64_t nf_maf_inverse_h_scratch_floats(int64_t B, int D, int hidden_padded, int num_blocks) {
    if (B < 0 || D < 1 || hidden_padded < 0 || num_blocks < 1 || num_blocks > 3) return NF_EINVAL;
    const int64_t nwt = (B + 31) / 32, Dp = (D + 31) / 32 * 32, NL = 1 + 2 * num_blocks;
    return nwt * 32 * (NL * (int64_t)hidden_padded + Dp + NL * nf::HT);
}

template <int NB>
static int maf_h_launch(const void *z, void *y, void *logdet, const void *blob, const int32_t *table, void *scratch, int64_t B, int D,
                        int hidden_padded, int acc, hipStream_t st) {
    using namespace nf;
    constexpr int NL = 1 + 2 * NB;
    const int64_t nwt = (B + 31) / 32;
    const int64_t Dp = (D + 31) / 32 * 32;
    float *S = (float *)scratch;
    float *Xs = S + nwt * 32 * (int64_t)NL * hidden_padded;
    // the feature scratch is read with zero weights before it is written (K0 is padded to 32): it must hold finite values
    if (hipMemsetAsync(Xs, 0, (size_t)nwt * 32 * Dp * sizeof(float), st) != hipSuccess) return NF_EIO;
    float *Ps = Xs + nwt * 32 * Dp;
    const int grid = (int)((nwt + HNW - 1) / HNW);
    const size_t lds_ring = (size_t)HNW * HRB * 256 * sizeof(float);
    static LdsOptIn opted;
    if (opt_in_lds(reinterpret_cast<const void *>(&maf_inverse_h_kernel<NB>), lds_ring + sizeof(float) * h_seq(NL), opted) != NF_OK)
        return NF_ENOTSUP;
    hipLaunchKernelGGL(maf_inverse_h_kernel<NB>, dim3(grid), dim3(64 * HNW), lds_ring, st, (const float *)z, (float *)y, (float *)logdet,
                       (const float *)blob, (const int *)table, S, Xs, Ps, B, acc);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// nf_maf_inverse on the half-sharing mapping: same blob / table format and semantics as nf_maf_inverse (maf_inverse.hip), for
// MADE conditioners of 1, 2 or 3 residual blocks (table[6]; flows/maf_pack.py).
extern "C" int nf_maf_inverse_h(const void *z, void *y, void *logdet, const void *blob, const int32_t *table, void *scratch,
                                int64_t B, int D, int hidden_padded, int num_blocks, int acc, nf_stream_t stream) {
    if (B < 0 || D < 2 || hidden_padded < 32 || hidden_padded % 32) return NF_EINVAL;
    if (acc < NF_LD_SUB || acc > NF_LD_ADD) return NF_EINVAL;
    if (num_blocks < 1 || num_blocks > 3) return NF_ENOTSUP;
    if (B == 0) return NF_OK;
    if (!z || !y || !logdet || !blob || !table || !scratch) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    if (num_blocks == 1) return maf_h_launch<1>(z, y, logdet, blob, table, scratch, B, D, hidden_padded, acc, st);
    if (num_blocks == 2) return maf_h_launch<2>(z, y, logdet, blob, table, scratch, B, D, hidden_padded, acc, st);
    return maf_h_launch<3>(z, y, logdet, blob, table, scratch, B, D, hidden_padded, acc, st);
}
