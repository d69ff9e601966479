// Incremental inverse of a masked autoregressive affine layer (MAF sampling direction) for gfx950.
//
// Reference semantics: normflows/flows/affine/autoregressive.py:29-38 runs D full MADE passes
// (nets/made.py:217-304, residual blocks :140-214) and keeps the last iterate; :114-128 is the element-wise affine
// inverse x = (z - shift) / (sigmoid(u + 2) + 1e-3), logabsdet = -sum log(scale).  The fixed point it reaches is
// computed here in ONE pass: a hidden unit of degree m depends only on features of degree <= m (made.py:63-81), so
// it is finalised right after feature m is known.
//
// One wave owns 64 samples for the whole layer (samples are independent: no grid-level synchronisation).  Hidden units
// are sorted by degree and cut into tiles of <= 32 units / <= 16 degrees (host side: flows/maf_pack.py).  Per tile:
//   block part  : contributions of all EARLIER tiles = dense [32 x K] x [K x 64] products on v_mfma_f32_32x32x2_f32
//                 (A = packed weights straight from L2, B = the wave's own activation scratch in HBM/L2, stored in
//                 MFMA B-operand order [k/8][half][sample][4] so both operands are 16-byte loads), transposed through
//                 8 KB of LDS so that afterwards lane = sample holds all 32 units of a layer in registers;
//   sequential  : the tile's degrees one after the other; the 32x32 diagonal blocks are read as SCALARS (uniform
//                 addresses), every hidden layer's activations live in VGPRs with static indices; after each degree
//                 the next feature is produced from its two final-layer rows.
// Work per sample = one MADE pass (0.6 MMAC for D=128, H=512) instead of D passes (153 MMAC).
//
// The same schedule inverts the autoregressive rational-quadratic spline layer (AR-NSF sampling direction,
// normflows/flows/neural_spline/autoregressive.py:94-134 over affine/autoregressive.py:29-38): there a feature needs
// 3K-1 | 3K | 3K+1 final-layer outputs, so every step has its own 32-row final block (block part on MFMA, left in
// the wave's LDS tile where lane = sample reads its column), followed by the element-wise inverse spline
// (utils/splines.py:16-219 through common.hpp::rqs_element).
#include "common.hpp"
#include "fused_common.hpp"

namespace nf {

constexpr int MT = 32;    // units per tile
constexpr int MS = 16;    // degrees per tile
constexpr int MW = 4;     // waves per workgroup
constexpr int M_HDR = 8, M_ENT = 24;
constexpr int M_SEQ = 5 * MT + MT + MT * MS + 4 * MT * MT + MT * MT;  // floats of a tile record after the A operands

typedef float f32x32 __attribute__((ext_vector_type(32)));

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// ldsw[u][sample] = sum_k A[u][k] * act[k][sample] for the wave's 64 samples; K is a multiple of 32.
// Tile pairing (PAIR / INIT, affine variant): the activation scratch is what the kernel streams from HBM (every tile re-reads
// ALL earlier tiles' activations: 5 GB per launch at config 5), so an even tile t also accumulates the NEXT tile's products
// over the same operands -- second A stream A2 (tile t+1's weights for the same earlier units), accumulators stored raw
// (C layout, 16-byte stores) in a per-wave stash -- and the odd tile starts from that stash (INIT) and only adds the 32 units
// of tile t.  Activation re-reads: sum 32 t over all tiles -> over the even ones + 32 per odd one (-47 %).
template <bool PAIR, bool INIT>
__device__ __forceinline__ void block_part_lds_x(const float *__restrict__ A, const float *__restrict__ A2, const float *Sl,
                                                 int K, float *ldsw, int lane, float *stash) {
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    const int half = lane >> 5, l31 = lane & 31;
    if constexpr (INIT) {
        const f32x4 *ps = reinterpret_cast<const f32x4 *>(stash) + lane;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f32x4 v0 = ps[r * 64], v1 = ps[(4 + r) * 64];
#pragma unroll
            for (int i = 0; i < 4; ++i) { c0[4 * r + i] = v0[i]; c1[4 * r + i] = v1[i]; }
        }
    }
    const f32x4 *pa = reinterpret_cast<const f32x4 *>(A) + half * 32 + l31;
    const f32x4 *pa2 = reinterpret_cast<const f32x4 *>(PAIR ? A2 : A) + half * 32 + l31;
    const f32x4 *pb = reinterpret_cast<const f32x4 *>(Sl) + half * 64 + l31;
#ifdef NF_MAF_ABL_NOBLOCK
    const int nkb = 0;
#else
    const int nkb = K >> 3;
#endif
    if (nkb > 0) {
        // K is a multiple of 32 (host packer), i.e. nkb of 4: eight named operand stages, each refilled right after it is
        // consumed (prefetch distance = 8 k-blocks = 4096 MFMA cycles, no register rotation moves).
        // With one wave per SIMD nothing else hides the HBM/L2 latency of the activation scratch.
        struct Stage { f32x4 a, a2, b0, b1; };   // a2 is dead (never loaded) unless PAIR
        auto ld = [&](int kb, Stage &st) {
            const int k = kb < nkb ? kb : nkb - 1;
            st.a = pa[k * 64]; st.b0 = pb[k * 128]; st.b1 = pb[k * 128 + 32];
            if constexpr (PAIR) st.a2 = pa2[k * 64];
        };
        auto mm = [&](const Stage &st) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                c0 = MFMA32(st.a[i], st.b0[i], c0);
                c1 = MFMA32(st.a[i], st.b1[i], c1);
                if constexpr (PAIR) {
                    c2 = MFMA32(st.a2[i], st.b0[i], c2);
                    c3 = MFMA32(st.a2[i], st.b1[i], c3);
                }
            }
        };
        if constexpr (PAIR) {
            // four stages of 16 MFMAs each: the same 4096 MFMA cycles of look-ahead as eight stages of 8, in 64 registers
            Stage s0, s1, s2, s3;
            ld(0, s0); ld(1, s1); ld(2, s2); ld(3, s3);
            for (int kb = 0; kb < nkb; kb += 4) {        // nkb is a multiple of 4
                mm(s0); ld(kb + 4, s0);
                mm(s1); ld(kb + 5, s1);
                mm(s2); ld(kb + 6, s2);
                mm(s3); ld(kb + 7, s3);
            }
        } else {
            Stage s0, s1, s2, s3, s4, s5, s6, s7;
            ld(0, s0); ld(1, s1); ld(2, s2); ld(3, s3); ld(4, s4); ld(5, s5); ld(6, s6); ld(7, s7);
            int kb = 0;
            for (; kb + 8 <= nkb; kb += 8) {
                mm(s0); ld(kb + 8, s0);
                mm(s1); ld(kb + 9, s1);
                mm(s2); ld(kb + 10, s2);
                mm(s3); ld(kb + 11, s3);
                mm(s4); ld(kb + 12, s4);
                mm(s5); ld(kb + 13, s5);
                mm(s6); ld(kb + 14, s6);
                mm(s7); ld(kb + 15, s7);
            }
            if (kb < nkb) { mm(s0); mm(s1); mm(s2); mm(s3); }
        }
    }
    if constexpr (PAIR) {
        f32x4 *ps = reinterpret_cast<f32x4 *>(stash) + lane;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ps[r * 64] = f32x4{c2[4 * r], c2[4 * r + 1], c2[4 * r + 2], c2[4 * r + 3]};
            ps[(4 + r) * 64] = f32x4{c3[4 * r], c3[4 * r + 1], c3[4 * r + 2], c3[4 * r + 3]};
        }
    }
    // C layout: row = (reg & 3) + 8 (reg >> 2) + 4 half, col = lane & 31  ->  LDS [unit][64 samples]  ->  lane = sample
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        ldsw[row * 64 + l31] = c0[r];
        ldsw[row * 64 + 32 + l31] = c1[r];
    }
    __builtin_amdgcn_wave_barrier();
}

// The same product with the activation operand (B) streamed through a per-wave LDS ring by LDS-DMA, MRB k-blocks ahead
// (affine variant).  With the operands loaded into registers eight k-blocks ahead the look-ahead is 4096 MFMA cycles (1.9 us):
// ablation with an L2-resident scratch ran the block part 0.38 ms faster per launch, i.e. HBM latency was exposed.  The ring
// costs no registers, so the look-ahead is MRB - 1 = 11 k-blocks; A (weights, L2-resident) stays on the register stages.
// Synchronisation: loads retire in order, exactly 2 (MRB - 1) DMA instructions are younger than the k-block about to be
// consumed while requests are being issued (the register loads of A in between only make the wait stricter), so
// `s_waitcnt vmcnt(2 (MRB - 1))` means it has landed; the tail drains.  A slot is refilled one step after it was consumed (its
// ds_reads completed before that step's MFMAs).
constexpr int MRB = 12;
template <bool PAIR, bool INIT>
__device__ __forceinline__ void block_part_ring(const float *__restrict__ A, const float *__restrict__ A2, const float *Sl,
                                                int K, float *ldsw, int lane, float *stash, float *ring) {
    typedef __attribute__((address_space(3))) void *lds_ptr;
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    const int half = lane >> 5, l31 = lane & 31;
    if constexpr (INIT) {
        const f32x4 *ps = reinterpret_cast<const f32x4 *>(stash) + lane;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f32x4 v0 = ps[r * 64], v1 = ps[(4 + r) * 64];
#pragma unroll
            for (int i = 0; i < 4; ++i) { c0[4 * r + i] = v0[i]; c1[4 * r + i] = v1[i]; }
        }
    }
    const f32x4 *pa = reinterpret_cast<const f32x4 *>(A) + half * 32 + l31;
    const f32x4 *pa2 = reinterpret_cast<const f32x4 *>(PAIR ? A2 : A) + half * 32 + l31;
    const int nkb = K >> 3;
    if (nkb > 0) {
        // inline asm, i.e. invisible to the compiler's wait-count pass (see maf_inverse_h.hip: with the builtin the compiler waits
        // vmcnt(0) at every first use of the register-loaded A operands while an LDS-DMA may be pending, draining the ring)
        const uint32_t ring_lds = (uint32_t)(uintptr_t)ring;
        auto dma = [&](int kb) {
            const float *src = Sl + (size_t)kb * 512 + lane * 4;
            const uint32_t dst = __builtin_amdgcn_readfirstlane(ring_lds + (uint32_t)(kb % MRB) * 2048u);
            // (m0 is a reserved register: the compiler does not honour it as a clobber, so it is saved and restored here)
            uint32_t m0_;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\t"
                         "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(m0_) : "s"(dst), "v"(src), "s"(dst + 1024u), "v"(src + 256) : "memory");
        };
        struct Stage { f32x4 a, a2; };
        auto ld = [&](int kb, Stage &st) {
            const int k = kb < nkb ? kb : nkb - 1;
            st.a = pa[k * 64];
            if constexpr (PAIR) st.a2 = pa2[k * 64];
        };
        // one step: request k-block kb + MRB - 1 (if the product has one) and wait for k-block kb.  While requests are still
        // being issued exactly 2 (MRB - 1) DMA instructions are younger than k-block kb's; in the tail nothing new is issued and
        // the wait is a full drain (everything outstanding is needed within the next MRB - 1 steps anyway).  No DMA is ever issued
        // past the product's end: with left-overs in flight two back-to-back products could exceed the 6-bit vmcnt counter.
        auto mm = [&](int kb, const Stage &st) {
            // in-order retirement; per k-block: dma(kb + MRB - 1) (2 requests) | wait | MFMAs | ld(kb + 4).  Younger than ld(kb) here:
            // 3 ld (x2 for a tile pair) + 4 x 2 requests; past the last request the three youngest ld may stay in flight
            constexpr int LPS = PAIR ? 2 : 1;
            if (kb + MRB - 1 < nkb) {
                dma(kb + MRB - 1);
                NF_WAIT_VMCNT(3 * LPS + 8);
            } else {
                NF_WAIT_VMCNT(3 * LPS);
            }
            const float *slot = ring + (kb % MRB) * 512 + (half * 64 + l31) * 4;
            const f32x4 b0 = *reinterpret_cast<const f32x4 *>(slot), b1 = *reinterpret_cast<const f32x4 *>(slot + 128);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                c0 = MFMA32(st.a[i], b0[i], c0);
                c1 = MFMA32(st.a[i], b1[i], c1);
                if constexpr (PAIR) {
                    c2 = MFMA32(st.a2[i], b0[i], c2);
                    c3 = MFMA32(st.a2[i], b1[i], c3);
                }
            }
        };
        Stage s0, s1, s2, s3;
        for (int j = 0; j < MRB - 4 && j < nkb; ++j) dma(j);     // the steady state's order: ld(i) / dma(MRB - 4 + i) alternating
        ld(0, s0); if (MRB - 4 < nkb) dma(MRB - 4);
        ld(1, s1); if (MRB - 3 < nkb) dma(MRB - 3);
        ld(2, s2); if (MRB - 2 < nkb) dma(MRB - 2);
        ld(3, s3);
        for (int kb = 0; kb < nkb; kb += 4) {        // nkb is a multiple of 4
            mm(kb, s0); ld(kb + 4, s0);
            mm(kb + 1, s1); ld(kb + 5, s1);
            mm(kb + 2, s2); ld(kb + 6, s2);
            mm(kb + 3, s3); ld(kb + 7, s3);
        }
    }
    if constexpr (PAIR) {
        f32x4 *ps = reinterpret_cast<f32x4 *>(stash) + lane;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ps[r * 64] = f32x4{c2[4 * r], c2[4 * r + 1], c2[4 * r + 2], c2[4 * r + 3]};
            ps[(4 + r) * 64] = f32x4{c3[4 * r], c3[4 * r + 1], c3[4 * r + 2], c3[4 * r + 3]};
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        ldsw[row * 64 + l31] = c0[r];
        ldsw[row * 64 + 32 + l31] = c1[r];
    }
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void block_part_lds(const float *__restrict__ A, const float *Sl, int K, float *ldsw, int lane) {
    block_part_lds_x<false, false>(A, nullptr, Sl, K, ldsw, lane, nullptr);
}

// Same product with lane = sample holding all 32 units in registers afterwards.
__device__ __forceinline__ void block_part(const float *__restrict__ A, const float *Sl, int K, float *ldsw, int lane,
                                           f32x32 &out) {
    block_part_lds(A, Sl, K, ldsw, lane);
#pragma unroll
    for (int u = 0; u < MT; ++u) out[u] = ldsw[u * 64 + lane];
    __builtin_amdgcn_wave_barrier();
}

// mode 1: first tile of a pair (also accumulates the next tile's products into the stash); mode 2: second tile (starts from
// the stash, K = the 32 units of the first tile); mode 0: unpaired.
__device__ __forceinline__ void block_part_mode(int mode, const float *__restrict__ A, const float *__restrict__ A2,
                                                const float *Sl, int K, float *ldsw, int lane, float *stash, float *ring,
                                                f32x32 &out) {
#ifdef NF_MAF_NO_RING
    if (mode == 1) block_part_lds_x<true, false>(A, A2, Sl, K, ldsw, lane, stash);
    else if (mode == 2) block_part_lds_x<false, true>(A, nullptr, Sl, K, ldsw, lane, stash);
    else block_part_lds_x<false, false>(A, nullptr, Sl, K, ldsw, lane, nullptr);
#else
    if (mode == 1) block_part_ring<true, false>(A, A2, Sl, K, ldsw, lane, stash, ring);
    else if (mode == 2) block_part_ring<false, true>(A, nullptr, Sl, K, ldsw, lane, stash, ring);
    else block_part_ring<false, false>(A, nullptr, Sl, K, ldsw, lane, nullptr, ring);
#endif
#pragma unroll
    for (int u = 0; u < MT; ++u) out[u] = ldsw[u * 64 + lane];
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void maf_finish(float us, float sh, float zf, float &xn, float &ld) {
    const float scale = 1.0f / (1.0f + __expf(-(us + 2.0f))) + 1e-3f;
    xn = (zf - sh) / scale;
    ld -= __logf(scale);
}

// Floats of the spline variant's dynamic LDS: the waves' transpose tiles, then the staged sequential-part weights.
static inline size_t arnsf_lds_floats(int R) {
    return (size_t)MW * MT * 64 + 5 * MT + MT * MS + 4 * MT * MT + MS * MT + (size_t)MS * R * MT;
}

// SPL = false: affine element (MAF); true: rational-quadratic spline element with `sp` and R final rows per feature.
template <bool SPL>
__global__ void __launch_bounds__(64 * MW)
maf_inverse_kernel(const float *__restrict__ z, float *__restrict__ y, float *__restrict__ logdet,
                   const float *__restrict__ blob, const int *__restrict__ table, float *S, float *Xs, float *Ps, int64_t B,
                   int acc, RqsParams<float> sp, int R) {
    __shared__ float lds[SPL ? 1 : MW][SPL ? 4 : MT * 64];
    __shared__ __attribute__((aligned(16))) float seqs[SPL ? 4 : M_SEQ];  // the tile's biases and diagonal blocks, shared by the 4 waves
    extern __shared__ __attribute__((aligned(16))) float dyn[];            // spline variant: everything lives here
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float *ldsw = SPL ? dyn + wid * (MT * 64) : lds[SPL ? 0 : wid];
    float *ringw = SPL ? nullptr : dyn + wid * (MRB * 512);      // affine variant: the wave's activation ring (dynamic LDS)
    (void)ringw;
    float *seqw = SPL ? dyn + MW * MT * 64 : seqs;
    const int64_t wt = (int64_t)blockIdx.x * MW + wid;
    const bool active = wt * 64 < B;  // idle waves of the last workgroup still take part in the staging barriers
    const int D = table[0], Dp = table[1], Hp = table[3], T = table[4];
    const int64_t sample = wt * 64 + lane;
    const bool valid = sample < B;
#ifdef NF_MAF_ABL_HOT   // every wave on the same scratch slab: L2-resident state, timing only
    const int64_t wts = 0;
#else
    const int64_t wts = active ? wt : 0;
#endif
    const float *zr = z + (valid ? sample : B - 1) * D;
    float *Sw = S + wts * ((int64_t)5 * Hp * 64);   // [layer][Hp/8][2][64][4]
    float *Xw = Xs + wts * ((int64_t)Dp * 64);      // [Dp/8][2][64][4]
    float *Pw = Ps ? Ps + wts * ((int64_t)5 * MT * 64) : nullptr;   // pair stash: [product][8][64][4] raw accumulators
    float ld = 0.0f, xcarry;
    if constexpr (SPL) {   // feature 0 depends on no hidden unit: its parameters are the final layer's bias
        const int K = sp.K;
        float lad;
        rqs_element<float>(sp, zr[0], [&](int k) { return blob[k]; }, [&](int k) { return blob[K + k]; },
                           [&](int j) { return blob[2 * K + j]; }, true, xcarry, lad);
        ld += lad;
    } else {
        maf_finish(blob[0], blob[1], zr[0], xcarry, ld);
    }
    if (active) Xw[lane * 4] = xcarry;
    if (valid) y[sample * D] = xcarry;

    for (int t = 0; t < T; ++t) {
        const int *te = table + M_HDR + M_ENT * t;
        const int dlo = te[0], ns = te[1], K0 = te[2];
        const int Kh = MT * t;
        const float *rec = blob + te[3];
        const float *A0 = rec;
        const float *Ah = A0 + K0 * MT;             // A1..A4, then AF (affine) or AF_0..AF_{ns-1} (spline): Kh*32 floats each
        // stage the sequential part's weights (affine: 23 KB), one copy per workgroup, read back as LDS broadcasts
        const int nblk = SPL ? 4 + ns : 5;
        const int nseq = SPL ? 5 * MT + MT * MS + 4 * MT * MT + ns * MT + ns * R * MT : M_SEQ;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // LDS-only barrier: a fence would wait for the activation stores in flight
        {
            const f32x4 *src = reinterpret_cast<const f32x4 *>(Ah + (size_t)nblk * Kh * MT);
            for (int i = threadIdx.x; i < nseq / 4; i += 64 * MW) reinterpret_cast<f32x4 *>(seqw)[i] = src[i];
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // LDS-only barrier: a fence would wait for the activation stores in flight
        if (!active) continue;
        const float *bias = seqw;
        const float *biasF = bias + 5 * MT;                            // affine layout only
        const float *W0d = SPL ? bias + 5 * MT : biasF + MT;
        const float *Wd = W0d + MT * MS;
        const float *WFd = SPL ? Wd + 4 * MT * MT + ns * MT : Wd + 4 * MT * MT;
        const float *biasFs = Wd + 4 * MT * MT;                       // spline layout: [ns][32]

        f32x16 zin;
#pragma unroll
        for (int j = 0; j < MS; ++j) zin[j] = (j < ns) ? zr[dlo + j] : 0.0f;

        __threadfence_block();  // the activation scratch written by the other lanes of this wave is read below
        f32x32 p0, p1, p2, p3, p4, pF;
        block_part(A0, Xw, K0, ldsw, lane, p0);
        if constexpr (SPL) {
            block_part(Ah + (size_t)0 * Kh * MT, Sw + (size_t)0 * Hp * 64, Kh, ldsw, lane, p1);
            block_part(Ah + (size_t)1 * Kh * MT, Sw + (size_t)1 * Hp * 64, Kh, ldsw, lane, p2);
            block_part(Ah + (size_t)2 * Kh * MT, Sw + (size_t)2 * Hp * 64, Kh, ldsw, lane, p3);
            block_part(Ah + (size_t)3 * Kh * MT, Sw + (size_t)3 * Hp * 64, Kh, ldsw, lane, p4);
        } else {
            // tile pairing (see block_part_lds_x): even tile = its own products + the next tile's over the same operands;
            // odd tile = the stash + the 32 units of its partner
            const int mode = !Pw ? 0 : ((t & 1) ? 2 : (t + 1 < T ? 1 : 0));
            const float *Ah2 = Ah;
            int Kh2 = 0;
            if (mode == 1) {
                const int *te2 = te + M_ENT;
                Ah2 = blob + te2[3] + (size_t)te2[2] * MT;
                Kh2 = Kh + MT;
            }
            const int koff = mode == 2 ? Kh - MT : 0;         // the partner's units: the last 32 of this tile's K range
            const int Kb = mode == 2 ? MT : Kh;
#define NF_MAF_BP(l, OUT)                                                                                                   \
            block_part_mode(mode, Ah + (size_t)(l) * Kh * MT + (size_t)koff * MT, Ah2 + (size_t)(l) * Kh2 * MT,                \
                            Sw + (size_t)(l) * Hp * 64 + (size_t)koff * 64, Kb, ldsw, lane, Pw + (size_t)(l) * MT * 64,    \
                            ringw, OUT)
            NF_MAF_BP(0, p1);
            NF_MAF_BP(1, p2);
            NF_MAF_BP(2, p3);
            NF_MAF_BP(3, p4);
            NF_MAF_BP(4, pF);
#undef NF_MAF_BP
        }
#pragma unroll
        for (int u = 0; u < MT; ++u) {
            p0[u] += bias[u];
            p1[u] += bias[MT + u];
            p2[u] += bias[2 * MT + u];
            p3[u] += bias[3 * MT + u];
            p4[u] += bias[4 * MT + u];
            if constexpr (!SPL) pF[u] += biasF[u];
        }
        f32x16 xg = {0};   // window features dlo-1 .. dlo+14 (0-based): xg[0] is the carry, xg[s+1] the output of step s
        xg[0] = xcarry;

        // The sequential part indexes the register vectors with the (wave-uniform) unit number: one compact copy of each
        // layer's code instead of 32 predicated ones, and a scalar loop over the units of the current degree.
#ifdef NF_MAF_ABL_NOSEQ
        for (int s = 0; s < (B == 12345 ? ns : 0); ++s) {
#else
        for (int s = 0; s < ns; ++s) {
#endif
            const unsigned m = (unsigned)te[4 + s];
            // initial layer: h0 = pre + W0[window] . x ; the residual h0 is folded into the pre-activation of block 1's
            // second linear (p2), p0 keeps relu(h0) = input of block 1's first linear
            for (unsigned mm = m; mm; mm &= mm - 1) {
                const int u = __builtin_amdgcn_readfirstlane(__builtin_ctz(mm));
                const f32x4 *w_ = reinterpret_cast<const f32x4 *>(W0d) + u * (MS / 4);
                float a = p0[u];
#pragma unroll
                for (int f = 0; f < MS; f += 4) {
                    const f32x4 w = w_[f / 4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) a = fmaf(w[i], xg[f + i], a);
                }
                p2[u] += a;
                p0[u] = fmaxf(a, 0.0f);
            }
#define MAF_DOT(WBASE, SRC)                                                                    \
            const f32x4 *w_ = reinterpret_cast<const f32x4 *>(WBASE) + u * (MT / 4);           \
            float a0 = 0.0f, a1 = 0.0f;                                                        \
            _Pragma("unroll") for (int v = 0; v < MT; v += 8) {                                \
                const f32x4 wa = w_[v / 4], wb = w_[v / 4 + 1];                                \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                \
                    a0 = fmaf(wa[i], SRC[v + i], a0);                                          \
                    a1 = fmaf(wb[i], SRC[v + 4 + i], a1);                                      \
                }                                                                              \
            }
            for (unsigned mm = m; mm; mm &= mm - 1) {
                const int u = __builtin_amdgcn_readfirstlane(__builtin_ctz(mm));
                MAF_DOT(Wd, p0)
                p1[u] = fmaxf(p1[u] + (a0 + a1), 0.0f);
            }
            for (unsigned mm = m; mm; mm &= mm - 1) {
                const int u = __builtin_amdgcn_readfirstlane(__builtin_ctz(mm));
                MAF_DOT(Wd + MT * MT, p1)
                const float h1 = p2[u] + (a0 + a1);
                p4[u] += h1;                 // residual stream after block 1 feeds block 2's output
                p2[u] = fmaxf(h1, 0.0f);
            }
            for (unsigned mm = m; mm; mm &= mm - 1) {
                const int u = __builtin_amdgcn_readfirstlane(__builtin_ctz(mm));
                MAF_DOT(Wd + 2 * MT * MT, p2)
                p3[u] = fmaxf(p3[u] + (a0 + a1), 0.0f);
            }
            for (unsigned mm = m; mm; mm &= mm - 1) {
                const int u = __builtin_amdgcn_readfirstlane(__builtin_ctz(mm));
                MAF_DOT(Wd + 3 * MT * MT, p3)
                p4[u] = p4[u] + (a0 + a1);   // = h2, the final layer's input (made.py:304: no activation before it)
            }
            {
                float xn;
                if constexpr (SPL) {
                    // this feature's R final rows: earlier tiles on MFMA (left in the LDS tile, lane = sample owns a
                    // column), the tile's own units from registers, then the inverse spline on the column
                    block_part_lds(Ah + (size_t)(4 + s) * Kh * MT, Sw + (size_t)4 * Hp * 64, Kh, ldsw, lane);
                    const float *wfs = WFd + (size_t)s * R * MT;
                    const float *bfs = biasFs + s * MT;
                    for (int u = 0; u < R; ++u) {
                        MAF_DOT(wfs, p4)
                        ldsw[u * 64 + lane] += (a0 + a1) + bfs[u];
                    }
                    const int K = sp.K;
                    float lad;
                    rqs_element<float>(sp, zin[s], [&](int k) { return ldsw[k * 64 + lane]; },
                                       [&](int k) { return ldsw[(K + k) * 64 + lane]; },
                                       [&](int j) { return ldsw[(2 * K + j) * 64 + lane]; }, true, xn, lad);
                    ld += lad;
                    __builtin_amdgcn_wave_barrier();
                } else {
                    float us, sh;
                    {
                        const int u = 2 * s;
                        MAF_DOT(WFd, p4)
                        us = pF[u] + (a0 + a1);
                    }
                    {
                        const int u = 2 * s + 1;
                        MAF_DOT(WFd, p4)
                        sh = pF[u] + (a0 + a1);
                    }
                    maf_finish(us, sh, zin[s], xn, ld);
                }
                // statically indexed (16 selects): see maf_inverse_h.hip -- the dynamically indexed register write of this line went
                // out of bounds in one build of the h mapping
#pragma unroll
                for (int j_ = 1; j_ < MS; ++j_) xg[j_] = (j_ == s + 1) ? xn : xg[j_];
                xcarry = xn;
                const int f = dlo + s;
                Xw[((size_t)((f >> 3) * 2 + ((f >> 2) & 1)) * 64 + lane) * 4 + (f & 3)] = xn;
                if (valid) y[sample * D + f] = xn;
            }
        }
        // ---- publish the tile: activations in B-operand order ----
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const size_t o = ((size_t)((4 * t + q) * 2 + hh) * 64 + lane) * 4;
                const int u0 = 8 * q + 4 * hh;
                *reinterpret_cast<f32x4 *>(Sw + (size_t)0 * Hp * 64 + o) = f32x4{p0[u0], p0[u0 + 1], p0[u0 + 2], p0[u0 + 3]};
                *reinterpret_cast<f32x4 *>(Sw + (size_t)1 * Hp * 64 + o) = f32x4{p1[u0], p1[u0 + 1], p1[u0 + 2], p1[u0 + 3]};
                *reinterpret_cast<f32x4 *>(Sw + (size_t)2 * Hp * 64 + o) = f32x4{p2[u0], p2[u0 + 1], p2[u0 + 2], p2[u0 + 3]};
                *reinterpret_cast<f32x4 *>(Sw + (size_t)3 * Hp * 64 + o) = f32x4{p3[u0], p3[u0 + 1], p3[u0 + 2], p3[u0 + 3]};
                *reinterpret_cast<f32x4 *>(Sw + (size_t)4 * Hp * 64 + o) = f32x4{p4[u0], p4[u0 + 1], p4[u0 + 2], p4[u0 + 3]};
            }
        }
    }
    if (valid) ld_store(logdet + sample, ld, acc);
}

}  // namespace nf

extern "C" int64_t nf_maf_inverse_scratch_floats(int64_t B, int D, int hidden_padded) {
    if (B < 0 || D < 1 || hidden_padded < 0) return NF_EINVAL;
    const int64_t nwt = (B + 63) / 64;
    const int64_t Dp = (D + 31) / 32 * 32;
    return nwt * 64 * ((int64_t)5 * hidden_padded + Dp + 5 * nf::MT);      // activations | features | pair stash
}

extern "C" int nf_maf_inverse(const void *z, void *y, void *logdet, const void *blob, const int32_t *table,
                              void *scratch, int64_t B, int D, int hidden_padded, int acc, nf_stream_t stream) {
    if (B < 0 || D < 2 || hidden_padded < 32 || hidden_padded % 32) return NF_EINVAL;
    if (acc < NF_LD_SUB || acc > NF_LD_ADD) return NF_EINVAL;
    if (B == 0) return NF_OK;
    if (!z || !y || !logdet || !blob || !table || !scratch) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    const int64_t nwt = (B + 63) / 64;
    const int64_t Dp = (D + 31) / 32 * 32;
    float *S = (float *)scratch;
    float *Xs = S + nwt * 64 * (int64_t)5 * hidden_padded;
    // the feature scratch is read with zero weights before it is written (K0 is padded to 32): it must hold finite values
    if (hipMemsetAsync(Xs, 0, (size_t)nwt * 64 * Dp * sizeof(float), st) != hipSuccess) return NF_EIO;
    const int grid = (int)((nwt + nf::MW - 1) / nf::MW);
#ifdef NF_MAF_NO_PAIR
    float *Ps = nullptr;
#else
    float *Ps = Xs + nwt * 64 * Dp;
#endif
    const size_t lds_ring = (size_t)nf::MW * nf::MRB * 512 * sizeof(float);
    static nf::LdsOptIn opted_aff;
    if (nf::opt_in_lds(reinterpret_cast<const void *>(&nf::maf_inverse_kernel<false>), lds_ring, opted_aff) != NF_OK)
        return NF_ENOTSUP;
    hipLaunchKernelGGL(nf::maf_inverse_kernel<false>, dim3(grid), dim3(64 * nf::MW), lds_ring, st, (const float *)z, (float *)y,
                       (float *)logdet, (const float *)blob, (const int *)table, S, Xs, Ps, B, acc, nf::RqsParams<float>{}, 2);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

extern "C" int nf_arnsf_inverse(const void *z, void *y, void *logdet, const void *blob, const int32_t *table,
                                void *scratch, int64_t B, int D, int hidden_padded, int K, int tails, double tail_bound,
                                double min_bin_width, double min_bin_height, double min_derivative, int acc,
                                nf_stream_t stream) {
    if (K < 1 || tails < NF_TAILS_NONE || tails > NF_TAILS_CIRCULAR) return NF_EINVAL;
    if (min_bin_width * K > 1.0 || min_bin_height * K > 1.0) return NF_EINVAL;   // utils/splines.py:121-124
    const int R = tails == NF_TAILS_LINEAR ? 3 * K - 1 : (tails == NF_TAILS_CIRCULAR ? 3 * K : 3 * K + 1);
    if (R > nf::MT) return NF_ENOTSUP;
    if (B < 0 || D < 2 || hidden_padded < 32 || hidden_padded % 32) return NF_EINVAL;
    if (acc < NF_LD_SUB || acc > NF_LD_ADD) return NF_EINVAL;
    if (B == 0) return NF_OK;
    if (!z || !y || !logdet || !blob || !table || !scratch) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    const int64_t nwt = (B + 63) / 64;
    const int64_t Dp = (D + 31) / 32 * 32;
    float *S = (float *)scratch;
    float *Xs = S + nwt * 64 * (int64_t)5 * hidden_padded;
    if (hipMemsetAsync(Xs, 0, (size_t)nwt * 64 * Dp * sizeof(float), st) != hipSuccess) return NF_EIO;
    auto sp = nf::make_rqs_params<float>(K, tails, tail_bound, 0, 1, 0, 1, min_bin_width, min_bin_height, min_derivative,
                                         1.0);
    const size_t lds = nf::arnsf_lds_floats(R) * sizeof(float);
    static nf::LdsOptIn opted;
    if (nf::opt_in_lds(reinterpret_cast<const void *>(&nf::maf_inverse_kernel<true>), lds, opted) != NF_OK) return NF_ENOTSUP;
    const int grid = (int)((nwt + nf::MW - 1) / nf::MW);
    hipLaunchKernelGGL(nf::maf_inverse_kernel<true>, dim3(grid), dim3(64 * nf::MW), lds, st, (const float *)z, (float *)y,
                       (float *)logdet, (const float *)blob, (const int *)table, S, Xs, (float *)nullptr, B, acc, sp, R);
    NF_CHECK_LAUNCH();
    return NF_OK;
}
