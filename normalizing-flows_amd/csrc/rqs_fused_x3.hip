// rqs_fused_x3.hip -- the fused NSF coupling layer of rqs_fused.hip with every GEMM evaluated on the bf16 matrix
// pipe by ERROR-COMPENSATED SPLITTING ("bf16x3"): each fp32 operand is written as hi + mid + lo (three bf16 values, 24
// significand bits, i.e. exactly the fp32 value up to 2^-24 relative), and a.b is accumulated in fp32 from the six
// products hi.hi + hi.mid + mid.hi + hi.lo + lo.hi + mid.mid (the three dropped terms are below 2^-23 |a||b|, the size
// of one fp32 rounding).  Weights are split once at pack time (round-to-nearest), activations in the kernel (truncation
// splits: one AND + one SUB per level, packed two-per-register with v_perm_b32).
//
// Why: on gfx950 the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at the fp32 VECTOR rate and -- measured with
// tools/ubench/overlap.py -- shares the vector ALU: VALU instructions issued beside it add to, instead of hiding
// behind, the MFMA time.  v_mfma_f32_32x32x16_bf16 is 16x faster per k and runs on the separate matrix pipe
// (tools/ubench/overlap_bf16.py: 8 VALU per MFMA pair are free), so 6 bf16 MFMAs per 16 k replace 8 fp32 MFMAs at
// 1/2.4 of the time and the spline epilogue / operand splitting overlap with them.  Results stay fp32-equivalent:
// the parity tests hold this path to the same golden vectors and tolerances as the exact-fp32 kernel.
//
// Structure (same work decomposition as rqs_fused.hip): 4 waves x 32 samples; Out^T = W Act^T; lane l owns sample
// l & 31 and, after a layer, hidden units 32 m + 8 q + 4 (l>>5) + r in C register 4 q + r of row-block m.  K step t
// (16 k) lets lane-half hh contract over the 8 units of registers 8 (t&1) .. 8 (t&1)+7 of row-block t>>1, so
// activations never leave registers.  Weights stream through a 3-slot LDS ring of 12 KB stages (one K step x 4
// row-blocks x 3 splits) filled by global_load_lds two stages ahead (counted vmcnt, raw s_barrier).
#include "fused_common.hpp"

namespace nf {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32;

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

constexpr int X3_SLOT_BYTES = 12288;          // ring slot = largest stage
#ifdef NF_X3_WIDE_FINAL
constexpr int X3_FINAL_BYTES = 12288;
#else
constexpr int X3_FINAL_BYTES = 9216;          // final-layer stage: 3 row-blocks x 3 splits x 1 KB
#endif
constexpr int X3_SLOT_FLOATS = X3_SLOT_BYTES / 4;
// The kernel consumes the blob's K-step stages in PAIRS ("super-stages": 24 KB, or 18 KB in the final layer): 8 waves
// x 3 LDS-DMA instructions each per super-stage, one barrier per 2 K steps.
constexpr int X3_NW = 8;
constexpr int X3_THREADS = 64 * X3_NW;
constexpr int X3_ROWS = 32 * X3_NW;
constexpr int X3_RING_FLOATS = 2 * X3_SLOT_FLOATS;  // one ring slot = one super-stage

// ---- x3 blob: header | small section (identical to the fp32 blob) | stages ------------------------------------
struct X3Layout {
    int nblk;
    __host__ __device__ FusedLayout f32() const { FusedLayout l; l.nblk = nblk; return l; }
    __host__ __device__ int64_t off_stage_bytes() const { return (int64_t)f32().off_stages() * 4; }
    __host__ __device__ int n_init() const { return 2; }
    __host__ __device__ int n_hidden() const { return 16 * nblk; }  // 8 per linear
    __host__ __device__ int n_final() const { return 64; }          // 8 groups x 8 K steps
    __host__ __device__ int n_base() const { return n_init() + n_hidden() + n_final(); }
    __host__ __device__ int64_t off_init() const { return 0; }
    __host__ __device__ int64_t off_hidden() const { return (int64_t)n_init() * X3_SLOT_BYTES; }
    __host__ __device__ int64_t off_final() const { return off_hidden() + (int64_t)n_hidden() * X3_SLOT_BYTES; }
    __host__ __device__ int64_t off_lu(int dir) const { return off_final() + (int64_t)n_final() * X3_FINAL_BYTES + (int64_t)dir * 2 * X3_SLOT_BYTES; }
    __host__ __device__ int64_t off_pad() const { return off_lu(2); }
    __host__ __device__ int64_t total_bytes() const { return off_stage_bytes() + off_pad() + 2 * X3_SLOT_BYTES; }
    // byte offset (from the first stage) of BASE stage b (init | hidden | final)
    __host__ __device__ int64_t base_off(int b) const {
        if (b < n_init() + n_hidden()) return (int64_t)b * X3_SLOT_BYTES;
        return off_final() + (int64_t)(b - n_init() - n_hidden()) * X3_FINAL_BYTES;
    }
};

// ---- fp32 -> three bf16 ----------------------------------------------------------------------------------------
__host__ __device__ inline unsigned short bf16_rne(float f) {
    union { float f; u32 u; } v;
    v.f = f;
    if ((v.u & 0x7f800000u) == 0x7f800000u) return (unsigned short)(v.u >> 16);  // inf / nan: truncate
    const u32 r = v.u + 0x7fffu + ((v.u >> 16) & 1u);
    return (unsigned short)(r >> 16);
}
__host__ __device__ inline float bf16_to_f32(unsigned short h) {
    union { float f; u32 u; } v;
    v.u = (u32)h << 16;
    return v.f;
}

// Re-slice one fp32 A-operand stage ([groups sg][64 lanes][4 floats]) into bf16x3 K-step images:
// dst[(rb_index * 3 + split) * 64 + lane][i] for K step t takes src[sg = 2 t + (i >> 2)][lane][i & 3].
__global__ void x3_convert_kernel(const float *__restrict__ src, unsigned short *__restrict__ dst, int n_rb,
                                  int groups_per_rb /* 16 (K=128), 4 (K=32), 8 (K=64) */, int64_t dst_step_stride_elems,
                                  int rb_per_step /* row-blocks stored per K-step stage */) {
    // grid-stride over (rb, t, lane, i)
    const int steps = groups_per_rb / 2;
    const int64_t total = (int64_t)n_rb * steps * 64 * 8;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(e & 7), lane = (int)((e >> 3) & 63);
        const int t = (int)((e >> 9) % steps), rb = (int)((e >> 9) / steps);
        const float w = src[((int64_t)rb * groups_per_rb + 2 * t + (i >> 2)) * 256 + lane * 4 + (i & 3)];
        const unsigned short hi = bf16_rne(w);
        const float r1 = w - bf16_to_f32(hi);
        const unsigned short mid = bf16_rne(r1);
        const float r2 = r1 - bf16_to_f32(mid);
        const unsigned short lo = bf16_rne(r2);
        // stage of this K step: group of rb_per_step row-blocks
        const int grp = rb / rb_per_step, rbl = rb % rb_per_step;
        unsigned short *stage = dst + ((int64_t)grp * steps + t) * dst_step_stride_elems;
        const int64_t base = ((int64_t)(rbl * 3) * 64 + lane) * 8 + i;
        stage[base] = hi;
        stage[base + 64 * 8] = mid;
        stage[base + 2 * 64 * 8] = lo;
    }
}

__global__ void x3_copy_kernel(const float *__restrict__ src, float *__restrict__ dst, int n) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[i];
}

// ---- in-kernel activation split (truncation): 8 fp32 values -> hi, mid, lo bf16x8 -------------------------
struct Split3 {
    bf16x8 hi, mid, lo;
};
__device__ __forceinline__ u32 f2u(float f) { return __builtin_bit_cast(u32, f); }
__device__ __forceinline__ float u2f(u32 u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ u32 pack_hi16(u32 even, u32 odd) {
    // {odd[31:16], even[31:16]} : bf16 element 2j in the low half, 2j+1 in the high half
    return __builtin_amdgcn_perm(odd, even, 0x07060302u);
}
__device__ __forceinline__ Split3 split8(const float (&v)[8]) {
    u32 h[8], m[8], l[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const u32 u = f2u(v[i]);
        h[i] = u;
        const float r1 = v[i] - u2f(u & 0xffff0000u);
        m[i] = f2u(r1);
        const float r2 = r1 - u2f(m[i] & 0xffff0000u);
        l[i] = f2u(r2);
    }
    typedef u32 u32x4 __attribute__((ext_vector_type(4)));
    u32x4 ph, pm, pl;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        ph[j] = pack_hi16(h[2 * j], h[2 * j + 1]);
        pm[j] = pack_hi16(m[2 * j], m[2 * j + 1]);
        pl[j] = pack_hi16(l[2 * j], l[2 * j + 1]);
    }
    Split3 s;
    s.hi = __builtin_bit_cast(bf16x8, ph);
    s.mid = __builtin_bit_cast(bf16x8, pm);
    s.lo = __builtin_bit_cast(bf16x8, pl);
    return s;
}

// acc += W_rowblock(K step) . act : six bf16 MFMAs.  abuf points at the row-block's [3 splits][64 lanes][8 bf16].
__device__ __forceinline__ void mm_x3(const unsigned short *abuf, int lane, const Split3 &b, f32x16 &acc) {
    const bf16x8 ahi = *reinterpret_cast<const bf16x8 *>(abuf + lane * 8);
    const bf16x8 amid = *reinterpret_cast<const bf16x8 *>(abuf + 64 * 8 + lane * 8);
    const bf16x8 alo = *reinterpret_cast<const bf16x8 *>(abuf + 2 * 64 * 8 + lane * 8);
    // small terms first, the dominant hi.hi last
    acc = MFMA16(alo, b.hi, acc);
    acc = MFMA16(ahi, b.lo, acc);
    acc = MFMA16(amid, b.mid, acc);
    acc = MFMA16(amid, b.hi, acc);
    acc = MFMA16(ahi, b.mid, acc);
    acc = MFMA16(ahi, b.hi, acc);
}

// NRB row-blocks of one K step, product-major: consecutive MFMAs write different accumulators, so the dependent-issue
// latency of v_mfma_f32_32x32x16_bf16 on one accumulator never stalls the matrix pipe.
struct A3 {
    bf16x8 hi, mid, lo;
};
__device__ __forceinline__ A3 load_a3(const unsigned short *abuf, int lane) {
    A3 a;
    a.hi = *reinterpret_cast<const bf16x8 *>(abuf + lane * 8);
#ifdef NF_X3_ABL_LDS13      // timing ablation (wrong results): one LDS read per row-block instead of three
    a.mid = a.hi;
    a.lo = a.hi;
#else
    a.mid = *reinterpret_cast<const bf16x8 *>(abuf + 64 * 8 + lane * 8);
    a.lo = *reinterpret_cast<const bf16x8 *>(abuf + 2 * 64 * 8 + lane * 8);
#endif
    return a;
}
__device__ __forceinline__ void mm_x3_4(const unsigned short *buf, int lane, const Split3 &b, f32x16 &c0, f32x16 &c1,
                                        f32x16 &c2, f32x16 &c3) {
#ifdef NF_X3_NO_INTERLEAVE
    mm_x3(buf + 0 * 3 * 512, lane, b, c0);
    mm_x3(buf + 1 * 3 * 512, lane, b, c1);
    mm_x3(buf + 2 * 3 * 512, lane, b, c2);
    mm_x3(buf + 3 * 3 * 512, lane, b, c3);
#else
    const A3 a0 = load_a3(buf, lane), a1 = load_a3(buf + 3 * 512, lane), a2 = load_a3(buf + 6 * 512, lane),
             a3 = load_a3(buf + 9 * 512, lane);
    c0 = MFMA16(a0.lo, b.hi, c0); c1 = MFMA16(a1.lo, b.hi, c1); c2 = MFMA16(a2.lo, b.hi, c2); c3 = MFMA16(a3.lo, b.hi, c3);
    c0 = MFMA16(a0.hi, b.lo, c0); c1 = MFMA16(a1.hi, b.lo, c1); c2 = MFMA16(a2.hi, b.lo, c2); c3 = MFMA16(a3.hi, b.lo, c3);
    c0 = MFMA16(a0.mid, b.mid, c0); c1 = MFMA16(a1.mid, b.mid, c1); c2 = MFMA16(a2.mid, b.mid, c2); c3 = MFMA16(a3.mid, b.mid, c3);
    c0 = MFMA16(a0.mid, b.hi, c0); c1 = MFMA16(a1.mid, b.hi, c1); c2 = MFMA16(a2.mid, b.hi, c2); c3 = MFMA16(a3.mid, b.hi, c3);
    c0 = MFMA16(a0.hi, b.mid, c0); c1 = MFMA16(a1.hi, b.mid, c1); c2 = MFMA16(a2.hi, b.mid, c2); c3 = MFMA16(a3.hi, b.mid, c3);
    c0 = MFMA16(a0.hi, b.hi, c0); c1 = MFMA16(a1.hi, b.hi, c1); c2 = MFMA16(a2.hi, b.hi, c2); c3 = MFMA16(a3.hi, b.hi, c3);
#endif
}
__device__ __forceinline__ void mm_x3_3(const unsigned short *buf, int lane, const Split3 &b, f32x16 &c0, f32x16 &c1,
                                        f32x16 &c2) {
#ifdef NF_X3_NO_INTERLEAVE
    mm_x3(buf + 0 * 3 * 512, lane, b, c0);
    mm_x3(buf + 1 * 3 * 512, lane, b, c1);
    mm_x3(buf + 2 * 3 * 512, lane, b, c2);
#else
    const A3 a0 = load_a3(buf, lane), a1 = load_a3(buf + 3 * 512, lane), a2 = load_a3(buf + 6 * 512, lane);
    c0 = MFMA16(a0.lo, b.hi, c0); c1 = MFMA16(a1.lo, b.hi, c1); c2 = MFMA16(a2.lo, b.hi, c2);
    c0 = MFMA16(a0.hi, b.lo, c0); c1 = MFMA16(a1.hi, b.lo, c1); c2 = MFMA16(a2.hi, b.lo, c2);
    c0 = MFMA16(a0.mid, b.mid, c0); c1 = MFMA16(a1.mid, b.mid, c1); c2 = MFMA16(a2.mid, b.mid, c2);
    c0 = MFMA16(a0.mid, b.hi, c0); c1 = MFMA16(a1.mid, b.hi, c1); c2 = MFMA16(a2.mid, b.hi, c2);
    c0 = MFMA16(a0.hi, b.mid, c0); c1 = MFMA16(a1.hi, b.mid, c1); c2 = MFMA16(a2.hi, b.mid, c2);
    c0 = MFMA16(a0.hi, b.hi, c0); c1 = MFMA16(a1.hi, b.hi, c1); c2 = MFMA16(a2.hi, b.hi, c2);
#endif
}

template <bool RELU>
__device__ __forceinline__ Split3 split_regs(const f32x16 &src, int half) {  // registers 8 half .. 8 half + 7
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        v[i] = half ? src[8 + i] : src[i];
        if (RELU) v[i] = fmaxf(v[i], 0.0f);
    }
    return split8(v);
}

// Up to F_MAX_LAYERS layers of one shape in ONE persistent launch (as rqs_fused.hip's FlowArgs): rows stay in the wave's LDS
// stash between layers, the weight stream runs straight through the layer boundaries.
struct X3Args {
    const float *pack[F_MAX_LAYERS];  // split-bf16 blobs in PROCESSING order
    unsigned long long parity;        // bit l: mask parity of layer l (0: transform features on odd columns)
    int nlayers;
};

// DIR: 0 = density (wrapper.inverse), 1 = sample (wrapper.forward).  LU: fuse each layer's LULinearPermute.
template <int DIR, bool LU>
__global__ void __launch_bounds__(X3_THREADS, 2)
rqs_fused_x3_kernel(const float *__restrict__ x, float *__restrict__ y, float *__restrict__ logdet, X3Args xa, int64_t B,
                    int nblk, RqsParams<float> p, int acc) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    X3Layout lay;
    lay.nblk = nblk;
    const FusedLayout fl = lay.f32();
    float *ring = smem;                            // 3 x 24 KB
    float *stash = ring + 3 * X3_RING_FLOATS;      // 8 waves x 32 x 64
    float *small = stash + X3_NW * 32 * 64;        // biases + tables
    const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nbase = lay.n_base();
    const int SPL = (nbase + (LU ? 2 : 0)) / 2;    // acquires (pairs of K-step stages) per layer
    float *st = stash + wid * 2048 + lane;

    // acquire index local to a layer -> byte offset in that layer's stage area and shape (wide: 12 KB stage = 12 x 1 KB pieces;
    // else 9 KB = 12 x 768 B pieces)
    auto stage_off = [&](int S, bool &wide) -> int64_t {
        const int s = 2 * S;  // first K-step stage of the pair (pairs are contiguous in the blob)
        int b = s;
        if (LU && DIR == 0) {
            if (s < 2) { wide = true; return lay.off_lu(0); }
            b = s - 2;
        }
        if (b >= nbase) {
            wide = true;
            return lay.off_lu(1);          // LU && DIR == 1: the layer's last acquire
        }
#ifdef NF_X3_WIDE_FINAL
        wide = true;
#else
        wide = b < lay.n_init() + lay.n_hidden();
#endif
        return lay.base_off(b);
    };
    // the DMA stream is issued in acquire order across the layers: (is_layer, is_local) = the next pair to request; past the
    // last layer the look-ahead lands in the last blob's padding
    int is_layer = 0, is_local = 0;
    // every wave issues exactly 3 LDS-DMA instructions per stage (counted vmcnt below relies on it)
    auto issue = [&](int s) {
        bool wide = true;
        const char *src;
        if (is_layer < xa.nlayers) {
            src = reinterpret_cast<const char *>(xa.pack[is_layer]) + lay.off_stage_bytes() + stage_off(is_local, wide);
            if (++is_local == SPL) { is_local = 0; ++is_layer; }
        } else {
            src = reinterpret_cast<const char *>(xa.pack[xa.nlayers - 1]) + lay.off_stage_bytes() + lay.off_pad();
        }
        float *slot = ring + (s % 3) * X3_RING_FLOATS;
        if (wide) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int piece = wid * 3 + i;  // 24 pieces of 1 KB
                __builtin_amdgcn_global_load_lds(src + piece * 1024 + lane * 16,
                                                 (__attribute__((address_space(3))) void *)((char *)slot + piece * 1024), 16, 0, 0);
            }
        } else {
            // 18 KB super-stage as 24 pieces of 768 B: 16-byte DMA with 48 active lanes (the 12-byte-per-lane form of
            // global_load_lds did not produce a contiguous LDS image on gfx950; every wave still issues 3 DMAs)
            if (lane < 48) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const int piece = wid * 3 + i;
                    __builtin_amdgcn_global_load_lds(src + piece * 768 + lane * 16,
                                                     (__attribute__((address_space(3))) void *)((char *)slot + piece * 768), 16, 0, 0);
                }
            }
        }
    };
    int stage = 0;
    auto acquire = [&]() -> const unsigned short * {
        // this wave's 3 pieces of `stage` have landed once at most the 3 pieces of stage+1 are outstanding
#ifndef NF_ABL_NOWAIT
        NF_WAIT_VMCNT(3);
#endif
#ifndef NF_ABL_NOBAR
        __builtin_amdgcn_s_barrier();
#endif
        asm volatile("" ::: "memory");
#ifndef NF_ABL_NODMA
        issue(stage + 2);
#endif
        const unsigned short *buf = reinterpret_cast<const unsigned short *>(ring + (stage % 3) * X3_RING_FLOATS);
        ++stage;
        return buf;
    };

    // ---- prologue: rows -> the wave's stash (they stay there between the layers) ----
#pragma unroll
    for (int Q = 0; Q < 4; ++Q) {
        const int64_t row = (int64_t)blockIdx.x * X3_ROWS + wid * 32 + (lane & 31);   // (recomputed in the epilogue: not
        const bool valid = row < B;                                                    //  kept live across the layers)
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
        if (valid) {
            const float *src = x + row * F_D + 16 * Q + 8 * hh;
            a = *reinterpret_cast<const f32x4 *>(src);
            b = *reinterpret_cast<const f32x4 *>(src + 4);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            st[(8 * Q + c) * 64] = a[c];
            st[(8 * Q + 4 + c) * 64] = b[c];
        }
    }
    float ld = 0.0f;

    // Counting: two stages are always in flight (3 LDS-DMA instructions per wave each).  Every acquire() waits until at
    // most 3 VMEM operations of this wave are outstanding -- memory operations retire in order, so the 3 pieces of
    // the stage about to be consumed (always older than the 3 pieces of the following stage) have landed -- then
    // passes the workgroup barrier (every wave's pieces landed, every wave done with the slot being refilled) and
    // issues stage+2.  (The small section's ordinary loads at a layer's start drain the queue once per layer: stricter,
    // never weaker.)
    const int lane_outer = lane;
    for (int layer = 0; layer < xa.nlayers; ++layer) {
    // every per-lane address below (bias rows, table rows) is derived from these two: opaque per layer, or the compiler hoists
    // all of them out of the layer loop and keeps ~20 VGPRs live across it (spills at the final layer's peak)
    int lane_l = lane_outer;
    asm volatile("" : "+v"(lane_l));
    const int lane = lane_l, hh = lane >> 5, tid = wid * 64 + lane;
    int nblk_l = nblk;
    asm volatile("" : "+s"(nblk_l));    // ... and the blob offsets (scalar) are recomputed per layer instead of held in SGPRs
    FusedLayout fl;
    fl.nblk = nblk_l;
    float *st = stash + wid * 2048 + lane;
    const float *pack = xa.pack[layer];
    const int par_t = ((xa.parity >> layer) & 1ull) ? 0 : 1;
    const int par_i = par_t ^ 1;
    __syncthreads();  // every wave is done with the previous layer's small section
    for (int i = tid; i < fl.small_floats(); i += X3_THREADS) small[i] = pack[F_HDR + i];
    if (layer == 0) {
        issue(0);
        issue(1);
    }
    __syncthreads();  // small section visible to all waves
    if (LU && DIR == 0) {
        float xin[32];
#pragma unroll
        for (int c = 0; c < 32; ++c) xin[c] = st[c * 64];
        const float *bsrc = small + fl.off_bias_lu(0) + hh * 16;
        f32x16 o0 = load_bias16(bsrc), o1 = load_bias16(bsrc + 32);
        {
            const unsigned short *buf = acquire();  // 4 K steps: [t][m][split][lane][8]
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = xin[8 * t + i];
                const Split3 b = split8(v);
                mm_x3(buf + ((t * 2 + 0) * 3) * 512, lane, b, o0);
                mm_x3(buf + ((t * 2 + 1) * 3) * 512, lane, b, o1);
            }
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            st[c * 64] = o0[c];
            st[(16 + c) * 64] = o1[c];
        }
        if (hh == 0) ld += pack[3];
    }
    __builtin_amdgcn_sched_barrier(0);   // nothing of the conditioner is hoisted above the LU product (register peak)

    // ---- unconditional spline on the identity half: sample direction first, density deferred ----
    float bx[16];
    {
        const float *tabs = small + fl.off_tables();
#pragma unroll
        for (int Q = 0; Q < 4; ++Q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 8 * Q + 2 * r + par_i;
                const int f = 8 * Q + 4 * hh + r;
                const float xi = st[c * 64];
                if (DIR == 1) {
                    float yi, l;
                    rqs_table_fast<true>(p, xi, tabs + f * F_TABW, yi, l);
                    st[c * 64] = yi;
                    ld += l;
                    bx[4 * Q + r] = yi;
                } else {
                    bx[4 * Q + r] = xi;
                }
            }
        // the log-det terms are finished here (see rqs_fused.hip: otherwise their operands stay live across the MFMA phases)
        if (DIR == 1) asm volatile("" : "+v"(ld));
    }

    // ---- initial layer (K = 32: two K steps, each stage = 4 row-blocks x 3 splits) ----
    f32x16 H0, H1, H2, H3;
    {
        const float *bsrc = small + fl.off_bias_init() + hh * 16;
        H0 = load_bias16(bsrc);
        H1 = load_bias16(bsrc + 32);
        H2 = load_bias16(bsrc + 64);
        H3 = load_bias16(bsrc + 96);
        const unsigned short *buf2 = acquire();
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const unsigned short *buf = buf2 + t * (X3_SLOT_BYTES / 2);
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = bx[8 * t + i];
            const Split3 b = split8(v);
            mm_x3_4(buf, lane, b, H0, H1, H2, H3);
        }
    }

    // ---- residual blocks ----
    for (int blk = 0; blk < nblk; ++blk) {
        f32x16 T0, T1, T2, T3;
        {
            const float *bsrc = small + fl.off_bias_hidden(2 * blk) + hh * 16;
            T0 = load_bias16(bsrc);
            T1 = load_bias16(bsrc + 32);
            T2 = load_bias16(bsrc + 64);
            T3 = load_bias16(bsrc + 96);
        }
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
            const unsigned short *buf2 = acquire();
            const f32x16 &src = tp == 0 ? H0 : (tp == 1 ? H1 : (tp == 2 ? H2 : H3));
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) {
                const Split3 b = split_regs<true>(src, tl);
                mm_x3_4(buf2 + tl * (X3_SLOT_BYTES / 2), lane, b, T0, T1, T2, T3);
            }
        }
        {
            const float *bsrc = small + fl.off_bias_hidden(2 * blk + 1) + hh * 16;
            H0 += load_bias16(bsrc);
            H1 += load_bias16(bsrc + 32);
            H2 += load_bias16(bsrc + 64);
            H3 += load_bias16(bsrc + 96);
        }
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
            const unsigned short *buf2 = acquire();
            const f32x16 &src = tp == 0 ? T0 : (tp == 1 ? T1 : (tp == 2 ? T2 : T3));
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) {
                const Split3 b = split_regs<true>(src, tl);
                mm_x3_4(buf2 + tl * (X3_SLOT_BYTES / 2), lane, b, H0, H1, H2, H3);
            }
        }
    }

    // ---- final layer: H is split once (8 K steps x 3 bf16x8), then 8 groups x 8 K steps x 3 row-blocks ----
    Split3 hs[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const f32x16 &src = (t >> 1) == 0 ? H0 : ((t >> 1) == 1 ? H1 : ((t >> 1) == 2 ? H2 : H3));
        hs[t] = split_regs<false>(src, t & 1);
    }
    float prm0[24], prm1[24];
    auto extract = [&](const f32x16 &A0, const f32x16 &A1, const f32x16 &A2) {
#pragma unroll
        for (int v = 0; v < 24; ++v) {
            prm0[v] = v < 16 ? A0[v] : A1[v - 16];
            prm1[v] = (v + 24) < 32 ? A1[v + 24 - 16] : A2[v + 24 - 32];
        }
    };
    auto element = [&](int g, int f, const float (&prm)[24]) {
#ifdef NF_ABL_NOEPI
#pragma unroll
        for (int v = 0; v < 24; ++v) asm volatile("" ::"v"(prm[v]));
        return;
#endif
        const int slot = 8 * (g >> 1) + 4 * (g & 1) + par_t + 2 * f;
        const float xt = st[slot * 64];
        float yt, l;
        rqs_regs<DIR == 1>(p, xt, prm, yt, l);
        st[slot * 64] = yt;
        ld += l;
    };
    auto uncond_pair = [&](int g) {
#ifdef NF_ABL_NOUNCOND
        return;
#endif
        const float *tabs = small + fl.off_tables();
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int r = 2 * (g & 1) + e;
            const int c = 8 * (g >> 1) + 2 * r + par_i;
            const int f = 8 * (g >> 1) + 4 * hh + r;
            float yi, l;
            rqs_table_fast<false>(p, st[c * 64], tabs + f * F_TABW, yi, l);
            st[c * 64] = yi;
            ld += l;
        }
    };
    auto group_mm = [&](int g, f32x16 &A0, f32x16 &A1, f32x16 &A2) {
        const float *bsrc = small + fl.off_bias_final() + (g * 3) * 32 + hh * 16;
        A0 = load_bias16(bsrc);
        A1 = load_bias16(bsrc + 32);
        A2 = load_bias16(bsrc + 64);
#pragma unroll
        for (int tp = 0; tp < 4; ++tp) {
            const unsigned short *buf2 = acquire();  // 2 K steps x [rb (3)][split (3)][lane][8]
            mm_x3_3(buf2, lane, hs[2 * tp], A0, A1, A2);
            mm_x3_3(buf2 + X3_FINAL_BYTES / 2, lane, hs[2 * tp + 1], A0, A1, A2);
        }
    };
    {
        f32x16 A0, A1, A2;
        group_mm(0, A0, A1, A2);
        extract(A0, A1, A2);
    }
    for (int g = 1; g < 8; ++g) {
        // the epilogue of group g-1 (VALU on registers) overlaps with this group's MFMAs on the matrix pipe
        f32x16 A0, A1, A2;
        element(g - 1, 0, prm0);
        element(g - 1, 1, prm1);
        if (DIR == 0) uncond_pair(g - 1);
        group_mm(g, A0, A1, A2);
        extract(A0, A1, A2);
    }
    element(7, 0, prm0);
    element(7, 1, prm1);
    if (DIR == 0) uncond_pair(7);

    // ---- end of the layer: the sample direction's LULinearPermute.forward on the finished rows ----
    if (LU && DIR == 1) {
        float yout[32];
#pragma unroll
        for (int c = 0; c < 32; ++c) yout[c] = st[c * 64];
        const float *bsrc = small + fl.off_bias_lu(1) + hh * 16;
        f32x16 o0 = load_bias16(bsrc), o1 = load_bias16(bsrc + 32);
        {
            const unsigned short *buf = acquire();
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = yout[8 * t + i];
                const Split3 b = split8(v);
                mm_x3(buf + ((t * 2 + 0) * 3) * 512, lane, b, o0);
                mm_x3(buf + ((t * 2 + 1) * 3) * 512, lane, b, o1);
            }
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            st[c * 64] = o0[c];
            st[(16 + c) * 64] = o1[c];
        }
        if (hh == 0) ld -= pack[3];
    }
    }  // layers

    // ---- epilogue ----
    ld += __shfl_xor(ld, 32, 64);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // drain the look-ahead DMAs before the workgroup retires
    int t_end = threadIdx.x;
    asm volatile("" : "+v"(t_end));     // (not the prologue's value kept live: see above)
    const int64_t row = (int64_t)blockIdx.x * X3_ROWS + (t_end >> 6) * 32 + (t_end & 31);
    const int hh_end = (t_end >> 5) & 1;
    const float *st_end = stash + (t_end >> 6) * 2048 + (t_end & 63);
    if (row < B) {
#pragma unroll
        for (int Q = 0; Q < 4; ++Q) {
            f32x4 a, b;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                a[c] = st_end[(8 * Q + c) * 64];
                b[c] = st_end[(8 * Q + 4 + c) * 64];
            }
            float *dst = y + row * F_D + 16 * Q + 8 * hh_end;
            *reinterpret_cast<f32x4 *>(dst) = a;
            *reinterpret_cast<f32x4 *>(dst + 4) = b;
        }
        if (hh_end == 0) ld_store(logdet + row, ld, acc);
    }
}

template <int DIR, bool LU>
static int launch_x3(const void *x, void *y, void *logdet, const X3Args &xa, int64_t B, int num_blocks,
                     const RqsParams<float> &p, int acc, size_t lds, hipStream_t st) {
    static LdsOptIn opted = {};  // one per <DIR, LU> instantiation
    if (opt_in_lds(reinterpret_cast<const void *>(&rqs_fused_x3_kernel<DIR, LU>), lds, opted) != NF_OK) return NF_ENOTSUP;
    const int grid = (int)((B + X3_ROWS - 1) / X3_ROWS);
    hipLaunchKernelGGL((rqs_fused_x3_kernel<DIR, LU>), dim3(grid), dim3(X3_THREADS), lds, st, (const float *)x, (float *)y,
                       (float *)logdet, xa, B, num_blocks, p, acc);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

}  // namespace nf

using namespace nf;

extern "C" int64_t nf_rqs_fused_x3_pack_size(int nI, int nT, int hidden, int num_blocks, int K) {
    if (nI != F_NI || nT != F_NI || hidden != F_H || K != F_K || num_blocks < 0 || num_blocks > 16) return NF_ENOTSUP;
    X3Layout lay;
    lay.nblk = num_blocks;
    return lay.total_bytes();
}

extern "C" int nf_rqs_fused_x3_pack(void *x3pack, const void *f32pack, int num_blocks, int has_lu, nf_stream_t stream) {
    if (num_blocks < 0 || num_blocks > 16) return NF_ENOTSUP;
    if (!x3pack || !f32pack) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    X3Layout lay;
    lay.nblk = num_blocks;
    const FusedLayout fl = lay.f32();
    const float *src = (const float *)f32pack;
    const float *sstages = src + fl.off_stages();
    char *dbase = (char *)x3pack + lay.off_stage_bytes();
    // header + small section verbatim
    hipLaunchKernelGGL(x3_copy_kernel, dim3(16), dim3(256), 0, st, src, (float *)x3pack, fl.off_stages());
    // init: fp32 stage 0 = 4 row-blocks x 4 k-groups -> 2 K-step stages of [4 rb][3][64][8]
    hipLaunchKernelGGL(x3_convert_kernel, dim3(16), dim3(256), 0, st, sstages, (unsigned short *)(dbase + lay.off_init()), 4,
                       4, (int64_t)X3_SLOT_BYTES / 2, 4);
    // hidden linears: 4 fp32 stages (row-blocks) each -> 8 K-step stages
    for (int l = 0; l < 2 * num_blocks; ++l)
        hipLaunchKernelGGL(x3_convert_kernel, dim3(64), dim3(256), 0, st, sstages + (size_t)(1 + 4 * l) * F_STAGE,
                           (unsigned short *)(dbase + lay.off_hidden() + (int64_t)l * 8 * X3_SLOT_BYTES), 4, 16,
                           (int64_t)X3_SLOT_BYTES / 2, 4);
    // final: 24 fp32 stages (g, rb) -> 8 groups x 8 K-step stages of [3 rb][3][64][8] (9 KB)
    hipLaunchKernelGGL(x3_convert_kernel, dim3(256), dim3(256), 0, st, sstages + (size_t)(1 + 8 * num_blocks) * F_STAGE,
                       (unsigned short *)(dbase + lay.off_final()), 24, 16, (int64_t)X3_FINAL_BYTES / 2, 3);
    if (has_lu) {
        // LU: one fp32 stage = 2 row-blocks x 8 k-groups -> 4 K steps; two steps share a 12 KB stage: [t&1][m][3][64][8]
        for (int dir = 0; dir < 2; ++dir) {
            const float *ls = sstages + (size_t)fl.lu_stage(dir) * F_STAGE;
            // rb_per_step = 2 and a step stride of half a slot put K steps (2 sp, 2 sp + 1) back to back in slot sp
            hipLaunchKernelGGL(x3_convert_kernel, dim3(16), dim3(256), 0, st, ls,
                               (unsigned short *)(dbase + lay.off_lu(dir)), 2, 8, (int64_t)X3_SLOT_BYTES / 4, 2);
        }
    }
    NF_CHECK_LAUNCH();
    return NF_OK;
}

static int x3_run(const void *x, void *y, void *logdet, const X3Args &xa, int fuse_lu, int64_t B, int D, int hidden,
                  int num_blocks, int K, double tail_bound, double min_bin_width, double min_bin_height, double min_derivative,
                  int direction, int acc, nf_stream_t stream) {
    if (D != F_D || hidden != F_H || K != F_K || num_blocks < 0 || num_blocks > 16) return NF_ENOTSUP;
    if (B < 0 || (direction != 0 && direction != 1)) return NF_EINVAL;
    if (acc < NF_LD_SUB || acc > NF_LD_ADD) return NF_EINVAL;
    if (B == 0) return NF_OK;
    if (!x || !y || !logdet) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    X3Layout lay;
    lay.nblk = num_blocks;
    auto p = make_rqs_params<float>(K, NF_TAILS_LINEAR, tail_bound, 0, 1, 0, 1, min_bin_width, min_bin_height,
                                    min_derivative, sqrt((double)hidden));
    const size_t lds = (size_t)(3 * X3_RING_FLOATS + X3_NW * 32 * 64 + lay.f32().small_floats()) * sizeof(float);
    if (lds > 160 * 1024) return NF_ENOTSUP;
    if (direction == 0)
        return fuse_lu ? launch_x3<0, true>(x, y, logdet, xa, B, num_blocks, p, acc, lds, st)
                       : launch_x3<0, false>(x, y, logdet, xa, B, num_blocks, p, acc, lds, st);
    return fuse_lu ? launch_x3<1, true>(x, y, logdet, xa, B, num_blocks, p, acc, lds, st)
                   : launch_x3<1, false>(x, y, logdet, xa, B, num_blocks, p, acc, lds, st);
}

extern "C" int nf_rqs_fused_x3(const void *x, void *y, void *logdet, const void *x3pack, int mask_parity, int fuse_lu,
                               int64_t B, int D, int hidden, int num_blocks, int K, double tail_bound,
                               double min_bin_width, double min_bin_height, double min_derivative, int direction, int acc,
                               nf_stream_t stream) {
    if (mask_parity != 0 && mask_parity != 1) return NF_EINVAL;
    if (!x3pack) return NF_EFAULT;
    X3Args xa;
    for (int l = 0; l < F_MAX_LAYERS; ++l) xa.pack[l] = nullptr;
    xa.pack[0] = (const float *)x3pack;
    xa.parity = mask_parity ? 1ull : 0ull;
    xa.nlayers = 1;
    return x3_run(x, y, logdet, xa, fuse_lu, B, D, hidden, num_blocks, K, tail_bound, min_bin_width, min_bin_height,
                  min_derivative, direction, acc, stream);
}

// Up to 64 layers of one shape in ONE persistent launch (the split-bf16 counterpart of nf_rqs_fused_chain): x3packs /
// mask_parities in PROCESSING order.
extern "C" int nf_rqs_fused_x3_chain(const void *x, void *y, void *logdet, const void *const *x3packs, const int *mask_parities,
                                     int num_layers, int fuse_lu, int64_t B, int D, int hidden, int num_blocks, int K,
                                     double tail_bound, double min_bin_width, double min_bin_height, double min_derivative,
                                     int direction, int acc, nf_stream_t stream) {
    if (num_layers < 1 || num_layers > F_MAX_LAYERS) return NF_EINVAL;
    if (!x3packs || !mask_parities) return NF_EFAULT;
    X3Args xa;
    xa.parity = 0ull;
    xa.nlayers = num_layers;
    for (int l = 0; l < F_MAX_LAYERS; ++l) xa.pack[l] = nullptr;
    for (int l = 0; l < num_layers; ++l) {
        if (!x3packs[l]) return NF_EFAULT;
        if (mask_parities[l] != 0 && mask_parities[l] != 1) return NF_EINVAL;
        xa.pack[l] = (const float *)x3packs[l];
        if (mask_parities[l]) xa.parity |= 1ull << l;
    }
    return x3_run(x, y, logdet, xa, fuse_lu, B, D, hidden, num_blocks, K, tail_bound, min_bin_width, min_bin_height,
                  min_derivative, direction, acc, stream);
}
