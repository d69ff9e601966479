// glow_conv.hip -- the conditioner of a GlowBlock, ConvNet2d([Cin, 256, 256, Cout], kernel sizes (3, 1, 3)) =
// conv3x3 -> LeakyReLU -> conv1x1 -> LeakyReLU -> conv3x3 (normflows/nets/cnn.py:5-63 as built by
// normflows/flows/affine/glow.py:41-62), as ONE kernel on exact-fp32 MFMA for gfx950 -- optionally with the rest of the
// GlowBlock (1x1 convolution + ActNorm, affine coupling) in the same launch, or a whole level of GlowBlocks (nf_glow_block / nf_glow_level, GlowLevel below).
// Three kernels share the decomposition and differ in how the pixels and rows are spread over waves: glow_convnet_kernel
// (256-pixel workgroups, described first), glow_convnet_small_kernel (64 pixels) and glow_convnet_tiny_kernel (16 pixels,
// rows split across the waves); nf_glow_convnet_layout picks one per call.
//
// A workgroup owns 256 pixels = 256 / (H W) WHOLE images (H W must divide 256: 16x16, 8x8, 4x4 ...), so the two 3x3
// convolutions never need pixels of another workgroup and the zero padding of cnn.py:33 is the image border itself.
// Per pixel the network is then an MLP with a gather in front and a scatter behind:
//   conv3x3 #1 : im2col -- the B operand of GEMM 1 is gathered from the zero-padded input image in LDS,
//                k = 9 c + tap (the natural (256, Cin, 3, 3) weight order), K1 = 9 Cin padded to a multiple of 8;
//   conv1x1    : GEMM 2, 256 x 256;
//   conv3x3 #2 : col2im -- GEMM 3 produces, per pixel, the 9 Cout partial products P[co, tap] = W3[co, :, tap] . h2,
//                and out[co, y, x] = b3[co] + sum_tap P[co, tap][y + ky - 1, x + kx - 1] is a 9-term gather over
//                neighbouring pixels through LDS (fixed summation order: deterministic).  Output rows are grouped 3
//                channels (27 rows, 5 padding) per 32-row MFMA block.
// The 256-channel hidden tensors never leave the chip: h1 = 8 x 16 accumulator registers per lane (wave = 32 pixels,
// v_mfma_f32_32x32x2_f32, transposed GEMMs as in rqs_fused.hip); h2 is produced 32 channels at a time and consumed
// at once as the K = 32 slice of GEMM 3 (its C registers ARE the B operand: lane-half hh contracts over exactly the
// units it holds).  Up to 4 output blocks (12 channels) are accumulated per sweep over h2; more output channels take
// further sweeps (h2 recomputed).  Weights stream through a 2-slot LDS ring by global_load_lds, one barrier per
// 16 KB stage, shared by the 8 waves.
//
// HBM traffic per pixel: 4 Cin bytes in, 4 Cout bytes out (the library path writes and re-reads the two 1 KB hidden
// tensors and runs two extra bias / activation passes).
#include "fused_common.hpp"

namespace nf {

constexpr int GC_HID = 256;          // hidden channels
constexpr int GC_NW = 8;             // waves per workgroup
constexpr int GC_PX = 32 * GC_NW;    // pixels per workgroup
constexpr int GC_HDR = 64;
constexpr int GC_STAGE = 4096;       // floats per stage (16 KB)
#ifndef NF_GC_RING
#define NF_GC_RING 3
#endif
constexpr int GC_RING = NF_GC_RING;  // weight-ring slots of the 256-pixel kernel
constexpr int GC_OBP = 4;            // output row-blocks (x 3 channels) per sweep
constexpr int GC_KG1MAX = 16;        // k-groups of GEMM 1 whose gathered operands are kept in registers (one stage)

struct GcMeta {
    int Cin, Cout;
    int K1;      // 9 Cin
    int nkg1;    // k-groups (of 8) of GEMM 1
    int nst1;    // stages per row-block of GEMM 1 (16 k-groups each)
    int OB;      // output row-blocks = ceil(Cout / 3)
    int npass;   // sweeps = ceil(OB / 4)
    int small;   // floats of the small section: b1 (256, register order) | b2 (256, register order) | b3 (12 npass)
    float slope;
};

static inline GcMeta gc_meta(int Cin, int Cout, double slope) {
    GcMeta m;
    m.Cin = Cin; m.Cout = Cout;
    m.K1 = 9 * Cin;
    m.nkg1 = (m.K1 + 7) / 8;
    m.nst1 = (m.nkg1 + 15) / 16;
    m.OB = (Cout + 2) / 3;
    m.npass = (m.OB + GC_OBP - 1) / GC_OBP;
    m.small = 2 * GC_HID + 3 * GC_OBP * m.npass;
    m.slope = (float)slope;
    return m;
}
__host__ __device__ inline int gc_small_padded(const GcMeta &m) { return (m.small + 63) / 64 * 64; }
__host__ __device__ inline int gc_off_stages(const GcMeta &m) { return GC_HDR + gc_small_padded(m); }
__host__ __device__ inline int gc_nstages_blob(const GcMeta &m) { return 8 * m.nst1 + 16 + 8 * m.npass; }

// One thread per blob float.  A stage always holds 32 rows x 128 k (GEMM 1 / 2) or 4 output blocks x 32 k (GEMM 3);
// only the order inside a stage depends on the MFMA shape the kernel uses:
//   wide  (v_mfma_f32_32x32x2): [k-group 0..15][lane][4], row = lane & 31, k = 8 kg + 4 (lane >> 5) + r4
//                               (GEMM 3: k-group = 4 mm + s); biases in accumulator-register order;
//   small (v_mfma_f32_16x16x4): [unit 0..15][lane][4], unit = 8 ob + kb, row = 16 ob + (lane & 15),
//                               k = 16 kb + 4 (lane >> 4) + r4 (GEMM 3: unit = 4 mm + 2 ob + kb); biases in natural order.
__global__ void gc_pack_kernel(const float *__restrict__ W1, const float *__restrict__ b1, const float *__restrict__ W2,
                               const float *__restrict__ b2, const float *__restrict__ W3, const float *__restrict__ b3,
                               float *__restrict__ blob, GcMeta m, int64_t total, int small_layout) {
    const int offs = gc_off_stages(m);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        float v = 0.0f;
        if (i < GC_HDR) {
            v = i == 0 ? 356.0f : (i == 1 ? (float)m.Cin : (i == 2 ? (float)m.Cout : 0.0f));
        } else if (i < offs) {
            const int j = (int)i - GC_HDR;
            if (j < 2 * GC_HID) {
                const int q = j & 255, reg = q & 15, hh = (q >> 4) & 1, mb = q >> 5;
                const int row = small_layout ? q : 32 * mb + 8 * (reg >> 2) + 4 * hh + (reg & 3);
                v = j < GC_HID ? b1[row] : b2[row];
            } else if (j - 2 * GC_HID < m.Cout) {
                v = b3[j - 2 * GC_HID];
            }
        } else {
            const int64_t t = i - offs;
            const int st = (int)(t / GC_STAGE), e = (int)(t % GC_STAGE);
            const int r4 = e & 3, lane = (e >> 2) & 63, kg = (e >> 8) & 15;
            const bool gemm3 = st >= 8 * m.nst1 + 16;
            int rho, kin, mm = 0;   // row inside the 32-row block, k inside the stage's k range, output block (GEMM 3)
            if (!small_layout) {
                rho = lane & 31;
                if (gemm3) { mm = kg >> 2; kin = 8 * (kg & 3) + 4 * (lane >> 5) + r4; }
                else kin = 8 * kg + 4 * (lane >> 5) + r4;
            } else {
                if (gemm3) { mm = kg >> 2; rho = 16 * ((kg >> 1) & 1) + (lane & 15); kin = 16 * (kg & 1) + 4 * (lane >> 4) + r4; }
                else { rho = 16 * (kg >> 3) + (lane & 15); kin = 16 * (kg & 7) + 4 * (lane >> 4) + r4; }
            }
            if (st < 8 * m.nst1) {
                const int mb = st / m.nst1, c = st % m.nst1;
                const int k = 128 * c + kin;
                if (k < m.K1) v = W1[(size_t)(32 * mb + rho) * m.K1 + k];
            } else if (!gemm3) {
                const int q = st - 8 * m.nst1, j = q >> 1, half = q & 1;
                v = W2[(size_t)(32 * j + rho) * GC_HID + 128 * half + kin];
            } else {
                const int q = st - 8 * m.nst1 - 16, pass = q >> 3, j = q & 7;
                const int cc = rho / 9, tap = rho - 9 * cc;
                const int co = 3 * (GC_OBP * pass + mm) + cc;
                if (rho < 27 && co < m.Cout) v = W3[((size_t)co * GC_HID + 32 * j + kin) * 9 + tap];
            }
        }
        blob[i] = v;
    }
}

#define GC_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
// Workgroup barrier of these kernels: LDS traffic of the wave drained, vector-memory operations left alone.  __syncthreads()
// is a fence, which the compiler implements as s_waitcnt vmcnt(0): every barrier then waits for everything a wave has in
// flight -- the younger stages of the weight rings (which made the depth of a ring irrelevant), the next block's first
// stages during the serial phases, the 16-unit register stream of the 16-pixel kernel.  Whatever crosses waves here goes
// through LDS; data that arrives by DMA is waited for explicitly (counted vmcnt) before the barrier that publishes it.
#define GL_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define GC_RING_BARRIER() GL_BARRIER()

// acc += Wblock(32 x 128) * B with the B operand of (k-group s, r) = bsrc[s >> 2][4 (s & 3) + r]: 16 ds_read_b128 + 64 MFMA
__device__ __forceinline__ void gc_mm128(const float *buf, int lane, f32x16 &acc, const f32x16 &b0, const f32x16 &b1,
                                         const f32x16 &b2, const f32x16 &b3) {
    // (requesting the A operands two k-groups ahead of their MFMAs was measured in round 3: slower here -- with two MFMA
    // waves per SIMD the other wave covers the read latency and the extra registers cost more)
    // Round 6: that measurement was void -- with the ring's requests issued through the builtin the compiler knew an LDS-DMA was
    // pending and answered EVERY later LDS read with s_waitcnt lgkmcnt(0) (the loop compiled to read -> full wait -> 4 MFMAs,
    // look-ahead or not).  With the requests as inline asm (gc_dma16) the look-ahead below is real: k-group s + 2's operand is
    // requested before k-group s's MFMAs (-DNF_GC_NOPIPE: the plain loop).
#ifdef NF_GC_NOPIPE
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(buf + s * 256 + lane * 4);
        const f32x16 &bs = (s >> 2) == 0 ? b0 : ((s >> 2) == 1 ? b1 : ((s >> 2) == 2 ? b2 : b3));
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = GC_MFMA(a[r], bs[4 * (s & 3) + r], acc);
    }
#else
    f32x4 a0 = *reinterpret_cast<const f32x4 *>(buf + lane * 4), a1 = *reinterpret_cast<const f32x4 *>(buf + 256 + lane * 4);
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const f32x4 a = a0;
        a0 = a1;
        if (s + 2 < 16) a1 = *reinterpret_cast<const f32x4 *>(buf + (s + 2) * 256 + lane * 4);
        __builtin_amdgcn_sched_barrier(0);
        const f32x16 &bs = (s >> 2) == 0 ? b0 : ((s >> 2) == 1 ? b1 : ((s >> 2) == 2 ? b2 : b3));
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = GC_MFMA(a[r], bs[4 * (s & 3) + r], acc);
        __builtin_amdgcn_sched_barrier(0);
    }
#endif
}

// One 16-byte-per-lane LDS-DMA request as inline asm (round 6, see gc_mm128): base = wave-uniform global pointer (SGPR pair),
// byte_off = the lane's 32-bit offset, dst = wave-uniform LDS address.  m0 is reserved: saved and restored.  Landing is waited for
// by the counted NF_WAIT_VMCNT of the ring's acquire, as before.
__device__ __forceinline__ void gc_dma16(const float *base, uint32_t byte_off, float *dst) {
    uint32_t m0_;
    const uint32_t ldsa = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)dst);
    // (the "s" constraint alone does not make the pointer scalar: a base the compiler cannot prove wave-uniform came out as a VGPR
    // pair in the instruction -- both halves through v_readfirstlane)
    const uint64_t b64 = (uint64_t)(uintptr_t)base;
    base = reinterpret_cast<const float *>((uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b64 >> 32)) << 32) |
                                                       (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b64)));
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(m0_) : "s"(ldsa), "v"(byte_off), "s"(base) : "memory");
}

__device__ __forceinline__ void gc_leaky(f32x16 &v, float slope) {
#pragma unroll
    for (int c = 0; c < 16; ++c) v[c] = fmaxf(v[c], v[c] * slope);   // 0 <= slope <= 1 (cnn.py:38 LeakyReLU)
}

// ---- optional fusion of the rest of the GlowBlock(s) around the conditioner --------------------------------------------
// GlowBlock = [AffineCouplingBlock(split "channel"), Invertible1x1Conv, ActNorm] (glow.py:11-84).  With parameters frozen
// the last two are ONE per-pixel affine map m = Wp z + bp with a constant log|det| per pixel (flows/glow.py::_fused_mix);
// the workgroup holds whole images, so the block runs inside the conditioner's launch:
//   direction 1 (GlowBlock.inverse, the density direction): mix first, conditioner on the mixed identity half,
//               coupling inverse (coupling.py:150-171) on the mixed other half;
//   direction 0 (GlowBlock.forward): conditioner on the raw identity half, coupling forward (:117-148), then the mix.
// A whole LEVEL of the multi-scale flow (core.py:588-616 / :553-586: up to GL_MAXB consecutive GlowBlocks of one shape) is
// one persistent launch: no block needs another workgroup's pixels, so the workgroup keeps its images in LDS (two
// ping-pong planes of C x pixels) across all blocks, z goes HBM -> LDS once and back once, the per-image log-det is
// accumulated in LDS in block order (deterministic) and stored once.  The glue around the level is folded into that one
// load / store: the input is the channel concatenation of two tensors (Merge, reshape.py:88-100) or the
// Squeeze.inverse view of an un-squeezed tensor (reshape.py:122-128); the output is split after cout0 channels into two
// tensors (Split, reshape.py:30-85 "channel") or written through the Squeeze.forward view (:116-121).
constexpr int GL_MAXB = 64;
struct GlowLevel {
    // DEVICE table of 4 nblocks pointers, block b (PROCESSING order) at [4 b .. 4 b + 3]: packed conditioner | (C, C) mix
    // matrix | (C) mix bias | log|det| of the mix per pixel (device scalar).  (A by-value array in the kernel arguments
    // makes the compiler hold all of it in SGPRs: hundreds of spilled scalars.)
    const float *const *tbl;
    __device__ __forceinline__ const float *blob(int b) const { return tbl[4 * b]; }
    __device__ __forceinline__ const float *Wp(int b) const { return tbl[4 * b + 1]; }
    __device__ __forceinline__ const float *bp(int b) const { return tbl[4 * b + 2]; }
    __device__ __forceinline__ const float *ldu(int b) const { return tbl[4 * b + 3]; }
    int nblocks;                  // 0: conditioner only (nf_glow_convnet)
    int C, c1;                    // channels of z; the identity half = the first c1 = ceil(C / 2) channels (reshape.py:31)
    int scale_map;                // NF_SCALE_EXP | NF_SCALE_SIGMOID | NF_SCALE_SIGMOID_INV
    int direction, acc;
    const float *in0, *in1;       // level input: channels [0, cin0) of in0 (B, cin0, H, W) then in1 (B, C - cin0, H, W); or,
    int cin0, in_sq;              //   in_sq = 1, the Squeeze.inverse view of in0 = (B, C / 4, 2H, 2W)
    float *out0, *out1;           // level output: channels [0, cout0) -> out0, the rest -> out1; or, out_sq = 1, through
    int cout0, out_sq;            //   the Squeeze.forward view into out0 = (B, C / 4, 2H, 2W)
    float *logdet;                // (B)
#ifdef NF_GL_TRACE
    unsigned long long *trace;    // debug builds: phase timestamps of workgroup 0 (100 MHz wall clock), 8 per block
#endif
};
#ifdef NF_GL_TRACE
static unsigned long long *g_gl_trace = nullptr;
extern "C" void nf_glow_debug_trace(void *buf) { g_gl_trace = (unsigned long long *)buf; }
#define GL_T(b, idx) do { if (lv.trace && blockIdx.x == 0 && threadIdx.x == 0) lv.trace[(b) * 8 + (idx)] = wall_clock64(); } while (0)
#else
#define GL_T(b, idx) do {} while (0)
#endif

__host__ __device__ inline int gl_wmix_padded(int C) { return (C * C + C + 63) / 64 * 64; }
__host__ __device__ inline int gb_lds_floats(const GlowLevel &lv, int Cout, int PXW) {
    // zA | zB (ping-pong planes) | prm (parameter planes) | ldt (log-det terms) | ldacc (per-image log-det) | 2 x (Wp | bp)
    // | the second bias section (both double-buffered: the next block's are DMA-prefetched under this block's GEMMs)
    return lv.nblocks ? (2 * lv.C + Cout + (lv.C - lv.c1)) * PXW + PXW + 2 * gl_wmix_padded(lv.C) + 2 * GC_HID + 64 +
                            (Cout + 63) / 64 * 64 : 0;
}

// Squeeze correspondence (reshape.py:116-128): small[b, 4 c + 2 i + j, h, w] <-> big[b, c, 2 h + i, 2 w + j]
__device__ __forceinline__ int64_t gl_sq_index(int64_t g, int C, int c, int H, int W, int h, int w) {
    return ((g * (C >> 2) + (c >> 2)) * (2 * H) + 2 * h + ((c >> 1) & 1)) * (int64_t)(2 * W) + 2 * w + (c & 1);
}

// Level input -> zin[c][p] (whole images of the workgroup), log-det accumulators cleared.
template <int PXW, int NT>
__device__ __forceinline__ void gl_load(const GlowLevel &lv, float *zin, float *ldacc, int H, int W, int64_t img0, int64_t B,
                                        int tid) {
    const int HW = H * W, C = lv.C;
    for (int i = tid; i < C * PXW; i += NT) {
        const int c = i / PXW, p = i - c * PXW, im = p / HW, q = p - im * HW;
        const int64_t g = img0 + im;
        float v = 0.0f;
        if (g < B) {
            if (lv.in_sq) { const int h = q / W; v = lv.in0[gl_sq_index(g, C, c, H, W, h, q - h * W)]; }
            else if (c < lv.cin0) v = lv.in0[(g * lv.cin0 + c) * HW + q];
            else v = lv.in1[(g * (C - lv.cin0) + (c - lv.cin0)) * HW + q];
        }
        zin[i] = v;
    }
    for (int i = tid; i < PXW; i += NT) ldacc[i] = 0.0f;
}

// zfin[c][p] -> level output; accumulated per-image log-dets -> logdet (one store per image).
template <int PXW, int NT>
__device__ __forceinline__ void gl_store(const GlowLevel &lv, const float *zfin, const float *ldacc, int H, int W, int64_t img0,
                                         int64_t B, int tid) {
    const int HW = H * W, C = lv.C, IPW = PXW / HW;
    for (int i = tid; i < C * PXW; i += NT) {
        const int c = i / PXW, p = i - c * PXW, im = p / HW, q = p - im * HW;
        const int64_t g = img0 + im;
        if (g < B) {
            const float v = zfin[i];
            if (lv.out_sq) { const int h = q / W; lv.out0[gl_sq_index(g, C, c, H, W, h, q - h * W)] = v; }
            else if (c < lv.cout0) lv.out0[(g * lv.cout0 + c) * HW + q] = v;
            else lv.out1[(g * (C - lv.cout0) + (c - lv.cout0)) * HW + q] = v;
        }
    }
    for (int im = tid; im < IPW; im += NT)
        if (img0 + im < B) ld_store(lv.logdet + img0 + im, ldacc[im], lv.acc);
}

// DMA prefetch (global -> LDS, no registers) of block b's bias section and mix matrix | bias into the given LDS buffers;
// 1 KB / 256 B pieces round-robin over the NW waves.  Completion: any later s_waitcnt vmcnt(0) + barrier.
template <int NW>
__device__ __forceinline__ void gl_prefetch(const GlowLevel &lv, int b, int nsmall, float *small_dst, float *wmix_dst, int wid,
                                            int lane) {
    typedef __attribute__((address_space(3))) void *lds_ptr;
    const float *blob = lv.blob(b) + GC_HDR;
    for (int piece = wid; piece * 256 < nsmall; piece += NW)      // nsmall is a multiple of 4
        if (piece * 256 + lane * 4 < nsmall)
            __builtin_amdgcn_global_load_lds(blob + piece * 256 + lane * 4, (lds_ptr)(small_dst + piece * 256), 16, 0, 0);
    const int C = lv.C, nw = C * C;
    const float *Wp = lv.Wp(b), *bp = lv.bp(b);
    for (int piece = wid; piece * 64 < nw; piece += NW)
        if (piece * 64 + lane < nw) __builtin_amdgcn_global_load_lds(Wp + piece * 64 + lane, (lds_ptr)(wmix_dst + piece * 64), 4, 0, 0);
    if (wid == NW - 1 && lane < C) __builtin_amdgcn_global_load_lds(bp + lane, (lds_ptr)(wmix_dst + nw), 4, 0, 0);   // C <= 64
}

// The per-pixel mix zalt[c][p] = bp[c] + sum_k Wp[c][k] zin[k][p] (k ascending, one fma per term: the order of the plain
// loop).  Thread = (pixel p, channel residue cg mod NG): the zin column of the pixel is read once per GL_MJ channels, the
// matrix rows as 16-byte reads, four k at a time -- a thread of the plain loop waited for two LDS reads per fma, and the
// prologue of a block was this loop (6 of 8 us at the 8x8 level of config 4).
constexpr int GL_MJ = 3;
template <int PXW, int NT>
__device__ __forceinline__ void gl_mix(const float *wmix, const float *zin, float *zalt, int C, int tid) {
    constexpr int NG = NT / PXW;
    static_assert(NT % PXW == 0, "pixels must divide the workgroup");
    const bool vec = (C & 3) == 0 && ((uint32_t)(uintptr_t)wmix & 15u) == 0;
    if (!vec) {
        for (int i = tid; i < C * PXW; i += NT) {
            const int c = i / PXW, p = i - c * PXW;
            float a = wmix[C * C + c];
            for (int k = 0; k < C; ++k) a = fmaf(wmix[c * C + k], zin[k * PXW + p], a);
            zalt[i] = a;
        }
        return;
    }
    const int p = tid % PXW, cg = tid / PXW;
    for (int c0 = cg; c0 < C; c0 += NG * GL_MJ) {
        int cj[GL_MJ];
        float a[GL_MJ];
#pragma unroll
        for (int j = 0; j < GL_MJ; ++j) {
            cj[j] = c0 + j * NG < C ? c0 + j * NG : c0;   // past the end: channel c0 again, not stored
            a[j] = wmix[C * C + cj[j]];
        }
        for (int k = 0; k < C; k += 4) {
            float z[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) z[i] = zin[(k + i) * PXW + p];
#pragma unroll
            for (int j = 0; j < GL_MJ; ++j) {
                const f32x4 w = *reinterpret_cast<const f32x4 *>(wmix + cj[j] * C + k);
#pragma unroll
                for (int i = 0; i < 4; ++i) a[j] = fmaf(w[i], z[i], a[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < GL_MJ; ++j)
            if (c0 + j * NG < C) zalt[(c0 + j * NG) * PXW + p] = a[j];
    }
}

// Start of block b (its mix matrix `wmix` was prefetched): direction 1 mixes zin -> zalt; the conditioner's zero-padded input
// images come from the identity half of the mixed (direction 1) / raw (direction 0) planes.  Ends WITHOUT a barrier
// (the caller's next barrier publishes xin).
// INFLIGHT: vector-memory operations issued AFTER this block's prefetch that may stay outstanding (weight-ring DMAs).
template <int PXW, int NT, int INFLIGHT>
__device__ __forceinline__ void gl_pre(const GlowLevel &lv, int b, const float *zin, float *zalt, const float *wmix, float *xin,
                                       int H, int W, int tid) {
    const int HW = H * W, PH = H + 2, PW = W + 2, IPW = PXW / HW, C = lv.C;
    // the prefetched sections of this block (block 0's were issued last, just before this call)
    if (b == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else NF_WAIT_VMCNT(INFLIGHT);
    GL_BARRIER();   // ... and zin (the level load / the previous block's output)
    const float *src = zin;
    if (lv.direction == 1) {
        gl_mix<PXW, NT>(wmix, zin, zalt, C, tid);
        GL_BARRIER();
        src = zalt;
    }
    const int per_img = lv.c1 * PH * PW, n = IPW * per_img;
    for (int i = tid; i < n; i += NT) {
        const int im = i / per_img, rem = i - im * per_img, c = rem / (PH * PW), rr = rem - c * PH * PW;
        const int yy = rr / PW - 1, xx = rr - (yy + 1) * PW - 1;
        xin[i] = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? src[c * PXW + im * HW + yy * W + xx] : 0.0f;
    }
}

// End of block b: coupling on the other half in place (parameter planes 2 i = shift, 2 i + 1 = scale, coupling.py:117-171);
// direction 0 then mixes zin -> zalt.  Either way the block's output is in zalt.  Per-image log-det in a fixed order (one
// wave per image), added to the level's accumulator.  Ends with a barrier.
template <int PXW, int NT>
__device__ __forceinline__ void gl_post(const GlowLevel &lv, int b, float *zin, float *zalt, const float *wmix, const float *prm,
                                        float *ldt, float *ldacc, int H, int W, int tid) {
    const int HW = H * W, IPW = PXW / HW, C = lv.C, c1 = lv.c1, n2 = C - c1;
    GL_BARRIER();   // parameter planes complete
    float *z2 = (lv.direction == 1 ? zalt : zin) + c1 * PXW;
    for (int e = tid; e < n2 * PXW; e += NT) {
        const int i = e / PXW, p = e - i * PXW;
        const float sh = prm[(2 * i) * PXW + p], sc = prm[(2 * i + 1) * PXW + p], v = z2[e];
        float o, l;
        if (lv.scale_map == NF_SCALE_EXP) {
            o = lv.direction == 0 ? v * M<float>::exp(sc) + sh : (v - sh) * M<float>::exp(-sc);
            l = sc;
        } else {
            const float sg = sigmoid(sc + 2.0f), lg = M<float>::log(sg);
            if (lv.scale_map == NF_SCALE_SIGMOID) { o = lv.direction == 0 ? v / sg + sh : (v - sh) * sg; l = -lg; }
            else { o = lv.direction == 0 ? v * sg + sh : (v - sh) / sg; l = lg; }
        }
        ldt[e] = lv.direction == 0 ? l : -l;
        z2[e] = o;
    }
    GL_BARRIER();
    if (lv.direction == 0) gl_mix<PXW, NT>(wmix, zin, zalt, C, tid);
    const int lane = tid & 63, wv = tid >> 6;
    const float ldu = *lv.ldu(b);
    for (int im = wv; im < IPW; im += NT / 64) {
        float a = 0.0f;
        for (int t = lane; t < n2 * HW; t += 64) {
            const int i = t / HW, q = t - i * HW;
            a += ldt[i * PXW + im * HW + q];
        }
        a = wave_sum(a);
        if (lane == 0) ldacc[im] += a + (float)HW * ldu;
    }
    GL_BARRIER();
}

// Padded input images of the conditioner from global memory (plain call).
template <int PXW, int NT>
__device__ __forceinline__ void gc_fill_xin_global(const float *__restrict__ x, int64_t xs_img, float *xin, int Cin, int H, int W,
                                                   int64_t img0, int64_t B, int tid) {
    const int HW = H * W, PH = H + 2, PW = W + 2, IPW = PXW / HW, per_img = Cin * PH * PW, n = IPW * per_img;
    for (int i = tid; i < n; i += NT) {
        const int im = i / per_img, rem = i - im * per_img, c = rem / (PH * PW), rr = rem - c * PH * PW;
        const int yy = rr / PW - 1, xx = rr - (yy + 1) * PW - 1;
        const int64_t g = img0 + im;
        float v = 0.0f;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W && g < B) v = x[g * xs_img + (int64_t)c * HW + yy * W + xx];
        xin[i] = v;
    }
}

// col2im of `nblk` consecutive output blocks starting at blk0 (their tap products at P + (b - blk0) 32 PXW): 9-term
// neighbour gather, + bias; to HBM (plain call) or to the LDS parameter planes prm[co][p] (fused block).
template <int PXW, int NT>
__device__ __forceinline__ void gc_gather_block(const float *P, int blk0, const GcMeta &mt, int H, int W, int64_t img0, int64_t B,
                                                const float *small, float *__restrict__ out, float *prm, int tid,
                                                int nblk = 1) {
    const int HW = H * W;
    for (int e = tid; e < nblk * 3 * PXW; e += NT) {
        const int bl = e / (3 * PXW), e3 = e - bl * 3 * PXW;
        const int cc = e3 / PXW, p = e3 - cc * PXW;
        const int co = 3 * (blk0 + bl) + cc;
        const int im = p / HW, q = p - im * HW, y = q / W, xq = q - y * W;
        const int64_t g = img0 + im;
        if (co < mt.Cout && (prm || g < B)) {
            float sum = small[2 * GC_HID + co];
            const float *pr = P + (bl * 32 + cc * 9) * PXW + p;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int yy = y + ky - 1, xx = xq + kx - 1;
                    if (yy >= 0 && yy < H && xx >= 0 && xx < W) sum += pr[(ky * 3 + kx) * PXW + (ky - 1) * W + (kx - 1)];
                }
            }
            if (prm) prm[co * PXW + p] = sum;
            else out[(g * mt.Cout + co) * HW + q] = sum;
        }
    }
}

// Opaque copy of a per-thread value, taken once per block of a level chain: everything derived from it is recomputed per
// block instead of being hoisted out of the block loop and carried (spilled) across the MFMA phases.
__device__ __forceinline__ int gl_opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

// LDS planes of the fused level behind the conditioner's own buffers (all kernels): zA | zB | prm | ldt | ldacc | wmix x 2 | second bias section
struct GlPlanes {
    float *zA, *zB, *prm, *ldt, *ldacc, *wmix, *wmix2, *small2;
};
__device__ __forceinline__ GlPlanes gl_planes(float *base, const GlowLevel &lv, int Cout, int PXW) {
    GlPlanes q;
    q.zA = base;
    q.zB = q.zA + lv.C * PXW;
    q.prm = q.zB + lv.C * PXW;
    q.ldt = q.prm + Cout * PXW;
    q.ldacc = q.ldt + (lv.C - lv.c1) * PXW;
    q.wmix = q.ldacc + PXW;
    q.wmix2 = q.wmix + gl_wmix_padded(lv.C);
    q.small2 = q.wmix2 + gl_wmix_padded(lv.C);
    return q;
}

template <bool PAIR1>     // GEMM 1's stages travel in pairs (K1 <= 64, see below)
__global__ void __launch_bounds__(64 * GC_NW, 2)
glow_convnet_kernel(const float *__restrict__ x, int64_t xs_img, float *__restrict__ out, const float *__restrict__ blob0,
                    GcMeta mt, int64_t B, int H, int W, GlowLevel lv) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int HW = H * W, PH = H + 2, PW = W + 2, IPW = GC_PX / HW;
    const int K1p = 8 * mt.nkg1;
    float *ring = smem;                          // GC_RING x 16 KB weight stages
    float *P = ring + GC_RING * GC_STAGE;        // 32 rows x 256 pixels: one output block's tap products
    float *small0 = P + 32 * GC_PX;              // biases
    int *koff = reinterpret_cast<int *>(small0 + gc_small_padded(mt));   // im2col offset of every k
    float *xin = reinterpret_cast<float *>(koff + K1p);                  // zero-padded input images [img][c][PH][PW]
    const GlPlanes pl = gl_planes(xin + IPW * mt.Cin * PH * PW, lv, mt.Cout, GC_PX);   // fused level only
    const int tid0 = threadIdx.x;
    const int wid = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int64_t img0 = (int64_t)blockIdx.x * IPW;
    const bool fused = lv.nblocks > 0;
    const int nb = fused ? lv.nblocks : 1;

    // GEMM 1 with K1 <= 64 (Cin <= 7: the 16x16 level of config 4) fills only half of a blob stage per row-block: two
    // row-blocks' used halves then travel as ONE ring stage (pieces 0..7 | 8..15): 4 barriers instead of 8 for the same MFMAs
    constexpr bool pair1 = PAIR1;                         // = mt.nkg1 <= 8 (the launcher)
    const int bl1 = 8 * mt.nst1;                          // GEMM 1's stages in the blob
    const int nst_l1 = pair1 ? 4 : bl1;                   // ... and in the stream
    const int total_stages = nst_l1 + 24 * mt.npass;     // per block
    const int all_stages = total_stages * nb;            // the stream runs straight through the block boundaries
    auto phys = [&](int s) -> int {   // stream position -> stage of the blob (GEMM 2's stages are re-streamed every sweep)
        if (s < nst_l1) return s;
        const int q = s - nst_l1, pass = q / 24, w = q - 24 * pass, j = w / 3, t = w - 3 * j;
        return t < 2 ? bl1 + 2 * j + t : bl1 + 16 + 8 * pass + j;
    };
    int stage = 0;   // global stage counter (all blocks)
    constexpr int PPW = 16 / GC_NW;  // 1 KB pieces per wave
    // Round 6: the blobs of the block in flight and of the next one are read from the level's pointer table ONCE per block and kept
    // in scalar registers.  issue() used to read lv.blob(bb) per stage: the table lives in memory the kernel may write, so the read
    // is a VECTOR load -- and the wait for it (vmcnt retires in order) drained every ring stage still in flight before the next
    // request went out: the ring never ran more than one stage ahead in the level chains.
    auto sptr = [](const float *q) -> const float * {
        const uint64_t b64 = (uint64_t)(uintptr_t)q;
        return reinterpret_cast<const float *>((uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(b64 >> 32)) << 32) |
                                                           (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)b64)));
    };
    int cur_b = 0;
    const float *blob_cur = fused ? sptr(lv.blob(0)) : blob0;
    const float *blob_nxt = (fused && nb > 1) ? sptr(lv.blob(1)) : blob_cur;
    auto issue = [&](int gs) {
        const int bb = gs / total_stages, s = gs - bb * total_stages;
        const float *stages = (bb == cur_b ? blob_cur : blob_nxt) + gc_off_stages(mt);
        static_assert(PPW == 2, "the paired GEMM 1 stages assume two pieces per wave");
        const float *src = (pair1 && s < nst_l1)
                               ? stages + (size_t)(2 * s + (wid >> 2)) * GC_STAGE + ((wid & 3) * PPW) * 256
                               : stages + (size_t)phys(s) * GC_STAGE + (wid * PPW) * 256;
        float *dst = ring + (gs % GC_RING) * GC_STAGE + (wid * PPW) * 256;
#ifdef NF_GC_BUILTIN_DMA
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            __builtin_amdgcn_global_load_lds(src + (tid0 & 63) * 4 + i * 256, (__attribute__((address_space(3))) void *)(dst + i * 256), 16, 0, 0);
#else
#pragma unroll
        for (int i = 0; i < PPW; ++i) gc_dma16(src, (uint32_t)(((tid0 & 63) * 4 + i * 256) * 4), dst + i * 256);
#endif
    };
    // GC_RING-slot ring, DMA GC_RING - 1 stages ahead: the wait leaves the younger stages' pieces in flight (vector-memory
    // operations retire in order; anything else a wave has outstanding only makes the wait stricter).  With one stage ahead
    // the L2 -> LDS latency of a stage (~2 us under load) had to fit under the 28..64 MFMAs of ONE stage: GEMM 1 ran at 2/3.
    auto acquire = [&]() -> const float * {
        if (stage + GC_RING - 2 < all_stages) NF_WAIT_VMCNT((GC_RING - 2) * PPW);
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // a raw barrier: __syncthreads() is a fence, which the compiler implements as s_waitcnt vmcnt(0) -- it waited for the
        // DMAs of ALL stages in flight at every stage and made the depth of the ring irrelevant (found in round 3)
        GC_RING_BARRIER();
        if (stage + GC_RING - 1 < all_stages) issue(stage + GC_RING - 1);   // runs on into the NEXT block's first stages
        const float *buf = ring + (stage % GC_RING) * GC_STAGE;
        ++stage;
        return buf;
    };

    // ---- prologue: first stage in flight; im2col offsets; the level's images into LDS ----
#pragma unroll
    for (int i = 0; i < GC_RING - 1; ++i)
        if (i < all_stages) issue(i);
    for (int k = tid0; k < K1p; k += 64 * GC_NW) {
        const int kk = k < mt.K1 ? k : mt.K1 - 1;   // padded k: zero weight, any valid address
        const int c = kk / 9, t = kk - 9 * c, ky = t / 3;
        koff[k] = c * PH * PW + ky * PW + (t - 3 * ky);
    }
    float *zin = pl.zA, *zalt = pl.zB;
    if (fused) {
        gl_prefetch<GC_NW>(lv, 0, mt.small, small0, pl.wmix, wid, tid0 & 63);
        gl_load<GC_PX, 64 * GC_NW>(lv, zin, pl.ldacc, H, W, img0, B, tid0);
    }

    for (int b = 0; b < nb; ++b) {
    GL_T(b, 0);
    if (b > 0) {        // (the stream has already run on into this block's first stages through blob_nxt)
        blob_cur = blob_nxt;
        cur_b = b;
        blob_nxt = (b + 1 < nb) ? sptr(lv.blob(b + 1)) : blob_cur;
    }
    const int tid = gl_opaque(tid0), lane = tid & 63, hh = lane >> 5;
    const int px = wid * 32 + (lane & 31);
    const int li = px / HW, pin = px - li * HW, py = pin / W, pxx = pin - py * W;
    const int base0 = li * mt.Cin * PH * PW + py * PW + pxx;
    const float *blob = blob_cur;
    // biases and mix of this block: prefetched (fused level; double-buffered), padded input images
    float *small = (fused && (b & 1)) ? pl.small2 : small0;
    const float *wmix = (b & 1) ? pl.wmix2 : pl.wmix;
    if (fused) {
        gl_pre<GC_PX, 64 * GC_NW, (GC_RING - 1) * (16 / GC_NW)>(lv, b, zin, zalt, wmix, xin, H, W, tid);
        // the next block's sections go out now (their buffers' last readers are behind the barriers above)
        if (b + 1 < nb) gl_prefetch<GC_NW>(lv, b + 1, mt.small, (b & 1) ? small0 : pl.small2, (b & 1) ? pl.wmix : pl.wmix2, wid, lane);
    } else {
        for (int i = tid; i < mt.small; i += 64 * GC_NW) small[i] = blob[GC_HDR + i];
        gc_fill_xin_global<GC_PX, 64 * GC_NW>(x, xs_img, xin, mt.Cin, H, W, img0, B, tid);
    }

    // ---- GEMM 1 (conv3x3 #1 by im2col): h1 = LeakyReLU(W1 col(x) + b1) ----
    f32x16 H0, H1, H2, H3, H4, H5, H6, H7;
    // the lane's im2col column (its pixel, k = 8 kg + 4 hh + r) is the B operand of all 8 row blocks: gathered once into
    // registers when K1 <= 128 (Cin <= 14), per use otherwise
    float bx[4 * GC_KG1MAX];
    const bool bx_cached = mt.nkg1 <= GC_KG1MAX;
    auto gemm1_pair = [&](f32x16 &acc0, f32x16 &acc1) {
        const float *buf = acquire();
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x16 &acc = half ? acc1 : acc0;
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                if (s < mt.nkg1) {
                    const f32x4 a = *reinterpret_cast<const f32x4 *>(buf + (8 * half + s) * 256 + lane * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc = GC_MFMA(a[r], bx[4 * s + r], acc);
                }
            }
            gc_leaky(acc, mt.slope);
        }
    };
    auto gemm1 = [&](f32x16 &acc) {
        if (bx_cached) {
            const float *buf = acquire();
#pragma unroll
            for (int s = 0; s < GC_KG1MAX; ++s) {
                if (s < mt.nkg1) {
                    const f32x4 a = *reinterpret_cast<const f32x4 *>(buf + s * 256 + lane * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc = GC_MFMA(a[r], bx[4 * s + r], acc);
                }
            }
        } else {
            for (int c = 0; c < mt.nst1; ++c) {
                const float *buf = acquire();
                const int ng = (mt.nkg1 - 16 * c) < 16 ? (mt.nkg1 - 16 * c) : 16;
                for (int s = 0; s < ng; ++s) {
                    const f32x4 a = *reinterpret_cast<const f32x4 *>(buf + s * 256 + lane * 4);
                    const int kb = 8 * (16 * c + s) + 4 * hh;
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc = GC_MFMA(a[r], xin[base0 + koff[kb + r]], acc);
                }
            }
        }
        gc_leaky(acc, mt.slope);
    };
    GL_BARRIER();   // the prologue's LDS writes
    GL_T(b, 1);
#pragma unroll
    for (int s = 0; s < GC_KG1MAX; ++s) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            bx[4 * s + r] = (bx_cached && s < mt.nkg1) ? xin[base0 + koff[8 * s + 4 * hh + r]] : 0.0f;
    }
    {
        const float *bsrc = small + hh * 16;
        H0 = load_bias16(bsrc);       H1 = load_bias16(bsrc + 32);  H2 = load_bias16(bsrc + 64);  H3 = load_bias16(bsrc + 96);
        H4 = load_bias16(bsrc + 128); H5 = load_bias16(bsrc + 160); H6 = load_bias16(bsrc + 192); H7 = load_bias16(bsrc + 224);
    }
    if constexpr (pair1) { gemm1_pair(H0, H1); gemm1_pair(H2, H3); gemm1_pair(H4, H5); gemm1_pair(H6, H7); }
    else { gemm1(H0); gemm1(H1); gemm1(H2); gemm1(H3); gemm1(H4); gemm1(H5); gemm1(H6); gemm1(H7); }
    GL_T(b, 2);

    // ---- sweeps over h2: GEMM 2 block by block, each block consumed at once by GEMM 3 ----
    for (int pass = 0; pass < mt.npass; ++pass) {
        f32x16 O0 = {0}, O1 = {0}, O2 = {0}, O3 = {0};
        for (int j = 0; j < 8; ++j) {
            f32x16 T = load_bias16(small + GC_HID + j * 32 + hh * 16);
            gc_mm128(acquire(), lane, T, H0, H1, H2, H3);
            gc_mm128(acquire(), lane, T, H4, H5, H6, H7);
            gc_leaky(T, mt.slope);
            const float *buf = acquire();
#pragma unroll
            for (int mm = 0; mm < GC_OBP; ++mm) {
                f32x16 &o = mm == 0 ? O0 : (mm == 1 ? O1 : (mm == 2 ? O2 : O3));
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const f32x4 a = *reinterpret_cast<const f32x4 *>(buf + (mm * 4 + s) * 256 + lane * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) o = GC_MFMA(a[r], T[4 * s + r], o);
                }
            }
        }
        // ---- col2im: per output block, tap products -> LDS, 9-term neighbour gather, + bias, -> HBM ----
        GL_T(b, 3);
#pragma unroll
        for (int mm = 0; mm < GC_OBP; ++mm) {
            const int blk = GC_OBP * pass + mm;
            if (blk < mt.OB) {
            const f32x16 &o = mm == 0 ? O0 : (mm == 1 ? O1 : (mm == 2 ? O2 : O3));
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) P[(8 * (reg >> 2) + 4 * hh + (reg & 3)) * GC_PX + px] = o[reg];
            GL_BARRIER();
            gc_gather_block<GC_PX, 64 * GC_NW>(P, blk, mt, H, W, img0, B, small, out, fused ? pl.prm : nullptr, tid);
            GL_BARRIER();
            }
        }
    }
    GL_T(b, 4);
    if (fused) {
        gl_post<GC_PX, 64 * GC_NW>(lv, b, zin, zalt, wmix, pl.prm, pl.ldt, pl.ldacc, H, W, tid);
        float *t_ = zin; zin = zalt; zalt = t_;
    }
    GL_T(b, 5);
    }  // blocks
    if (fused) gl_store<GC_PX, 64 * GC_NW>(lv, zin, pl.ldacc, H, W, img0, B, tid0);
}

// ---- small images: 64-pixel workgroups of 4 waves x 16 pixels on v_mfma_f32_16x16x4_f32 ------------------------------
// With B H W << 256 x 256 pixels the 256-pixel workgroups above cannot fill the chip (config 4: 8x8 images = 64
// workgroups).  Here a wave owns 16 pixels: h1 = 16 x f32x4 per lane, every output block fits the register file, so ONE
// sweep over h2 serves all output channels (OBT = capacity in 32-row output blocks).  C layout of the 16x16x4 MFMA:
// lane (g = lane >> 4, pixel = lane & 15) holds rows 4 g + r; the contraction k = 16 b + 4 g + r pairs lane group g with
// the rows it holds, as in the wide kernel.  One MFMA wave per SIMD: the weight ring is 4 stages deep to cover L2 latency.
// Next to the GS_NW MFMA waves the workgroup has GS_HW helper waves (round 3): they issue ALL of the ring's LDS-DMAs and
// prefetches (a vector-memory instruction costs the issuing wave 100-250 cycles, ~400 of the 2048 cycles of a stage when
// the MFMA waves issued them) and do half of the serial phases around the GEMMs (mix, padded images, col2im, coupling),
// which are latency-bound loops over the workgroup's threads.
#define GC_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
constexpr int GS_NW = 4;
#ifndef NF_GS_HW
#define NF_GS_HW 4
#endif
// helper waves per instantiation (0: the MFMA waves do everything, as in round 2): with helpers the workgroup has two waves
// per SIMD and 256 registers per lane, which the 16-block instantiation (64 output accumulators more) does not fit
template <int OBT> struct GsHelpers { static constexpr int value = OBT <= 8 ? NF_GS_HW : 0; };
constexpr int GS_PX = 16 * GS_NW;
constexpr int GS_RING = 4;
constexpr int GS_KB1MAX = 16;   // k-blocks of GEMM 1 whose gathered operands fit the register file (Cin <= 28)

__device__ __forceinline__ void gs_leaky(f32x4 &v, float slope) {
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = fmaxf(v[c], v[c] * slope);
}

template <int OBT>
__global__ void __launch_bounds__(64 * (GS_NW + GsHelpers<OBT>::value))
glow_convnet_small_kernel(const float *__restrict__ x, int64_t xs_img, float *__restrict__ out,
                          const float *__restrict__ blob0, GcMeta mt, int64_t B, int H, int W, GlowLevel lv) {
    constexpr int GS_HW = GsHelpers<OBT>::value;
    constexpr int GS_DW = GS_HW ? GS_HW : GS_NW;   // waves that issue DMAs
    constexpr int GS_NT = 64 * (GS_NW + GS_HW);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int HW = H * W, PH = H + 2, PW = W + 2, IPW = GS_PX / HW;
    const int nkb1 = (mt.K1 + 15) / 16, K1p = 16 * nkb1;
    float *ring = smem;                          // 4 x 16 KB weight stages
    float *P = ring + GS_RING * GC_STAGE;        // 32 rows x 64 pixels
    float *small0 = P + 32 * GS_PX;
    int *koff = reinterpret_cast<int *>(small0 + gc_small_padded(mt));
    float *xin = reinterpret_cast<float *>(koff + K1p);
    const GlPlanes pl = gl_planes(xin + IPW * mt.Cin * PH * PW, lv, mt.Cout, GS_PX);
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool mfma_wave = wid < GS_NW;                      // wave-uniform
    const bool dma_wave = GS_HW ? !mfma_wave : true;
    const int dwid = GS_HW ? wid - GS_NW : wid;              // index among the DMA waves
    const int px = (wid % GS_NW) * 16 + (lane & 15);
    const int li = px / HW, pin = px - li * HW, py = pin / W, pxx = pin - py * W;
    const int base0 = li * mt.Cin * PH * PW + py * PW + pxx;
    const int64_t img0 = (int64_t)blockIdx.x * IPW;
    const bool fused = lv.nblocks > 0;
    const int nb = fused ? lv.nblocks : 1;

    const int nst_l1 = 8 * mt.nst1, per_j = 2 + mt.npass;   // per h2 block: 2 stages of GEMM 2, npass stages of GEMM 3
    const int total_stages = nst_l1 + 8 * per_j;
    auto phys = [&](int s) -> int {
        if (s < nst_l1) return s;
        const int q = s - nst_l1, j = q / per_j, t = q - per_j * j;
        return t < 2 ? nst_l1 + 2 * j + t : nst_l1 + 16 + 8 * (t - 2) + j;
    };
    constexpr int PPW = 16 / GS_DW;
    for (int k = tid; k < K1p; k += GS_NT) {
        const int kk = k < mt.K1 ? k : mt.K1 - 1;
        const int c = kk / 9, t = kk - 9 * c, ky = t / 3;
        koff[k] = c * PH * PW + ky * PW + (t - 3 * ky);
    }
    float *zin = pl.zA, *zalt = pl.zB;
    if (fused) {
        if (dma_wave) gl_prefetch<GS_DW>(lv, 0, mt.small, small0, pl.wmix, dwid, lane);
        gl_load<GS_PX, GS_NT>(lv, zin, pl.ldacc, H, W, img0, B, tid);
    }

    for (int b = 0; b < nb; ++b) {
    GL_T(b, 0);
    const float *blob = fused ? lv.blob(b) : blob0;
    const float *stages = blob + gc_off_stages(mt);
    int stage = 0;   // per block: the ring is the col2im scratch at the end of a block, so the stream restarts
    auto issue = [&](int gs) {
        const float *src = stages + (size_t)phys(gs) * GC_STAGE + (dwid * PPW) * 256 + lane * 4;
        float *dst = ring + (gs % GS_RING) * GC_STAGE + (dwid * PPW) * 256;
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            __builtin_amdgcn_global_load_lds(src + i * 256, (__attribute__((address_space(3))) void *)(dst + i * 256), 16, 0, 0);
    };
    auto acquire = [&]() -> const float * {
        // stages stage+1 .. stage+RING-2 may stay in flight (loads retire in order); the tail drains everything
        if (dma_wave) {
            if (stage + GS_RING - 2 < total_stages) NF_WAIT_VMCNT((GS_RING - 2) * PPW);
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        // everyone's pieces of `stage` have landed; everyone is done with stage - 1 (slot reused below).  (One barrier per
        // PAIR of stages, 6 slots: measured in round 3, no gain -- the barrier count is not what is left.)
        GC_RING_BARRIER();
        if (dma_wave && stage + GS_RING - 1 < total_stages) issue(stage + GS_RING - 1);
        const float *buf = ring + (stage % GS_RING) * GC_STAGE;
        ++stage;
        return buf;
    };

    // ---- block prologue ----
    // the first stages go out FIRST (the ring is free: the previous block's col2im is behind a barrier) and land while
    // the biases, the mix and the padded images are prepared.  The ring's in-order accounting needs an empty queue in front.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (dma_wave) {
#pragma unroll
        for (int i = 0; i < GS_RING - 1; ++i)
            if (i < total_stages) issue(i);
    }
    float *small = (fused && (b & 1)) ? pl.small2 : small0;
    const float *wmix = (b & 1) ? pl.wmix2 : pl.wmix;
    if (fused) {
        gl_pre<GS_PX, GS_NT, (GS_RING - 1) * PPW>(lv, b, zin, zalt, wmix, xin, H, W, tid);
        if (dma_wave && b + 1 < nb)
            gl_prefetch<GS_DW>(lv, b + 1, mt.small, (b & 1) ? small0 : pl.small2, (b & 1) ? pl.wmix : pl.wmix2, dwid, lane);
    } else {
        for (int i = tid; i < mt.small; i += GS_NT) small[i] = blob[GC_HDR + i];
        gc_fill_xin_global<GS_PX, GS_NT>(x, xs_img, xin, mt.Cin, H, W, img0, B, tid);
    }
    GL_BARRIER();
    GL_T(b, 1);

    f32x4 O[2 * OBT];
    if (!mfma_wave) {
        for (int s = 0; s < total_stages; ++s) acquire();   // the helpers keep the ring turning
    } else {
    // ---- GEMM 1: h1 (16 blocks of 16 channels) ----
    auto lda = [&](const float *buf, int piece) { return *reinterpret_cast<const f32x4 *>(buf + piece * 256 + lane * 4); };
    f32x4 Hh[16];
#pragma unroll
    for (int b = 0; b < 16; ++b) Hh[b] = *reinterpret_cast<const f32x4 *>(small + 16 * b + 4 * g);
    if (nkb1 <= GS_KB1MAX) {
        // the lane's im2col column (its pixel, k = 16 kb + 4 g + r) is the B operand of all 16 row blocks: gathered once
        float bv[4 * GS_KB1MAX];
#pragma unroll
        for (int kb = 0; kb < GS_KB1MAX; ++kb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[4 * kb + r] = kb < nkb1 ? xin[base0 + koff[16 * kb + 4 * g + r]] : 0.0f;
        }
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) {
#pragma unroll
            for (int c = 0; c < GS_KB1MAX / 8; ++c) {
                if (c < mt.nst1) {
                    const float *buf = acquire();
                    f32x4 a0 = lda(buf, 0), a1 = lda(buf, 8);
#pragma unroll
                    for (int kb = 0; kb < 8; ++kb) {
                        // the next k-block's A operands are requested before this one's MFMAs (the compiler, left alone,
                        // reads, waits, multiplies: an LDS latency per 8 MFMAs with one MFMA wave per SIMD)
                        f32x4 n0 = a0, n1 = a1;
                        if (kb < 7) { n0 = lda(buf, kb + 1); n1 = lda(buf, 9 + kb); }
                        __builtin_amdgcn_sched_barrier(0);
                        if (8 * c + kb < nkb1) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                Hh[2 * mb] = GC_MFMA16(a0[r], bv[4 * (8 * c + kb) + r], Hh[2 * mb]);
                                Hh[2 * mb + 1] = GC_MFMA16(a1[r], bv[4 * (8 * c + kb) + r], Hh[2 * mb + 1]);
                            }
                        }
                        a0 = n0; a1 = n1;
                    }
                }
            }
            gs_leaky(Hh[2 * mb], mt.slope);
            gs_leaky(Hh[2 * mb + 1], mt.slope);
        }
    } else {
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) {
            for (int c = 0; c < mt.nst1; ++c) {
                const float *buf = acquire();
                const int nkb = (nkb1 - 8 * c) < 8 ? (nkb1 - 8 * c) : 8;
                for (int kb = 0; kb < nkb; ++kb) {
                    const int k0 = 16 * (8 * c + kb) + 4 * g;
                    float bw[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) bw[r] = xin[base0 + koff[k0 + r]];
                    const f32x4 a0 = *reinterpret_cast<const f32x4 *>(buf + kb * 256 + lane * 4);
                    const f32x4 a1 = *reinterpret_cast<const f32x4 *>(buf + (8 + kb) * 256 + lane * 4);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        Hh[2 * mb] = GC_MFMA16(a0[r], bw[r], Hh[2 * mb]);
                        Hh[2 * mb + 1] = GC_MFMA16(a1[r], bw[r], Hh[2 * mb + 1]);
                    }
                }
            }
            gs_leaky(Hh[2 * mb], mt.slope);
            gs_leaky(Hh[2 * mb + 1], mt.slope);
        }
    }

    GL_T(b, 2);
    // ---- one sweep: h2 block by block (GEMM 2), each consumed at once by GEMM 3 ----
#pragma unroll
    for (int i = 0; i < 2 * OBT; ++i) O[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < 8; ++j) {
        f32x4 T0 = *reinterpret_cast<const f32x4 *>(small + GC_HID + 32 * j + 4 * g);
        f32x4 T1 = *reinterpret_cast<const f32x4 *>(small + GC_HID + 32 * j + 16 + 4 * g);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const float *buf = acquire();
            f32x4 a0 = lda(buf, 0), a1 = lda(buf, 8);
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                f32x4 n0 = a0, n1 = a1;
                if (kb < 7) { n0 = lda(buf, kb + 1); n1 = lda(buf, 9 + kb); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    T0 = GC_MFMA16(a0[r], Hh[8 * half + kb][r], T0);
                    T1 = GC_MFMA16(a1[r], Hh[8 * half + kb][r], T1);
                }
                a0 = n0; a1 = n1;
            }
        }
        gs_leaky(T0, mt.slope);
        gs_leaky(T1, mt.slope);
#pragma unroll
        for (int q4 = 0; q4 < OBT / 4; ++q4) {
            if (q4 < mt.npass) {
            const float *buf = acquire();
            f32x4 a00 = lda(buf, 0), a01 = lda(buf, 1), a10 = lda(buf, 2), a11 = lda(buf, 3);
#pragma unroll
            for (int mm = 0; mm < 4; ++mm) {
                // units of output block mm: [ob][kb]; consecutive MFMAs alternate between the two accumulators
                f32x4 n00 = a00, n01 = a01, n10 = a10, n11 = a11;
                if (mm < 3) { n00 = lda(buf, mm * 4 + 4); n01 = lda(buf, mm * 4 + 5); n10 = lda(buf, mm * 4 + 6); n11 = lda(buf, mm * 4 + 7); }
                __builtin_amdgcn_sched_barrier(0);
                f32x4 &o0 = O[(4 * q4 + mm) * 2], &o1 = O[(4 * q4 + mm) * 2 + 1];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    o0 = GC_MFMA16(a00[r], T0[r], o0);
                    o1 = GC_MFMA16(a10[r], T0[r], o1);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    o0 = GC_MFMA16(a01[r], T1[r], o0);
                    o1 = GC_MFMA16(a11[r], T1[r], o1);
                }
                a00 = n00; a01 = n01; a10 = n10; a11 = n11;
            }
            }
        }
    }

    }  // MFMA waves
    GL_T(b, 3);
    // ---- col2im: the weight ring is dead now; when all output blocks' tap products fit its 64 KB they go there at once
    // and ONE flat gather follows (two barriers in all), otherwise block by block through P ----
    if (mt.OB * 32 * GS_PX <= GS_RING * GC_STAGE) {
        GL_BARRIER();   // every wave is done reading the last stage
#pragma unroll
        for (int blk = 0; blk < OBT; ++blk) {
            if (mfma_wave && blk < mt.OB) {
#pragma unroll
                for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                    for (int r = 0; r < 4; ++r) ring[(32 * blk + 16 * ob + 4 * g + r) * GS_PX + px] = O[2 * blk + ob][r];
            }
        }
        GL_BARRIER();
        gc_gather_block<GS_PX, GS_NT>(ring, 0, mt, H, W, img0, B, small, out, fused ? pl.prm : nullptr, tid, mt.OB);
    } else {
#pragma unroll
        for (int blk = 0; blk < OBT; ++blk) {
            if (blk < mt.OB) {
#pragma unroll
                for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (mfma_wave) P[(16 * ob + 4 * g + r) * GS_PX + px] = O[2 * blk + ob][r];
                GL_BARRIER();
                gc_gather_block<GS_PX, GS_NT>(P, blk, mt, H, W, img0, B, small, out, fused ? pl.prm : nullptr, tid);
                GL_BARRIER();
            }
        }
    }
    GL_T(b, 4);
    if (fused) {
        gl_post<GS_PX, GS_NT>(lv, b, zin, zalt, wmix, pl.prm, pl.ldt, pl.ldacc, H, W, tid);
        float *t_ = zin; zin = zalt; zalt = t_;
    } else {
        GL_BARRIER();
    }
    GL_T(b, 5);
    }  // blocks
    if (fused) gl_store<GS_PX, GS_NT>(lv, zin, pl.ldacc, H, W, img0, B, tid);
}

// ---- tiny images (H W | 16): one 16-pixel tile per workgroup, the waves split the ROWS of every GEMM -------------------
// Config 4's last level has 4096 pixels: pixel-parallel waves leave most SIMDs idle (64 workgroups of the kernel above),
// and a wave's chain of ~4000 dependent-issue MFMAs is the run time.  Here a workgroup owns ONE tile of 16 pixels (whole
// images again) and wave w computes rows [R w, R w + R), R = 256 / GT_NW, of h1 and h2 and its share of the output rows, for
// all 16 pixels.  Activations go through LDS in B-operand order acts[k / 16][(k / 4) % 4][pixel][k % 4] (one ds_read_b128
// per lane = the 4 B values of a 16-k block; one ds_write_b128 per lane stores a finished 16-row block), with a barrier
// between the GEMMs.  Every wave streams only ITS rows' weights, and each weight is used by exactly one wave of the
// workgroup: the A operands come straight from L2 into registers (global_load_dwordx4, 16 units = 16 KB per wave in
// flight; a first version went through private LDS-DMA rings and was bound by the ~25 GB/s per CU that path delivers).
// 256 workgroups x GT_NW waves fill every SIMD (GT_NW = 8: two waves per SIMD, one's barrier / LDS / load waits under the
// other's MFMAs); each workgroup streams all weights once (0.93 MB at the 4x4 level).  In a level chain the register
// prefetch runs straight on into the next block's stream.
constexpr int GT_PX = 16;
#ifndef NF_GT_NW
#define NF_GT_NW 8
#endif
constexpr int GT_NW = NF_GT_NW;
constexpr int GT_BPW = 16 / GT_NW;   // 16-row blocks of h1 / h2 per wave
constexpr int GT_PF = 16;          // units (16 rows x 16 k = [64 lanes][4] = 1 KB) a wave keeps in flight
constexpr int GT_SLOT = 1024;      // floats of 4 units (blob granularity)

struct GtMeta {
    int nkb1;    // 16-k blocks of GEMM 1, padded to a multiple of 16 (the register prefetch advances 16 units at a time)
    int NB3;     // 16-row output blocks per wave = ceil(2 OB / GT_NW)
    int slots;   // slots (4 units) per wave = (GT_BPW nkb1 + GT_BPW 16 + 16 NB3) / 4
};
__host__ __device__ inline GtMeta gt_meta(const GcMeta &m) {
    GtMeta t;
    t.nkb1 = 16 * ((m.K1 + 255) / 256);
    t.NB3 = (2 * m.OB + GT_NW - 1) / GT_NW;
    t.slots = (GT_BPW * t.nkb1 + GT_BPW * 16 + 16 * t.NB3) / 4;
    return t;
}
__host__ __device__ inline int64_t gt_total_floats(const GcMeta &m) {
    return (int64_t)gc_off_stages(m) + (int64_t)GT_NW * gt_meta(m).slots * GT_SLOT;
}

// blob: header | biases (natural order) | wave 0's slots | wave 1's | ..., each in consumption order:
// GEMM 1 (own GT_BPW row blocks x nkb1), GEMM 2 (own GT_BPW x 16), GEMM 3 (own NB3 x 16); unit (16 rows x 16 k): lane
// (i = lane & 15, g = lane >> 4) holds W[row i][16 kb + 4 g + r4].
__global__ void gt_pack_kernel(const float *__restrict__ W1, const float *__restrict__ b1, const float *__restrict__ W2,
                               const float *__restrict__ b2, const float *__restrict__ W3, const float *__restrict__ b3,
                               float *__restrict__ blob, GcMeta m, GtMeta t, int64_t total) {
    const int offs = gc_off_stages(m);
    constexpr int R = 16 * GT_BPW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        float v = 0.0f;
        if (i < GC_HDR) {
            v = i == 0 ? 357.0f : (i == 1 ? (float)m.Cin : (i == 2 ? (float)m.Cout : 0.0f));
        } else if (i < offs) {
            const int j = (int)i - GC_HDR;
            if (j < GC_HID) v = b1[j];
            else if (j < 2 * GC_HID) v = b2[j - GC_HID];
            else if (j - 2 * GC_HID < m.Cout) v = b3[j - 2 * GC_HID];
        } else {
            const int64_t e = i - offs;
            const int w = (int)(e / ((int64_t)t.slots * GT_SLOT));
            const int q0 = (int)(e % ((int64_t)t.slots * GT_SLOT));
            const int r4 = q0 & 3, lane = (q0 >> 2) & 63, ri = lane & 15, g = lane >> 4;
            int q = q0 >> 8;   // unit index in the wave's stream
            if (q < GT_BPW * t.nkb1) {
                const int bl = q / t.nkb1, kb = q - bl * t.nkb1;
                const int k = 16 * kb + 4 * g + r4;
                if (k < m.K1) v = W1[(size_t)(R * w + 16 * bl + ri) * m.K1 + k];
            } else if ((q -= GT_BPW * t.nkb1) < GT_BPW * 16) {
                const int bl = q >> 4, kb = q & 15;
                v = W2[(size_t)(R * w + 16 * bl + ri) * GC_HID + 16 * kb + 4 * g + r4];
            } else {
                q -= GT_BPW * 16;
                const int ol = q >> 4, kb = q & 15;
                const int o16 = w * t.NB3 + ol, blk = o16 >> 1, rho = 16 * (o16 & 1) + ri;
                const int cc = rho / 9, tap = rho - 9 * cc, co = 3 * blk + cc;
                if (rho < 27 && co < m.Cout) v = W3[((size_t)co * GC_HID + 16 * kb + 4 * g + r4) * 9 + tap];
            }
        }
        blob[i] = v;
    }
}

__global__ void __launch_bounds__(64 * GT_NW)
glow_convnet_tiny_kernel(const float *__restrict__ x, int64_t xs_img, float *__restrict__ out, const float *__restrict__ blob0,
                         GcMeta mt, GtMeta tm, int64_t B, int H, int W, GlowLevel lv) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int HW = H * W, PH = H + 2, PW = W + 2, IPW = GT_PX / HW;
    const int K1p = 16 * tm.nkb1;
    const int p_floats = GT_NW * tm.NB3 * 16 * GT_PX, a_floats = K1p * GT_PX + GC_HID * GT_PX;
    float *cols = smem;                                     // im2col columns in B-operand order (K1p x 16)
    float *h1s = cols + K1p * GT_PX;                        // h1 in B-operand order (256 x 16)
    float *P = cols;                                        // tap products: reuse cols + h1s once GEMM 2 is done
    float *h2s = cols + (p_floats > a_floats ? p_floats : a_floats);
    float *small0 = h2s + GC_HID * GT_PX;
    int *koff = reinterpret_cast<int *>(small0 + gc_small_padded(mt));
    float *xin = reinterpret_cast<float *>(koff + K1p);
    const GlPlanes pl = gl_planes(xin + IPW * mt.Cin * PH * PW, lv, mt.Cout, GT_PX);
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, j16 = lane & 15;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t img0 = (int64_t)blockIdx.x * IPW;
    const bool fused = lv.nblocks > 0;
    const int nb = fused ? lv.nblocks : 1;

    // the wave's weight stream: units in consumption order, block after block; the first 16 go out before anything else.
    // All stream bookkeeping is wave-uniform (scalar): pf_base = next unit to request, pf_left = units left in its block.
    const int nunits = 4 * tm.slots;
    auto stream_of = [&](int bb) -> const float * {
        return (fused ? lv.blob(bb) : blob0) + gc_off_stages(mt) + (size_t)wid * tm.slots * GT_SLOT;
    };
    const float *pf_base = stream_of(0);
    int pf_left = nunits, pf_stride = 256;
    const float *cross = nb > 1 ? stream_of(1) : pf_base + (size_t)(nunits - 1) * 256;   // where the stream goes on after this block
    bool cross_last = nb <= 1;          // ... which is the dummy tail (the last unit again and again)
    auto pf_next = [&]() -> f32x4 {
        // an explicitly GLOBAL load: the stream's base comes out of the level's pointer table, i.e. it is a generic pointer, and
        // a FLAT load counts on lgkmcnt as well -- every wait for an LDS read then waited for the whole register stream
        typedef const f32x4 __attribute__((address_space(1))) *gvec;
        const f32x4 v = *((gvec)(reinterpret_cast<const f32x4 *>(pf_base)) + lane);
        --pf_left;
        const bool cr = pf_left == 0;
        pf_base = cr ? cross : pf_base + pf_stride;
        pf_stride = (cr && cross_last) ? 0 : pf_stride;
        pf_left = cr ? (cross_last ? 0x40000000 : nunits) : pf_left;
        return v;
    };
    // (A per-workgroup rotation of the unit order, to keep the CUs of an XCD from asking the L2 for the same line at the
    // same moment, was measured: no gain -- and it made the summation order depend on the image's position in the batch.)
    f32x4 pf[GT_PF];
#pragma unroll
    for (int i = 0; i < GT_PF; ++i) pf[i] = pf_next();

    for (int k = tid; k < K1p; k += 64 * GT_NW) {
        const int kk = k < mt.K1 ? k : mt.K1 - 1;
        const int c = kk / 9, t = kk - 9 * c, ky = t / 3;
        koff[k] = c * PH * PW + ky * PW + (t - 3 * ky);
    }
    float *zin = pl.zA, *zalt = pl.zB;
    if (fused) {
        gl_prefetch<GT_NW>(lv, 0, mt.small, small0, pl.wmix, wid, lane);
        gl_load<GT_PX, 64 * GT_NW>(lv, zin, pl.ldacc, H, W, img0, B, tid);
    }

    // acc (16 rows x 16 pixels) = sum over nkb (a multiple of 16) 16-k blocks of A (the stream) x B (LDS, B-operand order);
    // every consumed unit is replaced by the load of the unit 16 ahead
    auto gemm_block = [&](const float *bsrc, int nkb, f32x4 &acc) {
        // four accumulators (one per r): a 16x16x4 MFMA has 40 cycles of dependent latency against 32 of issue, and the
        // B operand of the NEXT unit is read from LDS while this unit's MFMAs run
        f32x4 acc1 = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f}, acc3 = {0.f, 0.f, 0.f, 0.f};
        auto bload = [&](int kb) { return *reinterpret_cast<const f32x4 *>(bsrc + ((kb * 4 + g) * GT_PX + j16) * 4); };
        f32x4 bq = bload(0);
        for (int kb0 = 0; kb0 < nkb; kb0 += GT_PF) {
#pragma unroll
            for (int u = 0; u < GT_PF; ++u) {
                const f32x4 a = pf[u];
                pf[u] = pf_next();
                // B of the following unit (clamped at the block's end)
                const int kbn = kb0 + u + 1;
                const f32x4 bn = bload(kbn < nkb ? kbn : 0);
                __builtin_amdgcn_sched_barrier(0);   // keep both loads HERE: the scheduler otherwise sinks them to the group's end
                acc = GC_MFMA16(a[0], bq[0], acc);
                acc1 = GC_MFMA16(a[1], bq[1], acc1);
                acc2 = GC_MFMA16(a[2], bq[2], acc2);
                acc3 = GC_MFMA16(a[3], bq[3], acc3);
                bq = bn;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] += (acc1[r] + acc2[r]) + acc3[r];
    };

    for (int b = 0; b < nb; ++b) {
    GL_T(b, 0);
    const float *blob = fused ? lv.blob(b) : blob0;
    // where the prefetch continues when it runs off the end of THIS block's stream (it does so 16 units before the end)
    cross = (b + 1 < nb) ? stream_of(b + 1) : stream_of(b) + (size_t)(nunits - 1) * 256;
    cross_last = b + 1 >= nb;

    // ---- block prologue (shared): biases, padded images, im2col columns ----
    float *small = (fused && (b & 1)) ? pl.small2 : small0;
    const float *wmix = (b & 1) ? pl.wmix2 : pl.wmix;
    if (fused) {
        gl_pre<GT_PX, 64 * GT_NW, GT_PF>(lv, b, zin, zalt, wmix, xin, H, W, tid);
        if (b + 1 < nb) gl_prefetch<GT_NW>(lv, b + 1, mt.small, (b & 1) ? small0 : pl.small2, (b & 1) ? pl.wmix : pl.wmix2, wid, lane);
    } else {
        for (int i = tid; i < mt.small; i += 64 * GT_NW) small[i] = blob[GC_HDR + i];
        gc_fill_xin_global<GT_PX, 64 * GT_NW>(x, xs_img, xin, mt.Cin, H, W, img0, B, tid);
    }
    GL_BARRIER();
    for (int i = tid; i < K1p * GT_PX; i += 64 * GT_NW) {
        // element (kb, g', px, r) of the column block: k = 16 kb + 4 g' + r
        const int r = i & 3, px = (i >> 2) & 15, gg = (i >> 6) & 3, kb = i >> 8;
        const int li = px / HW, pin = px - li * HW, py = pin / W, pxx = pin - py * W;
        cols[i] = xin[li * mt.Cin * PH * PW + py * PW + pxx + koff[16 * kb + 4 * gg + r]];
    }
    GL_BARRIER();
    GL_T(b, 1);

    // ---- GEMM 1 -> h1s ----
    for (int bl = 0; bl < GT_BPW; ++bl) {
        const int b16 = GT_BPW * wid + bl;
        f32x4 acc = *reinterpret_cast<const f32x4 *>(small + 16 * b16 + 4 * g);
        gemm_block(cols, tm.nkb1, acc);
        gs_leaky(acc, mt.slope);
        *reinterpret_cast<f32x4 *>(h1s + ((b16 * 4 + g) * GT_PX + j16) * 4) = acc;
    }
    GL_BARRIER();
    GL_T(b, 2);
    // ---- GEMM 2 -> h2s ----
    for (int bl = 0; bl < GT_BPW; ++bl) {
        const int b16 = GT_BPW * wid + bl;
        f32x4 acc = *reinterpret_cast<const f32x4 *>(small + GC_HID + 16 * b16 + 4 * g);
        gemm_block(h1s, 16, acc);
        gs_leaky(acc, mt.slope);
        *reinterpret_cast<f32x4 *>(h2s + ((b16 * 4 + g) * GT_PX + j16) * 4) = acc;
    }
    GL_BARRIER();   // h2 complete; cols / h1s are dead: P may overwrite them
    // ---- GEMM 3 -> tap products ----
    for (int ol = 0; ol < tm.NB3; ++ol) {
        const int o16 = wid * tm.NB3 + ol;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        gemm_block(h2s, 16, acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) P[(16 * o16 + 4 * g + r) * GT_PX + j16] = acc[r];
    }
    GL_BARRIER();
    GL_T(b, 3);
    // ---- col2im over all output blocks in one flat loop ----
    gc_gather_block<GT_PX, 64 * GT_NW>(P, 0, mt, H, W, img0, B, small, out, fused ? pl.prm : nullptr, tid, mt.OB);
    GL_T(b, 4);
    if (fused) {
        gl_post<GT_PX, 64 * GT_NW>(lv, b, zin, zalt, wmix, pl.prm, pl.ldt, pl.ldacc, H, W, tid);
        float *t_ = zin; zin = zalt; zalt = t_;
    } else {
        GL_BARRIER();
    }
    GL_T(b, 5);
    }  // blocks
    if (fused) gl_store<GT_PX, 64 * GT_NW>(lv, zin, pl.ldacc, H, W, img0, B, tid);
}

static inline size_t gt_lds_bytes(const GcMeta &m, const GtMeta &t, int H, int W, const GlowLevel &fu) {
    const int IPW = GT_PX / (H * W), K1p = 16 * t.nkb1;
    const int p_floats = GT_NW * t.NB3 * 16 * GT_PX, a_floats = K1p * GT_PX + GC_HID * GT_PX;
    return ((size_t)(p_floats > a_floats ? p_floats : a_floats) + GC_HID * GT_PX +
            gc_small_padded(m) + K1p + (size_t)IPW * m.Cin * (H + 2) * (W + 2) + gb_lds_floats(fu, m.Cout, GT_PX)) *
               sizeof(float) + 16;
}

static inline size_t gs_lds_bytes(const GcMeta &m, int H, int W, const GlowLevel &fu) {
    const int IPW = GS_PX / (H * W);
    return ((size_t)GS_RING * GC_STAGE + 32 * GS_PX + gc_small_padded(m) + 16 * ((m.K1 + 15) / 16) +
            (size_t)IPW * m.Cin * (H + 2) * (W + 2) + gb_lds_floats(fu, m.Cout, GS_PX)) * sizeof(float) + 16;
}

static inline size_t gc_lds_bytes(const GcMeta &m, int H, int W, const GlowLevel &fu) {
    const int IPW = GC_PX / (H * W);
    return ((size_t)GC_RING * GC_STAGE + 32 * GC_PX + gc_small_padded(m) + 8 * m.nkg1 +
            (size_t)IPW * m.Cin * (H + 2) * (W + 2) + gb_lds_floats(fu, m.Cout, GC_PX)) * sizeof(float) + 16;
}

}  // namespace nf

using namespace nf;

static int gc_check(int Cin, int Cout, int hidden, double slope) {
    if (Cin < 1 || Cout < 1) return NF_EINVAL;
    if (!(slope >= 0.0 && slope <= 1.0)) return NF_EINVAL;
    if (hidden != GC_HID || Cin > 64 || Cout > 192) return NF_ENOTSUP;
    return NF_OK;
}

extern "C" int64_t nf_glow_convnet_pack_size(int Cin, int Cout, int hidden) {
    const int rc = gc_check(Cin, Cout, hidden, 0.0);
    if (rc) return rc;
    const GcMeta m = gc_meta(Cin, Cout, 0.0);
    const int64_t a = (int64_t)gc_off_stages(m) + (int64_t)gc_nstages_blob(m) * GC_STAGE, b = gt_total_floats(m);
    return (a > b ? a : b) * (int64_t)sizeof(float);   // one size for every layout
}

extern "C" int nf_glow_convnet_pack(void *wpack, const void *w1, const void *b1, const void *w2, const void *b2,
                                    const void *w3, const void *b3, int Cin, int Cout, int hidden, int layout,
                                    nf_stream_t stream) {
    const int rc = gc_check(Cin, Cout, hidden, 0.0);
    if (rc) return rc;
    if (layout < NF_GLOW_CONV_WIDE || layout > NF_GLOW_CONV_TINY) return NF_EINVAL;
    if (!wpack || !w1 || !b1 || !w2 || !b2 || !w3 || !b3) return NF_EFAULT;
    const GcMeta m = gc_meta(Cin, Cout, 0.0);
    if (layout == NF_GLOW_CONV_TINY) {
        const int64_t tot = gt_total_floats(m);
        hipLaunchKernelGGL(gt_pack_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, (const float *)w1, (const float *)b1,
                           (const float *)w2, (const float *)b2, (const float *)w3, (const float *)b3, (float *)wpack, m,
                           gt_meta(m), tot);
        NF_CHECK_LAUNCH();
        return NF_OK;
    }
    const int64_t total = (int64_t)gc_off_stages(m) + (int64_t)gc_nstages_blob(m) * GC_STAGE;
    hipLaunchKernelGGL(gc_pack_kernel, dim3(512), dim3(256), 0, (hipStream_t)stream, (const float *)w1, (const float *)b1,
                       (const float *)w2, (const float *)b2, (const float *)w3, (const float *)b3, (float *)wpack, m, total,
                       layout == NF_GLOW_CONV_SMALL ? 1 : 0);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

extern "C" int nf_glow_convnet_layout(int64_t B, int H, int W) {
    if (B < 0 || H < 1 || W < 1) return NF_EINVAL;
    const int64_t HW = (int64_t)H * W;
    if (HW <= GC_PX && GC_PX % HW == 0 && B * HW >= (int64_t)128 * GC_PX) return NF_GLOW_CONV_WIDE;   // >= 128 workgroups
    if (HW <= GT_PX && GT_PX % HW == 0 && B * HW < (int64_t)256 * GS_PX) return NF_GLOW_CONV_TINY;    // < 256 64-pixel workgroups
    if (HW <= GS_PX && GS_PX % HW == 0) return NF_GLOW_CONV_SMALL;
    if (HW <= GC_PX && GC_PX % HW == 0) return NF_GLOW_CONV_WIDE;
    return NF_ENOTSUP;
}

template <int OBT>
static int launch_small(const void *x, int64_t xs, void *out, const void *wpack, const GcMeta &m, int64_t B, int H, int W,
                        const GlowLevel &fu, hipStream_t st) {
    const size_t lds = gs_lds_bytes(m, H, W, fu);
    if (lds > 160 * 1024) return NF_ENOTSUP;
    static LdsOptIn opted = {};
    if (opt_in_lds(reinterpret_cast<const void *>(&glow_convnet_small_kernel<OBT>), lds, opted) != NF_OK) return NF_ENOTSUP;
    const int IPW = GS_PX / (H * W);
    const int64_t grid = (B + IPW - 1) / IPW;
    if (grid > 0x7fffffff) return NF_ERANGE;
    hipLaunchKernelGGL(glow_convnet_small_kernel<OBT>, dim3((unsigned)grid), dim3(64 * (GS_NW + GsHelpers<OBT>::value)), lds, st, (const float *)x, xs,
                       (float *)out, (const float *)wpack, m, B, H, W, fu);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

static int gc_launch(const void *x, int64_t xs, void *out, const void *wpack, const GcMeta &m, int64_t B, int H, int W,
                     int layout, const GlowLevel &fu, hipStream_t st) {
    const int PXW = layout == NF_GLOW_CONV_WIDE ? GC_PX : (layout == NF_GLOW_CONV_SMALL ? GS_PX : GT_PX);
    if (H * W > PXW || PXW % (H * W) != 0) return NF_ENOTSUP;   // whole images per workgroup
    if (layout == NF_GLOW_CONV_TINY) {
        const GtMeta t = gt_meta(m);
        const size_t lds = gt_lds_bytes(m, t, H, W, fu);
        if (lds > 160 * 1024) return NF_ENOTSUP;
        static LdsOptIn opted = {};
        if (opt_in_lds(reinterpret_cast<const void *>(&glow_convnet_tiny_kernel), lds, opted) != NF_OK) return NF_ENOTSUP;
        const int IPW = GT_PX / (H * W);
        const int64_t grid = (B + IPW - 1) / IPW;
        if (grid > 0x7fffffff) return NF_ERANGE;
        hipLaunchKernelGGL(glow_convnet_tiny_kernel, dim3((unsigned)grid), dim3(64 * GT_NW), lds, st, (const float *)x, xs,
                           (float *)out, (const float *)wpack, m, t, B, H, W, fu);
        NF_CHECK_LAUNCH();
        return NF_OK;
    }
    if (layout == NF_GLOW_CONV_SMALL) {
        if (m.OB <= 4) return launch_small<4>(x, xs, out, wpack, m, B, H, W, fu, st);
        if (m.OB <= 8) return launch_small<8>(x, xs, out, wpack, m, B, H, W, fu, st);
        if (m.OB <= 16) return launch_small<16>(x, xs, out, wpack, m, B, H, W, fu, st);
        return NF_ENOTSUP;
    }
    const size_t lds = gc_lds_bytes(m, H, W, fu);
    if (lds > 160 * 1024) return NF_ENOTSUP;
    const int IPW = GC_PX / (H * W);
    const int64_t grid = (B + IPW - 1) / IPW;
    if (grid > 0x7fffffff) return NF_ERANGE;
    if (m.nkg1 <= 8) {
        static LdsOptIn opted = {};
        if (opt_in_lds(reinterpret_cast<const void *>(&glow_convnet_kernel<true>), lds, opted) != NF_OK) return NF_ENOTSUP;
        hipLaunchKernelGGL(glow_convnet_kernel<true>, dim3((unsigned)grid), dim3(64 * GC_NW), lds, st, (const float *)x, xs,
                           (float *)out, (const float *)wpack, m, B, H, W, fu);
    } else {
        static LdsOptIn opted = {};
        if (opt_in_lds(reinterpret_cast<const void *>(&glow_convnet_kernel<false>), lds, opted) != NF_OK) return NF_ENOTSUP;
        hipLaunchKernelGGL(glow_convnet_kernel<false>, dim3((unsigned)grid), dim3(64 * GC_NW), lds, st, (const float *)x, xs,
                           (float *)out, (const float *)wpack, m, B, H, W, fu);
    }
    NF_CHECK_LAUNCH();
    return NF_OK;
}

extern "C" int nf_glow_convnet(const void *x, int64_t x_image_stride, void *out, const void *wpack, int64_t B, int Cin,
                               int H, int W, int Cout, int hidden, double leaky_slope, int layout, nf_stream_t stream) {
    const int rc = gc_check(Cin, Cout, hidden, leaky_slope);
    if (rc) return rc;
    if (layout < NF_GLOW_CONV_WIDE || layout > NF_GLOW_CONV_TINY) return NF_EINVAL;
    if (B < 0 || H < 1 || W < 1 || x_image_stride < (int64_t)Cin * H * W) return NF_EINVAL;
    if (B == 0) return NF_OK;
    if (!x || !out || !wpack) return NF_EFAULT;
    GlowLevel lv = {};
    return gc_launch(x, x_image_stride, out, wpack, gc_meta(Cin, Cout, leaky_slope), B, H, W, layout, lv, (hipStream_t)stream);
}

extern "C" int nf_glow_level(const void *in0, const void *in1, int cin0, int in_squeezed, void *out0, void *out1, int cout0,
                             int out_squeezed, void *logdet, const void *block_table, int nblocks, int64_t B, int C, int H,
                             int W, int hidden, double leaky_slope, int scale_map, int direction, int acc, int layout,
                             nf_stream_t stream) {
    if (C < 2) return NF_EINVAL;
    const int c1 = (C + 1) / 2, Cout = 2 * (C - c1);
    const int rc = gc_check(c1, Cout, hidden, leaky_slope);
    if (rc) return rc;
    if (layout < NF_GLOW_CONV_WIDE || layout > NF_GLOW_CONV_TINY) return NF_EINVAL;
    if (scale_map < NF_SCALE_EXP || scale_map > NF_SCALE_SIGMOID_INV) return NF_EINVAL;
    if ((direction != 0 && direction != 1) || acc < NF_LD_SUB || acc > NF_LD_ADD) return NF_EINVAL;
    if (B < 0 || H < 1 || W < 1) return NF_EINVAL;
    if (nblocks < 1 || nblocks > GL_MAXB) return NF_ERANGE;
    if ((in_squeezed || out_squeezed) && (C % 4) != 0) return NF_EINVAL;
    if (!in_squeezed && (cin0 < 1 || cin0 > C)) return NF_EINVAL;
    if (!out_squeezed && (cout0 < 1 || cout0 > C)) return NF_EINVAL;
    if (B == 0) return NF_OK;
    if (!in0 || !out0 || !logdet || !block_table) return NF_EFAULT;
    if (!in_squeezed && cin0 < C && !in1) return NF_EFAULT;
    if (!out_squeezed && cout0 < C && !out1) return NF_EFAULT;
    GlowLevel lv = {};
    lv.tbl = (const float *const *)block_table;
    lv.nblocks = nblocks;
    lv.C = C; lv.c1 = c1; lv.scale_map = scale_map; lv.direction = direction; lv.acc = acc;
    lv.in0 = (const float *)in0; lv.in1 = (const float *)in1; lv.cin0 = in_squeezed ? C : cin0; lv.in_sq = in_squeezed ? 1 : 0;
    lv.out0 = (float *)out0; lv.out1 = (float *)out1; lv.cout0 = out_squeezed ? C : cout0; lv.out_sq = out_squeezed ? 1 : 0;
    lv.logdet = (float *)logdet;
#ifdef NF_GL_TRACE
    lv.trace = g_gl_trace;
#endif
    return gc_launch(in0, (int64_t)C * H * W, out0, nullptr, gc_meta(c1, Cout, leaky_slope), B, H, W, layout, lv,
                     (hipStream_t)stream);
}
