// rows_linear.hip -- the conditioner's residual block over batch rows on exact-fp32 MFMA as ONE launch (nf_rows_block), with
// the neighbouring element-wise work folded in: the TRAINING path's forward and input gradients of nets/resnet.py:37-50 under
// core.py:87-102 `forward_kld` + `loss.backward()`, which the reference leaves to library GEMMs plus separate bias / ReLU /
// threshold / add kernels.  (Round 2 also carried a single-panel kernel, nf_rows_linear; the library GEMM beat it on every
// measured shape -- 42-49 vs 76-133 TFLOP/s, profiles/r02_kernel_bench.json -- so round 3 removed it from the library and the
// header: what wins is keeping a block's intermediate on chip, not re-implementing one GEMM.)
#include "fused_common.hpp"

using namespace nf;

namespace nf {

// ---- a whole residual block in one launch ---------------------------------------------------------------------------------
//   out1 = mask1( M1 pre1(in) + c1 ),   out2 = in + mask2( M2 pre2(out1) + c2 )          (H <= 128 columns everywhere)
// forward  (resnet.py:37-50): in = x,  M1 = W1, c1 = b1, pre1 = pre2 = ReLU, M2 = W2, c2 = b2:   out1 = t (the saved
//           pre-activation), out2 = x + W2 relu(t) + b2;
// backward (core.py:87-102 `loss.backward()` through the block): in = gy, M1 = W2^T, mask1 = (t > 0): out1 = gt (the
//           cotangent of t, kept for the weight gradients), M2 = W1^T, mask2 = (x > 0): out2 = gx = gy + (gt W1) (x > 0).
// Both panels stay in LDS (2 x 64 KB, one 8-wave workgroup per CU, two waves per SIMD); out1 never returns from HBM: its C
// registers ARE the B operand of the second product -- lane-half hh of a wave holds units 32 m + 8 q + 4 hh + r, and the
// second panel's LDS image is ordered so that MFMA step 16 m + 4 q + r contracts over exactly that unit (the trick of
// rqs_fused.hip).  Per row: one read of `in`, the mask sources, one write of out1 and out2.
constexpr int RB_NW = 8;

#ifdef NF_RB_TRACE
static unsigned long long *g_rb_trace = nullptr;
extern "C" void nf_rows_block_debug_trace(void *buf) { g_rb_trace = (unsigned long long *)buf; }
#define RB_T(i) do { if (a.trace && blockIdx.x == 0 && threadIdx.x == 0) a.trace[i] = wall_clock64(); } while (0)
#else
#define RB_T(i) do {} while (0)
#endif

struct RowsBlockArgs {
#ifdef NF_RB_TRACE
    unsigned long long *trace;
#endif
    const float *in; int64_t ldi;
    const float *M1; int64_t ldw1; int trans1;
    const float *c1;
    const float *m1; int64_t ldm1;
    float *out1; int64_t ldo1;
    const float *M2; int64_t ldw2; int trans2;
    const float *c2;
    const float *m2; int64_t ldm2;
    float *out2; int64_t ldo2;
    int64_t B;
    int H, relu1, relu2;
};

// Panel (H x H, zero padded to 128 x 128) -> LDS image [m][s4][lane][4]: element = Mp[32 m + (lane & 31)][k(s4, r, lane >> 5)],
// PERM = false: k = 4 s4 + r + 64 hk;  PERM = true: k = 32 (s4 >> 2) + 8 (s4 & 3) + 4 hk + r (the C-register order).
template <int NT>
struct RbPanelRegs { f32x4 v[128 / (NT / 32)]; };

template <int NT>
__device__ __forceinline__ void rb_fetch_panel(const float *__restrict__ W, int64_t ldw, int H, RbPanelRegs<NT> &pr, int tid) {
    const int c = tid & 31, g = tid / 32;       // chunk of 4 along the contiguous axis; NT / 32 lines per pass
    constexpr int LPP = NT / 32, NPASS = 128 / LPP;
    const bool vec_ok = ((ldw & 3) == 0) && (((uintptr_t)W & 15) == 0);
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int line = LPP * ps + g;
        pr.v[ps] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (line < H && 4 * c < H) {
            const float *src = W + (int64_t)line * ldw + 4 * c;
            if (4 * c + 3 < H && vec_ok) pr.v[ps] = *reinterpret_cast<const f32x4 *>(src);
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * c + r < H) pr.v[ps][r] = src[r];
            }
        }
    }
}

template <bool PERM, int NT>
__device__ __forceinline__ void rb_commit_panel(const RbPanelRegs<NT> &pr, int trans, float *Wl, int tid) {
    const int c = tid & 31, g = tid / 32;
    constexpr int LPP = NT / 32, NPASS = 128 / LPP;
    auto slot = [](int j, int k) -> int {       // LDS float index of Mp[j][k]
        int s4, hk, r;
        if (!PERM) { hk = k >> 6; s4 = (k & 63) >> 2; r = k & 3; }
        else { s4 = 4 * (k >> 5) + ((k & 31) >> 3); hk = (k >> 2) & 1; r = k & 3; }
        return (((j >> 5) * 16 + s4) * 64 + (j & 31) + 32 * hk) * 4 + r;
    };
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int line = LPP * ps + g;
        if (!trans) {       // line = output j, the chunk holds k = 4 c .. 4 c + 3 (one aligned group: one 16-byte store)
            *reinterpret_cast<f32x4 *>(Wl + slot(line, 4 * c)) = pr.v[ps];
        } else {            // line = k, the chunk holds j = 4 c .. 4 c + 3
#pragma unroll
            for (int e = 0; e < 4; ++e) Wl[slot(4 * c + e, line)] = pr.v[ps][e];
        }
    }
}

__global__ void __launch_bounds__(64 * RB_NW, 2)
rows_block_kernel(RowsBlockArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem_rb[];
    float *W1l = smem_rb, *W2l = smem_rb + 4 * 16 * 64 * 4;
    const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5;
    RB_T(0);
    // Both panels -> LDS, each thread's 16 loads in flight before its first LDS write.  (Deferring the second panel's
    // commit behind the first product was measured: the 32 registers it pins cost what the shorter prologue gains.)
    {
        RbPanelRegs<64 * RB_NW> p1r, p2r;
        rb_fetch_panel<64 * RB_NW>(a.M1, a.ldw1, a.H, p1r, tid);
        rb_fetch_panel<64 * RB_NW>(a.M2, a.ldw2, a.H, p2r, tid);
        rb_commit_panel<false, 64 * RB_NW>(p1r, a.trans1, W1l, tid);
        rb_commit_panel<true, 64 * RB_NW>(p2r, a.trans2, W2l, tid);
    }
    __syncthreads();
    RB_T(1);
    const int nmb = (a.H + 31) >> 5;
    const int64_t ntiles = (a.B + 32 * RB_NW - 1) / (32 * RB_NW);
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row = (tile * RB_NW + (tid >> 6)) * 32 + (lane & 31);
        const bool rv = row < a.B;
        const int64_t rc = rv ? row : a.B - 1;
#ifdef NF_RB_ABL_COALESCED   // ablation (wrong results): every global access of the tile as consecutive 16-byte pieces per lane
        const int64_t row0 = (tile * RB_NW + (tid >> 6)) * 32;
#define RB_ADDR(base, ld, c0, idx) ((base) + row0 * (ld) + ((idx) * 64 + lane) * 4)
#else
#define RB_ADDR(base, ld, c0, idx) ((base) + rc * (ld) + (c0))
#endif
        float xv[64];
        {
            const float *src = RB_ADDR(a.in, a.ldi, 64 * hh, 0);
#ifdef NF_RB_ABL_COALESCED
#define RB_XOFF(q) ((q) * 256)
#else
#define RB_XOFF(q) (4 * (q))
#endif
            const int kleft = a.H - 64 * hh;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (4 * q + 3 < kleft) v = *reinterpret_cast<const f32x4 *>(src + RB_XOFF(q));
#pragma unroll
                for (int r = 0; r < 4; ++r) xv[4 * q + r] = a.relu1 ? fmaxf(v[r], 0.0f) : v[r];
            }
        }
        // ---- first product; its C registers become the second product's B operand ----
        f32x16 T[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            T[m] = f32x16{0};
            if (m < nmb) {
                // the epilogue's inputs are requested BEFORE the MFMAs (clamped row, valid columns): one HBM round trip per
                // 32-row block was exposed when they were loaded where they are used
                f32x4 mk[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c0 = 32 * m + 8 * q + 4 * hh;
                    mk[q] = f32x4{1.f, 1.f, 1.f, 1.f};
                    if (a.m1 && c0 < a.H) mk[q] = *reinterpret_cast<const f32x4 *>(RB_ADDR(a.m1, a.ldm1, c0, m * 4 + q));
                }
                const float *wl = W1l + (size_t)m * 16 * 256 + lane * 4;
#pragma unroll
                for (int s4 = 0; s4 < 16; ++s4) {
                    const f32x4 w = *reinterpret_cast<const f32x4 *>(wl + s4 * 256);
#pragma unroll
                    for (int r = 0; r < 4; ++r) T[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[r], xv[4 * s4 + r], T[m], 0, 0, 0);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c0 = 32 * m + 8 * q + 4 * hh;     // H % 4 == 0: a run is inside the layer or outside
                    f32x4 v = {T[m][4 * q], T[m][4 * q + 1], T[m][4 * q + 2], T[m][4 * q + 3]};
                    if (c0 < a.H) {
                        if (a.c1) v += *reinterpret_cast<const f32x4 *>(a.c1 + c0);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = mk[q][r] > 0.0f ? v[r] : 0.0f;
                        if (rv) *reinterpret_cast<f32x4 *>(RB_ADDR(a.out1, a.ldo1, c0, m * 4 + q)) = v;
                    } else {
                        v = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) T[m][4 * q + r] = a.relu2 ? fmaxf(v[r], 0.0f) : v[r];
                }
            }
        }
        RB_T(3);
        // ---- second product + residual ----
        for (int mo = 0; mo < nmb; ++mo) {
            f32x4 mk[4], rs[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c0 = 32 * mo + 8 * q + 4 * hh;
                mk[q] = f32x4{1.f, 1.f, 1.f, 1.f};
                rs[q] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (c0 < a.H) {
                    if (a.m2) mk[q] = *reinterpret_cast<const f32x4 *>(RB_ADDR(a.m2, a.ldm2, c0, mo * 4 + q));
                    rs[q] = *reinterpret_cast<const f32x4 *>(RB_ADDR(a.in, a.ldi, c0, mo * 4 + q));      // the residual: `in` itself (cache-hot)
                }
            }
            f32x16 o = {0};
            const float *wl = W2l + (size_t)mo * 16 * 256 + lane * 4;
#pragma unroll
            for (int s4 = 0; s4 < 16; ++s4) {
                const f32x4 w = *reinterpret_cast<const f32x4 *>(wl + s4 * 256);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    o = __builtin_amdgcn_mfma_f32_32x32x2f32(w[r], T[s4 >> 2][4 * (s4 & 3) + r], o, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c0 = 32 * mo + 8 * q + 4 * hh;
                if (!rv || c0 >= a.H) continue;
                f32x4 v = {o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]};
                if (a.c2) v += *reinterpret_cast<const f32x4 *>(a.c2 + c0);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = mk[q][r] > 0.0f ? v[r] : 0.0f;
                v += rs[q];
                *reinterpret_cast<f32x4 *>(RB_ADDR(a.out2, a.ldo2, c0, mo * 4 + q)) = v;
            }
        }
        RB_T(4);
    }
}

}  // namespace nf

extern "C" int nf_rows_block(const void *in, int64_t ldi, const void *M1, int64_t ldw1, int trans1, const void *c1,
                             const void *mask1, int64_t ldm1, void *out1, int64_t ldo1, const void *M2, int64_t ldw2,
                             int trans2, const void *c2, const void *mask2, int64_t ldm2, void *out2, int64_t ldo2, int64_t B,
                             int H, int relu1, int relu2, nf_stream_t stream) {
    using namespace nf;
    if (B < 0 || H < 4 || (H & 3) || ldi < H || ldo1 < H || ldo2 < H) return NF_EINVAL;
    if (H > 128) return NF_ENOTSUP;
    if (B == 0) return NF_OK;
    if (!in || !M1 || !M2 || !out1 || !out2) return NF_EFAULT;
    if ((mask1 && ldm1 < H) || (mask2 && ldm2 < H)) return NF_EINVAL;
    if ((ldi | ldo1 | ldo2 | (mask1 ? ldm1 : 0) | (mask2 ? ldm2 : 0)) & 3) return NF_EINVAL;
    if (((uintptr_t)in | (uintptr_t)out1 | (uintptr_t)out2 | (uintptr_t)c1 | (uintptr_t)c2 | (uintptr_t)mask1 |
         (uintptr_t)mask2) & 15) return NF_EINVAL;
    RowsBlockArgs a;
    a.in = (const float *)in; a.ldi = ldi;
    a.M1 = (const float *)M1; a.ldw1 = ldw1; a.trans1 = trans1 ? 1 : 0;
    a.c1 = (const float *)c1; a.m1 = (const float *)mask1; a.ldm1 = ldm1; a.out1 = (float *)out1; a.ldo1 = ldo1;
    a.M2 = (const float *)M2; a.ldw2 = ldw2; a.trans2 = trans2 ? 1 : 0;
    a.c2 = (const float *)c2; a.m2 = (const float *)mask2; a.ldm2 = ldm2; a.out2 = (float *)out2; a.ldo2 = ldo2;
    a.B = B; a.H = H; a.relu1 = relu1 ? 1 : 0; a.relu2 = relu2 ? 1 : 0;
#ifdef NF_RB_TRACE
    a.trace = g_rb_trace;
#endif
    const size_t lds = (size_t)2 * 4 * 16 * 64 * 4 * sizeof(float);   // two 64 KB panels
    static LdsOptIn opted = {};
    if (opt_in_lds(reinterpret_cast<const void *>(&rows_block_kernel), lds, opted) != NF_OK) return NF_ENOTSUP;
    const int64_t ntiles = (B + 32 * RB_NW - 1) / (32 * RB_NW);
    const int grid = (int)(ntiles < 256 ? ntiles : 256);
    hipLaunchKernelGGL(rows_block_kernel, dim3(grid), dim3(64 * RB_NW), lds, (hipStream_t)stream, a);
    NF_CHECK_LAUNCH();
    return NF_OK;
}
