// rows_linear.hip -- one panel of a Linear layer over batch rows on exact-fp32 MFMA, with the neighbouring element-wise
// work folded in: the conditioner's GEMMs of the TRAINING path (forward and input gradients of nets/resnet.py:37-50,
// :92-104 under core.py:87-102 `forward_kld` + `loss.backward()`), which the reference (and round 1 of this repository)
// leaves to library GEMMs plus separate bias / ReLU / threshold / add kernels.
//
//   out[b, j] (op)= post( sum_k pre(x[b, k]) Wp[j][k] + bias[j] ),   j < Nc <= 128, k < Kc <= 128
//     pre  = ReLU on load (the block's activation in front of a linear, resnet.py:41-47)            [relu_in]
//     Wp   = W[n0 + j][k0 + k]  or, transposed, W[k0 + k][n0 + j]  (input gradient: gx = gy W)      [trans_w]
//     post = multiply by (mask_src[b, j] > 0)  (ReLU backward through the saved pre-activation)      [mask_src]
//            then + residual[b, j]             (the residual connection, forward and backward)       [residual]
//     op   = store, or add to what `out` holds (K split over several launches: out += panel)        [accumulate]
// Wider layers (the 736-row final layer) are covered by the caller launching one panel per 128 columns / 128 k.
//
// Mapping: a wave owns 32 rows; Out^T = Wp X^T with v_mfma_f32_32x32x2_f32, lane-half hh contracts over columns
// [KH hh, KH hh + KH) of its own row (KH = 64: sixteen 16-byte loads per lane, no LDS round trip for activations); the
// panel sits in LDS in A-operand order for the whole launch (64 KB; two workgroups per CU = two waves per SIMD cover each
// other's loads); a lane's C registers are 16-byte runs of its output row, so bias / mask / residual / out are 16-byte
// accesses at the same offsets.  Workgroups loop over 128-row tiles.  HBM-bound: ~(Kc + Nc (1 + extras)) 4 B per row.
#include "fused_common.hpp"

namespace nf {

constexpr int RL_NW = 4;          // waves per workgroup (128 rows per tile)
constexpr int RL_KH = 64;         // contraction columns per lane-half (Kc <= 128)

struct RowsLinearArgs {
    const float *x; int64_t ldx;            // (B, >= Kc) row pitch ldx
    const float *W; int64_t ldw;            // panel origin already applied; row pitch ldw
    const float *bias;                      // (Nc) or NULL
    const float *mask_src; int64_t ldm;     // (B, >= Nc) or NULL
    const float *residual; int64_t ldr;     // (B, >= Nc) or NULL
    float *out; int64_t ldo;                // (B, >= Nc)
    int64_t B;
    int Kc, Nc, relu_in, trans_w, accumulate;
};

__global__ void __launch_bounds__(64 * RL_NW, 2)
rows_linear_kernel(RowsLinearArgs a) {
    // Wl[m][s4][lane][4] = Wp[32 m + (lane & 31)][4 s4 + r + KH (lane >> 5)]   (4 x 16 x 64 x 4 floats = 64 KB)
    extern __shared__ __attribute__((aligned(16))) float Wl[];
    const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5;
    const int nmb = (a.Nc + 31) >> 5;       // 32-row output blocks in use
    // Panel -> LDS with 16-byte global loads along W's contiguous axis, all 16 of a thread in flight before the first LDS
    // write (a load -> store loop costs one L2 round trip per iteration: 60 us for 64 KB; this takes ~3).
    for (int i = tid; i < nmb * 16 * 64 * 4; i += 64 * RL_NW) Wl[i] = 0.0f;
    __syncthreads();
    {
        const int c = tid & 31, g = tid >> 5;   // chunk of 4 along the contiguous axis; 8 lines per pass
        f32x4 v[16];
        const int nlines = a.trans_w ? a.Kc : a.Nc, ncont = a.trans_w ? a.Nc : a.Kc;
        const bool vec_ok = ((a.ldw & 3) == 0) && (((uintptr_t)a.W & 15) == 0);
#pragma unroll
        for (int ps = 0; ps < 16; ++ps) {
            const int line = 8 * ps + g;
            v[ps] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (line < nlines && 4 * c < ncont) {
                const float *src = a.W + (int64_t)line * a.ldw + 4 * c;
                if (4 * c + 3 < ncont && vec_ok) v[ps] = *reinterpret_cast<const f32x4 *>(src);
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (4 * c + r < ncont) v[ps][r] = src[r];
                }
            }
        }
#pragma unroll
        for (int ps = 0; ps < 16; ++ps) {
            const int line = 8 * ps + g;
            if (!a.trans_w) {       // line = output j, the chunk holds k = 4 c .. 4 c + 3
                const int j = line, m = j >> 5, hk = c >> 4, s4 = c & 15;
                if (m < nmb) *reinterpret_cast<f32x4 *>(Wl + ((size_t)(m * 16 + s4) * 64 + (j & 31) + 32 * hk) * 4) = v[ps];
            } else {                // line = k, the chunk holds j = 4 c .. 4 c + 3
                const int k = line, hk = k >> 6, s4 = (k & 63) >> 2, r = k & 3;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int j = 4 * c + e, m = j >> 5;
                    if (m < nmb) Wl[((size_t)(m * 16 + s4) * 64 + (j & 31) + 32 * hk) * 4 + r] = v[ps][e];
                }
            }
        }
    }
    __syncthreads();
    const int64_t ntiles = (a.B + 32 * RL_NW - 1) / (32 * RL_NW);
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row = (tile * RL_NW + (tid >> 6)) * 32 + (lane & 31);
        const bool rv = row < a.B;
        const int64_t rc = rv ? row : a.B - 1;             // clamped: loads stay unconditional
        // ---- B operand: the lane's half row ----
        float xv[RL_KH];
        const float *src = a.x + rc * a.ldx + RL_KH * hh;
        const int kleft = a.Kc - RL_KH * hh;               // columns of this half that exist
#pragma unroll
        for (int q = 0; q < RL_KH / 4; ++q) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (4 * q + 3 < kleft) v = *reinterpret_cast<const f32x4 *>(src + 4 * q);
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * q + r < kleft) v[r] = src[4 * q + r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) xv[4 * q + r] = a.relu_in ? fmaxf(v[r], 0.0f) : v[r];
        }
        // ---- Out^T = Wp X^T, 32 output rows at a time ----
        for (int m = 0; m < nmb; ++m) {
            f32x16 o = {0};
            const float *wl = Wl + (size_t)m * 16 * 256 + lane * 4;
#pragma unroll
            for (int s4 = 0; s4 < 16; ++s4) {
                const f32x4 w = *reinterpret_cast<const f32x4 *>(wl + s4 * 256);
#pragma unroll
                for (int r = 0; r < 4; ++r) o = __builtin_amdgcn_mfma_f32_32x32x2f32(w[r], xv[4 * s4 + r], o, 0, 0, 0);
            }
            // ---- epilogue: C register 4 q + r = output column 32 m + 8 q + 4 hh + r of the lane's row ----
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c0 = 32 * m + 8 * q + 4 * hh;
                if (!rv || c0 >= a.Nc) continue;
                float *po = a.out + row * a.ldo + c0;
                if (c0 + 3 < a.Nc) {
                    f32x4 v = {o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]};
                    if (a.bias) v += *reinterpret_cast<const f32x4 *>(a.bias + c0);
                    if (a.mask_src) {
                        const f32x4 mk = *reinterpret_cast<const f32x4 *>(a.mask_src + row * a.ldm + c0);
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = mk[r] > 0.0f ? v[r] : 0.0f;
                    }
                    if (a.residual) v += *reinterpret_cast<const f32x4 *>(a.residual + row * a.ldr + c0);
                    if (a.accumulate) v += *reinterpret_cast<const f32x4 *>(po);
                    *reinterpret_cast<f32x4 *>(po) = v;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (c0 + r >= a.Nc) continue;
                        float v = o[4 * q + r];
                        if (a.bias) v += a.bias[c0 + r];
                        if (a.mask_src) v = a.mask_src[row * a.ldm + c0 + r] > 0.0f ? v : 0.0f;
                        if (a.residual) v += a.residual[row * a.ldr + c0 + r];
                        if (a.accumulate) v += po[r];
                        po[r] = v;
                    }
                }
            }
        }
    }
}

}  // namespace nf

using namespace nf;

extern "C" int nf_rows_linear(const void *x, int64_t ldx, const void *W, int64_t ldw, int trans_w, const void *bias,
                              const void *mask_src, int64_t ldm, const void *residual, int64_t ldr, void *out, int64_t ldo,
                              int64_t B, int Kc, int Nc, int relu_in, int accumulate, nf_stream_t stream) {
    if (B < 0 || Kc < 1 || Nc < 1 || ldx < Kc || ldo < Nc) return NF_EINVAL;
    if (Kc > 2 * RL_KH || Nc > 128) return NF_ENOTSUP;
    if (B == 0) return NF_OK;
    if (!x || !W || !out) return NF_EFAULT;
    if ((mask_src && ldm < Nc) || (residual && ldr < Nc)) return NF_EINVAL;
    // 16-byte accesses: row pitches and origins must keep 4-float alignment
    if ((ldx | ldo | (mask_src ? ldm : 0) | (residual ? ldr : 0)) & 3) return NF_EINVAL;
    if (((uintptr_t)x | (uintptr_t)out | (uintptr_t)bias | (uintptr_t)mask_src | (uintptr_t)residual) & 15) return NF_EINVAL;
    RowsLinearArgs a;
    a.x = (const float *)x; a.ldx = ldx;
    a.W = (const float *)W; a.ldw = ldw;
    a.bias = (const float *)bias;
    a.mask_src = (const float *)mask_src; a.ldm = ldm;
    a.residual = (const float *)residual; a.ldr = ldr;
    a.out = (float *)out; a.ldo = ldo;
    a.B = B; a.Kc = Kc; a.Nc = Nc; a.relu_in = relu_in ? 1 : 0; a.trans_w = trans_w ? 1 : 0; a.accumulate = accumulate ? 1 : 0;
    const size_t lds = (size_t)((Nc + 31) / 32) * 16 * 64 * 4 * sizeof(float);
    static LdsOptIn opted = {};
    if (opt_in_lds(reinterpret_cast<const void *>(&rows_linear_kernel), 64 * 1024, opted) != NF_OK) return NF_ENOTSUP;
    const int64_t ntiles = (B + 32 * RL_NW - 1) / (32 * RL_NW);
    const int grid = (int)(ntiles < 512 ? ntiles : 512);      // two workgroups per CU, each keeps its panel for all its tiles
    hipLaunchKernelGGL(rows_linear_kernel, dim3(grid), dim3(64 * RL_NW), lds, (hipStream_t)stream, a);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

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
