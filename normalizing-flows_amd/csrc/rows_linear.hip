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
    for (int i = tid; i < nmb * 16 * 64 * 4; i += 64 * RL_NW) {
        const int r = i & 3, l = (i >> 2) & 63, s4 = (i >> 8) & 15, m = i >> 12;
        const int j = 32 * m + (l & 31), k = 4 * s4 + r + RL_KH * (l >> 5);
        float v = 0.0f;
        if (j < a.Nc && k < a.Kc) v = a.trans_w ? a.W[(int64_t)k * a.ldw + j] : a.W[(int64_t)j * a.ldw + k];
        Wl[i] = v;
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
