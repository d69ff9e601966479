// rqs_fused.hip -- one NSF coupling layer (CoupledRationalQuadraticSpline, optionally followed/preceded by its
// LULinearPermute) as ONE kernel: ResidualNet conditioner on fp32 MFMA + rational-quadratic spline epilogue.
//
// Reference behaviour: normflows/flows/neural_spline/wrapper.py:79-85, nsf/coupling.py:71-128, :150-164, :221-253,
// :329-362, nets/resnet.py:37-50, :92-104, utils/splines.py:16-219; mixing.py:535-563 for the fused LU layer.
//
// Shape handled by this kernel (the benchmark shape; everything else takes the unfused path):
//   D = 64 features, alternating mask (32 identity / 32 transform), hidden = 128, K = 8 bins, linear tails,
//   ReLU ResidualNet with any number of blocks, no context, fp32.
//
// Work decomposition
//   workgroup = 4 waves, wave = 32 samples (MFMA N = 32).  All GEMMs are evaluated transposed,
//       Out^T[out, sample] = W[out, k] * Act^T[k, sample],
//   with v_mfma_f32_32x32x2_f32: A operand = weights (one VGPR), B operand = activations (one VGPR), C = 16 VGPRs
//   holding, for the lane's sample (lane & 31), the output rows (reg&3) + 8 (reg>>2) + 4 (lane>>5) of a 32-row block.
//   Because the contraction order is free, step t of the next layer lets lane-half hh contract over exactly the 64
//   hidden units whose values that half already holds in its C registers: activations never leave the register
//   file between layers (no LDS round trip, no cross-lane traffic); ReLU and bias are VALU ops on C registers.
//   The final layer's 736 rows are re-ordered (and padded to 768) at pack time so that after three 32-row blocks a
//   lane holds the complete 23-parameter sets of two (sample, feature) spline elements: the conditioner output is
//   never materialised (2944 B/sample/layer of HBM traffic in the unfused path -> 0).
//
// Weight stream
//   The layer's weights are packed once (nf_rqs_fused_pack) into MFMA A-operand order: 16 KB "stages", each a
//   32-row block x 128-k panel stored as [16 k-groups][64 lanes][4 floats], so that a wave's ds_read_b128 /
//   global_load_lds accesses are lane-linear (conflict-free, perfectly coalesced).  Stages are streamed through a
//   2-slot LDS ring with global_load_lds (DMA, no VGPR staging): while the 4 waves run the 64 MFMAs of stage s,
//   stage s+1 lands.  One barrier per stage (per 64 MFMAs = 4096 SIMD cycles).
//
// HBM traffic per sample: 256 B in + 256 B out + 8 B log-det (the 0.66 MB of weights per layer are read from L2).
// Arithmetic: 2 * 167 936 MAC per sample (768-row padded final layer) at the fp32 MFMA rate.
#include "fused_common.hpp"

// Rows the training variants write for the backward (386 MB per launch) are not read again by this launch: non-temporal stores
// (measured 29.7-29.8 vs 29.9-30.0 ms per training step, three alternating runs; -DNF_TRAIN_TEMPORAL_STORES: ordinary stores).
#ifndef NF_TRAIN_TEMPORAL_STORES
#define NF_TRAIN_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define NF_TRAIN_STORE(ptr, val) (*(ptr) = (val))
#endif

// This file is compiled twice: as it stands (8 waves = 256 rows per workgroup, every entry point) and through rqs_fused_nw4.hip
// (-DNF_FUSED_SECONDARY: 4 waves = 128 rows per workgroup, only the inference kernel under another name and its chain dispatch) for
// batches that leave the 8-wave workgroups more than half of the chip idle (round 6, late: nf_rqs_fused_chain).
#ifdef NF_FUSED_SECONDARY
#define rqs_fused_kernel rqs_fused_kernel_nw4
#endif

namespace nf {

#ifndef NF_FUSED_SECONDARY
// ---- pack kernels ---------------------------------------------------------------------------------------------
// A-operand image of one 32-row block: dst[s][lane][r4] = W[row(lane & 31)][kcol(s, lane >> 5, r4)]
__global__ void pack_init_kernel(const float *__restrict__ W /*128x32*/, const float *__restrict__ b, float *__restrict__ stage,
                                 float *__restrict__ bias_dst) {
    // stage 0: 4 row-blocks x (4 k-groups x 64 lanes x 4)
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < F_STAGE; i += gridDim.x * blockDim.x) {
        const int r4 = i & 3, lane = (i >> 2) & 63, s = (i >> 8) & 3, m = i >> 10;
        const int row = 32 * m + (lane & 31);
        const int k = 8 * s + 4 * (lane >> 5) + r4;  // identity-feature index
        stage[i] = W[row * F_NI + k];
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 128; i += gridDim.x * blockDim.x) {
        const int reg = i & 15, hh = (i >> 4) & 1, m = i >> 5;
        bias_dst[i] = b[32 * m + 8 * (reg >> 2) + 4 * hh + (reg & 3)];
    }
}

__global__ void pack_hidden_kernel(const float *__restrict__ W /*128x128*/, const float *__restrict__ b,
                                   float *__restrict__ stages /*4 stages*/, float *__restrict__ bias_dst) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 4 * F_STAGE; i += gridDim.x * blockDim.x) {
        const int r4 = i & 3, lane = (i >> 2) & 63, s = (i >> 8) & 15, m = i >> 12;
        const int row = 32 * m + (lane & 31);
        const int k = 8 * s + 4 * (lane >> 5) + r4;
        stages[i] = W[row * F_H + k];
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 128; i += gridDim.x * blockDim.x) {
        const int reg = i & 15, hh = (i >> 4) & 1, m = i >> 5;
        bias_dst[i] = b[32 * m + 8 * (reg >> 2) + 4 * hh + (reg & 3)];
    }
}

// Rows holding unnormalised widths / heights are pre-multiplied by wh_scale = log2(e) / sqrt(hidden): the division of
// nsf/coupling.py:334-339 and the exp -> exp2 conversion of the softmax cost nothing in the epilogue.
template <int KB>
__global__ void pack_final_kernel(const float *__restrict__ W /* 32 (3 KB - 1) x 128 */, const float *__restrict__ b,
                                  float *__restrict__ stages /* 3 KB stages */, float *__restrict__ bias_dst,
                                  float wh_scale) {
    constexpr int M = 3 * KB - 1;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 3 * KB * F_STAGE; i += gridDim.x * blockDim.x) {
        const int r4 = i & 3, lane = (i >> 2) & 63, s = (i >> 8) & 15, st = i >> 12;
        const int g = st / 3, rb = st % 3;
        const int row = final_row_k<KB>(g, rb, lane & 31);
        const int k = 8 * s + 4 * (lane >> 5) + r4;
        const float sc = (row >= 0 && (row % M) < 2 * KB) ? wh_scale : 1.0f;
        stages[i] = row >= 0 ? W[row * F_H + k] * sc : 0.0f;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 96 * KB; i += gridDim.x * blockDim.x) {
        const int reg = i & 15, hh = (i >> 4) & 1, st = i >> 5;
        const int g = st / 3, rb = st % 3;
        const int row = final_row_k<KB>(g, rb, 8 * (reg >> 2) + 4 * hh + (reg & 3));
        const float sc = (row >= 0 && (row % M) < 2 * KB) ? wh_scale : 1.0f;
        bias_dst[i] = row >= 0 ? b[row] * sc : 0.0f;
    }
}

__global__ void pack_tables_kernel(const float *__restrict__ uw, const float *__restrict__ uh, const float *__restrict__ ud,
                                   float *__restrict__ tab, RqsParams<float> p) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= F_NI) return;
    const int K = p.K;
    const float *wj = uw + j * K, *hj = uh + j * K, *dj = ud + j * (K - 1);
    auto wacc = [=](int k) { return wj[k]; };
    auto hacc = [=](int k) { return hj[k]; };
    auto dacc = [=](int k) { return dj[k]; };
    rqs_build_table<float>(p, wacc, hacc, dacc, tab + j * 3 * (K + 1));
}

__global__ void pack_header_kernel(float *__restrict__ hdr, int nblk) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        hdr[0] = 355.0f;
        hdr[1] = (float)nblk;
        hdr[2] = 0.0f;
        hdr[3] = 0.0f;
    }
}

// Header + final-layer stages + knot tables in ONE launch: what nf_rqs_fused_pack_final writes once per training step and
// layer (three launches before).  The last workgroup also does the two single-wave jobs.
__global__ void pack_final_all_kernel(const float *__restrict__ W, const float *__restrict__ b, float *__restrict__ stages,
                                      float *__restrict__ bias_dst, float wh_scale, float *__restrict__ hdr, int nblk,
                                      const float *__restrict__ uw, const float *__restrict__ uh,
                                      const float *__restrict__ ud, float *__restrict__ tab, RqsParams<float> p) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 24 * F_STAGE; i += gridDim.x * blockDim.x) {
        const int r4 = i & 3, lane = (i >> 2) & 63, s = (i >> 8) & 15, st = i >> 12;
        const int g = st / 3, rb = st % 3;
        const int row = final_row(g, rb, lane & 31);
        const int k = 8 * s + 4 * (lane >> 5) + r4;
        const float sc = (row >= 0 && (row % F_M) < 2 * F_K) ? wh_scale : 1.0f;
        stages[i] = row >= 0 ? W[row * F_H + k] * sc : 0.0f;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 768; i += gridDim.x * blockDim.x) {
        const int reg = i & 15, hh = (i >> 4) & 1, st = i >> 5;
        const int g = st / 3, rb = st % 3;
        const int row = final_row(g, rb, 8 * (reg >> 2) + 4 * hh + (reg & 3));
        const float sc = (row >= 0 && (row % F_M) < 2 * F_K) ? wh_scale : 1.0f;
        bias_dst[i] = row >= 0 ? b[row] * sc : 0.0f;
    }
    if (blockIdx.x != gridDim.x - 1) return;
    const int j = threadIdx.x;
    if (j == 0) {
        hdr[0] = 355.0f;
        hdr[1] = (float)nblk;
        hdr[2] = 0.0f;
        hdr[3] = 0.0f;
    }
    if (j < F_NI) {
        const float *wj = uw + j * F_K, *hj = uh + j * F_K, *dj = ud + j * (F_K - 1);
        auto wacc = [=](int k) { return wj[k]; };
        auto hacc = [=](int k) { return hj[k]; };
        auto dacc = [=](int k) { return dj[k]; };
        rqs_build_table<float>(p, wacc, hacc, dacc, tab + j * F_TABW);
    }
}


// LULinearPermute as ONE dense 64 x 64 matrix per direction (mixing.py:402-473, :535-563), composed in fp64:
//   density: y = L (U x[perm]) + b            -> W_d[i][perm[j]] = (L U)[i][j],            bias_d = b
//   sample : y[perm[j]] = (U^-1 L^-1 (x - b))_j -> W_s[perm[j]][k] = (U^-1 L^-1)[j][k],   bias_s = -W_s b
// Single workgroup; four 64 x 64 fp64 matrices in LDS.
__global__ void __launch_bounds__(256)
pack_lu_kernel(const int64_t *__restrict__ perm, const float *__restrict__ lower_entries,
               const float *__restrict__ upper_entries, const float *__restrict__ udiag_raw,
               const float *__restrict__ bias, float eps, float *__restrict__ stage_d, float *__restrict__ stage_s,
               float *__restrict__ bias_d, float *__restrict__ bias_s, float *__restrict__ hdr) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lu_raw[];
    constexpr int D = F_D, N = F_D * F_D;
    double *A = reinterpret_cast<double *>(lu_raw), *Bm = A + N, *Cm = Bm + N, *Dm = Cm + N;
    int *inv = reinterpret_cast<int *>(Dm + N);
    double *bs = reinterpret_cast<double *>(inv + D);  // sample bias, by output column
    __shared__ float sred[16];
    const int tid = threadIdx.x;
    for (int i = tid; i < N; i += 256) {
        const int r = i / D, c = i - r * D;
        double l = 0.0, u = 0.0;
        if (c < r) l = (double)lower_entries[r * (r - 1) / 2 + c];
        else if (c == r) { l = 1.0; u = (double)(softplus(udiag_raw[r]) + eps); }
        else u = (double)upper_entries[r * (D - 1) - r * (r - 1) / 2 + (c - r - 1)];
        A[i] = l;
        Bm[i] = u;
    }
    for (int i = tid; i < D; i += 256) inv[(int)perm[i]] = i;
    float part = 0.0f;
    for (int i = tid; i < D; i += 256) part += logf(softplus(udiag_raw[i]) + eps);  // mixing.py:514-532
    const float lad = block_sum(part, sred);
    if (tid == 0) { hdr[2] = 1.0f; hdr[3] = lad; }
    __syncthreads();
    // ---- density: C = L U ----
    for (int i = tid; i < N; i += 256) {
        const int r = i / D, c = i - r * D;
        double a = 0.0;
        for (int k = 0; k <= (r < c ? r : c); ++k) a += A[r * D + k] * Bm[k * D + c];
        Cm[i] = a;
    }
    __syncthreads();
    for (int i = tid; i < F_STAGE; i += 256) {
        const int r4 = i & 3, lane = (i >> 2) & 63, sg = (i >> 8) & 7, m = i >> 11;
        const int row = lu_out_col(m, lane & 31), k = lu_in_col(sg, lane >> 5, r4);
        stage_d[i] = (float)Cm[row * D + inv[k]];  // W_d[row][k] = (LU)[row][perm^-1[k]]
    }
    for (int i = tid; i < 64; i += 256) {
        const int reg = i & 15, hh = (i >> 4) & 1, m = i >> 5;
        bias_d[i] = bias[lu_out_col(m, 8 * (reg >> 2) + 4 * hh + (reg & 3))];
    }
    __syncthreads();
    // ---- sample: C = L^-1, D = U^-1 (column solves in fp64), A = D C ----
    for (int i = tid; i < N; i += 256) { Cm[i] = 0.0; Dm[i] = 0.0; }
    __syncthreads();
    for (int c = tid; c < D; c += 256) {
        for (int r = c; r < D; ++r) {
            double a = (r == c) ? 1.0 : 0.0;
            for (int k = c; k < r; ++k) a -= A[r * D + k] * Cm[k * D + c];
            Cm[r * D + c] = a;  // unit diagonal
        }
        for (int r = c; r >= 0; --r) {
            double a = (r == c) ? 1.0 : 0.0;
            for (int k = r + 1; k <= c; ++k) a -= Bm[r * D + k] * Dm[k * D + c];
            Dm[r * D + c] = a / Bm[r * D + r];
        }
    }
    __syncthreads();
    for (int i = tid; i < N; i += 256) {
        const int r = i / D, c = i - r * D;
        double a = 0.0;
        for (int k = (r > c ? r : c); k < D; ++k) a += Dm[r * D + k] * Cm[k * D + c];  // U^-1 upper, L^-1 lower
        A[i] = a;
    }
    __syncthreads();
    for (int j = tid; j < D; j += 256) {  // bias_s[perm[j]] = -sum_k (U^-1 L^-1)[j][k] b[k]
        double a = 0.0;
        for (int k = 0; k < D; ++k) a += A[j * D + k] * (double)bias[k];
        bs[(int)perm[j]] = -a;
    }
    __syncthreads();
    for (int i = tid; i < F_STAGE; i += 256) {
        const int r4 = i & 3, lane = (i >> 2) & 63, sg = (i >> 8) & 7, m = i >> 11;
        const int row = lu_out_col(m, lane & 31), k = lu_in_col(sg, lane >> 5, r4);
        stage_s[i] = (float)A[inv[row] * D + k];  // W_s[row][k] = (U^-1 L^-1)[perm^-1[row]][k]
    }
    for (int i = tid; i < 64; i += 256) {
        const int reg = i & 15, hh = (i >> 4) & 1, m = i >> 5;
        bias_s[i] = (float)bs[lu_out_col(m, 8 * (reg >> 2) + 4 * hh + (reg & 3))];
    }
}

// Training (round 6): the density-direction half of pack_lu_kernel for n layers in ONE launch (blockIdx.x = layer), once per step:
// the composed matrix W_d = (L U) with permuted columns as the LU stage of each layer's TRAINING blob (the whole-layer forward
// rqs_fused_kernel<0, true, 2> runs the LU in front of the coupling like inference does), its bias, the constant log|det| in the
// header -- and W_d row-major (64 x 64) for the backward (nf_lu_bwd_composed: gx = g W_d, dW_d = g^T x).
// table: n rows of 7 device pointers: perm, lower_entries, upper_entries, unconstrained_upper_diag, bias, blob, wd_out.
__global__ void __launch_bounds__(256)
pack_lu_train_multi_kernel(const void *const *__restrict__ table, float eps, int nblk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lu_raw[];
    constexpr int D = F_D, N = F_D * F_D;
    double *A = reinterpret_cast<double *>(lu_raw), *Bm = A + N, *Cm = Bm + N;
    int *inv = reinterpret_cast<int *>(Cm + N);
    __shared__ float sred[16];
    const void *const *row = table + (size_t)blockIdx.x * 7;
    const int64_t *perm = (const int64_t *)row[0];
    const float *lower_entries = (const float *)row[1], *upper_entries = (const float *)row[2], *udiag_raw = (const float *)row[3],
                *bias = (const float *)row[4];
    float *blob = (float *)row[5], *wd = (float *)row[6];
    FusedLayout lay;
    lay.nblk = nblk;
    float *stage_d = blob + lay.off_stages() + (size_t)lay.lu_stage(0) * F_STAGE, *bias_d = blob + F_HDR + lay.off_bias_lu(0);
    const int tid = threadIdx.x;
    for (int i = tid; i < N; i += 256) {
        const int r = i / D, c = i - r * D;
        double l = 0.0, u = 0.0;
        if (c < r) l = (double)lower_entries[r * (r - 1) / 2 + c];
        else if (c == r) { l = 1.0; u = (double)(softplus(udiag_raw[r]) + eps); }
        else u = (double)upper_entries[r * (D - 1) - r * (r - 1) / 2 + (c - r - 1)];
        A[i] = l;
        Bm[i] = u;
    }
    for (int i = tid; i < D; i += 256) inv[(int)perm[i]] = i;
    float part = 0.0f;
    for (int i = tid; i < D; i += 256) part += logf(softplus(udiag_raw[i]) + eps);  // mixing.py:514-532
    const float lad = block_sum(part, sred);
    __syncthreads();
    for (int i = tid; i < N; i += 256) {       // C = L U (fp64)
        const int r = i / D, c = i - r * D;
        double a = 0.0;
        for (int k = 0; k <= (r < c ? r : c); ++k) a += A[r * D + k] * Bm[k * D + c];
        Cm[i] = a;
    }
    __syncthreads();
    for (int i = tid; i < F_STAGE; i += 256) {
        const int r4 = i & 3, lane = (i >> 2) & 63, sg = (i >> 8) & 7, m = i >> 11;
        const int rr = lu_out_col(m, lane & 31), k = lu_in_col(sg, lane >> 5, r4);
        stage_d[i] = (float)Cm[rr * D + inv[k]];  // W_d[row][k] = (LU)[row][perm^-1[k]]
    }
    for (int i = tid; i < N; i += 256) {
        const int r = i / D, k = i - r * D;
        wd[i] = (float)Cm[r * D + inv[k]];
    }
    for (int i = tid; i < 64; i += 256) {
        const int reg = i & 15, hh = (i >> 4) & 1, m = i >> 5;
        bias_d[i] = bias[lu_out_col(m, 8 * (reg >> 2) + 4 * hh + (reg & 3))];
    }
    __syncthreads();                              // (orders this block's header words behind pack_all's: same stream, earlier launch)
    if (tid == 0) { blob[2] = 1.0f; blob[3] = lad; }
}
#endif  // !NF_FUSED_SECONDARY

// ---- the fused layer kernel -----------------------------------------------------------------------------------
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// Ask the scheduler to lay a region out as 64 x { 1 MFMA, NF_PIPE_VALU VALU/transcendental instructions }: the
// default (bottom-up) list scheduler clusters the epilogue at ~15 VALU per MFMA behind the last third of the region's
// MFMAs, which overflows the 64-cycle MFMA shadow when both waves of a SIMD are in that phase.
#ifndef NF_PIPE_VALU
#define NF_PIPE_VALU 5
#endif
#ifndef NF_USE_SCHED_PIPE  /* measured: no gain over the default schedule (tools/fused_ablate.py) */
#define NF_SCHED_PIPE() do {} while (0)
#else
#define NF_SCHED_PIPE()                                                         \
    do {                                                                        \
        _Pragma("unroll") for (int i_ = 0; i_ < 64; ++i_) {                     \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                  \
            __builtin_amdgcn_sched_group_barrier(0x402, NF_PIPE_VALU, 0);       \
        }                                                                       \
    } while (0)
#endif

// acc += Wblock(32 x 128) * B, where the B operand for (k-group s, r) is `bsrc[s >> 2][4 (s & 3) + r]`
// (optionally through ReLU): 16 ds_read_b128 + 64 MFMA.
// HB: hidden row-blocks in use (4 = 128 units; 2 / 1 = layers of <= 64 / <= 32 hidden units packed into the same 128-unit
// blob: the units beyond are zero rows / columns, so their k-groups and row-blocks are skipped -- exactly, not approximately).
template <bool RELU, int HB = 4, int LA = 2>
__device__ __forceinline__ void mm128(const float *buf, int lane, f32x16 &acc, const f32x16 &b0, const f32x16 &b1,
                                      const f32x16 &b2, const f32x16 &b3) {
#ifdef NF_EXP_SETPRIO
    __builtin_amdgcn_s_setprio(1);
#endif
#ifndef NF_NO_PREFETCH
    // A operands two k-groups ahead of their use (real since round 6: see NF_DMA16)
    // LA = 1: one k-group ahead (the instantiations that sit at the 256-register limit: 8 registers fewer)
    f32x4 a0 = *reinterpret_cast<const f32x4 *>(buf + lane * 4), a1 = a0;
    if (LA == 2) a1 = *reinterpret_cast<const f32x4 *>(buf + 256 + lane * 4);
#pragma unroll
    for (int s = 0; s < 4 * HB; ++s) {
        const f32x4 a = a0;
        if (LA == 2) {
            a0 = a1;
            if (s + 2 < 4 * HB) a1 = *reinterpret_cast<const f32x4 *>(buf + (s + 2) * 256 + lane * 4);
        } else if (s + 1 < 4 * HB) {
            a0 = *reinterpret_cast<const f32x4 *>(buf + (s + 1) * 256 + lane * 4);
        }
        __builtin_amdgcn_sched_barrier(0);   // keeps the request in FRONT of the k-group's MFMAs (without it: +0.5 % lost again)
#else
#pragma unroll
    for (int s = 0; s < 4 * HB; ++s) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(buf + s * 256 + lane * 4);
#endif
        const f32x16 &bs = (s >> 2) == 0 ? b0 : ((s >> 2) == 1 ? b1 : ((s >> 2) == 2 ? b2 : b3));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float bv = bs[4 * (s & 3) + r];
#ifndef NF_ABL_NORELU
            if (RELU) bv = fmaxf(bv, 0.0f);
#endif
            acc = MFMA(a[r], bv, acc);
        }
    }
#ifdef NF_EXP_SETPRIO
    __builtin_amdgcn_s_setprio(0);
#endif
}

// 2 row-blocks x 8 k-groups of the composed 64 x 64 LU matrix: out[slot 16 m + reg] = bias + sum_k W[.][k] xin[k-slot]
__device__ __forceinline__ void lu_mm(const float *buf, int lane, const float (&xin)[32], f32x16 &o0, f32x16 &o1) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        f32x16 &o = m == 0 ? o0 : o1;
#pragma unroll
        for (int sg = 0; sg < 8; ++sg) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(buf + (m * 8 + sg) * 256 + lane * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) o = MFMA(a[r], xin[4 * sg + r], o);
        }
    }
}

// Waves per workgroup (each wave = 32 samples).  8 waves share one weight stream: half the LDS-DMA instructions and
// barriers per wave of the 4-wave layout (measured: DMA issue costs 9 us of a 188 us launch at 4 waves).
#ifndef NF_FUSED_WAVES
#define NF_FUSED_WAVES 8
#endif
constexpr int F_NW = NF_FUSED_WAVES;
#ifndef NF_DMA_WAVES
#define NF_DMA_WAVES 4
#endif
constexpr int F_DMA_WAVES = NF_DMA_WAVES < F_NW ? NF_DMA_WAVES : F_NW;   // waves that issue the weight ring's LDS-DMAs
constexpr int F_THREADS = 64 * F_NW;
constexpr int F_ROWS = 32 * F_NW;

// A chain of up to F_MAX_LAYERS fused layers handled by ONE persistent launch: the workgroup keeps its 256 rows in the
// LDS stash across all layers (x is read from HBM once, y written once, the log-det lives in a register), the weight
// stream runs straight through the layer boundaries and the next layer's small section is DMA-prefetched.
struct FlowArgs {
    const float *blob[F_MAX_LAYERS];  // packed blobs in PROCESSING order
    unsigned long long parity;        // bit l: mask parity of layer l (0: transform features on odd columns)
    int nlayers;
    int ngroups;                      // final-layer groups in use (<= K): a layer of D <= 16 c live columns has transform features only
                                      // in the first c 16-column chunks = the first c K / 4 groups; the padding columns sit outside
                                      // the splines' interval (identity, log-det 0), so their groups are neither streamed nor computed
};

// DIR: 0 = density (wrapper.inverse), 1 = sample (wrapper.forward).  LU: fuse each layer's LULinearPermute
// (density: LULinearPermute.inverse BEFORE the coupling; sample: LULinearPermute.forward AFTER it).
// TRAIN (density direction, no LU, one layer): the training forward of the layer's LAST stage -- the hidden activations h2
// (B x 128, the output of the residual blocks, computed by the autograd-tracked trunk) come from HBM instead of the
// initial layer / blocks, the final layer + spline run as in inference, and the conditioner output is written for the
// backward in the lane's own order: cond_out[row][transform feature][24] (23 parameters + 1 pad, raw scale), six 16-byte
// stores per feature.  Replaces a library GEMM that materialises 193 MB plus the stand-alone spline kernel that reads them back.
// TRAIN = 2 with LU (round 6): the layer's LULinearPermute.inverse runs in the same launch in front of the coupling, as in inference
// (one dense 64 x 64 product, composed per step by nf_lu_pack_train_multi); its output -- the coupling's input, which the backward
// needs (spline, initial layer's weight gradient) -- is written to xlu_out (B x 64) from the LDS stash.
// TRAIN = 2: the WHOLE conditioner + transform as in inference (initial layer, residual blocks, final layer, spline) and, for
// the backward, every intermediate written on the way: act_out[t] (B x 128), t = 0: h0 (initial layer's output), then per
// block its pre-activation t and its output h (2 nblk + 1 tensors), rows through the wave's LDS transpose tile; cond_out as
// TRAIN = 1.  One launch replaces the library GEMM of the initial layer, the residual-block launches and the TRAIN = 1 launch.
template <int DIR, bool LU, int TRAIN = 0, int KB = F_K, int HB = 4>
__global__ void __launch_bounds__(F_THREADS, 2)
rqs_fused_kernel(const float *__restrict__ x, float *__restrict__ y, float *__restrict__ logdet, FlowArgs fa,
                 int64_t B, int nblk, RqsParams<float> p, int acc, const float *__restrict__ h_in = nullptr,
                 float *__restrict__ cond_out = nullptr, float unscale = 1.0f, float *__restrict__ act_out = nullptr,
                 float *__restrict__ xlu_out = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // operand look-ahead of the products (mm128): two k-groups where the registers allow it
    constexpr int LA = (HB == 4 && !(DIR == 0 && LU) && TRAIN == 0) ? 1 : 2;
    static_assert(KB == 4 || KB == 8 || KB == 16, "bins");
    static_assert(!TRAIN || (KB == F_K && DIR == 0 && (!LU || TRAIN == 2)),
                  "the training variants: 8 bins, density direction; the fused LU only with the whole-layer forward (TRAIN = 2)");
    static_assert(HB == 4 || ((HB == 2 || HB == 1) && !TRAIN), "hidden row-blocks");
    constexpr int MP = 3 * KB, FPL = 16 / KB, GQ = KB / 4, TABW = 3 * (KB + 1);   // slots per feature, features per lane-half and
                                                                                 // group, groups per 16-column chunk, table row
    FusedLayout lay;
    lay.nblk = nblk;
    lay.K = KB;
    float *ring = smem;                       // 2 x 4096
    float *stash = ring + 2 * F_STAGE;        // F_NW waves x 32 x 64
    float *small2 = stash + F_NW * 32 * 64;   // 2 x small_padded: biases + tables of the current / next layer
    const int small_pitch = lay.small_padded();
    const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: LDS-DMA bases are wave-uniform
    const int64_t row = (int64_t)blockIdx.x * F_ROWS + wid * 32 + (lane & 31);
    const bool valid = row < B;
    // logical stages of a layer: init | 2 nblk x HB hidden row-blocks | 3 KB final | (LU)
    const int nhid = 2 * nblk * HB;
    const int ngrp = TRAIN ? KB : fa.ngroups;
    const int nbase = 1 + nhid + 3 * ngrp;
    const int nstages = TRAIN == 1 ? 24 : nbase + (LU ? 1 : 0);
    const int total_stages = nstages * fa.nlayers;
    // base stage (without the LU) -> stage of the blob, which always holds four row-blocks per hidden Linear
    auto phys_base = [&](int b) -> int {
        if (HB == 4 || b == 0) return b;
        if (b <= nhid) return 1 + 4 * ((b - 1) / HB) + (b - 1) % HB;
        return 1 + 8 * nblk + (b - 1 - nhid);
    };
    // logical -> physical stage: the LU stage comes first in the density direction, last in the sample direction
    auto phys = [&](int s) -> int {
        if (TRAIN == 1) return 1 + 8 * nblk + s;      // the 24 final-layer stages only
        if (!LU) return phys_base(s);
        if (DIR == 0) return s == 0 ? lay.lu_stage(0) : phys_base(s - 1);
        return s < nbase ? phys_base(s) : lay.lu_stage(1);
    };
    float *st = stash + wid * 2048 + lane;  // value c (0..31) of this lane at st[c * 64]; c = 8 Q + column-in-chunk

    // ---- weight-stream helpers (2-slot ring, global -> LDS DMA); the stage counter runs across layers ----
    int stage = 0;
    auto issue = [&](int gs) {
        // only the first F_DMA_WAVES waves request (one per SIMD: while it sits in the vector-memory issue its partner on the SIMD
        // keeps the MFMAs going; all eight requesting: +0.6 % on the benchmark chain)
        constexpr int PPW = 16 / F_DMA_WAVES;  // 1 KB pieces per requesting wave (16 per stage)
        if (wid >= F_DMA_WAVES) return;
        const int layer = gs / nstages, s = gs - layer * nstages;
        const float *src = fa.blob[layer] + lay.off_stages() + (size_t)phys(s) * F_STAGE + (wid * PPW) * 256 + lane * 4;
        float *dst = ring + (gs & 1) * F_STAGE + (wid * PPW) * 256;
#ifdef NF_BUILTIN_DMA
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            __builtin_amdgcn_global_load_lds(src + i * 256, (__attribute__((address_space(3))) void *)(dst + i * 256), 16, 0, 0);
#else
        // round 6: requests as inline asm (common.hpp NF_DMA16) so that the products' operand look-ahead survives compilation
#pragma unroll
        for (int i = 0; i < PPW; ++i) NF_DMA16(src - lane * 4 + i * 256, lane * 16, dst + i * 256);
#endif
    };
    auto issue_small = [&](int layer) {  // small section of `layer` -> LDS buffer layer & 1 (1 KB pieces round-robin)
        const float *src = fa.blob[layer] + F_HDR;
        float *dst = small2 + (layer & 1) * small_pitch;
#ifdef NF_BUILTIN_DMA
        for (int piece = wid; piece * 256 < small_pitch; piece += F_NW)
            __builtin_amdgcn_global_load_lds(src + piece * 256 + lane * 4,
                                             (__attribute__((address_space(3))) void *)(dst + piece * 256), 16, 0, 0);
#else
        for (int piece = wid; piece * 256 < small_pitch; piece += F_NW) NF_DMA16(src + piece * 256, lane * 16, dst + piece * 256);
#endif
    };
    // `after`: vector-memory operations (training stores) this wave issued AFTER the requests of the stage it now waits for;
    // they retire in order, so they may stay in flight.  Only trusted for full tiles (every store instruction of the wave
    // has active lanes, so the count is exact).  The barrier is the raw one: __syncthreads() is a fence and the compiler
    // implements it as vmcnt(0) while an LDS-DMA may be pending -- it would wait for the stores after all.
    const bool full_tile = TRAIN ? (int64_t)(blockIdx.x + 1) * F_ROWS <= B : false;     // (inference: dead, no register)
    auto acquire = [&](int after = 0) -> const float * {
        if (TRAIN && after == 14 && full_tile) NF_WAIT_VMCNT(14);
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifndef NF_ABL_NOBAR
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
#ifndef NF_ABL_NODMA
        if (stage + 1 < total_stages) issue(stage + 1);
#endif
        const float *buf = ring + (stage & 1) * F_STAGE;
        ++stage;
        return buf;
    };

    // ---- prologue: first stage and first small section in flight, x rows -> LDS stash ----
    issue(0);
    issue_small(0);
#pragma unroll
    for (int Q = 0; Q < 4; ++Q) {
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

    for (int layer = 0; layer < fa.nlayers; ++layer) {
    const float *small = small2 + (layer & 1) * small_pitch;
    const int par_t = ((fa.parity >> layer) & 1ull) ? 0 : 1;  // column parity of the transform features
    const int par_i = par_t ^ 1;
    const float lu_lad = LU ? fa.blob[layer][3] : 0.0f;       // constant log|det| of the layer's LU (header word 3)
    if (layer > 0 || !LU || DIR == 1) {
        // the small section must have landed before its first use; in the LU-first (density) order of layer 0 the
        // acquire() below does it.  (Later layers' sections were issued a whole layer ago.)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (LU && DIR == 0) {
        // LULinearPermute.inverse (mixing.py:560-563) as one dense 64 x 64 product on MFMA; C register `reg` of
        // row-block m is the new value of slot 16 m + reg
        float xin[32];
#pragma unroll
        for (int c = 0; c < 32; ++c) xin[c] = st[c * 64];
        const float *buf = acquire();  // also publishes this layer's small section
        const float *bsrc = small + lay.off_bias_lu(0) + hh * 16;
        f32x16 o0 = load_bias16(bsrc), o1 = load_bias16(bsrc + 32);
        lu_mm(buf, lane, xin, o0, o1);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            st[c * 64] = o0[c];
            st[(16 + c) * 64] = o1[c];
        }
        if (hh == 0) ld += lu_lad;  // once per sample (the two lane halves are summed at the end)
    }
    // prefetch the NEXT layer's small section into the other buffer (its previous user, layer - 1, is done: every
    // wave has passed a barrier of this layer)
    if (layer + 1 < fa.nlayers) issue_small(layer + 1);

    // ---- unconditional spline on the identity half (nsf/coupling.py:88-92 density / :112-116 sample) ----
    // sample: CDF^-1 first, its output feeds the conditioner.  density: the conditioner sees the raw values; the CDF
    // itself is evaluated later in the shadow of the final layer's MFMAs (uncond_pair below).
    float bx[16];
    {
        const float *tabs = small + lay.off_tables();
#pragma unroll
        for (int Q = 0; Q < 4; ++Q) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 8 * Q + 2 * r + par_i;
                const int f = 8 * Q + 4 * hh + r;  // identity feature index
                const float xi = st[c * 64];
                if (DIR == 1) {
                    float yi, l;
                    rqs_table_fast<true, KB>(p, xi, tabs + f * TABW, yi, l);
                    st[c * 64] = yi;
                    ld += l;
                    bx[4 * Q + r] = yi;
                } else {
                    bx[4 * Q + r] = xi;
                }
            }
        }
        // sample direction: the 16 log-det terms must be FINISHED here.  Nothing reads `ld` before the end of the layer, so
        // the compiler otherwise sinks the two logs of every element below the conditioner and carries their operands
        // (4 values x 16 elements) across all the MFMA phases: 75 spilled VGPRs, 304 B of scratch per lane.
        if (DIR == 1) asm volatile("" : "+v"(ld));
    }

    // ---- initial layer: H = W0 xi + b0 (K = 32: 4 row-blocks x 4 k-groups in ONE stage) ----
    f32x16 H0, H1, H2, H3;
    // TRAIN = 2: a (B x 128) row tensor from the wave's C-layout registers (register 4 q + r of block m = unit 32 m + 8 q + 4 hh + r
    // of the lane's row), one 32-unit row-block at a time through the transpose tile (row pitch 36 floats: conflict-free
    // enough), written back as 128-byte row pieces: 8 lanes per row, 8 rows per instruction
    auto store_act = [&](int t, const f32x16 &V0, const f32x16 &V1, const f32x16 &V2, const f32x16 &V3) {
        if constexpr (TRAIN == 2) {
#ifdef NF_ABL_NOACT
            return;
#endif
            int l_ = lane;      // per-lane addresses derived afresh at every call: hoisted out of the kernel they cost ~30 VGPRs
            asm volatile("" : "+v"(l_));
            float *tw = small2 + small_pitch + wid * 1536;
            float *twr = tw + (l_ & 31) * 36 + 4 * (l_ >> 5);
            const int rl = l_ >> 3, cl = l_ & 7;
            const float *tww = tw + rl * 36 + 4 * cl;
            const int64_t row0 = (int64_t)blockIdx.x * F_ROWS + wid * 32 + rl;
            float *dst = act_out + ((size_t)t * (size_t)B + row0) * F_H + 4 * cl;
#ifdef NF_TRAIN_ACT_DIRECT      // ablation: 16-byte pieces straight from the C-layout registers (32 bytes per row and instruction)
            {
                const int64_t rowd = (int64_t)blockIdx.x * F_ROWS + wid * 32 + (l_ & 31);
                float *dd = act_out + ((size_t)t * (size_t)B + rowd) * F_H + 4 * (l_ >> 5);
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const f32x16 &V = m == 0 ? V0 : (m == 1 ? V1 : (m == 2 ? V2 : V3));
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (rowd < B)
                            NF_TRAIN_STORE(reinterpret_cast<f32x4 *>(dd + 32 * m + 8 * q), (f32x4{V[4 * q], V[4 * q + 1], V[4 * q + 2], V[4 * q + 3]}));
                }
                return;
            }
#endif
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const f32x16 &V = m == 0 ? V0 : (m == 1 ? V1 : (m == 2 ? V2 : V3));
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<f32x4 *>(twr + 8 * q) = f32x4{V[4 * q], V[4 * q + 1], V[4 * q + 2], V[4 * q + 3]};
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    if (row0 + 8 * it < B)
                        NF_TRAIN_STORE(reinterpret_cast<f32x4 *>(dst + (size_t)(8 * it) * F_H + 32 * m), *reinterpret_cast<const f32x4 *>(tww + 8 * it * 36));
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
    };
    // TRAIN = 2 with the fused LU: the coupling's input rows (the LU's output, in the stash since the LU stage) for the backward;
    // issued behind a stage's barrier + DMA request like store_act (the stash rows are not modified before the final layer's groups)
    auto store_xlu = [&]() {
        if constexpr (TRAIN == 2 && LU) {
            if (valid) {
#pragma unroll
                for (int Q = 0; Q < 4; ++Q) {
                    f32x4 a, b;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        a[c] = st[(8 * Q + c) * 64];
                        b[c] = st[(8 * Q + 4 + c) * 64];
                    }
                    float *dst = xlu_out + row * F_D + 16 * Q + 8 * hh;
                    NF_TRAIN_STORE(reinterpret_cast<f32x4 *>(dst), a);
                    NF_TRAIN_STORE(reinterpret_cast<f32x4 *>(dst + 4), b);
                }
            }
        }
    };
    if constexpr (TRAIN == 1) {
        // h2 from HBM in C-register order: register 4 q + r of block m = unit 32 m + 8 q + 4 hh + r of the lane's row
        const float *hr = h_in + (valid ? row : 0) * F_H + 4 * hh;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 a0 = *reinterpret_cast<const f32x4 *>(hr + 8 * q), a1 = *reinterpret_cast<const f32x4 *>(hr + 32 + 8 * q),
                        a2 = *reinterpret_cast<const f32x4 *>(hr + 64 + 8 * q), a3 = *reinterpret_cast<const f32x4 *>(hr + 96 + 8 * q);
#pragma unroll
            for (int r = 0; r < 4; ++r) { H0[4 * q + r] = a0[r]; H1[4 * q + r] = a1[r]; H2[4 * q + r] = a2[r]; H3[4 * q + r] = a3[r]; }
        }
    } else {
    {
        const float *bsrc = small + lay.off_bias_init() + hh * 16;
        H0 = load_bias16(bsrc);
        H1 = load_bias16(bsrc + 32);
        H2 = load_bias16(bsrc + 64);
        H3 = load_bias16(bsrc + 96);
        const float *buf = acquire();
#pragma unroll
        for (int m = 0; m < HB; ++m) {
            f32x16 &acc_m = m == 0 ? H0 : (m == 1 ? H1 : (m == 2 ? H2 : H3));
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const f32x4 a = *reinterpret_cast<const f32x4 *>(buf + (m * 4 + s) * 256 + lane * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) acc_m = MFMA(a[r], bx[4 * s + r], acc_m);
            }
        }
    }

    // TRAIN = 2: every store_act() goes out right BEHIND the next stage's barrier and DMA request -- issued in front of it, the
    // acquire's wait for the DMA also waited for the write acknowledgements of 16 KB per wave (vector-memory operations retire in order)
    // ---- residual blocks: H += W2 relu(W1 relu(H) + b1) + b2 (resnet.py:37-50) ----
    for (int blk = 0; blk < nblk; ++blk) {
        f32x16 T0, T1, T2, T3;
        {
            const float *bsrc = small + lay.off_bias_hidden(2 * blk) + hh * 16;
            T0 = load_bias16(bsrc);
            T1 = load_bias16(bsrc + 32);
            T2 = load_bias16(bsrc + 64);
            T3 = load_bias16(bsrc + 96);
        }
        {
            const float *buf = acquire();
            if (blk == 0) store_xlu();
            store_act(2 * blk, H0, H1, H2, H3);         // the block's input (blk = 0: the initial layer's output)
            mm128<true, HB, LA>(buf, lane, T0, H0, H1, H2, H3);
        }
        if constexpr (HB >= 2) mm128<true, HB, LA>(acquire(), lane, T1, H0, H1, H2, H3);
        if constexpr (HB == 4) {
            mm128<true, HB, LA>(acquire(), lane, T2, H0, H1, H2, H3);
            mm128<true, HB, LA>(acquire(), lane, T3, H0, H1, H2, H3);
        }
        {
            const float *bsrc = small + lay.off_bias_hidden(2 * blk + 1) + hh * 16;
            H0 += load_bias16(bsrc);
            H1 += load_bias16(bsrc + 32);
            H2 += load_bias16(bsrc + 64);
            H3 += load_bias16(bsrc + 96);
        }
#ifdef NF_EXP_RELU_PER_USE
        store_act(1 + 2 * blk, T0, T1, T2, T3);
        mm128<true, HB, LA>(acquire(), lane, H0, T0, T1, T2, T3);
        if constexpr (HB >= 2) mm128<true, HB, LA>(acquire(), lane, H1, T0, T1, T2, T3);
        if constexpr (HB == 4) {
            mm128<true, HB, LA>(acquire(), lane, H2, T0, T1, T2, T3);
            mm128<true, HB, LA>(acquire(), lane, H3, T0, T1, T2, T3);
        }
#else
        const float *buf2 = acquire();
        store_act(1 + 2 * blk, T0, T1, T2, T3);      // the pre-activation (the backward's ReLU mask and weight-gradient operand)
        // T is dead after this linear: ReLU it once in place (64 v_max) instead of once per use (256)
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            T0[c] = fmaxf(T0[c], 0.0f);
            T1[c] = fmaxf(T1[c], 0.0f);
            T2[c] = fmaxf(T2[c], 0.0f);
            T3[c] = fmaxf(T3[c], 0.0f);
        }
        mm128<false, HB, LA>(buf2, lane, H0, T0, T1, T2, T3);
        if constexpr (HB >= 2) mm128<false, HB, LA>(acquire(), lane, H1, T0, T1, T2, T3);
        if constexpr (HB == 4) {
            mm128<false, HB, LA>(acquire(), lane, H2, T0, T1, T2, T3);
            mm128<false, HB, LA>(acquire(), lane, H3, T0, T1, T2, T3);
        }
#endif
#ifdef NF_EXP_RELU_PER_USE
        store_act(2 + 2 * blk, H0, H1, H2, H3);
#endif
    }
    }  // TRAIN != 1

    // ---- final layer in 8 groups of 3 row-blocks; each group yields the parameters of 2 spline elements ----
    // Software pipeline: the spline evaluations of group g-1 (pure VALU work on registers) are placed in the same
    // scheduling regions as the 3 x 64 MFMAs of group g, so that they issue in the shadow of the 64-cycle MFMAs
    // instead of leaving the matrix pipe idle (ablation: epilogue + unconditional splines exposed = 22 % of the
    // kernel).  Region 1: element 0, region 2: element 1, region 3: two unconditional-spline elements (density).
    float prm[FPL][MP];
    auto extract = [&](const f32x16 &A0, const f32x16 &A1, const f32x16 &A2) {
        // lane's parameter list v = 16 rb + reg: feature f = v / MP, parameter v % MP
#pragma unroll
        for (int v = 0; v < 48; ++v) prm[v / MP][v % MP] = v < 16 ? A0[v] : (v < 32 ? A1[v - 16] : A2[v - 32]);
    };
    auto element = [&](int g, int f) {
#ifdef NF_ABL_NOEPI
#pragma unroll
        for (int v = 0; v < MP; ++v) asm volatile("" ::"v"(prm[f][v]));
        return;
#endif
        const int slot = 8 * (g / GQ) + 2 * ((g % GQ) * FPL + f) + par_t;
        const float xt = st[slot * 64];
        float yt, l;
#ifdef NF_EPI_SCALAR
        rqs_regs<DIR == 1, KB>(p, xt, prm[f], yt, l);
#else
        rqs_regs_t<DIR == 1, KB>(p, xt, prm[f], yt, l);      // (round 5: binary bin descent; K = 4 / 16 and the training forward come here)
#endif
        st[slot * 64] = yt;
        ld += l;
    };
    // both elements of a group at once on packed arithmetic + the binary bin descent (fused_common.hpp rqs_regs2, round 5);
    // -DNF_EPI_SCALAR = the element-by-element evaluation of rounds 1-4 (ablation)
    auto element_pair = [&](int g) {
#if defined(NF_ABL_NOEPI) || defined(NF_EPI_SCALAR)
        element(g, 0);
        element(g, 1);
#else
        if constexpr (FPL == 2) {
            const int s0 = 8 * (g / GQ) + 2 * ((g % GQ) * FPL) + par_t, s1 = s0 + 2;
            float y0, y1, l0, l1;
            rqs_regs2<DIR == 1, KB>(p, st[s0 * 64], st[s1 * 64], prm[0], prm[FPL - 1], y0, y1, l0, l1);
            st[s0 * 64] = y0;
            st[s1 * 64] = y1;
            ld += l0;
            ld += l1;
        }
#endif
    };
    auto uncond_group = [&](int g) {  // density only: the FPL identity slots of chunk Q = g / GQ that go with this group
#ifdef NF_ABL_NOUNCOND
        return;
#endif
        const float *tabs = small + lay.off_tables();
#pragma unroll
        for (int e = 0; e < FPL; ++e) {
            const int r = FPL * (g % GQ) + e;
            const int c = 8 * (g / GQ) + 2 * r + par_i;
            const int f = 8 * (g / GQ) + 4 * hh + r;
            float yi, l;
            rqs_table_fast<false, KB>(p, st[c * 64], tabs + f * TABW, yi, l);
            st[c * 64] = yi;
            ld += l;
        }
    };
#ifdef NF_F32_SWPIPE
    static_assert(KB == F_K && HB == 4, "the software-pipelined order is written for 8 bins, 128 hidden units");
    {   // group 0: MFMAs only
        const float *bsrc = small + lay.off_bias_final() + hh * 16;
        f32x16 A0 = load_bias16(bsrc), A1 = load_bias16(bsrc + 32), A2 = load_bias16(bsrc + 64);
        mm128<false, 4, LA>(acquire(), lane, A0, H0, H1, H2, H3);
        mm128<false, 4, LA>(acquire(), lane, A1, H0, H1, H2, H3);
        mm128<false, 4, LA>(acquire(), lane, A2, H0, H1, H2, H3);
        extract(A0, A1, A2);
    }
    for (int g = 1; g < 8; ++g) {
        const float *bsrc = small + lay.off_bias_final() + (g * 3) * 32 + hh * 16;
        f32x16 A0 = load_bias16(bsrc), A1 = load_bias16(bsrc + 32), A2 = load_bias16(bsrc + 64);
        {
            const float *buf = acquire();
            mm128<false, 4, LA>(buf, lane, A0, H0, H1, H2, H3);
            element(g - 1, 0);
            NF_SCHED_PIPE();
        }
        {
            const float *buf = acquire();
            mm128<false, 4, LA>(buf, lane, A1, H0, H1, H2, H3);
            element(g - 1, 1);
            NF_SCHED_PIPE();
        }
        {
            const float *buf = acquire();
            mm128<false, 4, LA>(buf, lane, A2, H0, H1, H2, H3);
            if (DIR == 0) uncond_group(g - 1);
            NF_SCHED_PIPE();
        }
        extract(A0, A1, A2);
    }
    element(7, 0);
    element(7, 1);
    if (DIR == 0) uncond_group(7);
#else
    // The exact-fp32 MFMA shares the vector ALU (tools/ubench/overlap.py): there is nothing to hide the epilogue
    // behind, so each group's spline elements are evaluated straight from its accumulators (no parameter copies, lower
    // register pressure); the software-pipelined order (NF_F32_SWPIPE) only pays on the bf16 matrix pipe.
    for (int g = 0; g < ngrp; ++g) {   // (up to) KB groups of 3 row-blocks
        const float *bsrc = small + lay.off_bias_final() + (g * 3) * 32 + hh * 16;
        f32x16 A0 = load_bias16(bsrc), A1 = load_bias16(bsrc + 32), A2 = load_bias16(bsrc + 64);
        {
            // behind group g - 1's 14 parameter stores (training); group 0: the last residual block's output goes out here
            const float *buf = acquire(g > 0 ? 14 : 0);
#ifndef NF_EXP_RELU_PER_USE
            if (g == 0) {
                if (nblk == 0) store_xlu();
                store_act(2 * nblk, H0, H1, H2, H3);
            }
#endif
            mm128<false, HB, LA>(buf, lane, A0, H0, H1, H2, H3);
        }
        mm128<false, HB, LA>(acquire(), lane, A1, H0, H1, H2, H3);
        mm128<false, HB, LA>(acquire(), lane, A2, H0, H1, H2, H3);
        extract(A0, A1, A2);
#ifdef NF_ABL_NOCOND
        if constexpr (false) {
#else
        if constexpr (TRAIN) {
#endif
#ifdef NF_TRAIN_DIRECT_STORES
            if (valid) {   // the two features' parameter sets, raw scale, for the backward kernel (pitch 24 floats per feature)
                float *dst = cond_out + row * (F_NI * 24) + (8 * (g >> 1) + 4 * hh + 2 * (g & 1)) * 24;
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    const float u_ = q < 4 ? unscale : 1.0f;
                    *reinterpret_cast<f32x4 *>(dst + 4 * q) =
                        f32x4{prm[0][4 * q] * u_, prm[0][4 * q + 1] * u_, prm[0][4 * q + 2] * u_, prm[0][4 * q + 3] * u_};
                    *reinterpret_cast<f32x4 *>(dst + 24 + 4 * q) =
                        f32x4{prm[1][4 * q] * u_, prm[1][4 * q + 1] * u_, prm[1][4 * q + 2] * u_, prm[1][4 * q + 3] * u_};
                }
            }
#else
            // The two features' parameter sets (raw scale, pitch 24 floats per feature) for the backward kernel, through a 6 KB
            // transpose tile per wave: a lane holds ITS row's 192 contiguous bytes (rows 3 KB apart), so direct 16-byte stores
            // touch 64 partial lines per instruction (280 MB of fabric writes for 218 MB of rows, 40 us of the launch);
            // re-read row-major, 12 lanes cover one row's 192 bytes and an instruction writes five rows' full lines.
            int l_ = lane;      // (addresses derived afresh per group: see store_act)
            asm volatile("" : "+v"(l_));
            float *tw = small2 + small_pitch + wid * 1536;      // the second small buffer is free (one layer) + the extra LDS
            const int64_t row0 = (int64_t)blockIdx.x * F_ROWS + wid * 32;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if ((l_ >> 5) == half) {
                    float *tr = tw + (l_ & 31) * 48;
#pragma unroll
                    for (int q = 0; q < 6; ++q) {
                        const float u_ = q < 4 ? unscale : 1.0f;
                        *reinterpret_cast<f32x4 *>(tr + 4 * q) =
                            f32x4{prm[0][4 * q] * u_, prm[0][4 * q + 1] * u_, prm[0][4 * q + 2] * u_, prm[0][4 * q + 3] * u_};
                        *reinterpret_cast<f32x4 *>(tr + 24 + 4 * q) =
                            f32x4{prm[1][4 * q] * u_, prm[1][4 * q + 1] * u_, prm[1][4 * q + 2] * u_, prm[1][4 * q + 3] * u_};
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const int fo = (8 * (g >> 1) + 4 * half + 2 * (g & 1)) * 24;
                const int rl = l_ / 12, ql = l_ - 12 * rl;          // 5 rows x 12 sixteen-byte pieces per instruction
#pragma unroll
                for (int it = 0; it < 7; ++it) {
                    const int r = 5 * it + rl;
                    if (l_ < 60 && r < 32 && row0 + r < B)
                        NF_TRAIN_STORE(reinterpret_cast<f32x4 *>(cond_out + (row0 + r) * (F_NI * 24) + fo + 4 * ql),
                                       *reinterpret_cast<const f32x4 *>(tw + r * 48 + 4 * ql));
                }
                __builtin_amdgcn_wave_barrier();
            }
#endif
        }
        if constexpr (FPL == 2 && TRAIN != 2) {      // (the training forward's row stores leave no registers for the pair: 4 spills)
            element_pair(g);
        } else {
#pragma unroll
            for (int f = 0; f < FPL; ++f) element(g, f);
        }
        if (DIR == 0) uncond_group(g);
    }
#endif

    if (LU && DIR == 1) {
        // LULinearPermute.forward (mixing.py:555-558): triangular solves + inverse permutation as one dense product
        float yin[32];
#pragma unroll
        for (int c = 0; c < 32; ++c) yin[c] = st[c * 64];
        const float *bsrc = small + lay.off_bias_lu(1) + hh * 16;
        f32x16 o0 = load_bias16(bsrc), o1 = load_bias16(bsrc + 32);
        lu_mm(acquire(), lane, yin, o0, o1);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            st[c * 64] = o0[c];
            st[(16 + c) * 64] = o1[c];
        }
        if (hh == 0) ld -= lu_lad;
    }
    }  // layers

    // ---- epilogue: rows back to HBM, per-sample log-det (both lane halves of a sample) ----
    ld += __shfl_xor(ld, 32, 64);
    if (valid) {
#pragma unroll
        for (int Q = 0; Q < 4; ++Q) {
            f32x4 a, b;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                a[c] = st[(8 * Q + c) * 64];
                b[c] = st[(8 * Q + 4 + c) * 64];
            }
            float *dst = y + row * F_D + 16 * Q + 8 * hh;
            *reinterpret_cast<f32x4 *>(dst) = a;
            *reinterpret_cast<f32x4 *>(dst + 4) = b;
        }
        if (hh == 0) ld_store(logdet + row, ld, acc);
    }
}

#ifndef NF_FUSED_SECONDARY
// The whole blob of a layer (without the LU) in ONE launch: what nf_rqs_fused_pack does with 4 + 2 nblk launches.  The training
// step re-packs every layer's weights once per step, so the launches count (8 bins only: the training variants' shape).
struct PackAllArgs {
    const float *w_init, *b_init, *w_final, *b_final, *uw, *uh, *ud;
    const float *w_blk[32], *b_blk[32];
    int nblk;
    // optional by-products for the backward (all three or none): the initial weight TRANSPOSED on full rows (64, hidden; only the
    // identity features' rows are written, the rest of the caller's buffer stays zero) and the final weight as the 24 A-operand
    // stages of nf_final_bwd (32 x 24 x hidden floats: the buffer round 2 filled with 24-row groups for a library GEMM)
    float *wfull, *wpad;
    const int64_t *iidx;
};
__device__ __forceinline__ void pack_all_body(const PackAllArgs &a, float *__restrict__ blob, float wh_scale,
                                              const RqsParams<float> &p) {
    FusedLayout lay;
    lay.nblk = a.nblk;
    float *small = blob + F_HDR, *stages = blob + lay.off_stages();
    const int nlin = 2 * a.nblk, nst = 1 + 4 * nlin + 24;
    const int64_t tid0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = tid0; i < (int64_t)nst * F_STAGE; i += nth) {
        const int st = (int)(i >> 12), j = (int)(i & (F_STAGE - 1));
        const int r4 = j & 3, lane = (j >> 2) & 63;
        float v;
        if (st == 0) {                       // initial layer: 4 row-blocks x 4 k-groups
            const int s = (j >> 8) & 3, m = j >> 10;
            v = a.w_init[(32 * m + (lane & 31)) * F_NI + 8 * s + 4 * (lane >> 5) + r4];
        } else if (st <= 4 * nlin) {         // hidden Linear l, row-block m
            const int l = (st - 1) >> 2, m = (st - 1) & 3, s = (j >> 8) & 15;
            v = a.w_blk[l][(32 * m + (lane & 31)) * F_H + 8 * s + 4 * (lane >> 5) + r4];
        } else {                             // final layer: group g, row-block rb
            const int f = st - 1 - 4 * nlin, g = f / 3, rb = f % 3, s = (j >> 8) & 15;
            const int row = final_row(g, rb, lane & 31);
            const float sc = (row >= 0 && (row % F_M) < 2 * F_K) ? wh_scale : 1.0f;
            v = row >= 0 ? a.w_final[row * F_H + 8 * s + 4 * (lane >> 5) + r4] * sc : 0.0f;
        }
        stages[i] = v;
    }
    if (a.wfull) {
        for (int64_t i = tid0; i < F_H * F_NI; i += nth) a.wfull[a.iidx[i & 31] * F_H + (i >> 5)] = a.w_init[i];
        // the final weight as the A operand of the backward's gh = g W_final (final_bwd.hip, v_mfma_f32_16x16x4_f32): 24 stages
        // [k-step vv (8)][unit-block quad uq (2)][lane (64)][unit block j (4)]; stage 3 g + rb holds k-steps v = 8 rb + vv of group g,
        // lane (m = lane & 15, hq = lane >> 4) supplies W[final_bwd_row(g, v, hq)][16 (4 uq + j) + m], raw scale
        for (int64_t i = tid0; i < (int64_t)24 * F_STAGE; i += nth) {
            const int st = (int)(i >> 12), j = (int)(i & (F_STAGE - 1));
            const int u4 = j & 3, lane = (j >> 2) & 63, uq = (j >> 8) & 1, vv = j >> 9;
            const int row = final_bwd_row(st / 3, 8 * (st % 3) + vv, lane >> 4);
            a.wpad[i] = row >= 0 ? a.w_final[row * F_H + 16 * (4 * uq + u4) + (lane & 15)] : 0.0f;
        }
    }
    const int nbias = 128 + 128 * nlin + 768;
    for (int64_t i = tid0; i < nbias; i += nth) {
        const int reg = i & 15, hh = (i >> 4) & 1;
        float v;
        if (i < 128 + 128 * nlin) {
            const int blk = (int)(i >> 7), m = (int)((i >> 5) & 3);
            const float *b = blk == 0 ? a.b_init : a.b_blk[blk - 1];
            v = b[32 * m + 8 * (reg >> 2) + 4 * hh + (reg & 3)];
        } else {
            const int st = (int)((i - 128 - 128 * nlin) >> 5), g = st / 3, rb = st % 3;
            const int row = final_row(g, rb, 8 * (reg >> 2) + 4 * hh + (reg & 3));
            const float sc = (row >= 0 && (row % F_M) < 2 * F_K) ? wh_scale : 1.0f;
            v = row >= 0 ? a.b_final[row] * sc : 0.0f;
        }
        small[i] = v;       // bias_init | bias_hidden | bias_final are contiguous at the start of the small section
    }
    if (blockIdx.x == gridDim.x - 1) {
        const int j = threadIdx.x;
        if (j == 0) {
            blob[0] = 355.0f;
            blob[1] = (float)a.nblk;
            blob[2] = 0.0f;
            blob[3] = 0.0f;
        }
        if (j < F_NI) {
            const float *wj = a.uw + j * F_K, *hj = a.uh + j * F_K, *dj = a.ud + j * (F_K - 1);
            auto wacc = [=](int k) { return wj[k]; };
            auto hacc = [=](int k) { return hj[k]; };
            auto dacc = [=](int k) { return dj[k]; };
            rqs_build_table<float>(p, wacc, hacc, dacc, small + lay.off_tables() + j * F_TABW);
        }
    }
}

__global__ void pack_all_kernel(PackAllArgs a, float *__restrict__ blob, float wh_scale, RqsParams<float> p) {
    pack_all_body(a, blob, wh_scale, p);
}

// Every layer of a model in ONE launch (blockIdx.y = layer): the training step re-packs all layers once per step, 32 launches of
// ~12 us otherwise.  table: per layer NF_PACK_ALL_COLS(num_blocks) device pointers (include/nf_mi355x.h).
__global__ void pack_all_multi_kernel(const void *const *__restrict__ table, int cols, int nblk, float wh_scale, RqsParams<float> p) {
    __shared__ PackAllArgs a;
    __shared__ float *blob;
    const void *const *row = table + (size_t)blockIdx.y * cols;
    if (threadIdx.x == 0) {
        blob = (float *)row[0];
        a.w_init = (const float *)row[1]; a.b_init = (const float *)row[2];
        a.w_final = (const float *)row[3]; a.b_final = (const float *)row[4];
        a.uw = (const float *)row[5]; a.uh = (const float *)row[6]; a.ud = (const float *)row[7];
        a.wfull = (float *)row[8]; a.wpad = (float *)row[9]; a.iidx = (const int64_t *)row[10];
        a.nblk = nblk;
        for (int l = 0; l < 2 * nblk; ++l) {
            a.w_blk[l] = (const float *)row[11 + l];
            a.b_blk[l] = (const float *)row[11 + 2 * nblk + l];
        }
    }
    __syncthreads();
    pack_all_body(a, blob, wh_scale, p);
}
#endif  // !NF_FUSED_SECONDARY

}  // namespace nf

using namespace nf;

static inline bool fused_bins_ok(int K) { return K == 4 || K == 8 || K == 16; }   // instantiations of the exact-fp32 kernel

// The whole-layer training forward (TRAIN = 2; LU: with the layer's LULinearPermute.inverse in front) on THIS build's workgroup size.
template <bool LU>
static int train_fwd_launch(const void *x, void *xlu_out, void *y, void *logdet, void *cond_out, void *act_out, const FlowArgs &fa,
                            int64_t B, int hidden, int num_blocks, const RqsParams<float> &p, int acc, hipStream_t st) {
    FusedLayout lay;
    lay.nblk = num_blocks;
    const size_t lds = (size_t)(2 * F_STAGE + F_NW * 32 * 64 + lay.small_padded() + F_NW * 1536) * sizeof(float);
    if (lds > 160 * 1024) return NF_ENOTSUP;
    static LdsOptIn opted = {};
    if (opt_in_lds(reinterpret_cast<const void *>(&rqs_fused_kernel<0, LU, 2>), lds, opted) != NF_OK) return NF_ENOTSUP;
    const int grid = (int)((B + F_ROWS - 1) / F_ROWS);
    hipLaunchKernelGGL((rqs_fused_kernel<0, LU, 2>), dim3(grid), dim3(F_THREADS), lds, st, (const float *)x, (float *)y,
                       (float *)logdet, fa, B, num_blocks, p, acc, (const float *)nullptr, (float *)cond_out,
                       (float)(sqrt((double)hidden) / 1.4426950408889634), (float *)act_out, (float *)xlu_out);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

#ifdef NF_FUSED_SECONDARY
// (internal: called by nf_rqs_fused_train_full_fwd / _pair_fwd of the primary build)
extern "C" int nf_rqs_fused_train_fwd_nw4_(int lu, const void *x, void *xlu_out, void *y, void *logdet, void *cond_out, void *act_out,
                                           const void *fa, int64_t B, int hidden, int num_blocks, const void *p, int acc,
                                           nf_stream_t stream) {
    const FlowArgs &f = *static_cast<const FlowArgs *>(fa);
    const RqsParams<float> &pp = *static_cast<const RqsParams<float> *>(p);
    return lu ? train_fwd_launch<true>(x, xlu_out, y, logdet, cond_out, act_out, f, B, hidden, num_blocks, pp, acc, (hipStream_t)stream)
              : train_fwd_launch<false>(x, xlu_out, y, logdet, cond_out, act_out, f, B, hidden, num_blocks, pp, acc, (hipStream_t)stream);
}
#endif

#ifndef NF_FUSED_SECONDARY
extern "C" int nf_rqs_fused_train_fwd_nw4_(int lu, const void *x, void *xlu_out, void *y, void *logdet, void *cond_out, void *act_out,
                                           const void *fa, int64_t B, int hidden, int num_blocks, const void *p, int acc,
                                           nf_stream_t stream);
// Batches of at most NF_FUSED_SMALL_ROWS rows run on 128-row workgroups (rqs_fused_nw4.hip; the note at nf_rqs_fused_chain).
#ifndef NF_FUSED_SMALL_ROWS
#define NF_FUSED_SMALL_ROWS 32768
#endif
static int g_small_batch = 1;

// One-launch pack of a whole layer (8 bins, no LU) for the training forward nf_rqs_fused_train_full_fwd.
extern "C" int nf_rqs_fused_pack_all(void *wpack, const void *w_init, const void *b_init, const void *const *w_blocks,
                                     const void *const *b_blocks, const void *w_final, const void *b_final, const void *uw,
                                     const void *uh, const void *ud, int hidden, int num_blocks, int K, double tail_bound,
                                     double min_bin_width, double min_bin_height, double min_derivative, void *wfull, void *wpad,
                                     const void *identity_idx, nf_stream_t stream) {
    if (hidden != F_H || K != F_K || num_blocks < 0 || num_blocks > 16) return NF_ENOTSUP;
    if (!wpack || !w_init || !b_init || !w_final || !b_final || !uw || !uh || !ud) return NF_EFAULT;
    if ((wfull || wpad || identity_idx) && !(wfull && wpad && identity_idx)) return NF_EFAULT;
    if (num_blocks > 0 && (!w_blocks || !b_blocks)) return NF_EFAULT;
    if (min_bin_width * K > 1.0 || min_bin_height * K > 1.0) return NF_EINVAL;
    PackAllArgs a;
    a.w_init = (const float *)w_init; a.b_init = (const float *)b_init;
    a.w_final = (const float *)w_final; a.b_final = (const float *)b_final;
    a.uw = (const float *)uw; a.uh = (const float *)uh; a.ud = (const float *)ud;
    a.nblk = num_blocks;
    a.wfull = (float *)wfull; a.wpad = (float *)wpad; a.iidx = (const int64_t *)identity_idx;
    for (int l = 0; l < 32; ++l) { a.w_blk[l] = nullptr; a.b_blk[l] = nullptr; }
    for (int l = 0; l < 2 * num_blocks; ++l) {
        if (!w_blocks[l] || !b_blocks[l]) return NF_EFAULT;
        a.w_blk[l] = (const float *)w_blocks[l];
        a.b_blk[l] = (const float *)b_blocks[l];
    }
    auto p = make_rqs_params<float>(K, NF_TAILS_LINEAR, tail_bound, 0, 1, 0, 1, min_bin_width, min_bin_height,
                                    min_derivative, 1.0);
    hipLaunchKernelGGL(pack_all_kernel, dim3(16 * (1 + 8 * num_blocks + 24)), dim3(256), 0, (hipStream_t)stream, a, (float *)wpack,
                       (float)(1.4426950408889634 / sqrt((double)hidden)), p);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// nf_rqs_fused_pack_all for n layers of one shape in ONE launch; table (device): n rows of 11 + 4 num_blocks pointers --
// wpack, w_init, b_init, w_final, b_final, uw, uh, ud, wfull, wpad, identity_idx, then the 2 num_blocks hidden weights and the
// 2 num_blocks hidden biases in layer order (wfull / wpad / identity_idx may be NULL together).
extern "C" int nf_rqs_fused_pack_all_multi(const void *table, int n_layers, int hidden, int num_blocks, int K, double tail_bound,
                                           double min_bin_width, double min_bin_height, double min_derivative, nf_stream_t stream) {
    if (hidden != F_H || K != F_K || num_blocks < 0 || num_blocks > 16) return NF_ENOTSUP;
    if (n_layers < 0 || n_layers > 65535) return NF_EINVAL;
    if (n_layers == 0) return NF_OK;
    if (!table) return NF_EFAULT;
    if (min_bin_width * K > 1.0 || min_bin_height * K > 1.0) return NF_EINVAL;
    auto p = make_rqs_params<float>(K, NF_TAILS_LINEAR, tail_bound, 0, 1, 0, 1, min_bin_width, min_bin_height,
                                    min_derivative, 1.0);
    hipLaunchKernelGGL(pack_all_multi_kernel, dim3(16 * (1 + 8 * num_blocks + 24), n_layers), dim3(256), 0, (hipStream_t)stream,
                       (const void *const *)table, 11 + 4 * num_blocks, num_blocks,
                       (float)(1.4426950408889634 / sqrt((double)hidden)), p);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// Training forward of a WHOLE layer (core.py:87-102 over nets/resnet.py:53-104 + nsf/coupling.py:83-98): the inference
// launch plus the tensors the backward needs -- act_out (2 num_blocks + 1, B, 128): h0, then per block its pre-activation t and
// its output h; cond_out (B, 32, 24) as nf_rqs_fused_train_fwd.  wpack: nf_rqs_fused_pack_all / nf_rqs_fused_pack (no LU).
extern "C" int nf_rqs_fused_train_full_fwd(const void *x, void *y, void *logdet, void *cond_out, void *act_out, const void *wpack,
                                           int mask_parity, int64_t B, int D, int hidden, int num_blocks, int K, double tail_bound,
                                           double min_bin_width, double min_bin_height, double min_derivative, int acc,
                                           nf_stream_t stream) {
    if (D != F_D || hidden != F_H || K != F_K || num_blocks < 0 || num_blocks > 16) return NF_ENOTSUP;
    if (B < 0 || (mask_parity != 0 && mask_parity != 1) || acc < NF_LD_SUB || acc > NF_LD_ADD) return NF_EINVAL;
    if (B == 0) return NF_OK;
    if (!x || !y || !logdet || !cond_out || !act_out || !wpack) return NF_EFAULT;
    FlowArgs fa;
    fa.parity = mask_parity ? 1ull : 0ull;
    fa.nlayers = 1;
    fa.ngroups = F_K;
    for (int l = 0; l < F_MAX_LAYERS; ++l) fa.blob[l] = nullptr;
    fa.blob[0] = (const float *)wpack;
    auto p = make_rqs_params<float>(K, NF_TAILS_LINEAR, tail_bound, 0, 1, 0, 1, min_bin_width, min_bin_height,
                                    min_derivative, sqrt((double)hidden));
    if (g_small_batch && B <= NF_FUSED_SMALL_ROWS)
        return nf_rqs_fused_train_fwd_nw4_(0, x, nullptr, y, logdet, cond_out, act_out, &fa, B, hidden, num_blocks, &p, acc, stream);
    return train_fwd_launch<false>(x, nullptr, y, logdet, cond_out, act_out, fa, B, hidden, num_blocks, p, acc, (hipStream_t)stream);
}

// Whole-layer training forward WITH the layer's LULinearPermute.inverse in front of the coupling (round 6): the launch of
// nf_rqs_fused_train_full_fwd plus one LU stage; xlu_out (B, 64) = the LU's output = the coupling's input (the backward's `x`).
// wpack: nf_rqs_fused_pack_all[_multi] then nf_lu_pack_train_multi (LU stage, bias, log|det| in the header).
extern "C" int nf_rqs_fused_train_pair_fwd(const void *x, void *xlu_out, void *y, void *logdet, void *cond_out, void *act_out,
                                           const void *wpack, int mask_parity, int64_t B, int D, int hidden, int num_blocks, int K,
                                           double tail_bound, double min_bin_width, double min_bin_height, double min_derivative,
                                           int acc, nf_stream_t stream) {
    if (D != F_D || hidden != F_H || K != F_K || num_blocks < 0 || num_blocks > 16) return NF_ENOTSUP;
    if (B < 0 || (mask_parity != 0 && mask_parity != 1) || acc < NF_LD_SUB || acc > NF_LD_ADD) return NF_EINVAL;
    if (B == 0) return NF_OK;
    if (!x || !xlu_out || !y || !logdet || !cond_out || !act_out || !wpack) return NF_EFAULT;
    FlowArgs fa;
    fa.parity = mask_parity ? 1ull : 0ull;
    fa.nlayers = 1;
    fa.ngroups = F_K;
    for (int l = 0; l < F_MAX_LAYERS; ++l) fa.blob[l] = nullptr;
    fa.blob[0] = (const float *)wpack;
    auto p = make_rqs_params<float>(K, NF_TAILS_LINEAR, tail_bound, 0, 1, 0, 1, min_bin_width, min_bin_height,
                                    min_derivative, sqrt((double)hidden));
    if (g_small_batch && B <= NF_FUSED_SMALL_ROWS)
        return nf_rqs_fused_train_fwd_nw4_(1, x, xlu_out, y, logdet, cond_out, act_out, &fa, B, hidden, num_blocks, &p, acc, stream);
    return train_fwd_launch<true>(x, xlu_out, y, logdet, cond_out, act_out, fa, B, hidden, num_blocks, p, acc, (hipStream_t)stream);
}

// The LU stage of n training blobs in one launch (once per step, behind nf_rqs_fused_pack_all_multi).  table (device): n rows of
// 7 pointers perm (int64), lower_entries, upper_entries, unconstrained_upper_diag, bias, wpack, wd_out (64 x 64 floats: W_d
// row-major, y = W_d x + bias, for nf_lu_bwd_composed).  D = 64, 8 bins.
extern "C" int nf_lu_pack_train_multi(const void *table, int n_layers, int num_blocks, int D, double eps, nf_stream_t stream) {
    if (D != F_D || num_blocks < 0 || num_blocks > 16) return NF_ENOTSUP;
    if (n_layers < 0 || n_layers > 65535) return NF_EINVAL;
    if (n_layers == 0) return NF_OK;
    if (!table) return NF_EFAULT;
    const size_t lds = (size_t)3 * F_D * F_D * sizeof(double) + F_D * sizeof(int) + 64;
    static LdsOptIn opted = {};
    if (opt_in_lds(reinterpret_cast<const void *>(&pack_lu_train_multi_kernel), lds, opted) != NF_OK) return NF_ENOTSUP;
    hipLaunchKernelGGL(pack_lu_train_multi_kernel, dim3(n_layers), dim3(256), lds, (hipStream_t)stream, (const void *const *)table,
                       (float)eps, num_blocks);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// Host-side view of the final layer's row order (tests): row of the (32 (3 K - 1), 128) weight held by MFMA row `rho` of
// row-block `rb` of group `g`, -1 for a padding row, NF_ENOTSUP for an unsupported bin count.
extern "C" int nf_rqs_fused_final_row(int K, int g, int rb, int rho) {
    if (!fused_bins_ok(K)) return NF_ENOTSUP;
    if (g < 0 || g >= K || rb < 0 || rb > 2 || rho < 0 || rho > 31) return NF_EINVAL;
    return K == 4 ? final_row_k<4>(g, rb, rho) : (K == 8 ? final_row_k<8>(g, rb, rho) : final_row_k<16>(g, rb, rho));
}

extern "C" int64_t nf_rqs_fused_pack_size(int nI, int nT, int hidden, int num_blocks, int K) {
    if (nI != F_NI || nT != F_NI || hidden != F_H || !fused_bins_ok(K) || num_blocks < 0 || num_blocks > 16) return NF_ENOTSUP;
    FusedLayout lay;
    lay.nblk = num_blocks;
    lay.K = K;
    return lay.total_floats() * (int64_t)sizeof(float);
}

extern "C" int nf_rqs_fused_pack(void *wpack, const void *w_init, const void *b_init, const void *const *w_blocks,
                                 const void *const *b_blocks, const void *w_final, const void *b_final, const void *uw,
                                 const void *uh, const void *ud, int nI, int nT, int hidden, int num_blocks, int K,
                                 double tail_bound, double min_bin_width, double min_bin_height, double min_derivative,
                                 nf_stream_t stream) {
    if (nI != F_NI || nT != F_NI || hidden != F_H || !fused_bins_ok(K) || num_blocks < 0 || num_blocks > 16) return NF_ENOTSUP;
    if (!wpack || !w_init || !b_init || !w_final || !b_final || !uw || !uh || !ud) return NF_EFAULT;
    if (num_blocks > 0 && (!w_blocks || !b_blocks)) return NF_EFAULT;
    if (min_bin_width * K > 1.0 || min_bin_height * K > 1.0) return NF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    FusedLayout lay;
    lay.nblk = num_blocks;
    lay.K = K;
    float *blob = (float *)wpack;
    float *small = blob + F_HDR;
    float *stages = blob + lay.off_stages();
    hipLaunchKernelGGL(pack_header_kernel, dim3(1), dim3(64), 0, st, blob, num_blocks);
    hipLaunchKernelGGL(pack_init_kernel, dim3(16), dim3(256), 0, st, (const float *)w_init, (const float *)b_init, stages,
                       small + lay.off_bias_init());
    for (int l = 0; l < 2 * num_blocks; ++l) {
        if (!w_blocks[l] || !b_blocks[l]) return NF_EFAULT;
        hipLaunchKernelGGL(pack_hidden_kernel, dim3(64), dim3(256), 0, st, (const float *)w_blocks[l],
                           (const float *)b_blocks[l], stages + (size_t)(1 + 4 * l) * F_STAGE,
                           small + lay.off_bias_hidden(l));
    }
#define NF_PACK_FINAL(KB)                                                                                            \
    hipLaunchKernelGGL(pack_final_kernel<KB>, dim3(48 * KB), dim3(256), 0, st, (const float *)w_final,                   \
                       (const float *)b_final, stages + (size_t)(1 + 8 * num_blocks) * F_STAGE,                          \
                       small + lay.off_bias_final(), (float)(1.4426950408889634 / sqrt((double)hidden)))
    if (K == 4) NF_PACK_FINAL(4);
    else if (K == 8) NF_PACK_FINAL(8);
    else NF_PACK_FINAL(16);
#undef NF_PACK_FINAL
    auto p = make_rqs_params<float>(K, NF_TAILS_LINEAR, tail_bound, 0, 1, 0, 1, min_bin_width, min_bin_height,
                                    min_derivative, 1.0);
    hipLaunchKernelGGL(pack_tables_kernel, dim3(1), dim3(64), 0, st, (const float *)uw, (const float *)uh,
                       (const float *)ud, small + lay.off_tables(), p);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// Only what the training forward (nf_rqs_fused_train_fwd) reads: header, final-layer stages + bias, knot tables.
extern "C" int nf_rqs_fused_pack_final(void *wpack, const void *w_final, const void *b_final, const void *uw, const void *uh,
                                       const void *ud, int hidden, int num_blocks, int K, double tail_bound,
                                       double min_bin_width, double min_bin_height, double min_derivative, nf_stream_t stream) {
    if (hidden != F_H || K != F_K || num_blocks < 0 || num_blocks > 16) return NF_ENOTSUP;
    if (!wpack || !w_final || !b_final || !uw || !uh || !ud) return NF_EFAULT;
    if (min_bin_width * K > 1.0 || min_bin_height * K > 1.0) return NF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    FusedLayout lay;
    lay.nblk = num_blocks;
    float *blob = (float *)wpack;
    float *small = blob + F_HDR;
    float *stages = blob + lay.off_stages();
    auto p = make_rqs_params<float>(K, NF_TAILS_LINEAR, tail_bound, 0, 1, 0, 1, min_bin_width, min_bin_height,
                                    min_derivative, 1.0);
    hipLaunchKernelGGL(pack_final_all_kernel, dim3(384), dim3(256), 0, st, (const float *)w_final, (const float *)b_final,
                       stages + (size_t)(1 + 8 * num_blocks) * F_STAGE, small + lay.off_bias_final(),
                       (float)(1.4426950408889634 / sqrt((double)hidden)), blob, num_blocks, (const float *)uw,
                       (const float *)uh, (const float *)ud, small + lay.off_tables(), p);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

extern "C" int nf_rqs_fused_pack_lu(void *wpack, int num_blocks, const int64_t *perm, const void *lower_entries,
                                    const void *upper_entries, const void *unconstrained_upper_diag, const void *bias,
                                    int D, double eps, int K, nf_stream_t stream) {
    if (D != F_D || num_blocks < 0 || num_blocks > 16 || !fused_bins_ok(K)) return NF_ENOTSUP;
    if (!wpack || !perm || !lower_entries || !upper_entries || !unconstrained_upper_diag || !bias) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    FusedLayout lay;
    lay.nblk = num_blocks;
    lay.K = K;          // the layout of the blob the LU stages are added to (nf_rqs_fused_pack with the same K)
    float *blob = (float *)wpack;
    float *small = blob + F_HDR;
    float *stages = blob + lay.off_stages();
    const size_t lds = (size_t)4 * F_D * F_D * sizeof(double) + F_D * sizeof(int) + F_D * sizeof(double) + 64;
    static LdsOptIn opted = {};
    if (opt_in_lds(reinterpret_cast<const void *>(&pack_lu_kernel), lds, opted) != NF_OK) return NF_ENOTSUP;
    hipLaunchKernelGGL(pack_lu_kernel, dim3(1), dim3(256), lds, st, perm, (const float *)lower_entries,
                       (const float *)upper_entries, (const float *)unconstrained_upper_diag, (const float *)bias,
                       (float)eps, stages + (size_t)lay.lu_stage(0) * F_STAGE, stages + (size_t)lay.lu_stage(1) * F_STAGE,
                       small + lay.off_bias_lu(0), small + lay.off_bias_lu(1), blob);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

#endif  // !NF_FUSED_SECONDARY

template <int DIR, bool LU, int KB, int HB>
static int launch_fused(const void *x, void *y, void *logdet, const FlowArgs &fa, int64_t B, int num_blocks,
                        const RqsParams<float> &p, int acc, size_t lds, hipStream_t st) {
    static LdsOptIn opted = {};  // one per <DIR, LU, KB, HB> instantiation
    if (opt_in_lds(reinterpret_cast<const void *>(&rqs_fused_kernel<DIR, LU, false, KB, HB>), lds, opted) != NF_OK) return NF_ENOTSUP;
    const int grid = (int)((B + F_ROWS - 1) / F_ROWS);
    hipLaunchKernelGGL((rqs_fused_kernel<DIR, LU, false, KB, HB>), dim3(grid), dim3(F_THREADS), lds, st, (const float *)x, (float *)y,
                       (float *)logdet, fa, B, num_blocks, p, acc, (const float *)nullptr, (float *)nullptr, 1.0f);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// The instantiation for (direction, LU, bins, hidden units) of THIS build's workgroup size.
static int chain_dispatch(const void *x, void *y, void *logdet, const FlowArgs &fa, int64_t B, int hidden, int num_blocks, int K,
                          const RqsParams<float> &p, int direction, int fuse_lu, int acc, hipStream_t st) {
    FusedLayout lay;
    lay.nblk = num_blocks;
    lay.K = K;
    const size_t lds = (size_t)(2 * F_STAGE + F_NW * 32 * 64 + 2 * lay.small_padded()) * sizeof(float);
    if (lds > 160 * 1024) return NF_ENOTSUP;
#define NF_FUSED_DISPATCH(KB, HB)                                                                                     \
    do {                                                                                                              \
        if (direction == 0)                                                                                           \
            return fuse_lu ? launch_fused<0, true, KB, HB>(x, y, logdet, fa, B, num_blocks, p, acc, lds, st)          \
                           : launch_fused<0, false, KB, HB>(x, y, logdet, fa, B, num_blocks, p, acc, lds, st);        \
        return fuse_lu ? launch_fused<1, true, KB, HB>(x, y, logdet, fa, B, num_blocks, p, acc, lds, st)              \
                       : launch_fused<1, false, KB, HB>(x, y, logdet, fa, B, num_blocks, p, acc, lds, st);            \
    } while (0)
    if (hidden == F_H / 2) {
        if (K == 4) NF_FUSED_DISPATCH(4, 2);
        if (K == 16) NF_FUSED_DISPATCH(16, 2);
        NF_FUSED_DISPATCH(8, 2);
    }
    if (hidden == F_H / 4) {
        if (K == 4) NF_FUSED_DISPATCH(4, 1);
        if (K == 16) NF_FUSED_DISPATCH(16, 1);
        NF_FUSED_DISPATCH(8, 1);
    }
    if (K == 4) NF_FUSED_DISPATCH(4, 4);
    if (K == 16) NF_FUSED_DISPATCH(16, 4);
    NF_FUSED_DISPATCH(8, 4);
#undef NF_FUSED_DISPATCH
}

#ifdef NF_FUSED_SECONDARY
// (internal: called by nf_rqs_fused_chain of the primary build; fa / p by pointer across the translation units)
extern "C" int nf_rqs_fused_chain_nw4_(const void *x, void *y, void *logdet, const void *fa, int64_t B, int hidden, int num_blocks, int K,
                                       const void *p, int direction, int fuse_lu, int acc, nf_stream_t stream) {
    static_assert(F_NW == 4, "the secondary build is the 4-wave one");
    return chain_dispatch(x, y, logdet, *static_cast<const FlowArgs *>(fa), B, hidden, num_blocks, K,
                          *static_cast<const RqsParams<float> *>(p), direction, fuse_lu, acc, (hipStream_t)stream);
}
#else
extern "C" int nf_rqs_fused_chain_nw4_(const void *x, void *y, void *logdet, const void *fa, int64_t B, int hidden, int num_blocks, int K,
                                       const void *p, int direction, int fuse_lu, int acc, nf_stream_t stream);

// Batches of at most NF_FUSED_SMALL_ROWS rows run on 128-row workgroups (rqs_fused_nw4.hip).  A workgroup owns its rows for the whole
// chain, so a pass takes the same 5.4 ms for ANY batch the 8-wave workgroups hold in one round (<= 65 536 rows); at <= 32 768 rows half
// of the CUs have no workgroup at all.  Four waves per workgroup put those rows on twice as many CUs, one wave per SIMD instead of two
// sharing its matrix pipe: 3.0 ms per pass (32 768 rows: 6.0 -> 10.8 M rows/s).  Above that the 8-wave layout wins (one weight stream
// and one barrier per 256 rows: 65 536 rows 12.07 vs 10.92 M rows/s).  nf_rqs_fused_small_batch(0) keeps every batch on the 8-wave
// kernel (differential tests).
extern "C" int nf_rqs_fused_small_batch(int enable) {
    const int old = g_small_batch;
    if (enable >= 0) g_small_batch = enable ? 1 : 0;
    return old;
}

extern "C" int nf_rqs_fused_chain(const void *x, void *y, void *logdet, const void *const *wpacks,
                                  const int *mask_parities, int num_layers, int fuse_lu, int64_t B, int D, int hidden,
                                  int num_blocks, int K, double tail_bound, double min_bin_width, double min_bin_height,
                                  double min_derivative, int direction, int acc, nf_stream_t stream) {
    // hidden = 128, or 64 / 32: the blobs are the 128-unit layout (nf_rqs_fused_pack) whose units >= hidden are all-zero rows and
    // columns (a narrower conditioner zero-padded by the caller); the kernel then skips those row-blocks and k-groups
    // D = 64, or 16 / 32 / 48: rows are still 64 floats wide, columns >= D are padding that holds values outside the splines'
    // interval (the caller's _pad_rows); the final-layer groups of the padding chunks are skipped
    if ((D != F_D && D != 16 && D != 32 && D != 48) || (hidden != F_H && hidden != F_H / 2 && hidden != F_H / 4) || !fused_bins_ok(K) ||
        num_blocks < 0 || num_blocks > 16)
        return NF_ENOTSUP;
    if (num_layers < 1 || num_layers > F_MAX_LAYERS) return NF_ERANGE;
    if (min_bin_width * K > 1.0 || min_bin_height * K > 1.0) return NF_EINVAL;
    if (B < 0 || (direction != 0 && direction != 1)) return NF_EINVAL;
    if (acc < NF_LD_SUB || acc > NF_LD_ADD) return NF_EINVAL;
    if (B == 0) return NF_OK;
    if (!x || !y || !logdet || !wpacks || !mask_parities) return NF_EFAULT;
    FlowArgs fa;
    fa.parity = 0ull;
    fa.nlayers = num_layers;
    fa.ngroups = (D / 16) * (K / 4);
    for (int l = 0; l < F_MAX_LAYERS; ++l) fa.blob[l] = nullptr;
    for (int l = 0; l < num_layers; ++l) {
        if (!wpacks[l]) return NF_EFAULT;
        if (mask_parities[l] != 0 && mask_parities[l] != 1) return NF_EINVAL;
        fa.blob[l] = (const float *)wpacks[l];
        if (mask_parities[l]) fa.parity |= 1ull << l;
    }
    auto p = make_rqs_params<float>(K, NF_TAILS_LINEAR, tail_bound, 0, 1, 0, 1, min_bin_width, min_bin_height,
                                    min_derivative, sqrt((double)F_H));
    if (g_small_batch && B <= NF_FUSED_SMALL_ROWS)
        return nf_rqs_fused_chain_nw4_(x, y, logdet, &fa, B, hidden, num_blocks, K, &p, direction, fuse_lu, acc, stream);
    return chain_dispatch(x, y, logdet, fa, B, hidden, num_blocks, K, p, direction, fuse_lu, acc, (hipStream_t)stream);
}

extern "C" int nf_rqs_fused_train_fwd(const void *x, const void *h2, void *y, void *logdet, void *cond_out, const void *wpack,
                                      int mask_parity, int64_t B, int D, int hidden, int num_blocks, int K, double tail_bound,
                                      double min_bin_width, double min_bin_height, double min_derivative, int acc,
                                      nf_stream_t stream) {
    if (D != F_D || hidden != F_H || K != F_K || num_blocks < 0 || num_blocks > 16) return NF_ENOTSUP;
    if (B < 0 || (mask_parity != 0 && mask_parity != 1) || acc < NF_LD_SUB || acc > NF_LD_ADD) return NF_EINVAL;
    if (B == 0) return NF_OK;
    if (!x || !h2 || !y || !logdet || !cond_out || !wpack) return NF_EFAULT;
    FlowArgs fa;
    fa.parity = mask_parity ? 1ull : 0ull;
    fa.nlayers = 1;
    fa.ngroups = F_K;
    for (int l = 0; l < F_MAX_LAYERS; ++l) fa.blob[l] = nullptr;
    fa.blob[0] = (const float *)wpack;
    FusedLayout lay;
    lay.nblk = num_blocks;
    auto p = make_rqs_params<float>(K, NF_TAILS_LINEAR, tail_bound, 0, 1, 0, 1, min_bin_width, min_bin_height,
                                    min_derivative, sqrt((double)hidden));
    // + the waves' 6 KB transpose tiles for the row stores (they start in the unused second small buffer)
    const size_t lds = (size_t)(2 * F_STAGE + F_NW * 32 * 64 + lay.small_padded() + F_NW * 1536) * sizeof(float);
    if (lds > 160 * 1024) return NF_ENOTSUP;
    static LdsOptIn opted = {};
    if (opt_in_lds(reinterpret_cast<const void *>(&rqs_fused_kernel<0, false, true>), lds, opted) != NF_OK) return NF_ENOTSUP;
    const int grid = (int)((B + F_ROWS - 1) / F_ROWS);
    hipLaunchKernelGGL((rqs_fused_kernel<0, false, true>), dim3(grid), dim3(F_THREADS), lds, (hipStream_t)stream, (const float *)x,
                       (float *)y, (float *)logdet, fa, B, num_blocks, p, acc, (const float *)h2, (float *)cond_out,
                       (float)(sqrt((double)hidden) / 1.4426950408889634));
    NF_CHECK_LAUNCH();
    return NF_OK;
}

extern "C" int nf_rqs_fused(const void *x, void *y, void *logdet, const void *wpack, int mask_parity, int fuse_lu,
                            int64_t B, int D, int hidden, int num_blocks, int K, double tail_bound, double min_bin_width,
                            double min_bin_height, double min_derivative, int direction, int acc, nf_stream_t stream) {
    const void *packs[1] = {wpack};
    const int par[1] = {mask_parity};
    if (!wpack) return NF_EFAULT;
    return nf_rqs_fused_chain(x, y, logdet, packs, par, 1, fuse_lu, B, D, hidden, num_blocks, K, tail_bound, min_bin_width,
                              min_bin_height, min_derivative, direction, acc, stream);
}
#endif  // !NF_FUSED_SECONDARY
