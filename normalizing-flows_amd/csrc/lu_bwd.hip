// Backward of LULinearPermute's batch side in the density direction (mixing.py:535-563: u = U x[perm], y = L u + b) for
// D = 64 as ONE pass over the rows -- the resblock_bwd.hip scheme at width 64, without masks:
//
//   gu = gy Lm             (= L^T gy per row)            dL  = gy^T u,  db = colsum(gy)
//   gx = gu Up             (= P U^T gu per row)          dUp = gu^T x   (nf_lu_param_grads maps it through perm)
//
// Lm (D, D) = L, Up (D, D) = U with its columns permuted (nf_lu_factors); u = the forward's intermediate.  The separate kernels
// (nf_rows_matvec2 + nf_linear_wgrad_pair) read gy twice and round-trip gu: 112 MB at B = 65 536; here 64 MB.
//
// Workgroup = 4 waves, persistent over 64-row tiles, two workgroups per CU (70 KB of LDS each).  gy / u / x arrive by LDS-DMA
// into row-major tiles of pitch 68 floats.  Input-gradient products: wave (rh, cb) owns rows 32 rh.. x columns 32 cb.., A by
// ds_read_b128 (lane = row, 4 consecutive k), B = the weight slice in 32 registers.  Weight gradients: wave (mb, nb) owns one
// 32 x 32 block of the 64 x 64 output, both operands row-wise ds_read_b32; partial tiles per workgroup, summed in a fixed
// order by nf::wgrad_reduce_kernel.
#include "common.hpp"
#include "fused_common.hpp"

namespace nf {

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

int wgrad_reduce_launch(const float *part, float *dW, float *db, int64_t nW, int M, int chunks, int N, int np, int64_t zpart,
                        int64_t zdW, int64_t zdb, const int *colmap, int Nout, hipStream_t st);       // wgrad.hip

constexpr int LB_R = 64, LB_D = 64, LB_P = 68, LB_TILE = LB_R * LB_P;
constexpr int LB_NI = LB_R * (LB_P / 4) / 64;        // 17 DMA instructions of 64 lanes x 16 B per tile

struct LuBwdArgs {
    const float *gy, *u, *x, *Lm, *Up;
    float *gx;
    float *part;           // [2][grid][64 * 64 + 64]: (dL, db) then (dUp, -)
    int64_t B;
};

#define LB_BARRIER_ALL() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")

__global__ void __launch_bounds__(256, 2)
lu_bwd_kernel(LuBwdArgs a) {
    typedef __attribute__((address_space(3))) void *lds_ptr;
    extern __shared__ __attribute__((aligned(16))) float smem_lb[];
    float *Gt = smem_lb, *Ut = Gt + LB_TILE, *Xt = Ut + LB_TILE, *Dt = Xt + LB_TILE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rh = wid >> 1, cb = wid & 1;          // rows / columns of the input-gradient products; (m, n) block of the weight gradients
    const int grid = gridDim.x;
    const int64_t ntiles = a.B / LB_R;

    unsigned goff[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int s = 64 * (wid + 4 * q) + lane, row = s / 17, c = s - 17 * row;
        goff[q] = (unsigned)((row < LB_R ? row : 0) * LB_D + 4 * (c < 16 ? c : 15));
    }
    auto issue = [&](const float *src, float *tile) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_global_load_lds(src + goff[q], (lds_ptr)(tile + 256 * (wid + 4 * q)), 16, 0, 0);
        if (wid == 0) __builtin_amdgcn_global_load_lds(src + goff[4], (lds_ptr)(tile + 256 * 16), 16, 0, 0);   // instruction 17
    };

    int64_t tile = blockIdx.x;
    if (tile < ntiles) {
        issue(a.gy + tile * (LB_R * LB_D), Gt);
        issue(a.u + tile * (LB_R * LB_D), Ut);
        issue(a.x + tile * (LB_R * LB_D), Xt);
    }
    float WLr[32], WUr[32];
    {
        const int i = lane & 31, hh = lane >> 5;
#pragma unroll
        for (int Q = 0; Q < 8; ++Q)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int k = 8 * Q + 4 * hh + s;
                WLr[4 * Q + s] = a.Lm[k * LB_D + 32 * cb + i];
                WUr[4 * Q + s] = a.Up[k * LB_D + 32 * cb + i];
            }
    }
    f32x16 accL = {0}, accU = {0};
    float bs = 0.f;
    LB_BARRIER_ALL();

    for (; tile < ntiles; tile += grid) {
        const bool more = tile + grid < ntiles;
        int l_ = lane;
        asm volatile("" : "+v"(l_));          // per-tile addresses (see resblock_bwd.hip)
        const int i = l_ & 31, hh = l_ >> 5;
        // ---- gu = gy Lm -> Dt ----
        {
            f32x16 C = {0};
            const float *ap = Gt + (32 * rh + i) * LB_P + 4 * hh;
#pragma unroll
            for (int Q = 0; Q < 8; ++Q) {
                const f32x4 av = *reinterpret_cast<const f32x4 *>(ap + 8 * Q);
#pragma unroll
                for (int s = 0; s < 4; ++s) C = MFMA32(av[s], WLr[4 * Q + s], C);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) Dt[(32 * rh + 8 * (r >> 2) + 4 * hh + (r & 3)) * LB_P + 32 * cb + i] = C[r];
        }
        // ---- dL += gy^T u, db += colsum(gy) ----
        {
            const float *ap = Gt + hh * LB_P + 32 * rh + i, *bp = Ut + hh * LB_P + 32 * cb + i;
#pragma unroll 8
            for (int kp = 0; kp < LB_R / 2; ++kp) {
                const float a0 = ap[kp * 2 * LB_P];
                bs += a0;
                accL = MFMA32(a0, bp[kp * 2 * LB_P], accL);
            }
        }
        LB_BARRIER_ALL();          // Dt complete, x landed; every wave is done with Gt and Ut
        if (more) {
            issue(a.gy + (tile + grid) * (LB_R * LB_D), Gt);
            issue(a.u + (tile + grid) * (LB_R * LB_D), Ut);
        }
        // ---- gx = gu Up ----
        {
            f32x16 C = {0};
            const float *ap = Dt + (32 * rh + i) * LB_P + 4 * hh;
#pragma unroll
            for (int Q = 0; Q < 8; ++Q) {
                const f32x4 av = *reinterpret_cast<const f32x4 *>(ap + 8 * Q);
#pragma unroll
                for (int s = 0; s < 4; ++s) C = MFMA32(av[s], WUr[4 * Q + s], C);
            }
            float *gp = a.gx + (tile * LB_R + 32 * rh + 4 * hh) * LB_D + 32 * cb + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) gp[(8 * (r >> 2) + (r & 3)) * LB_D] = C[r];
        }
        // ---- dUp += gu^T x ----
        {
            const float *ap = Dt + hh * LB_P + 32 * rh + i, *bp = Xt + hh * LB_P + 32 * cb + i;
#pragma unroll 8
            for (int kp = 0; kp < LB_R / 2; ++kp) accU = MFMA32(ap[kp * 2 * LB_P], bp[kp * 2 * LB_P], accU);
        }
        LB_BARRIER_ALL();          // next tile's gy / u landed; every wave is done with Xt and Dt
        if (more) issue(a.x + (tile + grid) * (LB_R * LB_D), Xt);
    }

    constexpr int64_t nW = LB_D * LB_D, stride = nW + LB_D;
    float *oL = a.part + (int64_t)blockIdx.x * stride, *oU = a.part + ((int64_t)grid + blockIdx.x) * stride;
    const int i = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int mm = 32 * rh + (r & 3) + 8 * (r >> 2) + 4 * hh;
        oL[mm * LB_D + 32 * cb + i] = accL[r];
        oU[mm * LB_D + 32 * cb + i] = accU[r];
    }
    bs += __shfl_xor(bs, 32);
    if (cb == 0 && hh == 0) oL[nW + 32 * rh + i] = bs;
    if (cb == 1 && hh == 0) oU[nW + 32 * rh + i] = 0.0f;       // the second problem has no bias; keep the slot defined
}

// The density direction's FORWARD on the same tiles (training keeps u for the backward): u = x UpT (= U x[perm] per row),
// y = u LT + b (= L u + b), logdet (op)= ld_sign * *ld_const.  x tiles by LDS-DMA, both weight slices in registers, u goes to
// LDS (A operand of the second product) and to HBM; 50 MB of traffic per launch at B = 65 536 = its algorithmic bytes.
struct LuFwdArgs {
    const float *x, *UpT, *LT, *bias, *ld_const;
    float *u, *y, *logdet;
    float ld_sign;
    int acc;
    int64_t B;
};

__global__ void __launch_bounds__(256, 2)
lu_fwd_kernel(LuFwdArgs a) {
    typedef __attribute__((address_space(3))) void *lds_ptr;
    extern __shared__ __attribute__((aligned(16))) float smem_lb[];
    float *Xt = smem_lb, *Dt = Xt + LB_TILE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rh = wid >> 1, cb = wid & 1;
    const int grid = gridDim.x;
    const int64_t ntiles = a.B / LB_R;
    unsigned goff[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int s = 64 * (wid + 4 * q) + lane, row = s / 17, c = s - 17 * row;
        goff[q] = (unsigned)((row < LB_R ? row : 0) * LB_D + 4 * (c < 16 ? c : 15));
    }
    auto issue = [&](const float *src, float *tile) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_global_load_lds(src + goff[q], (lds_ptr)(tile + 256 * (wid + 4 * q)), 16, 0, 0);
        if (wid == 0) __builtin_amdgcn_global_load_lds(src + goff[4], (lds_ptr)(tile + 256 * 16), 16, 0, 0);
    };
    int64_t tile = blockIdx.x;
    if (tile < ntiles) issue(a.x + tile * (LB_R * LB_D), Xt);
    float W1r[32], W2r[32], bv;
    {
        const int i = lane & 31, hh = lane >> 5;
#pragma unroll
        for (int Q = 0; Q < 8; ++Q)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int k = 8 * Q + 4 * hh + s;
                W1r[4 * Q + s] = a.UpT[k * LB_D + 32 * cb + i];
                W2r[4 * Q + s] = a.LT[k * LB_D + 32 * cb + i];
            }
        bv = a.bias ? a.bias[32 * cb + i] : 0.0f;
    }
    const float ldv = a.logdet ? a.ld_sign * a.ld_const[0] : 0.0f;
    for (; tile < ntiles; tile += grid) {
        const bool more = tile + grid < ntiles;
        LB_BARRIER_ALL();          // x landed; every wave is done with Dt
        int l_ = lane;
        asm volatile("" : "+v"(l_));
        const int i = l_ & 31, hh = l_ >> 5;
        const int64_t r0 = tile * LB_R + 32 * rh + 4 * hh;
        {
            f32x16 C = {0};
            const float *ap = Xt + (32 * rh + i) * LB_P + 4 * hh;
#pragma unroll
            for (int Q = 0; Q < 8; ++Q) {
                const f32x4 av = *reinterpret_cast<const f32x4 *>(ap + 8 * Q);
#pragma unroll
                for (int s = 0; s < 4; ++s) C = MFMA32(av[s], W1r[4 * Q + s], C);
            }
            float *up = a.u + r0 * LB_D + 32 * cb + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                Dt[(32 * rh + 8 * (r >> 2) + 4 * hh + (r & 3)) * LB_P + 32 * cb + i] = C[r];
                up[(8 * (r >> 2) + (r & 3)) * LB_D] = C[r];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");          // u complete in LDS; every wave is done with Xt
        if (more) issue(a.x + (tile + grid) * (LB_R * LB_D), Xt);
        {
            f32x16 C = {0};
            const float *ap = Dt + (32 * rh + i) * LB_P + 4 * hh;
#pragma unroll
            for (int Q = 0; Q < 8; ++Q) {
                const f32x4 av = *reinterpret_cast<const f32x4 *>(ap + 8 * Q);
#pragma unroll
                for (int s = 0; s < 4; ++s) C = MFMA32(av[s], W2r[4 * Q + s], C);
            }
            float *yp = a.y + r0 * LB_D + 32 * cb + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) yp[(8 * (r >> 2) + (r & 3)) * LB_D] = C[r] + bv;
        }
        if (a.logdet && tid < LB_R) ld_store(a.logdet + tile * LB_R + tid, ldv, a.acc);
    }
}

static int lb_grid(int64_t B) {
    const int64_t nt = B / LB_R;
    return (int)(nt < 512 ? nt : 512);
}

}  // namespace nf

extern "C" int64_t nf_lu_bwd_scratch_floats(int64_t B) {
    using namespace nf;
    if (B < LB_R || B % LB_R) return NF_EINVAL;
    return 2 * (int64_t)lb_grid(B) * ((int64_t)LB_D * LB_D + LB_D) + LB_D;
}

extern "C" int nf_lu_bwd(const void *gy, const void *u, const void *x, const void *Lm, const void *Up, void *gx, void *dL, void *db,
                         void *dUp, void *scratch, int64_t B, int D, nf_stream_t stream) {
    using namespace nf;
    if (D != LB_D || B < LB_R || B % LB_R) return NF_ENOTSUP;
    if (!gy || !u || !x || !Lm || !Up || !gx || !dL || !db || !dUp || !scratch) return NF_EFAULT;
    if (((uintptr_t)gy | (uintptr_t)u | (uintptr_t)x | (uintptr_t)gx) & 15) return NF_EINVAL;
    if ((((uintptr_t)dL ^ (uintptr_t)dUp) | ((uintptr_t)db ^ (uintptr_t)scratch)) & 3) return NF_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int grid = lb_grid(B);
    LuBwdArgs a;
    a.gy = (const float *)gy; a.u = (const float *)u; a.x = (const float *)x;
    a.Lm = (const float *)Lm; a.Up = (const float *)Up;
    a.gx = (float *)gx; a.part = (float *)scratch; a.B = B;
    const size_t lds = (size_t)4 * LB_TILE * sizeof(float);
    static LdsOptIn opted = {};
    if (opt_in_lds(reinterpret_cast<const void *>(&lu_bwd_kernel), lds, opted) != NF_OK) return NF_ENOTSUP;
    hipLaunchKernelGGL(lu_bwd_kernel, dim3(grid), dim3(256), lds, st, a);
    NF_CHECK_LAUNCH();
    const int64_t stride = (int64_t)LB_D * LB_D + LB_D;
    // one reduction for both problems (blockIdx.y); the second has no bias: its (zero) column sums go to 64 spare floats at the
    // end of the scratch
    float *dummy = (float *)scratch + 2 * grid * stride;
    return wgrad_reduce_launch(a.part, (float *)dL, (float *)db, (int64_t)LB_D * LB_D, LB_D, grid, LB_D, 2, grid * stride,
                               (float *)dUp - (float *)dL, dummy - (float *)db, nullptr, 0, st);
}

extern "C" int nf_lu_fwd(const void *x, const void *UpT, const void *LT, const void *bias, void *u, void *y, void *logdet,
                         const void *ld_const, double ld_sign, int acc, int64_t B, int D, nf_stream_t stream) {
    using namespace nf;
    if (D != LB_D || B < LB_R || B % LB_R) return NF_ENOTSUP;
    if (acc < NF_LD_SUB || acc > NF_LD_ADD) return NF_EINVAL;
    if (!x || !UpT || !LT || !u || !y || (logdet && !ld_const)) return NF_EFAULT;
    if (((uintptr_t)x | (uintptr_t)u | (uintptr_t)y) & 15) return NF_EINVAL;
    LuFwdArgs a;
    a.x = (const float *)x; a.UpT = (const float *)UpT; a.LT = (const float *)LT; a.bias = (const float *)bias;
    a.ld_const = (const float *)ld_const; a.u = (float *)u; a.y = (float *)y; a.logdet = (float *)logdet;
    a.ld_sign = (float)ld_sign; a.acc = acc; a.B = B;
    hipLaunchKernelGGL(lu_fwd_kernel, dim3(lb_grid(B)), dim3(256), (size_t)2 * LB_TILE * sizeof(float), (hipStream_t)stream, a);
    NF_CHECK_LAUNCH();
    return NF_OK;
}
