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

// Round 6: the same backward for the COMPOSED layer y = W_d x + b (W_d = L U with permuted columns: what the fused training forward
// rqs_fused_kernel<0, true, 2> multiplies by): ONE product and ONE weight gradient per tile instead of two each, no intermediate u:
//   gx = g W_d  (= W_d^T g per row),   dW_d = g^T x,   db = colsum(g);
// the factors' gradients follow on the parameter side (nf_lu_param_grads_composed: dL = dM U^T, dU = L^T dM with
// dM[:, j] = dW_d[:, perm[j]] -- 64^3 products once per layer instead of two more passes over the batch).
// Reads g and x (34 MB at B = 65 536), writes gx (17 MB): HBM-bound; the tiles are double-buffered (one barrier per tile: the next
// tile's 2 x 17 DMA instructions are in flight while this one multiplies), two workgroups per CU.
struct LuBwdCArgs {
    const float *g, *x, *Wd;
    float *gx;
    float *part;           // [grid][64 * 64 + 64]: (dW_d, db)
    int64_t B;
};

__global__ void __launch_bounds__(256, 2)
lu_bwd_c_kernel(LuBwdCArgs a) {
    typedef __attribute__((address_space(3))) void *lds_ptr;
    extern __shared__ __attribute__((aligned(16))) float smem_lb[];
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
        if (wid == 0) __builtin_amdgcn_global_load_lds(src + goff[4], (lds_ptr)(tile + 256 * 16), 16, 0, 0);   // instruction 17
    };
    int64_t tile = blockIdx.x;
    if (tile < ntiles) {
        issue(a.g + tile * (LB_R * LB_D), smem_lb);
        issue(a.x + tile * (LB_R * LB_D), smem_lb + LB_TILE);
    }
    float Wr[32];
    {
        const int i = lane & 31, hh = lane >> 5;
#pragma unroll
        for (int Q = 0; Q < 8; ++Q)
#pragma unroll
            for (int s = 0; s < 4; ++s) Wr[4 * Q + s] = a.Wd[(8 * Q + 4 * hh + s) * LB_D + 32 * cb + i];
    }
    f32x16 acc = {0};
    float bs = 0.f;
    int buf = 0;
    for (; tile < ntiles; tile += grid, buf ^= 1) {
        float *Gt = smem_lb + buf * 2 * LB_TILE, *Xt = Gt + LB_TILE;
        LB_BARRIER_ALL();          // this tile landed; every wave is done with the other buffer set
        if (tile + grid < ntiles) {
            float *Gn = smem_lb + (buf ^ 1) * 2 * LB_TILE;
            issue(a.g + (tile + grid) * (LB_R * LB_D), Gn);
            issue(a.x + (tile + grid) * (LB_R * LB_D), Gn + LB_TILE);
        }
        int l_ = lane;
        asm volatile("" : "+v"(l_));
        const int i = l_ & 31, hh = l_ >> 5;
        // ---- gx = g W_d ----
        {
            f32x16 C = {0};
            const float *ap = Gt + (32 * rh + i) * LB_P + 4 * hh;
#pragma unroll
            for (int Q = 0; Q < 8; ++Q) {
                const f32x4 av = *reinterpret_cast<const f32x4 *>(ap + 8 * Q);
#pragma unroll
                for (int s = 0; s < 4; ++s) C = MFMA32(av[s], Wr[4 * Q + s], C);
            }
            float *gp = a.gx + (tile * LB_R + 32 * rh + 4 * hh) * LB_D + 32 * cb + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) gp[(8 * (r >> 2) + (r & 3)) * LB_D] = C[r];
        }
        // ---- dW_d += g^T x, db += colsum(g) ----
        {
            const float *ap = Gt + hh * LB_P + 32 * rh + i, *bp = Xt + hh * LB_P + 32 * cb + i;
#pragma unroll 8
            for (int kp = 0; kp < LB_R / 2; ++kp) {
                const float a0 = ap[kp * 2 * LB_P];
                bs += a0;
                acc = MFMA32(a0, bp[kp * 2 * LB_P], acc);
            }
        }
    }
    constexpr int64_t nW = LB_D * LB_D, stride = nW + LB_D;
    float *o = a.part + (int64_t)blockIdx.x * stride;
    const int i = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[(32 * rh + (r & 3) + 8 * (r >> 2) + 4 * hh) * LB_D + 32 * cb + i] = acc[r];
    bs += __shfl_xor(bs, 32);
    if (cb == 0 && hh == 0) o[nW + 32 * rh + i] = bs;
}

// dW_d (reduced) -> the packed parameter gradients of LULinearPermute (mixing.py:402-473 under autograd): W_d[:, perm[j]] =
// (L U)[:, j], so dM[r][j] = dW_d[r][perm[j]];  dL = dM U^T (strictly lower part -> g_lower), dU = L^T dM (strictly upper part ->
// g_upper; diagonal + gl_sum / diag through softplus' -> g_udiag), gl_sum = sum of the (B) log-det cotangent.
// 16 workgroups x 256 threads, one output entry per thread (row r = 4 w + tid / 64, column c = tid % 64): L, U, dM in LDS (pitch 65:
// the column walks are conflict-free), BOTH 64-term dot products of an entry evaluated over the full k range without a branch -- the
// triangular zeros of L and U do the masking -- so the LDS reads pipeline (the first version walked the triangular ranges in
// divergent, latency-bound loops in ONE workgroup: 29 us per layer).  Lm / Um: the dense factors nf_lu_factors[_multi] assembled
// this step.
__global__ void __launch_bounds__(256)
lu_c_param_grads_kernel(const float *__restrict__ dWd, const float *__restrict__ Lm, const float *__restrict__ Um,
                        const int64_t *__restrict__ perm, const float *__restrict__ gld, int64_t B,
                        const float *__restrict__ udiag_raw, float eps, float *__restrict__ g_lower, float *__restrict__ g_upper,
                        float *__restrict__ g_udiag) {
    constexpr int D = LB_D, N = D * D, P = D + 1;
    __shared__ float Ls[D * P], Us[D * P], Ms[D * P];
    __shared__ float sred[16];
    const int tid = threadIdx.x;
    const int r = 4 * blockIdx.x + (tid >> 6), c = tid & 63;
    float gl = 0.0f;
    // (every workgroup owns four diagonal entries, so every one of them sums the cotangent -- in the same order: the same value)
    if (gld) {
        float p0 = 0.0f, p1 = 0.0f, p2 = 0.0f, p3 = 0.0f;
        const int64_t B4 = (reinterpret_cast<uintptr_t>(gld) & 15) == 0 ? B / 4 : 0;
        const f32x4 *g4 = reinterpret_cast<const f32x4 *>(gld);
#pragma unroll 8
        for (int64_t b = tid; b < B4; b += 256) {
            const f32x4 v = g4[b];
            p0 += v[0]; p1 += v[1]; p2 += v[2]; p3 += v[3];
        }
        for (int64_t b = 4 * B4 + tid; b < B; b += 256) p0 += gld[b];
        gl = block_sum((p0 + p1) + (p2 + p3), sred);
    }
    for (int i = tid; i < N; i += 256) {
        const int rr = i / D, cc = i - rr * D;
        Ls[rr * P + cc] = Lm[i];
        Us[rr * P + cc] = Um[i];
        Ms[rr * P + cc] = dWd[rr * D + (int)perm[cc]];
    }
    __syncthreads();
    // dL[r][c] = sum_k dM[r][k] U[c][k] (U[c][k] = 0 for k < c);  dU[r][c] = sum_k L[k][r] dM[k][c] (L[k][r] = 0 for k < r, 1 at k = r)
    float sl0 = 0.0f, sl1 = 0.0f, su0 = 0.0f, su1 = 0.0f;
#pragma unroll
    for (int k = 0; k < D; k += 2) {
        sl0 = fmaf(Ms[r * P + k], Us[c * P + k], sl0);
        sl1 = fmaf(Ms[r * P + k + 1], Us[c * P + k + 1], sl1);
        su0 = fmaf(Ls[k * P + r], Ms[k * P + c], su0);
        su1 = fmaf(Ls[(k + 1) * P + r], Ms[(k + 1) * P + c], su1);
    }
    const float sl = sl0 + sl1, su = su0 + su1;
    if (c < r) g_lower[r * (r - 1) / 2 + c] = sl;
    else if (c > r) g_upper[r * (D - 1) - r * (r - 1) / 2 + (c - r - 1)] = su;
    else {
        const float u = udiag_raw[r], d = softplus(u) + eps;
        g_udiag[r] = (su + gl / d) * (u > 20.0f ? 1.0f : sigmoid(u));
    }
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

extern "C" int64_t nf_lu_bwd_composed_scratch_floats(int64_t B) {
    using namespace nf;
    if (B < LB_R || B % LB_R) return NF_EINVAL;
    return (int64_t)lb_grid(B) * ((int64_t)LB_D * LB_D + LB_D);
}

// Backward of the composed LULinearPermute (density direction; the layer as the fused training forward evaluates it: y = W_d x + b),
// D = 64, one pass over the rows: gx (B, 64) = g W_d, dWd (64, 64) = g^T x, db (64) = colsum(g); W_d (64, 64) row-major from
// nf_lu_pack_train_multi.  B a multiple of 64; scratch: nf_lu_bwd_composed_scratch_floats(B).  Deterministic.
// ... the pass over the rows alone: partial tiles [nf_lu_bwd_composed_grid(B)][64 * 64 + 64] (dW_d then the column sums of g) stay
// in `scratch` for the caller's reduction (nf_pair_train_bwd: the layer's one reduction launch).
extern "C" int nf_lu_bwd_composed_grid(int64_t B) {
    using namespace nf;
    if (B < LB_R || B % LB_R) return NF_EINVAL;
    return lb_grid(B);
}
extern "C" int nf_lu_bwd_composed_partials(const void *g, const void *x, const void *Wd, void *gx, void *scratch, int64_t B, int D,
                                           nf_stream_t stream) {
    using namespace nf;
    if (D != LB_D || B < LB_R || B % LB_R) return NF_ENOTSUP;
    if (!g || !x || !Wd || !gx || !scratch) return NF_EFAULT;
    if (((uintptr_t)g | (uintptr_t)x | (uintptr_t)gx) & 15) return NF_EINVAL;
    LuBwdCArgs a;
    a.g = (const float *)g; a.x = (const float *)x; a.Wd = (const float *)Wd; a.gx = (float *)gx; a.part = (float *)scratch; a.B = B;
    const size_t lds = (size_t)4 * LB_TILE * sizeof(float);
    static LdsOptIn opted = {};
    if (opt_in_lds(reinterpret_cast<const void *>(&lu_bwd_c_kernel), lds, opted) != NF_OK) return NF_ENOTSUP;
    hipLaunchKernelGGL(lu_bwd_c_kernel, dim3(lb_grid(B)), dim3(256), lds, (hipStream_t)stream, a);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

extern "C" int nf_lu_bwd_composed(const void *g, const void *x, const void *Wd, void *gx, void *dWd, void *db, void *scratch, int64_t B,
                                  int D, nf_stream_t stream) {
    using namespace nf;
    if (!dWd || !db) return NF_EFAULT;
    const int rc = nf_lu_bwd_composed_partials(g, x, Wd, gx, scratch, B, D, stream);
    if (rc != NF_OK) return rc;
    return wgrad_reduce_launch((const float *)scratch, (float *)dWd, (float *)db, (int64_t)LB_D * LB_D, LB_D, lb_grid(B), LB_D, 1, 0, 0, 0,
                               nullptr, 0, (hipStream_t)stream);
}

// (g_lower, g_upper, g_udiag) of LULinearPermute from the composed matrix's gradient dWd (nf_lu_bwd_composed), the dense factors
// Lm, Um (nf_lu_factors[_multi]: out, out + D * D), the permutation and the (B) log-det cotangent gld (summed here; may be NULL).
extern "C" int nf_lu_param_grads_composed(const void *dWd, const void *Lm, const void *Um, const int64_t *perm, const void *gld,
                                          int64_t B, const void *unconstrained_upper_diag, double eps, void *g_lower, void *g_upper,
                                          void *g_udiag, int D, nf_stream_t stream) {
    using namespace nf;
    if (D != LB_D) return NF_ENOTSUP;
    if (!dWd || !Lm || !Um || !perm || !unconstrained_upper_diag || !g_lower || !g_upper || !g_udiag) return NF_EFAULT;
    hipLaunchKernelGGL(lu_c_param_grads_kernel, dim3(16), dim3(256), 0, (hipStream_t)stream, (const float *)dWd, (const float *)Lm,
                       (const float *)Um, perm, (const float *)gld, gld ? B : 0, (const float *)unconstrained_upper_diag, (float)eps,
                       (float *)g_lower, (float *)g_upper, (float *)g_udiag);
    NF_CHECK_LAUNCH();
    return NF_OK;
}
