// inv1x1.hip -- Glow Invertible1x1Conv (normflows/flows/mixing.py:88-133).
//
// nf_inv1x1_assemble: single-workgroup kernel building W (C x C) from (P, L, U, sign_S, log_S) in LDS; for the
//   sampling direction the two triangular factors are inverted by substitution in fp64 (the reference calls
//   torch.inverse on .double() copies every forward, mixing.py:94-101), rounded to the working dtype and
//   multiplied as (U^-1 L^-1) P^T in that dtype, like the reference.
// nf_inv1x1_conv: per-pixel C x C mat-vec with W^T held in LDS (broadcast 16-B reads); lanes run along the
//   pixel axis so every channel plane is read and written with unit stride.  Algorithmic HBM bytes: one read
//   and one write of z (the ceil(C/8) re-reads of a pixel column hit L1/L2).
#include "common.hpp"

namespace nf {

template <typename T>
__device__ __forceinline__ void inv1x1_assemble_body(const T *__restrict__ P, const T *__restrict__ L, const T *__restrict__ U,
                                                     const T *__restrict__ sign_S, const T *__restrict__ log_S, T *__restrict__ W,
                                                     T *__restrict__ logdet_unit, int C, int inverse) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int n = C * C;
    double *Dd = reinterpret_cast<double *>(smem_raw);  // fp64 triangular factor
    double *Ed = Dd + n;                                // fp64 inverse of it
    T *Lt = reinterpret_cast<T *>(Ed + n);              // L' or L'^-1 in T
    T *Ut = Lt + n;                                     // U' or U'^-1 in T
    T *Tm = Ut + n;                                     // temp product
    T *Pt = reinterpret_cast<T *>(Dd);                  // P staged in LDS (round 5: the products' inner loops were C dependent global
    __shared__ T sred[16];                              // loads per output entry, 18 us per call at C = 48); it shares the fp64 factor's
    const int tid = threadIdx.x;                        // space: unused in the density direction, free again before the last product
    if (!inverse)
        for (int i = tid; i < n; i += 256) Pt[i] = P[i];

    T part = T(0);
    for (int i = tid; i < C; i += 256) part += log_S[i];
    const T ls = block_sum(part, sred);
    if (tid == 0 && logdet_unit) *logdet_unit = inverse ? -ls : ls;

    // L' = tril(L,-1) + I ; U' = triu(U,1) + diag(sign_S * exp(log_S))   (mixing.py:90-93)
    for (int i = tid; i < n; i += 256) {
        const int r = i / C, c = i - r * C;
        Lt[i] = c < r ? L[i] : (c == r ? T(1) : T(0));
        Ut[i] = c > r ? U[i] : (c == r ? sign_S[r] * M<T>::exp(log_S[r]) : T(0));
    }
    __syncthreads();

    if (!inverse) {
        // W = (P @ L') @ U'.  P comes from torch.lu_unpack (mixing.py:76-80): rows with a single 1 -- then P @ L' is a row gather (the
        // same values bit for bit: the product only adds exact zeros); any other P takes the product.  U' is upper triangular: the
        // second product stops at the diagonal (C^3 LDS-bound multiply-adds on 256 threads were 16 us per call at C = 48).
        __shared__ int perm[128], one_hot;         // (C <= 75: the LDS bound of launch_assemble)
        if (tid == 0) one_hot = 1;
        __syncthreads();
        for (int r = tid; r < C; r += 256) {
            int pk = -1, cnt = 0;
            for (int k = 0; k < C; ++k) {
                const T v = Pt[r * C + k];
                if (v != T(0)) { ++cnt; pk = v == T(1) ? k : -1; }
            }
            perm[r] = pk;
            if (cnt != 1 || pk < 0) one_hot = 0;
        }
        __syncthreads();
        const bool gather = one_hot != 0;
        for (int i = tid; i < n; i += 256) {
            const int r = i / C, c = i - r * C;
            T a = T(0);
            if (gather) a = Lt[perm[r] * C + c];
            else
                for (int k = 0; k < C; ++k) a += Pt[r * C + k] * Lt[k * C + c];
            Tm[i] = a;
        }
        __syncthreads();
        for (int i = tid; i < n; i += 256) {
            const int r = i / C, c = i - r * C;
            T a = T(0);
            for (int k = 0; k <= c; ++k) a += Tm[r * C + k] * Ut[k * C + c];
            W[i] = a;
        }
        return;
    }

    // ---- L'^-1 in fp64: thread c solves L' x = e_c (forward substitution) ----
    for (int i = tid; i < n; i += 256) { Dd[i] = (double)Lt[i]; Ed[i] = 0.0; }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        for (int r = c; r < C; ++r) {
            double a = (r == c) ? 1.0 : 0.0;
            for (int k = c; k < r; ++k) a -= Dd[r * C + k] * Ed[k * C + c];
            Ed[r * C + c] = a / Dd[r * C + r];
        }
    }
    __syncthreads();
    for (int i = tid; i < n; i += 256) Lt[i] = (T)Ed[i];
    __syncthreads();
    // ---- U'^-1 in fp64: thread c solves U' x = e_c (back substitution) ----
    for (int i = tid; i < n; i += 256) { Dd[i] = (double)Ut[i]; Ed[i] = 0.0; }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        for (int r = c; r >= 0; --r) {
            double a = (r == c) ? 1.0 : 0.0;
            for (int k = r + 1; k <= c; ++k) a -= Dd[r * C + k] * Ed[k * C + c];
            Ed[r * C + c] = a / Dd[r * C + r];
        }
    }
    __syncthreads();
    for (int i = tid; i < n; i += 256) { Ut[i] = (T)Ed[i]; Pt[i] = P[i]; }       // (Dd is done: P moves in)
    __syncthreads();
    // ---- W = (U'^-1 @ L'^-1) @ P^T ----
    for (int i = tid; i < n; i += 256) {
        const int r = i / C, c = i - r * C;
        T a = T(0);
        for (int k = 0; k < C; ++k) a += Ut[r * C + k] * Lt[k * C + c];
        Tm[i] = a;
    }
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        const int r = i / C, c = i - r * C;
        T a = T(0);
        for (int k = 0; k < C; ++k) a += Tm[r * C + k] * Pt[c * C + k];
        W[i] = a;
    }
}

template <typename T>
__global__ void __launch_bounds__(256)
inv1x1_assemble_kernel(const T *__restrict__ P, const T *__restrict__ L, const T *__restrict__ U,
                       const T *__restrict__ sign_S, const T *__restrict__ log_S, T *__restrict__ W,
                       T *__restrict__ logdet_unit, int C, int inverse) {
    inv1x1_assemble_body<T>(P, L, U, sign_S, log_S, W, logdet_unit, C, inverse);
}

// Round 6 (late): the matrices of up to 32 Invertible1x1Convs of one size in ONE launch, workgroup b = layer b -- a Glow level's 32
// blocks assemble their W from parameters only, so the training step need not spend one 4-30 us single-workgroup launch per block on
// it (96 per step of BASELINE configs[3]).  Density direction (inverse = 0) only.
constexpr int I1_MULTI = 32;
struct Inv1x1AssembleMulti {
    const void *P[I1_MULTI], *L[I1_MULTI], *U[I1_MULTI], *sign_S[I1_MULTI], *log_S[I1_MULTI];
    void *W[I1_MULTI], *ld[I1_MULTI];
};

template <typename T>
__global__ void __launch_bounds__(256)
inv1x1_assemble_multi_kernel(Inv1x1AssembleMulti m, int C) {
    const int b = blockIdx.x;
    inv1x1_assemble_body<T>((const T *)m.P[b], (const T *)m.L[b], (const T *)m.U[b], (const T *)m.sign_S[b], (const T *)m.log_S[b],
                            (T *)m.W[b], (T *)m.ld[b], C, 0);
}

constexpr int OT = 8;  // output channels per lane

template <typename T>
__global__ void __launch_bounds__(256)
inv1x1_conv_kernel(const T *__restrict__ z, const T *__restrict__ W, const T *__restrict__ logdet_unit,
                   T *__restrict__ y, T *__restrict__ logdet_scalar, T *__restrict__ logdet, int64_t B, int C,
                   int64_t HW, int acc, const T *__restrict__ obias, int wt) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *Wt = reinterpret_cast<T *>(smem_raw);  // [c][Cp] : Wt[c][o] = W[o][c], row pitch Cp = roundup(C, OT)
    const int Cp = (C + OT - 1) / OT * OT;
    for (int i = threadIdx.x; i < C * Cp; i += blockDim.x) {
        const int c = i / Cp, o = i - c * Cp;
        Wt[i] = o < C ? (wt ? W[c * C + o] : W[o * C + c]) : T(0);      // wt: the product with W^T (the backward's gz = W^T gy)
    }
    __syncthreads();
    const int64_t npix = B * HW;
    const int otiles = Cp / OT;
    const int64_t nwork = npix * otiles;
    for (int64_t wi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; wi < nwork; wi += (int64_t)gridDim.x * blockDim.x) {
        const int ot = (int)(wi / npix);
        const int64_t pix = wi - (int64_t)ot * npix;
        const int64_t b = pix / HW, p = pix - b * HW;
        const T *zb = z + b * (int64_t)C * HW + p;
        T a[OT];
#pragma unroll
        for (int j = 0; j < OT; ++j) a[j] = (obias && ot * OT + j < C) ? obias[ot * OT + j] : T(0);
        const T *wrow = Wt + ot * OT;
        int c = 0;
        for (; c + 4 <= C; c += 4) {         // four channel loads in flight (round 6: the loop was one dependent load per iteration)
            const T z0 = zb[(int64_t)c * HW], z1 = zb[(int64_t)(c + 1) * HW], z2 = zb[(int64_t)(c + 2) * HW], z3 = zb[(int64_t)(c + 3) * HW];
#pragma unroll
            for (int j = 0; j < OT; ++j) a[j] += wrow[c * Cp + j] * z0;
#pragma unroll
            for (int j = 0; j < OT; ++j) a[j] += wrow[(c + 1) * Cp + j] * z1;
#pragma unroll
            for (int j = 0; j < OT; ++j) a[j] += wrow[(c + 2) * Cp + j] * z2;
#pragma unroll
            for (int j = 0; j < OT; ++j) a[j] += wrow[(c + 3) * Cp + j] * z3;
        }
        for (; c < C; ++c) {
            const T zc = zb[(int64_t)c * HW];
#pragma unroll
            for (int j = 0; j < OT; ++j) a[j] += wrow[c * Cp + j] * zc;
        }
        T *yb = y + b * (int64_t)C * HW + p;
#pragma unroll
        for (int j = 0; j < OT; ++j) {
            const int o = ot * OT + j;
            if (o < C) yb[(int64_t)o * HW] = a[j];
        }
    }
    const T ldv = logdet_unit ? (*logdet_unit) * (T)HW : T(0);
    if (blockIdx.x == 0 && threadIdx.x == 0 && logdet_scalar) *logdet_scalar = ldv;
    if (logdet)
        for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < B; r += (int64_t)gridDim.x * blockDim.x)
            ld_store(logdet + r, ldv, acc);
}

template <typename T>
static int launch_assemble(const void *P, const void *L, const void *U, const void *sign_S, const void *log_S, void *W,
                           void *logdet_unit, int C, int inverse, hipStream_t st) {
    const size_t lds = (size_t)C * C * (2 * sizeof(double) + 3 * sizeof(T));
    if (lds > 158 * 1024) return NF_ENOTSUP;
    static LdsOptIn opted = {};
    if (opt_in_lds(reinterpret_cast<const void *>(&inv1x1_assemble_kernel<T>), lds, opted) != NF_OK) return NF_ENOTSUP;
    hipLaunchKernelGGL(inv1x1_assemble_kernel<T>, dim3(1), dim3(256), lds, st, (const T *)P, (const T *)L, (const T *)U,
                       (const T *)sign_S, (const T *)log_S, (T *)W, (T *)logdet_unit, C, inverse);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

template <typename T>
static int launch_conv(const void *z, const void *W, const void *logdet_unit, void *y, void *logdet_scalar, void *logdet,
                       int64_t B, int C, int64_t HW, int acc, hipStream_t st, const void *obias = nullptr, int wt = 0) {
    const int Cp = (C + OT - 1) / OT * OT;
    const size_t lds = (size_t)C * Cp * sizeof(T);
    if (lds > 160 * 1024) return NF_ENOTSUP;
    static LdsOptIn opted = {};
    if (opt_in_lds(reinterpret_cast<const void *>(&inv1x1_conv_kernel<T>), lds, opted) != NF_OK) return NF_ENOTSUP;
    const int64_t nwork = B * HW * (Cp / OT);
    const int grid = grid_for(nwork > 0 ? nwork : 1, 256);
    hipLaunchKernelGGL(inv1x1_conv_kernel<T>, dim3(grid), dim3(256), lds, st, (const T *)z, (const T *)W,
                       (const T *)logdet_unit, (T *)y, (T *)logdet_scalar, (T *)logdet, B, C, HW, acc, (const T *)obias, wt);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

}  // namespace nf

extern "C" int nf_inv1x1_assemble(const void *P, const void *L, const void *U, const void *sign_S, const void *log_S,
                                  void *W, void *logdet_unit, int C, int inverse, int dtype, nf_stream_t stream) {
    if (C < 1 || (inverse != 0 && inverse != 1)) return NF_EINVAL;
    if (!P || !L || !U || !sign_S || !log_S || !W) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == NF_F32) return nf::launch_assemble<float>(P, L, U, sign_S, log_S, W, logdet_unit, C, inverse, st);
    if (dtype == NF_F64) return nf::launch_assemble<double>(P, L, U, sign_S, log_S, W, logdet_unit, C, inverse, st);
    return NF_ENOTSUP;
}

// nf_inv1x1_assemble (density direction, float32) for n layers of one size C in ceil(n / 32) launches: P, L, U, sign_S, log_S, W,
// logdet_unit = HOST arrays of n device pointers (mixing.py:88-104 per layer).
extern "C" int nf_inv1x1_assemble_multi(const void *const *P, const void *const *L, const void *const *U, const void *const *sign_S,
                                        const void *const *log_S, void *const *W, void *const *logdet_unit, int n, int C,
                                        nf_stream_t stream) {
    if (n < 0 || C < 1) return NF_EINVAL;
    if (n == 0) return NF_OK;
    if (!P || !L || !U || !sign_S || !log_S || !W || !logdet_unit) return NF_EFAULT;
    for (int i = 0; i < n; ++i)
        if (!P[i] || !L[i] || !U[i] || !sign_S[i] || !log_S[i] || !W[i]) return NF_EFAULT;
    const size_t lds = (size_t)C * C * (2 * sizeof(double) + 3 * sizeof(float));
    if (lds > 158 * 1024) return NF_ENOTSUP;
    static nf::LdsOptIn opted = {};
    if (nf::opt_in_lds(reinterpret_cast<const void *>(&nf::inv1x1_assemble_multi_kernel<float>), lds, opted) != NF_OK) return NF_ENOTSUP;
    for (int i0 = 0; i0 < n; i0 += nf::I1_MULTI) {
        const int m = n - i0 < nf::I1_MULTI ? n - i0 : nf::I1_MULTI;
        nf::Inv1x1AssembleMulti t = {};
        for (int i = 0; i < m; ++i) {
            t.P[i] = P[i0 + i]; t.L[i] = L[i0 + i]; t.U[i] = U[i0 + i]; t.sign_S[i] = sign_S[i0 + i]; t.log_S[i] = log_S[i0 + i];
            t.W[i] = W[i0 + i]; t.ld[i] = logdet_unit[i0 + i];
        }
        hipLaunchKernelGGL(nf::inv1x1_assemble_multi_kernel<float>, dim3(m), dim3(256), lds, (hipStream_t)stream, t, C);
        NF_CHECK_LAUNCH();
    }
    return NF_OK;
}

extern "C" int nf_inv1x1_conv(const void *z, const void *W, const void *logdet_unit, void *y, void *logdet_scalar,
                              void *logdet, int64_t B, int C, int64_t HW, int acc, int dtype, nf_stream_t stream) {
    if (B < 0 || C < 1 || HW < 1 || acc < NF_LD_SUB || acc > NF_LD_ADD) return NF_EINVAL;
    if (!W) return NF_EFAULT;
    if (B > 0 && (!z || !y)) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == NF_F32) return nf::launch_conv<float>(z, W, logdet_unit, y, logdet_scalar, logdet, B, C, HW, acc, st);
    if (dtype == NF_F64) return nf::launch_conv<double>(z, W, logdet_unit, y, logdet_scalar, logdet, B, C, HW, acc, st);
    return NF_ENOTSUP;
}

// y = W^T z per pixel (no log-det): the input gradient of the 1x1 convolution under autograd (mixing.py:106-133 through
// loss.backward()) without a transposed copy of W per call (round 6: 96 tiny transposes per step of config 4).
extern "C" int nf_inv1x1_conv_t(const void *z, const void *W, void *y, int64_t B, int C, int64_t HW, int dtype, nf_stream_t stream) {
    if (B < 0 || C < 1 || HW < 1) return NF_EINVAL;
    if (!W) return NF_EFAULT;
    if (B > 0 && (!z || !y)) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == NF_F32) return nf::launch_conv<float>(z, W, nullptr, y, nullptr, nullptr, B, C, HW, 0, st, nullptr, 1);
    if (dtype == NF_F64) return nf::launch_conv<double>(z, W, nullptr, y, nullptr, nullptr, B, C, HW, 0, st, nullptr, 1);
    return NF_ENOTSUP;
}

// y = W z + bias per pixel: the 1x1 convolution with the neighbouring ActNorm folded into W and bias by the caller.
extern "C" int nf_inv1x1_conv_affine(const void *z, const void *W, const void *bias, const void *logdet_unit, void *y,
                                     void *logdet_scalar, void *logdet, int64_t B, int C, int64_t HW, int acc, int dtype,
                                     nf_stream_t stream) {
    if (B < 0 || C < 1 || HW < 1 || acc < NF_LD_SUB || acc > NF_LD_ADD) return NF_EINVAL;
    if (!W) return NF_EFAULT;
    if (B > 0 && (!z || !y)) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == NF_F32)
        return nf::launch_conv<float>(z, W, logdet_unit, y, logdet_scalar, logdet, B, C, HW, acc, st, bias);
    if (dtype == NF_F64)
        return nf::launch_conv<double>(z, W, logdet_unit, y, logdet_scalar, logdet, B, C, HW, acc, st, bias);
    return NF_ENOTSUP;
}
