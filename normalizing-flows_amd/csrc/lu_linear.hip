// lu_linear.hip -- LULinearPermute (normflows/flows/mixing.py:535-563, :229-244, :402-473, :514-532).
//
// One launch = gather-permute + two triangular mat-vecs (density) or two triangular solves (sample) + bias +
// the batch-constant log-det, replacing ~35 eager launches per layer of the reference.
//
// Workgroup = 256 lanes = 256 samples.  L (strictly lower, unit diagonal implied) and U (upper incl. diagonal)
// are assembled from the packed parameter vectors into ONE D*D LDS matrix; every lane reads the same word per
// step (LDS broadcast).  Each lane keeps its sample's vector as one LDS column (pitch 256 words, so a wave
// touches 64 consecutive banks).  The permutation is applied while the row is scattered into the column.
// HBM traffic per sample: D*4 in + D*4 out + 8 (log-det rmw); the weights are read once per workgroup.
#include "common.hpp"

namespace nf {

template <typename T>
__global__ void __launch_bounds__(256)
lu_linear_permute_kernel(const T *__restrict__ x, T *__restrict__ y, T *__restrict__ logdet,
                         const int64_t *__restrict__ perm, const T *__restrict__ lower_entries,
                         const T *__restrict__ upper_entries, const T *__restrict__ udiag_raw,
                         const T *__restrict__ bias, int64_t B, int D, T eps, int direction, int acc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T *sLU = reinterpret_cast<T *>(smem_raw);  // D*D : [r][c] = L[r][c] (c<r), U[r][c] (c>=r)
    T *sb = sLU + (size_t)D * D;               // D
    T *sred = sb + D;                          // 16
    const int NT = blockDim.x;                 // lanes (= samples) per workgroup: 256, 128 or 64
    T *sv = sred + 16;                         // D * NT : column-per-lane vectors
    int *sinv = reinterpret_cast<int *>(sv + (size_t)D * NT);  // D : inverse permutation

    const int tid = threadIdx.x;
    // ---- assemble L, U (mixing.py:402-412) ----
    for (int i = tid; i < D * D; i += NT) {
        const int r = i / D, c = i - r * D;
        T v;
        if (c < r) v = lower_entries[(size_t)r * (r - 1) / 2 + c];
        else if (c == r) v = softplus(udiag_raw[r]) + eps;
        else v = upper_entries[(size_t)r * (D - 1) - (size_t)r * (r - 1) / 2 + (c - r - 1)];
        sLU[i] = v;
    }
    for (int i = tid; i < D; i += NT) { sb[i] = bias[i]; sinv[(int)perm[i]] = i; }
    // logabsdet = sum log(upper_diag) (mixing.py:514-532)
    T part = T(0);
    for (int i = tid; i < D; i += NT) part += M<T>::log(softplus(udiag_raw[i]) + eps);
    T lad = block_sum(part, sred);
    if (direction) lad = -lad;
    __syncthreads();

    const int64_t ntiles = (B + NT - 1) / NT;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t b = tile * NT + tid;
        if (b < B) {
            T *v = sv + tid;  // element j at v[j*256]
            const T *row = x + b * D;
            T *out = y + b * D;
            if (direction == 0) {
                // density (LULinearPermute.inverse): t = x[perm]; u = U t; y = L u + bias
                for (int c = 0; c < D; ++c) v[(size_t)sinv[c] * NT] = row[c];  // t_j = x[perm[j]]
                // u_i = sum_{j>=i} U[i][j] t_j : needs t_j for j >= i only, ascending i is safe in place
                for (int i = 0; i < D; ++i) {
                    T a = T(0);
                    const T *Ui = sLU + (size_t)i * D;
                    for (int j = i; j < D; ++j) a += Ui[j] * v[(size_t)j * NT];
                    v[(size_t)i * NT] = a;
                }
                // y_i = u_i + sum_{j<i} L[i][j] u_j + bias_i : descending i is safe in place
                for (int i = D - 1; i >= 0; --i) {
                    T a = T(0);
                    const T *Li = sLU + (size_t)i * D;
                    for (int j = 0; j < i; ++j) a += Li[j] * v[(size_t)j * NT];
                    out[i] = (a + v[(size_t)i * NT]) + sb[i];
                }
            } else {
                // sample (LULinearPermute.forward): solve L u = x - bias, solve U t = u, y[perm[j]] = t_j
                for (int c = 0; c < D; ++c) v[(size_t)c * NT] = row[c] - sb[c];
                for (int i = 0; i < D; ++i) {
                    T a = v[(size_t)i * NT];
                    const T *Li = sLU + (size_t)i * D;
                    for (int j = 0; j < i; ++j) a -= Li[j] * v[(size_t)j * NT];
                    v[(size_t)i * NT] = a;
                }
                for (int i = D - 1; i >= 0; --i) {
                    T a = v[(size_t)i * NT];
                    const T *Ui = sLU + (size_t)i * D;
                    for (int j = i + 1; j < D; ++j) a -= Ui[j] * v[(size_t)j * NT];
                    v[(size_t)i * NT] = a / Ui[i];
                }
                for (int c = 0; c < D; ++c) out[c] = v[(size_t)sinv[c] * NT];
            }
            ld_store(logdet + b, lad, acc);
        }
    }
}

template <typename T>
static int launch_lu(const void *x, void *y, void *logdet, const int64_t *perm, const void *lo, const void *up,
                     const void *ud, const void *bias, int64_t B, int D, double eps, int direction, int acc,
                     hipStream_t st) {
    int NT = 256;
    auto lds_bytes = [&](int nt) { return ((size_t)D * D + D + 16 + (size_t)D * nt) * sizeof(T) + (size_t)D * sizeof(int); };
    while (NT > 64 && lds_bytes(NT) > 80 * 1024) NT >>= 1;  // keep two workgroups per CU when possible
    const size_t lds = lds_bytes(NT);
    if (lds > 160 * 1024) return NF_ENOTSUP;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(&lu_linear_permute_kernel<T>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return NF_ENOTSUP;
    }
    const int64_t ntiles = (B + NT - 1) / NT;
    const int grid = (int)(ntiles < 2048 ? ntiles : 2048);
    hipLaunchKernelGGL(lu_linear_permute_kernel<T>, dim3(grid), dim3(NT), lds, st, (const T *)x, (T *)y, (T *)logdet,
                       perm, (const T *)lo, (const T *)up, (const T *)ud, (const T *)bias, B, D, (T)eps, direction, acc);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

}  // namespace nf

extern "C" int nf_lu_linear_permute(const void *x, void *y, void *logdet, const int64_t *perm,
                                    const void *lower_entries, const void *upper_entries,
                                    const void *unconstrained_upper_diag, const void *bias, int64_t B, int D,
                                    double eps, int direction, int acc, int dtype, nf_stream_t stream) {
    if (B < 0 || D < 1) return NF_EINVAL;
    if (direction != 0 && direction != 1) return NF_EINVAL;
    if (acc < NF_LD_SUB || acc > NF_LD_ADD) return NF_EINVAL;
    if (B == 0) return NF_OK;
    if (!x || !y || !logdet || !perm || !unconstrained_upper_diag || !bias) return NF_EFAULT;
    if (D > 1 && (!lower_entries || !upper_entries)) return NF_EFAULT;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == NF_F32)
        return nf::launch_lu<float>(x, y, logdet, perm, lower_entries, upper_entries, unconstrained_upper_diag, bias, B,
                                    D, eps, direction, acc, st);
    if (dtype == NF_F64)
        return nf::launch_lu<double>(x, y, logdet, perm, lower_entries, upper_entries, unconstrained_upper_diag, bias,
                                     B, D, eps, direction, acc, st);
    return NF_ENOTSUP;
}
