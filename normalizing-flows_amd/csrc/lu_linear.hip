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


// ---------------------------------------------------------------------------------------------------------
// Tile variant for D <= 64: one wave = 64 samples whose rows live in an LDS tile of pitch P = roundup(D,4)+4 words
// (lane s reads its own row with ds_read_b128: 16-lane groups hit 16 distinct 16-byte slots -> conflict-free; the
// triangular factors are zero-padded full matrices in LDS read with broadcast ds_read_b128), updated IN PLACE:
//   density: t = x[perm] (permutation applied while staging) ; u_i = sum_{j>=i} U_ij t_j ascending ;
//            y_i = u_i + sum_{j<i} L_ij u_j + b_i descending ; rows stored back coalesced
//   sample : v = x - b ; forward substitution ascending ; back substitution descending ; y[perm[j]] = t_j on store.
// 0.5 LDS instruction per FMA instead of 2, global rows read/written with unit stride.
template <typename T, int VEC>
struct alignas(sizeof(T) * VEC) VecT { T v[VEC]; };

template <typename T>
__global__ void __launch_bounds__(256)
lu_tile_kernel(const T *__restrict__ x, T *__restrict__ y, T *__restrict__ logdet, const int64_t *__restrict__ perm,
               const T *__restrict__ lower_entries, const T *__restrict__ upper_entries, const T *__restrict__ udiag_raw,
               const T *__restrict__ bias, int64_t B, int D, T eps, int direction, int acc) {
    constexpr int V = 16 / sizeof(T);  // elements per 16-byte LDS access (4 floats / 2 doubles)
    typedef VecT<T, V> vec_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int Dp = (D + 3) / 4 * 4;      // padded row length of the matrices (row groups of 4, vectors of V | 4)
    const int P = Dp + V;                // tile pitch
    T *sL = reinterpret_cast<T *>(smem_raw);  // Dp x Dp strictly lower (zeros elsewhere)
    T *sU = sL + (size_t)Dp * Dp;             // Dp x Dp upper incl. diagonal (zeros elsewhere, pad rows: unit diagonal)
    T *sb = sU + (size_t)Dp * Dp;             // Dp
    T *sred = sb + Dp;                        // 16
    T *tiles = sred + 16;                     // nwaves x 64 x P
    int *sperm = reinterpret_cast<int *>(tiles + (size_t)(blockDim.x >> 6) * 64 * P);  // D : perm
    int *sinv = sperm + D;                                                               // D : inverse perm
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nth = blockDim.x;
    // Every block rebuilds the two factors; one source load per matrix element, sixteen in flight per thread so the
    // global latency is paid ~Dp*Dp/(16*nth) times instead of once per element.
    for (int base = 0; base < Dp * Dp; base += nth * 16) {
        T val[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int i = base + q * nth + tid;
            const int r = i / Dp, c = i - r * Dp;
            const T *src = nullptr;
            if (i < Dp * Dp && r < D && c < D) {
                if (c < r) src = lower_entries + ((size_t)r * (r - 1) / 2 + c);
                else if (c == r) src = udiag_raw + r;
                else src = upper_entries + ((size_t)r * (D - 1) - (size_t)r * (r - 1) / 2 + (c - r - 1));
            }
            val[q] = src ? *src : T(0);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int i = base + q * nth + tid;
            if (i < Dp * Dp) {
                const int r = i / Dp, c = i - r * Dp;
                const bool in = r < D && c < D;
                sL[i] = (in && c < r) ? val[q] : T(0);
                sU[i] = in ? (c == r ? softplus(val[q]) + eps : (c > r ? val[q] : T(0))) : (c == r ? T(1) : T(0));
            }
        }
    }
    for (int i = tid; i < Dp; i += nth) sb[i] = i < D ? bias[i] : T(0);
    for (int i = tid; i < D; i += nth) { sperm[i] = (int)perm[i]; sinv[(int)perm[i]] = i; }
    T part = T(0);
    for (int i = tid; i < D; i += nth) part += M<T>::log(softplus(udiag_raw[i]) + eps);
    T lad = block_sum(part, sred);
    if (direction) lad = -lad;
    __syncthreads();

    T *tile = tiles + (size_t)wid * 64 * P;
    T *row = tile + (size_t)lane * P;
    for (int c = D; c < P; ++c) row[c] = T(0);  // pad columns stay zero for the whole kernel
    const int64_t nwt = (B + 63) / 64;
    const int nwaves = nth >> 6;
    for (int64_t wt = (int64_t)blockIdx.x * nwaves + wid; wt < nwt; wt += (int64_t)gridDim.x * nwaves) {
        const int64_t b0 = wt * 64;
        const int ts = (int)((B - b0) < 64 ? (B - b0) : 64);
        // ---- stage rows: the 64 x D block is one contiguous span, read flat with unit stride (8 loads in flight per lane);
        //      density applies the permutation on the way in: t_j = x[perm[j]]  <=>  t[inv[c]] = x[c]
        const int span = ts * D;
#ifndef NF_LU_ABL_NOSTAGE
        for (int base = 0; base < 64 * D; base += 64 * 8) {
            T v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = base + q * 64 + lane;
                v[q] = e < span ? x[b0 * D + e] : T(0);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = base + q * 64 + lane;
                if (e < 64 * D) {
                    const int r = e / D, c = e - r * D;
                    if (direction == 0) tile[(size_t)r * P + sinv[c]] = v[q];
                    else tile[(size_t)r * P + c] = v[q] - sb[c];
                }
            }
        }
#endif
        // the tile is private to this wave: no workgroup barrier, wave-level ordering of LDS accesses suffices
        __builtin_amdgcn_wave_barrier();
        // Rows are processed four at a time (4 independent accumulators share every vector read of the sample row), which
        // is what hides the LDS latency at the one-wave-per-SIMD occupancy the tile size allows.
        auto dot4 = [&](const T *mat, int i0, int jb, int je, T (&a)[4]) {
            int j = jb;
            for (; j + 2 * V <= je; j += 2 * V) {  // two vector columns per trip: 10 LDS reads in flight, 8V FMAs
                const vec_t t0 = *reinterpret_cast<const vec_t *>(row + j);
                const vec_t t1 = *reinterpret_cast<const vec_t *>(row + j + V);
                vec_t m0[4], m1[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    m0[r] = *reinterpret_cast<const vec_t *>(mat + (size_t)(i0 + r) * Dp + j);
                    m1[r] = *reinterpret_cast<const vec_t *>(mat + (size_t)(i0 + r) * Dp + j + V);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
#pragma unroll
                    for (int q = 0; q < V; ++q) a[r] += m0[r].v[q] * t0.v[q];
#pragma unroll
                    for (int q = 0; q < V; ++q) a[r] += m1[r].v[q] * t1.v[q];
                }
            }
            for (; j < je; j += V) {
                const vec_t t = *reinterpret_cast<const vec_t *>(row + j);
                vec_t m[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) m[r] = *reinterpret_cast<const vec_t *>(mat + (size_t)(i0 + r) * Dp + j);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int q = 0; q < V; ++q) a[r] += m[r].v[q] * t.v[q];
            }
        };
#ifndef NF_LU_ABL_NOCOMPUTE
        const int Dg = Dp;
        if (direction == 0) {
            for (int i0 = 0; i0 < Dg; i0 += 4) {  // u_i = sum_{j >= i0} U_ij t_j   (U_ij = 0 for j < i)
                T a[4] = {T(0), T(0), T(0), T(0)};
                dot4(sU, i0, i0, Dp, a);
#pragma unroll
                for (int r = 0; r < 4; ++r) row[i0 + r] = a[r];
            }
            for (int i0 = Dg - 4; i0 >= 0; i0 -= 4) {  // y_i = u_i + sum_{j < i} L_ij u_j + b_i   (L_ij = 0 for j >= i)
                T a[4] = {T(0), T(0), T(0), T(0)};
                const int je = i0 + 4;
                dot4(sL, i0, 0, je, a);
#pragma unroll
                for (int r = 0; r < 4; ++r) row[i0 + r] = (a[r] + row[i0 + r]) + sb[i0 + r];
            }
        } else {
            for (int i0 = 0; i0 < Dg; i0 += 4) {  // forward substitution, unit lower: block part then the 4x4 triangle
                T a[4] = {T(0), T(0), T(0), T(0)};
                dot4(sL, i0, 0, i0, a);
                T u[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    T acc2 = row[i0 + r] - a[r];
#pragma unroll
                    for (int c = 0; c < r; ++c) acc2 -= sL[(size_t)(i0 + r) * Dp + i0 + c] * u[c];
                    u[r] = acc2;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) row[i0 + r] = u[r];
            }
            for (int i0 = Dg - 4; i0 >= 0; i0 -= 4) {  // back substitution
                T a[4] = {T(0), T(0), T(0), T(0)};
                if (i0 + 4 < Dp) dot4(sU, i0, i0 + 4, Dp, a);
                T t[4];
#pragma unroll
                for (int r = 3; r >= 0; --r) {
                    T acc2 = row[i0 + r] - a[r];
#pragma unroll
                    for (int c = r + 1; c < 4; ++c) acc2 -= sU[(size_t)(i0 + r) * Dp + i0 + c] * t[c];
                    t[r] = acc2 / sU[(size_t)(i0 + r) * Dp + i0 + r];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) row[i0 + r] = t[r];
            }
        }
#endif
        __builtin_amdgcn_wave_barrier();
        // ---- store rows flat (unit stride); the sample direction un-permutes on the way out: y[c] = t[inv[c]] ----
#ifdef NF_LU_ABL_NOSTORE
        if (lad == T(12345))
#endif
        for (int e = lane; e < span; e += 64) {
            const int r = e / D, c = e - r * D;
            y[b0 * D + e] = direction == 0 ? tile[(size_t)r * P + c] : tile[(size_t)r * P + sinv[c]];
        }
        if (lane < ts) ld_store(logdet + b0 + lane, lad, acc);
        __builtin_amdgcn_wave_barrier();
    }
}

template <typename T>
static int launch_lu(const void *x, void *y, void *logdet, const int64_t *perm, const void *lo, const void *up,
                     const void *ud, const void *bias, int64_t B, int D, double eps, int direction, int acc,
                     hipStream_t st) {
    if (D <= 64) {  // LDS tile variant
        constexpr int V = 16 / sizeof(T);
        const int Dp = (D + 3) / 4 * 4, P = Dp + V;
        auto tile_lds = [&](int nw) { return ((size_t)2 * Dp * Dp + Dp + 16 + (size_t)nw * 64 * P) * sizeof(T) + (size_t)2 * D * sizeof(int) + 16; };
        int nw = 4;  // one 64-sample tile per wave; the factors are rebuilt per block, so prefer wide blocks
        while (nw > 1 && tile_lds(nw) > 160 * 1024) nw >>= 1;
        const size_t lds = tile_lds(nw);
        static LdsOptIn opted = {};
        if (opt_in_lds(reinterpret_cast<const void *>(&lu_tile_kernel<T>), lds, opted) != NF_OK) return NF_ENOTSUP;
        const int64_t nwt = (B + 63) / 64;
        int64_t g = (nwt + nw - 1) / nw;
        if (g > 2048) g = 2048;
        hipLaunchKernelGGL(lu_tile_kernel<T>, dim3((int)g), dim3(64 * nw), lds, st, (const T *)x, (T *)y, (T *)logdet, perm,
                           (const T *)lo, (const T *)up, (const T *)ud, (const T *)bias, B, D, (T)eps, direction, acc);
        NF_CHECK_LAUNCH();
        return NF_OK;
    }
    int NT = 256;
    auto lds_bytes = [&](int nt) { return ((size_t)D * D + D + 16 + (size_t)D * nt) * sizeof(T) + (size_t)D * sizeof(int); };
    while (NT > 64 && lds_bytes(NT) > 80 * 1024) NT >>= 1;  // keep two workgroups per CU when possible
    const size_t lds = lds_bytes(NT);
    if (lds > 160 * 1024) return NF_ENOTSUP;
    static LdsOptIn opted = {};
    if (opt_in_lds(reinterpret_cast<const void *>(&lu_linear_permute_kernel<T>), lds, opted) != NF_OK) return NF_ENOTSUP;
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
