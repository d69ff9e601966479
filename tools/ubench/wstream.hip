// Microbenchmark for "weights straight from L2 into VGPRs": every wave streams the SAME packed weight stream
// ([k-group][64 lanes][4 floats], 1 KB per load instruction) through a ring of RING prefetched registers and issues
// MF MFMAs (v_mfma_f32_32x32x2_f32, two column tiles) per load.  Compare with the MFMA-only time.
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

template <int MODE>   // 0: MFMA only (operands in registers), 1: loads + MFMA
__global__ void __launch_bounds__(256) wstream_kernel(const float *__restrict__ w, float *out, int nloads, int layers) {
    const int lane = threadIdx.x & 63;
    const f32x4 *p = reinterpret_cast<const f32x4 *>(w) + lane;
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    const float b0 = (float)lane * 1e-3f, b1 = b0 + 1.0f;
    f32x4 r0, r1, r2, r3, r4, r5, r6, r7;
    auto ld = [&](int i, f32x4 &r) { r = p[(size_t)i * 64]; };
    auto mm = [&](const f32x4 &a) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            c0 = MFMA(a[i], b0, c0);
            c1 = MFMA(a[i], b1, c1);
        }
    };
    const int total = nloads * layers;
    if (MODE == 1) { ld(0, r0); ld(1, r1); ld(2, r2); ld(3, r3); ld(4, r4); ld(5, r5); ld(6, r6); ld(7, r7); }
    else { r0 = r1 = r2 = r3 = r4 = r5 = r6 = r7 = f32x4{1.f, 2.f, 3.f, 4.f}; }
    for (int i = 0; i < total; i += 8) {
#define STEP(R, K)                                               \
        mm(R);                                                    \
        if (MODE == 1) { int n = i + 8 + K; ld(n < total ? n : total - 1, R); }
        STEP(r0, 0) STEP(r1, 1) STEP(r2, 2) STEP(r3, 3) STEP(r4, 4) STEP(r5, 5) STEP(r6, 6) STEP(r7, 7)
#undef STEP
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

extern "C" void run(int mode, const void *w, void *out, int nloads, int layers, int grid, int block, void *stream) {
    if (mode == 0) hipLaunchKernelGGL(wstream_kernel<0>, dim3(grid), dim3(block), 0, (hipStream_t)stream, (const float *)w, (float *)out, nloads, layers);
    else hipLaunchKernelGGL(wstream_kernel<1>, dim3(grid), dim3(block), 0, (hipStream_t)stream, (const float *)w, (float *)out, nloads, layers);
}
