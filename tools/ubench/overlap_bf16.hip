// Microbenchmark: VALU fillers beside v_mfma_f32_32x32x16_bf16 (the guide says the matrix pipe is separate from the VALU).
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int NV>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
    f32x16 acc = {0}, acc2 = {0};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + threadIdx.x + i); b[i] = (short)(0x3f00 + i); }
    float v0 = seed + threadIdx.x, v1 = seed * 0.5f, v2 = v0 * v1, v3 = v0 - v1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0 || MODE == 2) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
                if (MODE == 2 || MODE == 0) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc2, 0, 0, 0);
            }
            if (MODE == 1 || MODE == 2) {
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(v1), "v"(v2));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v3) : "v"(v1), "v"(v2));
                }
            }
        }
    }
    float s = v0 + v3;
    for (int i = 0; i < 16; ++i) s += acc[i] + acc2[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
#define INST(M, N) template __global__ void k<M, N>(float*, int, float);
INST(0, 0) INST(1, 2) INST(1, 4) INST(1, 8) INST(2, 2) INST(2, 4) INST(2, 8)
extern "C" void run(int mode, int nv, float* out, int iters, int grid, int block, void* stream) {
    hipStream_t st = (hipStream_t)stream;
#define L(M, N) if (mode == M && nv == N) hipLaunchKernelGGL((k<M, N>), dim3(grid), dim3(block), 0, st, out, iters, 1.0f);
    L(0, 0) L(1, 2) L(1, 4) L(1, 8) L(2, 2) L(2, 4) L(2, 8)
}
