// Microbenchmark: does VALU work issue in the shadow of v_mfma_f32_32x32x2_f32 on gfx950?
// Variants (template MODE): 0 = MFMA chain only, 1 = VALU only, 2 = interleaved (NV VALU per MFMA), 3 = trans only, 4 = MFMA + trans
#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int NV>
__global__ void __launch_bounds__(256) k(float* out, int iters, float seed) {
    f32x16 acc = {0};
    float a = seed + threadIdx.x, b = seed * 0.5f;
    float v0 = a, v1 = b, v2 = a * b, v3 = a - b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0 || MODE == 2 || MODE == 4) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            if (MODE == 1 || MODE == 2) {
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    // independent scalar-f32 fma chains (asm keeps the compiler from packing/combining them)
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v0) : "v"(v1), "v"(v2));
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v3) : "v"(v1), "v"(v2));
                }
            }
            if (MODE == 3 || MODE == 4) {
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    asm volatile("v_exp_f32 %0, %0" : "+v"(v0));
                    asm volatile("v_exp_f32 %0, %0" : "+v"(v3));
                }
            }
        }
    }
    float s = v0 + v3;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

#define INST(M, N) template __global__ void k<M, N>(float*, int, float);
INST(0, 0) INST(1, 2) INST(1, 4) INST(1, 8) INST(2, 2) INST(2, 4) INST(2, 8) INST(3, 1) INST(3, 2) INST(4, 1) INST(4, 2)

extern "C" void run(int mode, int nv, float* out, int iters, int grid, int block, void* stream) {
    hipStream_t st = (hipStream_t)stream;
#define L(M, N) if (mode == M && nv == N) hipLaunchKernelGGL((k<M, N>), dim3(grid), dim3(block), 0, st, out, iters, 1.0f);
    L(0, 0) L(1, 2) L(1, 4) L(1, 8) L(2, 2) L(2, 4) L(2, 8) L(3, 1) L(3, 2) L(4, 1) L(4, 2)
}
