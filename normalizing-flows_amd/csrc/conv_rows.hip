// conv_rows.hip -- the 3x3 convolutions of GlowBlock's conditioner (nets/cnn.py:5-63: Conv2d(C/2, 256, 3, padding 1) -> LeakyReLU ->
// Conv2d(256, 256, 1) -> LeakyReLU -> Conv2d(256, C, 3, padding 1)) under autograd (core.py:87-102 + loss.backward() through
// flows/affine/glow.py:10-100), as a per-pixel MLP with a gather in front and a gather-sum behind: rows = pixels (b, y, x),
//   col[r][tap C + c]  = in[b][c][y + dy][x + dx]                      (tap = 3 (dy + 1) + (dx + 1); zero outside the image)
//   a1 = W1c col + b1;  a2 = W2 relu(a1) + b2;  P = W3t relu(a2)       (the plain-MLP mode of made_fwd.hip / made_bwd.hip)
//   out[b][o][y][x]    = b3[o] + sum_tap P[(b, y + dy, x + dx)][tap Cout + o]
// with W1c[o][tap C + c] = conv1.weight[o][c][tap], W3t[tap Cout + o][c] = conv3.weight[o][c][tap].  The backward pass uses the same
// two kernels with the offsets negated (`flip`): dP = gather(g_out), g_in = gather_sum(g_col).  Element-wise HBM-bound passes over
// at most 9 C values per pixel (C <= 48): the hidden tensors (256 channels) never take part.
#include "common.hpp"

namespace nf {

// col[r][k] for k < 9 C (k >= 9 C up to ld: untouched by the caller's choice of ld = 9 C); one thread per element, k fastest
__global__ void __launch_bounds__(256)
conv3x3_gather_kernel(const float *__restrict__ in, float *__restrict__ col, int64_t B, int C, int H, int W, int ld, int flip,
                      int64_t sb) {
    const int K = 9 * C;
    const int64_t N = B * H * W * K;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < N; o += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = o / K;
        const int k = (int)(o - r * K);
        const int tap = k / C, c = k - tap * C;
        int dy = tap / 3 - 1, dx = tap % 3 - 1;
        if (flip) { dy = -dy; dx = -dx; }
        const int xw = (int)(r % W), yh = (int)((r / W) % H);
        const int64_t b = r / ((int64_t)W * H);
        const int yy = yh + dy, xx = xw + dx;
        float v = 0.0f;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = in[b * sb + ((int64_t)c * H + yy) * W + xx];
        col[r * ld + k] = v;
    }
}

// out[b][c][y][x] = bias[c] + sum_tap P[(b, y + dy, x + dx)][tap C + c] over the taps whose neighbour lies inside the image
__global__ void __launch_bounds__(256)
conv3x3_gather_sum_kernel(const float *__restrict__ P, const float *__restrict__ bias, float *__restrict__ out, int64_t B, int C, int H,
                          int W, int ld, int flip) {
    const int64_t N = B * C * H * W;
    for (int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; o < N; o += (int64_t)gridDim.x * blockDim.x) {
        const int xw = (int)(o % W), yh = (int)((o / W) % H), c = (int)((o / ((int64_t)W * H)) % C);
        const int64_t b = o / ((int64_t)W * H * C);
        float s = bias ? bias[c] : 0.0f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            int dy = tap / 3 - 1, dx = tap % 3 - 1;
            if (flip) { dy = -dy; dx = -dx; }
            const int yy = yh + dy, xx = xw + dx;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) s += P[((b * H + yy) * W + xx) * ld + tap * C + c];
        }
        out[o] = s;
    }
}

// Round 6 (late): the two kernels above cost 17 + 23 us per call in config 4's training step (384 calls: 7.8 ms of 51.6) -- 64-bit
// divisions per element in the gather, and in the gather-sum lanes along x reading rows of P that lie `ld` floats apart (4 useful
// bytes per 64-byte sector, each sector re-read by every channel's threads).  Same arithmetic, other mappings:
//   gather: a thread keeps one k = tap C + c (its tap / channel decomposition is computed once) and walks over pixels with 32-bit
//     index arithmetic; writes are 9 C contiguous floats per pixel.
//   gather-sum: a block takes 64 pixels x C channels, sums with the CHANNEL fastest across lanes (each tap read is a contiguous run
//     of C floats of one row of P), transposes through LDS and writes NCHW runs along the pixels.
__global__ void __launch_bounds__(256)
conv3x3_gather_rows_kernel(const float *__restrict__ in, float *__restrict__ col, int npx, int C, int H, int W, int ld, int flip,
                           int64_t sb, int KP) {
    const int K = 9 * C, HW = H * W;
    const int kl = threadIdx.x % KP, pl = threadIdx.x / KP, ppb = 256 / KP;
    for (int k = kl; k < K; k += KP) {
        const int tap = k / C, c = k - tap * C;
        int dy = tap / 3 - 1, dx = tap - 3 * (tap / 3) - 1;
        if (flip) { dy = -dy; dx = -dx; }
        const int coff = c * HW;
        for (int r = blockIdx.x * ppb + pl; r < npx; r += gridDim.x * ppb) {
            const int b = r / HW, pin = r - b * HW, yh = pin / W, xw = pin - yh * W;
            const int yy = yh + dy, xx = xw + dx;
            float v = 0.0f;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = in[(int64_t)b * sb + coff + yy * W + xx];
            col[(int64_t)r * ld + k] = v;
        }
    }
}

// gather, whole images through LDS (C H W <= 12288 floats): a block reads its image once, coalesced, and writes the 9 C-float rows of
// its share of the pixels from LDS -- the rows kernel above reads every input value nine times as scattered 4-byte loads.  A thread
// keeps one k; the pixels' (y, x) come from a table built with the image (no division in the element loop).
__global__ void __launch_bounds__(256)
conv3x3_gather_img_kernel(const float *__restrict__ in, float *__restrict__ col, int C, int H, int W, int ld, int flip, int64_t sb,
                          int KP, int S) {
    extern __shared__ float img[];                 // [C][H][W] then HW packed (y << 16 | x)
    const int HW = H * W, n = C * HW, K = 9 * C;
    int *pyx = reinterpret_cast<int *>(img + n);
    const int b = blockIdx.x / S, part = blockIdx.x - b * S;
    const float *src = in + (int64_t)b * sb;
    for (int i = threadIdx.x; i < n; i += 256) img[i] = src[i];
    for (int i = threadIdx.x; i < HW; i += 256) { const int y = i / W; pyx[i] = (y << 16) | (i - y * W); }
    __syncthreads();
    const int p0 = (int)((int64_t)HW * part / S), p1 = (int)((int64_t)HW * (part + 1) / S);
    const int kl = threadIdx.x % KP, pl = threadIdx.x / KP, ppb = 256 / KP;
    float *dst = col + (int64_t)b * HW * ld;
    for (int k = kl; k < K; k += KP) {
        const int tap = k / C, c = k - tap * C;
        int dy = tap / 3 - 1, dx = tap - 3 * (tap / 3) - 1;
        if (flip) { dy = -dy; dx = -dx; }
        const float *ic = img + c * HW;
        for (int pin = p0 + pl; pin < p1; pin += ppb) {
            const int yx = pyx[pin], yy = (yx >> 16) + dy, xx = (yx & 0xffff) + dx;
            float v = 0.0f;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = ic[yy * W + xx];
            dst[(int64_t)pin * ld + k] = v;
        }
    }
}

// GS_PX pixels per block round of the gather-sum: the largest of 64 / 32 / 16 that still gives the launch >= 2048 blocks (each block
// round is a chain of dependent-latency steps: the more rounds in flight, the better the loads overlap)
template <int GS_PX>
__global__ void __launch_bounds__(256)
conv3x3_gather_sum_rows_kernel(const float *__restrict__ P, const float *__restrict__ bias, float *__restrict__ out, int npx, int C,
                               int H, int W, int ld, int flip) {
    extern __shared__ float tile[];          // [C][GS_PX + 1]
    const int HW = H * W, n = GS_PX * C;
    for (int r0 = blockIdx.x * GS_PX; r0 < npx; r0 += gridDim.x * GS_PX) {
        for (int idx = threadIdx.x; idx < n; idx += 256) {
            const int px = idx / C, c = idx - px * C, r = r0 + px;
            float s = 0.0f;
            if (r < npx) {
                const int b = r / HW, pin = r - b * HW, yh = pin / W, xw = pin - yh * W;
                s = bias ? bias[c] : 0.0f;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {       // (the order of the one-thread-per-element kernel: same bits)
                    int dy = tap / 3 - 1, dx = tap % 3 - 1;
                    if (flip) { dy = -dy; dx = -dx; }
                    const int yy = yh + dy, xx = xw + dx;
                    if (yy >= 0 && yy < H && xx >= 0 && xx < W) s += P[((int64_t)(b * HW + yy * W + xx)) * ld + tap * C + c];
                }
            }
            tile[c * (GS_PX + 1) + px] = s;
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < n; idx += 256) {
            const int c = idx / GS_PX, px = idx - c * GS_PX, r = r0 + px;
            if (r < npx) {
                const int b = r / HW, pin = r - b * HW;
                out[((int64_t)b * C + c) * HW + pin] = tile[c * (GS_PX + 1) + px];
            }
        }
        __syncthreads();
    }
}

// out[c] = sum over b, y, x of g[b][c][y][x]: the last convolution's bias gradient (the reference's autograd: conv2d's bias backward).
// One block per channel, 16-byte loads along the pixels, a fixed summation order (deterministic); replaces `gout.sum((0, 2, 3))`
// (a generic strided torch reduction: 15-31 us per call in config 4's training step).
typedef float cs_f32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(1024)
channel_sum_kernel(const float *__restrict__ g, float *__restrict__ out, int B, int C, int HW) {
    __shared__ float sc[32];
    const int c = blockIdx.x;
    float s0 = 0.0f, s1 = 0.0f;
    if ((HW & 3) == 0) {
        const int q = HW >> 2, n = B * q;
        for (int i = threadIdx.x; i < n; i += 1024) {
            const int b = i / q, p = i - b * q;
            const cs_f32x4 v = *reinterpret_cast<const cs_f32x4 *>(g + ((int64_t)b * C + c) * HW + 4 * p);
            s0 += v[0] + v[1];
            s1 += v[2] + v[3];
        }
    } else {
        const int n = B * HW;
        for (int i = threadIdx.x; i < n; i += 1024) {
            const int b = i / HW, p = i - b * HW;
            s0 += g[((int64_t)b * C + c) * HW + p];
        }
    }
    const float tot = block_sum(s0 + s1, sc);
    if (threadIdx.x == 0) out[c] = tot;
}

}  // namespace nf

// col (B H W, ld >= 9 C) from the NCHW tensor `in` (batch stride `batch_stride` elements >= C H W: a channel split of a wider tensor is
// read in place); flip = 1: offsets negated (the backward pass's gather of the output cotangent).
extern "C" int nf_conv3x3_gather(const void *in, void *col, int64_t B, int C, int H, int W, int ld, int flip, int64_t batch_stride,
                                 nf_stream_t stream) {
    if (B < 0 || C < 1 || H < 1 || W < 1 || ld < 9 * C || (flip != 0 && flip != 1) || batch_stride < (int64_t)C * H * W) return NF_EINVAL;
    if (B == 0) return NF_OK;
    if (!in || !col) return NF_EFAULT;
#ifndef NF_CONV_ELEMENTWISE      // (-DNF_CONV_ELEMENTWISE: the one-thread-per-element kernels of rounds 4-5)
    if (B * H * W < (1ll << 30) && (int64_t)C * H * W < (1ll << 30)) {
        const int npx = (int)(B * H * W), K = 9 * C;
        int KP = 1;
        while (KP < K && KP < 256) KP <<= 1;
        if ((int64_t)C * H * W <= 12288 && H < 32768 && W < 65536 && B <= 1 << 20) {
            int S = 1;                                       // blocks per image: >= 1024 blocks where the images allow
            while (B * S < 1024 && 2 * S <= H * W / 16 && S < 16) S *= 2;
            hipLaunchKernelGGL(nf::conv3x3_gather_img_kernel, dim3((unsigned)(B * S)), dim3(256),
                               (size_t)(C * H * W + H * W) * sizeof(float), (hipStream_t)stream, (const float *)in, (float *)col, C, H, W,
                               ld, flip, batch_stride, KP, S);
            NF_CHECK_LAUNCH();
            return NF_OK;
        }
        const int ppb = 256 / KP;
        int grid = (npx + ppb - 1) / ppb;
        if (grid > 256 * 16) grid = 256 * 16;
        hipLaunchKernelGGL(nf::conv3x3_gather_rows_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const float *)in, (float *)col,
                           npx, C, H, W, ld, flip, batch_stride, KP);
        NF_CHECK_LAUNCH();
        return NF_OK;
    }
#endif
    hipLaunchKernelGGL(nf::conv3x3_gather_kernel, dim3(nf::grid_for(B * H * W * 9 * C, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float *)in, (float *)col, B, C, H, W, ld, flip, batch_stride);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// NCHW `out` (B, C, H, W) from the per-pixel tap products P (B H W, ld >= 9 C); bias (C) may be NULL; flip as above.
extern "C" int nf_conv3x3_gather_sum(const void *P, const void *bias, void *out, int64_t B, int C, int H, int W, int ld, int flip,
                                     nf_stream_t stream) {
    if (B < 0 || C < 1 || H < 1 || W < 1 || ld < 9 * C || (flip != 0 && flip != 1)) return NF_EINVAL;
    if (B == 0) return NF_OK;
    if (!P || !out) return NF_EFAULT;
#ifndef NF_CONV_ELEMENTWISE
    if (B * H * W < (1ll << 30) && C <= 128) {
        const int npx = (int)(B * H * W);
        const int gpx = npx / 64 >= 2048 ? 64 : (npx / 32 >= 2048 ? 32 : 16);
        int grid = (npx + gpx - 1) / gpx;
        if (grid > 256 * 64) grid = 256 * 64;
        const size_t lds = (size_t)C * (gpx + 1) * sizeof(float);
        if (gpx == 64)
            hipLaunchKernelGGL(nf::conv3x3_gather_sum_rows_kernel<64>, dim3(grid), dim3(256), lds, (hipStream_t)stream, (const float *)P,
                               (const float *)bias, (float *)out, npx, C, H, W, ld, flip);
        else if (gpx == 32)
            hipLaunchKernelGGL(nf::conv3x3_gather_sum_rows_kernel<32>, dim3(grid), dim3(256), lds, (hipStream_t)stream, (const float *)P,
                               (const float *)bias, (float *)out, npx, C, H, W, ld, flip);
        else
            hipLaunchKernelGGL(nf::conv3x3_gather_sum_rows_kernel<16>, dim3(grid), dim3(256), lds, (hipStream_t)stream, (const float *)P,
                               (const float *)bias, (float *)out, npx, C, H, W, ld, flip);
        NF_CHECK_LAUNCH();
        return NF_OK;
    }
#endif
    hipLaunchKernelGGL(nf::conv3x3_gather_sum_kernel, dim3(nf::grid_for(B * C * H * W, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float *)P, (const float *)bias, (float *)out, B, C, H, W, ld, flip);
    NF_CHECK_LAUNCH();
    return NF_OK;
}

// out (C) = sum of g (B, C, H, W) over batch and pixels (float32, contiguous NCHW, 16-byte aligned when H W % 4 == 0).
extern "C" int nf_channel_sum(const void *g, void *out, int64_t B, int C, int64_t HW, nf_stream_t stream) {
    if (B < 0 || C < 1 || HW < 1 || B * HW >= (1ll << 31)) return NF_EINVAL;
    if (!out || (B > 0 && !g)) return NF_EFAULT;
    if ((HW & 3) == 0 && ((uintptr_t)g & 15)) return NF_EINVAL;
    hipLaunchKernelGGL(nf::channel_sum_kernel, dim3(C), dim3(1024), 0, (hipStream_t)stream, (const float *)g, (float *)out, (int)B, C,
                       (int)HW);
    NF_CHECK_LAUNCH();
    return NF_OK;
}
