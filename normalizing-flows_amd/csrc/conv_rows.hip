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

}  // namespace nf

// col (B H W, ld >= 9 C) from the NCHW tensor `in` (batch stride `batch_stride` elements >= C H W: a channel split of a wider tensor is
// read in place); flip = 1: offsets negated (the backward pass's gather of the output cotangent).
extern "C" int nf_conv3x3_gather(const void *in, void *col, int64_t B, int C, int H, int W, int ld, int flip, int64_t batch_stride,
                                 nf_stream_t stream) {
    if (B < 0 || C < 1 || H < 1 || W < 1 || ld < 9 * C || (flip != 0 && flip != 1) || batch_stride < (int64_t)C * H * W) return NF_EINVAL;
    if (B == 0) return NF_OK;
    if (!in || !col) return NF_EFAULT;
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
    hipLaunchKernelGGL(nf::conv3x3_gather_sum_kernel, dim3(nf::grid_for(B * C * H * W, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float *)P, (const float *)bias, (float *)out, B, C, H, W, ld, flip);
    NF_CHECK_LAUNCH();
    return NF_OK;
}
