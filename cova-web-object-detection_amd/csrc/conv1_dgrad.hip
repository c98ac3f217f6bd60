// Gradient w.r.t. the page images -- what autograd gives the reference for free when `images.requires_grad`
// (models.py:94-122 through nn.Conv2d(3, 64, 7, 2, 3)); NOT on the training hot path (train.py never asks for it: conv1 of
// the step computes its weight gradient only), so these two kernels are plain vector code, correct and bounded by L2 reads:
//   cova_pool_bwd_dy1:  dy1 = abc[0] * route(dp, idx) + abc[1] * y1 + abc[2]      (the BatchNorm + ReLU + MaxPool backward apply
//                       that cova_conv1_wgrad_poolbwd forms on load, written out: [B,H1,W1,64])
//   cova_conv1_dgrad:   dimg[b,c,y,x] = sum_{co,kh,kw} dy1[b,(y+3-kh)/2,(x+3-kw)/2,co] * w[co,c,kh,kw]   over the taps whose
//                       (y+3-kh), (x+3-kw) are even and land inside the map (transposed 7x7 / stride-2 convolution), NCHW.
#include "common.h"

namespace {

// one thread = one y1 pixel x 4 channels
__global__ __launch_bounds__(256) void pool_bwd_dy1_kernel(const float *__restrict__ dp, const uint8_t *__restrict__ idx,
                                                           const float *__restrict__ y1, const float *__restrict__ abc,
                                                           float *__restrict__ dy1, int B, int H1, int W1, int H2, int W2)
{
    const int c4 = threadIdx.x & 15;
    const float4 A4 = *reinterpret_cast<const float4 *>(abc + c4 * 4);
    const float4 B4 = *reinterpret_cast<const float4 *>(abc + 64 + c4 * 4);
    const float4 C4 = *reinterpret_cast<const float4 *>(abc + 128 + c4 * 4);
    const float Av[4] = {A4.x, A4.y, A4.z, A4.w}, Bv[4] = {B4.x, B4.y, B4.z, B4.w}, Cv[4] = {C4.x, C4.y, C4.z, C4.w};
    const long long npix = (long long)B * H1 * W1;
    for (long long p = (((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4); p < npix;
         p += ((long long)gridDim.x * blockDim.x) >> 4) {
        const int X = (int)(p % W1);
        const long long pr = p / W1;
        const int Y = (int)(pr % H1), b = (int)(pr / H1);
        float g[4] = {0.f, 0.f, 0.f, 0.f};
        // the windows (oy, ox) of the 3x3 / stride 2 / pad 1 pooling that contain (Y, X): oy = (Y + 1 - ky) / 2 for the ky in
        // {0, 1, 2} of matching parity; summed in the fixed order ky, kx ascending
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            if (((Y + 1 - ky) & 1) != 0 || Y + 1 - ky < 0) continue;
            const int oy = (Y + 1 - ky) >> 1;
            if (oy >= H2) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                if (((X + 1 - kx) & 1) != 0 || X + 1 - kx < 0) continue;
                const int ox = (X + 1 - kx) >> 1;
                if (ox >= W2) continue;
                const size_t o = (((size_t)b * H2 + oy) * W2 + ox) * 64 + c4 * 4;
                const uchar4 id = *reinterpret_cast<const uchar4 *>(idx + o);
                const float4 d = *reinterpret_cast<const float4 *>(dp + o);
                const int code = ky * 3 + kx;
                g[0] += id.x == code ? d.x : 0.f;
                g[1] += id.y == code ? d.y : 0.f;
                g[2] += id.z == code ? d.z : 0.f;
                g[3] += id.w == code ? d.w : 0.f;
            }
        }
        const float4 v = *reinterpret_cast<const float4 *>(y1 + (size_t)p * 64 + c4 * 4);
        const float yv[4] = {v.x, v.y, v.z, v.w};
        float o4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o4[j] = fmaf(Av[j], g[j], fmaf(Bv[j], yv[j], Cv[j]));      // same expression as the weight gradient's
        *reinterpret_cast<float4 *>(dy1 + (size_t)p * 64 + c4 * 4) = make_float4(o4[0], o4[1], o4[2], o4[3]);
    }
}

// block = 16 x 16 image pixels; thread = one pixel, its 3 channels; the weights in LDS as [kh][kw][co][c padded to 4]
__global__ __launch_bounds__(256) void conv1_dgrad_kernel(const float *__restrict__ dy1, const float *__restrict__ w,
                                                          float *__restrict__ dimg, int H, int W, int H1, int W1)
{
    __shared__ __attribute__((aligned(16))) float s_w[49 * 64 * 4];        // 50,176 B
    for (int e = threadIdx.x; e < 49 * 64; e += 256) {
        const int co = e & 63, t = e >> 6;                                  // t = kh * 7 + kw
        *reinterpret_cast<float4 *>(s_w + e * 4) = make_float4(w[(co * 3 + 0) * 49 + t], w[(co * 3 + 1) * 49 + t],
                                                                w[(co * 3 + 2) * 49 + t], 0.f);
    }
    __syncthreads();
    const int x = blockIdx.x * 16 + (threadIdx.x & 15), y = blockIdx.y * 16 + (threadIdx.x >> 4), b = blockIdx.z;
    if (x >= W || y >= H) return;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    const float *db = dy1 + (size_t)b * H1 * W1 * 64;
    for (int kh = (y + 3) & 1; kh < 7; kh += 2) {                           // (y + 3 - kh) even
        const int oy = (y + 3 - kh) >> 1;
        if (y + 3 - kh < 0 || oy >= H1) continue;
        for (int kw = (x + 3) & 1; kw < 7; kw += 2) {
            const int ox = (x + 3 - kw) >> 1;
            if (x + 3 - kw < 0 || ox >= W1) continue;
            const float *dpix = db + ((size_t)oy * W1 + ox) * 64;
            const float *wt = s_w + (kh * 7 + kw) * 64 * 4;
#pragma unroll 4
            for (int c4 = 0; c4 < 16; ++c4) {
                const float4 d = *reinterpret_cast<const float4 *>(dpix + c4 * 4);
                const float dv[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 ww = *reinterpret_cast<const float4 *>(wt + (c4 * 4 + j) * 4);
                    a0 = fmaf(dv[j], ww.x, a0);
                    a1 = fmaf(dv[j], ww.y, a1);
                    a2 = fmaf(dv[j], ww.z, a2);
                }
            }
        }
    }
    const size_t plane = (size_t)H * W;
    float *o = dimg + (size_t)b * 3 * plane + (size_t)y * W + x;
    o[0] = a0;
    o[plane] = a1;
    o[2 * plane] = a2;
}

}  // namespace

COVA_API int cova_conv_out_size(int in_size, int kernel, int stride, int pad);

// dy1 NHWC [B,H1,W1,64] = abc[0]*route(dp, idx) + abc[1]*y1 + abc[2]: the operand cova_conv1_wgrad_poolbwd forms on load,
// materialised (dp [B,H2,W2,64] already ReLU-masked, idx from cova_bn_relu_maxpool_fwd, abc [3,64])
COVA_API int cova_pool_bwd_dy1(const float *dp, const uint8_t *idx, const float *y1, const float *abc, float *dy1, int B,
                               int H1, int W1, void *stream)
{
    COVA_REQUIRE(dp && idx && y1 && abc && dy1 && B > 0 && H1 > 0 && W1 > 0);
    const int H2 = cova_conv_out_size(H1, 3, 2, 1), W2 = cova_conv_out_size(W1, 3, 2, 1);
    const long long npix = (long long)B * H1 * W1;
    const long long want = cdivll(npix, 16);
    const int grid = (int)(want < 65536 ? want : 65536);
    hipLaunchKernelGGL(pool_bwd_dy1_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, dp, idx, y1, abc, dy1, B, H1, W1,
                       H2, W2);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

// dimg NCHW [B,3,H,W] = the gradient of nn.Conv2d(3,64,7,stride 2,pad 3) w.r.t. its input: dy1 NHWC [B,H1,W1,64], w OIHW
COVA_API int cova_conv1_dgrad(const float *dy1, const float *w_oihw, float *dimg, int B, int H, int W, void *stream)
{
    COVA_REQUIRE(dy1 && w_oihw && dimg && B > 0 && H > 0 && W > 0 && B <= 65535);
    const int H1 = cova_conv_out_size(H, 7, 2, 3), W1 = cova_conv_out_size(W, 7, 2, 3);
    hipLaunchKernelGGL(conv1_dgrad_kernel, dim3(cdiv(W, 16), cdiv(H, 16), B), dim3(256), 0, (hipStream_t)stream, dy1, w_oihw,
                       dimg, H, W, H1, W1);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}
