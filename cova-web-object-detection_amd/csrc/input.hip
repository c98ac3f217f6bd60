// Device-side input pipeline: what the reference does per batch on
// CPU DataLoader workers before `.to(device)` (train.py:47-52), done on the GPU instead so that
// only uint8 pixels and the raw box rows cross PCIe.
//   * ToTensor (datasets.py:41-45,96-97): uint8 HWC -> float32 CHW, value / 255
//   * WebDataset.__getitem__ box part (datasets.py:110-128): labels = last column,
//     [x,y,w,h] -> [x1,y1,x2,y2], context window of `context_size` preorder neighbours on both sides
//   * custom_collate_fn (datasets.py:159-178): page index column, neighbour ids offset to
//     batch-global ids (-1 pads untouched)
// HBM-bound byte/integer work: one coalesced pass each, bit-exact with the reference.
#include "common.h"

namespace {

// 4 consecutive pixels per thread: 12 bytes in (3 x 32-bit loads), one float4 store per channel plane.
__global__ __launch_bounds__(256) void u8_nhwc_to_f32_nchw_kernel(const uint8_t *__restrict__ src,
                                                                   float *__restrict__ dst,
                                                                   long long npix4, long long plane,
                                                                   long long total_pix)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npix4;
         i += (long long)gridDim.x * blockDim.x) {
        const long long p = i * 4;                 // first pixel (flat over B*H*W); H*W % 4 == 0
        const long long b = p / plane, q = p - b * plane;
        const uint32_t *s32 = reinterpret_cast<const uint32_t *>(src + p * 3);
        const uint32_t w0 = s32[0], w1 = s32[1], w2 = s32[2];
        uint8_t v[12];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            v[k] = (w0 >> (8 * k)) & 255;
            v[4 + k] = (w1 >> (8 * k)) & 255;
            v[8 + k] = (w2 >> (8 * k)) & 255;
        }
        float *d = dst + b * 3 * plane + q;
#pragma unroll
        for (int c = 0; c < 3; ++c)
            *reinterpret_cast<float4 *>(d + c * plane) =
                make_float4((float)v[c] / 255.f, (float)v[3 + c] / 255.f, (float)v[6 + c] / 255.f,
                            (float)v[9 + c] / 255.f);
    }
    (void)total_pix;
}

// generic tail-safe variant (H*W not a multiple of 4 or unaligned buffers)
__global__ __launch_bounds__(256) void u8_nhwc_to_f32_nchw_scalar_kernel(
    const uint8_t *__restrict__ src, float *__restrict__ dst, long long plane, long long total_pix)
{
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < total_pix;
         p += (long long)gridDim.x * blockDim.x) {
        const long long b = p / plane, q = p - b * plane;
#pragma unroll
        for (int c = 0; c < 3; ++c) dst[(b * 3 + c) * plane + q] = (float)src[p * 3 + c] / 255.f;
    }
}

// one thread per (box, slot): slot < K writes a neighbour id, slot == K writes the box row + label
__global__ __launch_bounds__(256) void collate_boxes_kernel(
    const float *__restrict__ rows, const int *__restrict__ page_offsets, int B, int cs,
    float *__restrict__ bboxes, long long *__restrict__ labels, long long *__restrict__ ctx, int N)
{
    const int K = 2 * cs;
    const long long total = (long long)N * (K + 1);
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const int g = (int)(t / (K + 1)), slot = (int)(t - (long long)g * (K + 1));
        int lo = 0, hi = B;                        // page of box g: page_offsets[lo] <= g < [lo+1]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (page_offsets[mid] <= g) lo = mid; else hi = mid;
        }
        const int base = page_offsets[lo], n = page_offsets[lo + 1] - base, i = g - base;
        if (slot == K) {
            const float x = rows[g * 5 + 0], y = rows[g * 5 + 1];
            bboxes[g * 5 + 0] = (float)lo;
            bboxes[g * 5 + 1] = x;
            bboxes[g * 5 + 2] = y;
            bboxes[g * 5 + 3] = x + rows[g * 5 + 2];          // datasets.py:115, float32 add
            bboxes[g * 5 + 4] = y + rows[g * 5 + 3];
            labels[g] = (long long)rows[g * 5 + 4];            // torch.LongTensor(float) truncates
        } else {
            // neighbours: max(0,i-cs)..i-1 then i+1..min(n,i+cs+1)-1, then -1 pads
            const int nleft = min(i, cs), nright = min(n - 1 - i, cs);
            long long v = -1;
            if (slot < nleft) v = base + (i - nleft + slot);
            else if (slot < nleft + nright) v = base + (i + 1 + (slot - nleft));
            ctx[(long long)g * K + slot] = v;
        }
    }
}

inline int grid_for(long long n)
{
    long long g = (n + 255) / 256;
    if (g > 256 * 32) g = 256 * 32;
    return (int)(g < 1 ? 1 : g);
}

}  // namespace

// u8 [B,H,W,3] -> f32 [B,3,H,W], value/255 (torchvision ToTensor as used at datasets.py:41-45)
COVA_API int cova_images_u8_to_f32(const uint8_t *u8_nhwc, float *f32_nchw, int B, int H, int W,
                                   void *stream)
{
    COVA_REQUIRE(u8_nhwc && f32_nchw && B > 0 && H > 0 && W > 0);
    const long long plane = (long long)H * W, total = plane * B;
    hipStream_t st = (hipStream_t)stream;
    if (plane % 4 == 0 && ((uintptr_t)u8_nhwc & 3) == 0 && ((uintptr_t)f32_nchw & 15) == 0)
        hipLaunchKernelGGL(u8_nhwc_to_f32_nchw_kernel, dim3(grid_for(total / 4)), dim3(256), 0, st,
                           u8_nhwc, f32_nchw, total / 4, plane, total);
    else
        hipLaunchKernelGGL(u8_nhwc_to_f32_nchw_scalar_kernel, dim3(grid_for(total)), dim3(256), 0, st,
                           u8_nhwc, f32_nchw, plane, total);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

// rows [N,5] = x,y,w,h,label (the bboxes/*.csv rows, datasets.py:52-60) of B pages back to back,
// page p owning rows page_offsets[p] .. page_offsets[p+1]-1 (device int32 [B+1]).
// -> bboxes [N,5] = page,x1,y1,x2,y2; labels [N] i64; ctx [N, 2*context_size] i64 (batch-global, -1 pads)
COVA_API int cova_collate_boxes(const float *rows, const int *page_offsets, int B, int N,
                                int context_size, float *bboxes, long long *labels, long long *ctx,
                                void *stream)
{
    COVA_REQUIRE(rows && page_offsets && bboxes && labels && B > 0 && N >= 0 && context_size >= 0);
    COVA_REQUIRE(context_size == 0 || ctx);
    if (N == 0) return COVA_OK;
    hipLaunchKernelGGL(collate_boxes_kernel, dim3(grid_for((long long)N * (2 * context_size + 1))),
                       dim3(256), 0, (hipStream_t)stream, rows, page_offsets, B, context_size, bboxes,
                       labels, ctx, N);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}
