// Diagnostic probes (not on the product path): sustained f32-MFMA rate of this GPU under load,
// the practical ceiling the conv kernels are compared with besides the 157.3 TFLOP/s spec peak.
#include "common.h"

namespace {
// 8 waves per block, 2 per SIMD; each wave keeps 4 independent accumulators busy.
__global__ __launch_bounds__(512) void mfma_f32_probe_kernel(float *out, int iters, float a0, float b0)
{
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = (float)(t + r);
    float a = a0 + (threadIdx.x & 7) * 1e-3f, b = b0 - (threadIdx.x & 3) * 1e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc[0] = mfma32(a, b, acc[0]);
            acc[1] = mfma32(b, a, acc[1]);
            acc[2] = mfma32(a, a, acc[2]);
            acc[3] = mfma32(b, b, acc[3]);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    if (s == 123.456f) out[0] = s;      // keeps the chain live, practically never true
}
}  // namespace

// launches `blocks` x 512 threads; every wave issues iters*16 MFMAs (each 4096 FLOP)
COVA_API int cova_probe_mfma_f32(float *scratch, int blocks, int iters, void *stream)
{
    COVA_REQUIRE(scratch && blocks > 0 && iters > 0);
    hipLaunchKernelGGL(mfma_f32_probe_kernel, dim3(blocks), dim3(512), 0, (hipStream_t)stream, scratch,
                       iters, 0.999f, 1.001f);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}
