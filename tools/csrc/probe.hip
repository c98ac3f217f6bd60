// Diagnostic probes (not on the product path): sustained f32-MFMA rate of this GPU under load,
// the practical ceiling the conv kernels are compared with besides the 157.3 TFLOP/s spec peak.
#include "../../cova-web-object-detection_amd/csrc/common.h"

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
// Mixed probe: blocks [0, mfma_blocks) run the MFMA loop, the remaining blocks stream `buf`
// (float4 loads, grid-stride) -- both kinds co-resident on every CU.  Answers whether MFMA issue and
// HBM streaming overlap on this chip or add up (power / fabric coupling).
__global__ __launch_bounds__(512) void mix_probe_kernel(float *out, const float4 *buf, long long n4,
                                                        int mfma_blocks, int iters, int passes,
                                                        float a0, float b0)
{
    if ((int)blockIdx.x < mfma_blocks) {
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
        if (s == 123.456f) out[0] = s;
    } else {
        const long long nb = gridDim.x - mfma_blocks, bid = blockIdx.x - mfma_blocks;
        float s = 0.f;
        for (int p = 0; p < passes; ++p)
            for (long long i = bid * 512 + threadIdx.x; i < n4; i += nb * 512) {
                const float4 v = buf[i];
                s += v.x + v.y + v.z + v.w;
            }
        if (s == 123.456f) out[1] = s;
    }
}
// Same-wave probe: every wave interleaves its MFMA chain with `loads_per_iter` independent float4
// loads per 16 MFMAs (consumed four iterations later) -- the conv kernels' structure in miniature.
template <int LPI>
__global__ __launch_bounds__(512) void mfma_load_probe_kernel(float *out, const float4 *buf, long long n4,
                                                              int iters, float a0, float b0)
{
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = (float)(t + r);
    float a = a0 + (threadIdx.x & 7) * 1e-3f, b = b0 - (threadIdx.x & 3) * 1e-3f;
    const long long stride = (long long)gridDim.x * 512;
    long long idx = (long long)blockIdx.x * 512 + threadIdx.x;
    float4 ring[4][LPI > 0 ? LPI : 1];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < (LPI > 0 ? LPI : 1); ++j) ring[k][j] = make_float4(0.f, 0.f, 0.f, 0.f);
    float s = 0.f;
    for (int i = 0; i < iters; i += 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int j = 0; j < LPI; ++j) {
                s += ring[k][j].x + ring[k][j].w;
                ring[k][j] = buf[idx];
                idx += stride;
                if (idx >= n4) idx -= n4;
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[0] = mfma32(a, b, acc[0]);
                acc[1] = mfma32(b, a, acc[1]);
                acc[2] = mfma32(a, a, acc[2]);
                acc[3] = mfma32(b, b, acc[3]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    if (s == 123.456f) out[0] = s;
}
// MFMA chain with VPM independent VALU FMAs issued after every MFMA (same wave): do the vector ALU
// and the matrix pipe overlap, or do their issue cycles add up?
template <int VPM>
__global__ __launch_bounds__(512) void mfma_valu_probe_kernel(float *out, int iters, float a0, float b0)
{
    f32x4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = f32x4{(float)t, 1.f, 2.f, 3.f};
    float a = a0 + (threadIdx.x & 7) * 1e-3f, b = b0 - (threadIdx.x & 3) * 1e-3f;
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = a * (k + 1);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            acc[u & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u & 7], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < VPM; ++k) v[(u + k) & 7] = fmaf(v[(u + k) & 7], b, a);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) s += acc[t][0] + acc[t][3];
#pragma unroll
    for (int k = 0; k < 8; ++k) s += v[k];
    if (s == 123.456f) out[0] = s;
}

// VALU only (same FMA stream, no MFMA) for reference
template <int VPM>
__global__ __launch_bounds__(512) void valu_probe_kernel(float *out, int iters, float a0, float b0)
{
    float a = a0 + (threadIdx.x & 7) * 1e-3f, b = b0 - (threadIdx.x & 3) * 1e-3f;
    float v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = a * (k + 1);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
#pragma unroll
            for (int k = 0; k < VPM; ++k) v[(u + k) & 7] = fmaf(v[(u + k) & 7], b, a);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += v[k];
    if (s == 123.456f) out[0] = s;
}
}  // namespace

// blocks x 8 waves, per wave iters*16 v_mfma_f32_16x16x4_f32, each followed by valu_per_mfma
// independent v_fma_f32 (0, 2, 4 or 6); mfma == 0 runs the VALU stream alone
COVA_API int cova_probe_mfma_valu(float *scratch, int blocks, int iters, int valu_per_mfma, int mfma,
                                  void *stream)
{
    COVA_REQUIRE(scratch && blocks > 0 && iters > 0);
    hipStream_t st = (hipStream_t)stream;
    const dim3 g(blocks), b(512);
#define COVA_PV(N)                                                                                   \
    if (valu_per_mfma == N) {                                                                        \
        if (mfma) hipLaunchKernelGGL(mfma_valu_probe_kernel<N>, g, b, 0, st, scratch, iters, 0.999f, 1.001f); \
        else hipLaunchKernelGGL(valu_probe_kernel<N>, g, b, 0, st, scratch, iters, 0.999f, 1.001f);  \
    }
    COVA_PV(0) COVA_PV(2) COVA_PV(4) COVA_PV(6)
#undef COVA_PV
    COVA_REQUIRE(valu_per_mfma == 0 || valu_per_mfma == 2 || valu_per_mfma == 4 || valu_per_mfma == 6);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

// blocks x 8 waves; per wave iters*16 MFMAs and iters*loads_per_iter float4 loads per lane
COVA_API int cova_probe_mfma_load(float *scratch, const float *buf, long long n4, int blocks, int iters,
                                  int loads_per_iter, void *stream)
{
    COVA_REQUIRE(scratch && buf && blocks > 0 && iters > 0 && n4 > 0);
    const float4 *b4 = reinterpret_cast<const float4 *>(buf);
    hipStream_t st = (hipStream_t)stream;
    if (loads_per_iter == 0)
        hipLaunchKernelGGL(mfma_load_probe_kernel<0>, dim3(blocks), dim3(512), 0, st, scratch, b4, n4, iters, 0.999f, 1.001f);
    else if (loads_per_iter == 1)
        hipLaunchKernelGGL(mfma_load_probe_kernel<1>, dim3(blocks), dim3(512), 0, st, scratch, b4, n4, iters, 0.999f, 1.001f);
    else if (loads_per_iter == 2)
        hipLaunchKernelGGL(mfma_load_probe_kernel<2>, dim3(blocks), dim3(512), 0, st, scratch, b4, n4, iters, 0.999f, 1.001f);
    else
        return COVA_ERR_BAD_ARG;
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

// mfma_blocks x (iters*16 MFMAs per wave)  +  stream_blocks streaming n4 float4 `passes` times
COVA_API int cova_probe_mix(float *scratch, const float *buf, long long n4, int mfma_blocks,
                            int stream_blocks, int iters, int passes, void *stream)
{
    COVA_REQUIRE(scratch && buf && mfma_blocks >= 0 && stream_blocks >= 0 && mfma_blocks + stream_blocks > 0);
    hipLaunchKernelGGL(mix_probe_kernel, dim3(mfma_blocks + stream_blocks), dim3(512), 0,
                       (hipStream_t)stream, scratch, reinterpret_cast<const float4 *>(buf), n4,
                       mfma_blocks, iters, passes, 0.999f, 1.001f);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

// launches `blocks` x 512 threads; every wave issues iters*16 MFMAs (each 4096 FLOP)
COVA_API int cova_probe_mfma_f32(float *scratch, int blocks, int iters, void *stream)
{
    COVA_REQUIRE(scratch && blocks > 0 && iters > 0);
    hipLaunchKernelGGL(mfma_f32_probe_kernel, dim3(blocks), dim3(512), 0, (hipStream_t)stream, scratch,
                       iters, 0.999f, 1.001f);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

// Lane-pattern probe: copies an NHWC-64 tensor (npix pixels) in runs of 32 pixels per wave with the
// lane -> (pixel, channel) mapping of  mode 0: the Winograd epilogue (16 lanes = 16 pixels 512 B apart,
// lane>>4 = 16-byte slice),  mode 1: fully contiguous (16 lanes = one pixel),  mode 2: 4 consecutive
// lanes = one 64-byte segment.  8 float4 loads in flight per lane, then 8 stores.  lds_bytes of dynamic
// LDS limit the blocks per CU (the conv kernels run 2 waves per SIMD).
__global__ __launch_bounds__(512) void lane_pattern_probe_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                                 long long nruns, int mode, int loads_only)
{
    extern __shared__ float dyn_lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (mode == 5 || mode == 6) {
        // the 64->256 forward convolution's OUTPUT pattern on a [R,256] map: a wave owns 32 rows x 256 channels.
        // mode 5: channel per lane (lane = channel n*32 + p of rows (r&3) + 8 (r>>2) + 4 h): 4-byte stores, two 128-byte
        //         runs per instruction; mode 6: the same bytes as float4 stores, 16 lanes per 256-byte run
        for (long long run = (long long)blockIdx.x * 8 + wave; run < nruns / 4; run += (long long)gridDim.x * 8) {
            float *dst = out + run * 32 * 256;
            if (mode == 5) {
#pragma unroll 1
                for (int n = 0; n < 8; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        dst[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 256 + n * 32 + (lane & 31)] = (float)(lane + r);
            } else {
#pragma unroll 1
                for (int n = 0; n < 8; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        *reinterpret_cast<float4 *>(dst + (4 * n + r) * 256 + lane * 4) = make_float4(1.f, 2.f, 3.f, (float)lane);
            }
        }
        return;
    }
    if (mode == 3 || mode == 4) {
        // the 1x1-convolution operand of a [R,256] map: a wave takes 32 rows of 1 KB, four chunks of 8 float4 per lane.
        // mode 3: row per lane (lane = row p | K-half h: 8 x 16 B walk one 128-byte line of the lane's own row);
        // mode 4: the same bytes with 8 adjacent lanes on one 128-byte line
        for (long long run = (long long)blockIdx.x * 8 + wave; run < nruns / 4; run += (long long)gridDim.x * 8) {
            const float *src = in + run * 32 * 256;
#pragma unroll 1
            for (int c = 0; c < 4; ++c) {
                float4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    int off;
                    if (mode == 3) off = (lane & 31) * 256 + (lane >> 5) * 128 + c * 32 + 4 * j;
                    else off = (4 * j + (lane >> 4)) * 256 + ((lane >> 3) & 1) * 128 + c * 32 + 4 * (lane & 7);
                    v[j] = *reinterpret_cast<const float4 *>(src + off);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
            }
        }
        if (acc.x == 12345.678f) *reinterpret_cast<float4 *>(out + threadIdx.x * 4) = acc;
        return;
    }
    for (long long run = (long long)blockIdx.x * 8 + wave; run < nruns; run += (long long)gridDim.x * 8) {
        const float *src = in + run * 32 * 64;
        float *dst = out + run * 32 * 64;
        int off[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (mode == 0) off[j] = (2 * (lane & 15) + (j & 1)) * 64 + (j >> 1) * 16 + (lane >> 4) * 4;
            else if (mode == 1) off[j] = ((lane >> 4) + 4 * j) * 64 + (lane & 15) * 4;
            else off[j] = ((lane >> 2) + 16 * (j & 1)) * 64 + (lane & 3) * 4 + 16 * (j >> 1);
        }
        float4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4 *>(src + off[j]);
        if (loads_only) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { acc.x += v[j].x; acc.y += v[j].y; acc.z += v[j].z; acc.w += v[j].w; }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) *reinterpret_cast<float4 *>(dst + off[j]) = v[j];
        }
    }
    if (loads_only && acc.x == 12345.678f) *reinterpret_cast<float4 *>(out + threadIdx.x * 4) = acc;
    if (dyn_lds[0] == 12345.678f && threadIdx.x == 9999) out[0] = 0.f;
}

COVA_API int cova_probe_lane_pattern(const float *in, float *out, long long npix, int mode, int loads_only,
                                     int blocks, int lds_bytes, void *stream)
{
    COVA_REQUIRE(in && out && npix > 0 && npix % 32 == 0 && mode >= 0 && mode <= 6 && blocks > 0 && lds_bytes >= 0);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(lane_pattern_probe_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(lane_pattern_probe_kernel, dim3(blocks), dim3(512), (size_t)lds_bytes, (hipStream_t)stream,
                       in, out, npix / 32, mode, loads_only);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

// ---- LDS poison: every CU's 160 KB of LDS filled with a bit pattern (a quiet NaN as float, a huge value as int) in front of a
// product launch -- a kernel that reads LDS it did not write then produces NaN instead of silently inheriting its predecessor's
// data (tools/poison_check.py with POISON_LDS=1)
__global__ __launch_bounds__(1024) void lds_fill_kernel(unsigned pattern, unsigned *sink)
{
    extern __shared__ unsigned s_all[];
    for (int i = threadIdx.x; i < 160 * 1024 / 4; i += 1024) s_all[i] = pattern;
    __syncthreads();
    if (sink != nullptr && s_all[(threadIdx.x * 37) % (160 * 1024 / 4)] != pattern) sink[0] = 1;       // (keeps the stores)
}

COVA_API int cova_probe_lds_fill(int pattern_bits, int blocks, void *stream)
{
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void *)lds_fill_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return 1;
        attr = true;
    }
    hipLaunchKernelGGL(lds_fill_kernel, dim3(blocks), dim3(1024), 160 * 1024, (hipStream_t)stream, (unsigned)pattern_bits,
                       (unsigned *)nullptr);
    return (int)hipGetLastError();
}
