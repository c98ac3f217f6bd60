// Dispatch-gap probe (round 6): what makes the ~6-10 us between the end of a big launch and the start of the next one in the
// bench step?  Sequences  A ; tiny ; tiny  for several kinds of A -- run under `rocprofv3 --kernel-trace`,
// tools/gap_probe.py prints the gap in front of the first `tiny` by kind of A.
//   wr<0> 400 MB plain stores | wr<1> nontemporal stores | wr<2> sc0 sc1 (write-through) stores | wr<0> 4 MB | rd 400 MB |
//   lds: 256 blocks x 512 threads x 160 KB of LDS, no memory traffic | wr<0> 400 MB followed by ANOTHER wr<0> (big after big)
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void wr_kernel(float4 *__restrict__ out, size_t n4, float v)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 x = make_float4(v, v + 1.f, v + 2.f, (float)i);
        if (MODE == 0) out[i] = x;
        else if (MODE == 1) {
            typedef float v4f __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(v4f{x.x, x.y, x.z, x.w}, reinterpret_cast<v4f *>(out + i));
        }
        else {
            typedef float v4f __attribute__((ext_vector_type(4)));
            const v4f xv = {x.x, x.y, x.z, x.w};
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(out + i), "v"(xv) : "memory");
        }
    }
}

__global__ __launch_bounds__(256) void rd_kernel(const float4 *__restrict__ in, size_t n4, float *__restrict__ sink)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 x = in[i];
        s += x.x + x.y + x.z + x.w;
    }
    if (s == 123.456f) sink[0] = s;
}

__global__ __launch_bounds__(512, 1) void lds_kernel(float *__restrict__ sink, int iters)
{
    __shared__ float buf[40000];              // 160,000 B: one block per CU
    for (int i = threadIdx.x; i < 40000; i += 512) buf[i] = (float)i;
    __syncthreads();
    float s = 0.f;
    for (int k = 0; k < iters; ++k) s += buf[(threadIdx.x * 33 + k * 7) % 40000];
    if (s == 123.456f) sink[0] = s;
}

__global__ void tiny_kernel(float *p) { if (threadIdx.x == 0 && p[1] == 123.f) p[0] = 1.f; }
__global__ void mark_kernel(float *p, int tag) { if (threadIdx.x == 0 && p[1] == 123.f) p[0] = (float)tag; }

int main()
{
    const size_t big = (size_t)400 << 20, small = (size_t)4 << 20;
    float4 *buf; float *sink;
    (void)hipMalloc(&buf, big); (void)hipMalloc(&sink, 256); (void)hipMemset(sink, 0, 256); (void)hipMemset(buf, 0, big);
    const int grid = 256 * 8;
    for (int rep = 0; rep < 6; ++rep) {
        for (int kind = 0; kind < 8; ++kind) {
            hipLaunchKernelGGL(mark_kernel, dim3(1), dim3(64), 0, 0, sink, kind);
            switch (kind) {
            case 0: hipLaunchKernelGGL(wr_kernel<0>, dim3(grid), dim3(256), 0, 0, buf, big / 16, 1.f); break;
            case 1: hipLaunchKernelGGL(wr_kernel<1>, dim3(grid), dim3(256), 0, 0, buf, big / 16, 1.f); break;
            case 2: hipLaunchKernelGGL(wr_kernel<2>, dim3(grid), dim3(256), 0, 0, buf, big / 16, 1.f); break;
            case 3: hipLaunchKernelGGL(wr_kernel<0>, dim3(grid), dim3(256), 0, 0, buf, small / 16, 1.f); break;
            case 4: hipLaunchKernelGGL(rd_kernel, dim3(grid), dim3(256), 0, 0, buf, big / 16, sink); break;
            case 5: hipLaunchKernelGGL(lds_kernel, dim3(256), dim3(512), 0, 0, sink, 2000); break;
            case 6: hipLaunchKernelGGL(wr_kernel<0>, dim3(grid), dim3(256), 0, 0, buf, big / 16, 1.f);
                    hipLaunchKernelGGL(wr_kernel<0>, dim3(grid), dim3(256), 0, 0, buf, big / 16, 2.f); break;
            case 7: hipLaunchKernelGGL(wr_kernel<1>, dim3(grid), dim3(256), 0, 0, buf, big / 16, 1.f);
                    hipLaunchKernelGGL(wr_kernel<1>, dim3(grid), dim3(256), 0, 0, buf, big / 16, 2.f); break;
            }
            hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, 0, sink);
            hipLaunchKernelGGL(tiny_kernel, dim3(1), dim3(64), 0, 0, sink);
        }
        (void)hipDeviceSynchronize();
    }
    printf("done\n");
    return 0;
}
