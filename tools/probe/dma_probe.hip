// LDS-DMA throughput probe (round 6): how many bytes per clock does a CU move with global_load_lds_dwordx4 (1 KB per wave
// instruction, L2-resident source) against global_load_dwordx4 into registers (+ ds_write_b128 of the same bytes)?
// One block per CU, W waves per block, every wave issues N copies of 1 KB back to back from a 64 KB window (L1 / L2 hits),
// waits, repeats R times.  Prints bytes per clock per CU (s_memtime: 100 MHz -> converted with the measured kernel time instead:
// bytes / (seconds * 2.4e9)).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(3))) void lds_void;
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int N>
__global__ __launch_bounds__(512, 1) void probe(const float *__restrict__ src, float *__restrict__ sink, int reps, int window_kb)
{
    __shared__ __attribute__((aligned(1024))) float lds[8 * N * 256];      // N KB per wave
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds_base = (unsigned)(size_t)(lds_void *)(lds + wave * N * 256);
    const char *base = reinterpret_cast<const char *>(src) + (size_t)blockIdx.x * window_kb * 1024;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < reps; ++r) {
        const unsigned off0 = (unsigned)(((r * 8 + wave) * N * 1024) % (window_kb * 1024)) + (unsigned)lane * 16u;
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < N; ++i)
                asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_base + i * 1024), "v"(off0 + i * 1024), "s"(base) : "memory", "m0");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            f32x4 v[N];
#pragma unroll
            for (int i = 0; i < N; ++i) v[i] = *reinterpret_cast<const f32x4 *>(base + off0 + i * 1024);
            if (MODE == 2) {
#pragma unroll
                for (int i = 0; i < N; ++i) *reinterpret_cast<f32x4 *>(lds + wave * N * 256 + i * 256 + lane * 4) = v[i];
            } else {
#pragma unroll
                for (int i = 0; i < N; ++i) acc += v[i];
            }
        }
    }
    __syncthreads();
    if (MODE != 1) acc[0] += lds[threadIdx.x];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}

template <int MODE, int N>
static void run(const char *what, int waves, const float *src, float *sink, int window_kb)
{
    const int reps = 2000, grid = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) {
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL((probe<MODE, N>), dim3(grid), dim3(64 * waves), 0, 0, src, sink, reps, window_kb);
        (void)hipEventRecord(e1, 0);
        (void)hipEventSynchronize(e1);
    }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes_per_cu = (double)reps * waves * N * 1024.0;
    printf("%-44s waves %d  N %d  %8.3f ms  %6.1f B/clk/CU at 2.4 GHz  (%5.1f clk per 1 KB instruction per CU)\n", what, waves, N, ms,
           bytes_per_cu / (ms * 1e-3 * 2.4e9), ms * 1e-3 * 2.4e9 / (reps * waves * N));
}

int main()
{
    float *src, *sink;
    const int window_kb = 64;
    (void)hipMalloc(&src, (size_t)256 * window_kb * 1024);
    (void)hipMalloc(&sink, 256);
    (void)hipMemset(src, 0, (size_t)256 * window_kb * 1024);
    for (int waves : {1, 2, 4, 8}) {
        if (waves == 1) { run<0, 8>("global_load_lds_dwordx4 (LDS-DMA)", 1, src, sink, window_kb); run<1, 8>("global_load_dwordx4 -> registers", 1, src, sink, window_kb); run<2, 8>("global_load_dwordx4 + ds_write_b128", 1, src, sink, window_kb); }
        if (waves == 2) { run<0, 8>("global_load_lds_dwordx4 (LDS-DMA)", 2, src, sink, window_kb); run<1, 8>("global_load_dwordx4 -> registers", 2, src, sink, window_kb); run<2, 8>("global_load_dwordx4 + ds_write_b128", 2, src, sink, window_kb); }
        if (waves == 4) { run<0, 8>("global_load_lds_dwordx4 (LDS-DMA)", 4, src, sink, window_kb); run<1, 8>("global_load_dwordx4 -> registers", 4, src, sink, window_kb); run<2, 8>("global_load_dwordx4 + ds_write_b128", 4, src, sink, window_kb); }
        if (waves == 8) { run<0, 8>("global_load_lds_dwordx4 (LDS-DMA)", 8, src, sink, window_kb); run<1, 8>("global_load_dwordx4 -> registers", 8, src, sink, window_kb); run<2, 8>("global_load_dwordx4 + ds_write_b128", 8, src, sink, window_kb); }
    }
    return 0;
}
