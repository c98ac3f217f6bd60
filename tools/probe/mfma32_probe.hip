// Tools only: issue rate of v_mfma_f32_32x32x16_bf16 in ONE wave as a function of the number of independent accumulators
// it rotates over (1, 2, 3, 4, 6), alone on its SIMD and beside a second wave doing the same.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/mfma32_probe.hip -o build/mfma32_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

template <int NACC, int WAVES>
__global__ __launch_bounds__(512) void probe(int iters, float seed, float* out) {
    extern __shared__ float lds[];
    const int w = threadIdx.x >> 6;
    float r = 0.f;
    if (w < 4 * WAVES) {
        f16v acc[NACC];
#pragma unroll
        for (int a = 0; a < NACC; ++a)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[a][k] = 0.f;
        bf8 x, y;
#pragma unroll
        for (int k = 0; k < 8; ++k) { x[k] = (__bf16)(seed + k); y[k] = (__bf16)(seed - k); }
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 48 / NACC; ++j)
#pragma unroll
                for (int a = 0; a < NACC; ++a)
                    acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
        }
#pragma unroll
        for (int a = 0; a < NACC; ++a) r += acc[a][a];
    }
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int NACC, int WAVES>
void run(int iters, float* out) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute((const void*)probe<NACC, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    probe<NACC, WAVES><<<256, 512, 100 * 1024>>>(iters / 8, 0.001f, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    probe<NACC, WAVES><<<256, 512, 100 * 1024>>>(iters, 0.001f, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const int n = 48 / NACC * NACC;
    printf("%d accumulators, %d wave(s) per SIMD: %7.1f ns per MFMA per wave  (%5.1f cycles at 2.28 GHz; pipe share %3.0f %%)\n", NACC, WAVES,
           ms * 1e6 / iters / n, ms * 1e6 / iters / n * 2.28, 100.0 * WAVES * 32.0 / (ms * 1e6 / iters / n * 2.28));
}

int main() {
    float* out; (void)hipMalloc(&out, 4096);
    const int it = 2000;
    run<1, 1>(it, out); run<2, 1>(it, out); run<3, 1>(it, out); run<4, 1>(it, out); run<6, 1>(it, out);
    run<1, 2>(it, out); run<2, 2>(it, out); run<3, 2>(it, out); run<4, 2>(it, out); run<6, 2>(it, out);
    return 0;
}
