// Tools only: issue rate of v_mfma_f32_16x16x32_bf16 from ONE wave, alone on its SIMD and beside a second wave doing the same:
//   R<n>   the wave rotates over n independent accumulators (n = 1: one dependent chain),
//   C6     six accumulators, each takes SIX back-to-back dependent MFMAs before the next one (the split F(4x4) loop's order),
//   C6v<k> as C6 with k independent v_fma_f32 between the chains,
//   I<n>_<k> n accumulators round robin with k independent v_fma_f32 after EVERY MFMA,
//   X2     two dependent chains interleaved A B A B ... (every dependent pair has one foreign MFMA between it).
// Cycles are shader-clock ticks (s_memtime) of wave 0.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/mfma16_probe.hip -o build/mfma16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

#define MF(acc) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc, 0, 0, 0)

template <int MODE, int WAVES>
__global__ __launch_bounds__(512) void probe(int iters, float seed, float* out, unsigned long long* cyc) {
    extern __shared__ float lds[];
    const int w = threadIdx.x >> 6;
    float r = 0.f;
    if (w < 4 * WAVES) {
        f4v acc[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) acc[a] = f4v{0.f, 0.f, 0.f, 0.f};
        bf8 x, y;
#pragma unroll
        for (int k = 0; k < 8; ++k) { x[k] = (__bf16)(seed + k); y[k] = (__bf16)(seed - k); }
        float f0 = seed, f1 = seed * 2, f2 = seed * 3, f3 = seed * 4;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < iters; ++i) {
            if (MODE >= 1 && MODE <= 6) {               // R<n>: 36 MFMAs round robin over n accumulators
#pragma unroll
                for (int j = 0; j < 36 / MODE; ++j)
#pragma unroll
                    for (int a = 0; a < MODE; ++a) MF(acc[a]);
            } else if (MODE >= 10 && MODE < 20) {       // C6 / C6v<k>
#pragma unroll
                for (int a = 0; a < 6; ++a) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 6; ++j) MF(acc[a]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < MODE - 10; ++k) {
                        f0 = __builtin_fmaf(f0, f1, f2); f1 = __builtin_fmaf(f1, f2, f3); f2 = __builtin_fmaf(f2, f3, f0); f3 = __builtin_fmaf(f3, f0, f1);
                    }
                }
            } else if (MODE >= 100) {                   // I<n>_<k>: n accumulators round robin, k independent fmas after EVERY MFMA
                constexpr int NA = MODE / 100, K = MODE % 100;
#pragma unroll
                for (int j = 0; j < 36 / NA; ++j)
#pragma unroll
                    for (int a = 0; a < NA; ++a) {
                        MF(acc[a]);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int k = 0; k < K; ++k) {
                            if ((k & 3) == 0) f0 = __builtin_fmaf(f0, seed, seed);
                            else if ((k & 3) == 1) f1 = __builtin_fmaf(f1, seed, seed);
                            else if ((k & 3) == 2) f2 = __builtin_fmaf(f2, seed, seed);
                            else f3 = __builtin_fmaf(f3, seed, seed);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
            } else {                                    // X2
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                    for (int j = 0; j < 6; ++j) { MF(acc[2 * p]); MF(acc[2 * p + 1]); }
            }
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
#pragma unroll
        for (int a = 0; a < 6; ++a) r += acc[a][a & 3];
        r += f0 + f1 + f2 + f3;
    }
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int MODE, int WAVES>
void run(const char* name, int iters, float* out, unsigned long long* cyc) {
    (void)hipFuncSetAttribute((const void*)probe<MODE, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    probe<MODE, WAVES><<<256, 512, 100 * 1024>>>(iters / 8, 0.001f, out, cyc);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    probe<MODE, WAVES><<<256, 512, 100 * 1024>>>(iters, 0.001f, out, cyc);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c = 0;
    (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-6s %d wave(s) per SIMD: %6.1f ticks, %6.2f ns per MFMA per wave; chip %7.1f TFLOP/s (dense bf16)\n", name, WAVES,
           (double)c / iters / 36.0, ms * 1e6 / iters / 36.0, 2.0 * 8192 * 36.0 * iters * 256 * 4 * WAVES / (ms * 1e-3) * 1e-12);
}

int main() {
    float* out; (void)hipMalloc(&out, 4096);
    unsigned long long* cyc; (void)hipMalloc(&cyc, 64);
    const int it = 4000;
#define BOTH(M, name) run<M, 1>(name, it, out, cyc); run<M, 2>(name, it, out, cyc);
    BOTH(1, "R1") BOTH(2, "R2") BOTH(3, "R3") BOTH(4, "R4") BOTH(6, "R6") BOTH(10, "C6") BOTH(12, "C6v2") BOTH(16, "C6v6") BOTH(30, "X2")
    BOTH(102, "I1_2") BOTH(103, "I1_3") BOTH(104, "I1_4") BOTH(106, "I1_6") BOTH(202, "I2_2") BOTH(203, "I2_3") BOTH(204, "I2_4") BOTH(206, "I2_6")
    BOTH(603, "I6_3") BOTH(604, "I6_4") BOTH(606, "I6_6") BOTH(608, "I6_8")
    return 0;
}
