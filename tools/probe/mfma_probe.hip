// Tools only: do the two waves of a SIMD overlap matrix and vector work?  f32 MFMA (16x16x4) against bf16 MFMA
// (16x16x32), each beside a VALU-only partner wave, beside an LDS-read partner, and interleaved in one wave.
//   hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize tools/probe/mfma_probe.hip -o gpurun_out/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

enum { IDLE = 0, MF32 = 1, MBF = 2, VALU = 3, MF32_V = 4, MBF_V = 5, LDSR = 6, MBF_L = 7 };

template <int R>
__device__ __forceinline__ float role(int iters, float seed, float* lds) {
    float r = 0.f;
    if constexpr (R == MF32) {
        f4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        float x = seed, y = seed * 0.5f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, a3, 0, 0, 0);
            }
        }
        r = a0[0] + a1[1] + a2[2] + a3[3];
    } else if constexpr (R == MBF) {
        f4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        bf8 x, y;
#pragma unroll
        for (int k = 0; k < 8; ++k) { x[k] = (__bf16)(seed + k); y[k] = (__bf16)(seed - k); }
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y, x, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, x, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y, y, a3, 0, 0, 0);
            }
        }
        r = a0[0] + a1[1] + a2[2] + a3[3];
    } else if constexpr (R == VALU) {
        float c[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) c[k] = seed + k;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 32; ++j) {
#pragma unroll
                for (int k = 0; k < 8; ++k) c[k] = fmaf(c[k], seed, 1.0f);
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) r += c[k];
    } else if constexpr (R == MF32_V || R == MBF_V) {
        // one wave: per MFMA, NV independent fmas
        f4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        float c[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) c[k] = seed + k;
        float x = seed, y = seed * 0.5f;
        bf8 bx, by;
#pragma unroll
        for (int k = 0; k < 8; ++k) { bx[k] = (__bf16)(seed + k); by[k] = (__bf16)(seed - k); }
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if constexpr (R == MF32_V) {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
#pragma unroll
                    for (int k = 0; k < 4; ++k) c[k] = fmaf(c[k], seed, 1.0f);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
#pragma unroll
                    for (int k = 4; k < 8; ++k) c[k] = fmaf(c[k], seed, 1.0f);
                    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, a2, 0, 0, 0);
#pragma unroll
                    for (int k = 0; k < 4; ++k) c[k] = fmaf(c[k], seed, 1.0f);
                    a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, a3, 0, 0, 0);
#pragma unroll
                    for (int k = 4; k < 8; ++k) c[k] = fmaf(c[k], seed, 1.0f);
                } else {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx, by, a0, 0, 0, 0);
#pragma unroll
                    for (int k = 0; k < 2; ++k) c[k] = fmaf(c[k], seed, 1.0f);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(by, bx, a1, 0, 0, 0);
#pragma unroll
                    for (int k = 2; k < 4; ++k) c[k] = fmaf(c[k], seed, 1.0f);
                    a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx, bx, a2, 0, 0, 0);
#pragma unroll
                    for (int k = 4; k < 6; ++k) c[k] = fmaf(c[k], seed, 1.0f);
                    a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(by, by, a3, 0, 0, 0);
#pragma unroll
                    for (int k = 6; k < 8; ++k) c[k] = fmaf(c[k], seed, 1.0f);
                }
            }
        }
        r = a0[0] + a1[1] + a2[2] + a3[3];
#pragma unroll
        for (int k = 0; k < 8; ++k) r += c[k];
    } else if constexpr (R == LDSR) {
        // 64 ds_read_b128 per iteration (1 KiB per wave each)
        f4 acc = {0, 0, 0, 0};
        const f4* p = (const f4*)lds + (threadIdx.x & 63);
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 64; ++j) {
                f4 v = *(volatile const f4*)(p + ((j * 64) & 1023));
                acc += v;
            }
        }
        r = acc[0] + acc[1] + acc[2] + acc[3];
    } else if constexpr (R == MBF_L) {
        // one wave: bf16 MFMAs whose B operand is read from LDS every time (16 B per lane per MFMA)
        f4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
        bf8 bx;
#pragma unroll
        for (int k = 0; k < 8; ++k) bx[k] = (__bf16)(seed + k);
        const bf8* p = (const bf8*)lds + (threadIdx.x & 63);
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                bf8 b0 = *(volatile const bf8*)(p + ((j * 256) & 1023));
                bf8 b1 = *(volatile const bf8*)(p + ((j * 256 + 64) & 1023));
                bf8 b2 = *(volatile const bf8*)(p + ((j * 256 + 128) & 1023));
                bf8 b3 = *(volatile const bf8*)(p + ((j * 256 + 192) & 1023));
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx, b0, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx, b1, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx, b2, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bx, b3, a3, 0, 0, 0);
            }
        }
        r = a0[0] + a1[1] + a2[2] + a3[3];
    }
    return r;
}

template <int RA, int RB>
__global__ __launch_bounds__(512) void probe(int iters, float seed, float* out) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = seed + i;
    __syncthreads();
    const int w = threadIdx.x >> 6;
    float r;
    if (w < 4) r = role<RA>(iters, seed, lds);
    else r = role<RB>(iters, seed, lds);
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int RA, int RB>
float run(const char* name, int iters, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)probe<RA, RB>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    probe<RA, RB><<<256, 512, 100 * 1024>>>(iters / 8, 0.001f, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<RA, RB><<<256, 512, 100 * 1024>>>(iters, 0.001f, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %8.3f ms  %9.1f ns/iter\n", name, ms, ms * 1e6 / iters);
    return ms;
}

int main() {
    float* out; hipMalloc(&out, 4096);
    const int it = 2000;
    // per iteration: MF32 = 64 MFMAs (2048 cyc), MBF = 64 MFMAs (1024 cyc), VALU = 256 fma, LDSR = 64 reads
    run<MF32, IDLE>("f32 mfma | idle", it, out);
    run<MF32, MF32>("f32 mfma | f32 mfma", it, out);
    run<IDLE, VALU>("idle | valu", it, out);
    run<VALU, VALU>("valu | valu", it, out);
    run<MF32, VALU>("f32 mfma | valu", it, out);
    run<MBF, IDLE>("bf16 mfma | idle", it, out);
    run<MBF, MBF>("bf16 mfma | bf16 mfma", it, out);
    run<MBF, VALU>("bf16 mfma | valu", it, out);
    run<MF32_V, IDLE>("f32 mfma+4valu each | idle", it, out);
    run<MBF_V, IDLE>("bf16 mfma+2valu each | idle", it, out);
    run<MF32_V, MF32_V>("f32 mfma+4valu x2", it, out);
    run<MBF_V, MBF_V>("bf16 mfma+2valu x2", it, out);
    run<IDLE, LDSR>("idle | lds b128", it, out);
    run<LDSR, LDSR>("lds b128 x2", it, out);
    run<MBF, LDSR>("bf16 mfma | lds b128", it, out);
    run<MF32, LDSR>("f32 mfma | lds b128", it, out);
    run<MBF_L, IDLE>("bf16 mfma B from lds | idle", it, out);
    run<MBF_L, MBF_L>("bf16 mfma B from lds x2", it, out);
    run<MBF_L, VALU>("bf16 mfma B from lds | valu", it, out);
    return 0;
}
