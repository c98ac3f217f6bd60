// Internal helpers shared by the gfx950 kernels of the CoVA hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define COVA_OK 0
#define COVA_ERR_BAD_ARG 10001

#define COVA_LAUNCH_CHECK()                         \
    do {                                            \
        hipError_t e__ = hipGetLastError();         \
        if (e__ != hipSuccess) return (int)e__;     \
    } while (0)

#define COVA_REQUIRE(cond)                          \
    do {                                            \
        if (!(cond)) return COVA_ERR_BAD_ARG;       \
    } while (0)

// Ablation switches of the conv kernels (tools/conv_bench.py) exist only in builds made with
// -DCOVA_ABLATE; in the production library the mask folds to 0 and the branches disappear.
#ifdef COVA_ABLATE
#define COVA_ABL(x) (x)
#else
#define COVA_ABL(x) 0
#endif

#define COVA_API extern "C" __attribute__((visibility("default")))

// neighbour slots per node the wave-per-node GAT kernels hold in registers: up to sixteen 64-lane passes (-cs <= 512: far
// beyond the boxes of a page -- slots past a page's size are pads; models.py:171-177 itself takes any n_context)
#define COVA_GAT_MAX_K 1024

// v_mfma_f32_32x32x2_f32: D[32x32] += A[32x2] * B[2x32], exact f32 FMA chain.
// lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31];
// D register r of lane l is D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31].
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ int mfma32_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Sum over the 16 lanes of a DPP row (lanes 16r..16r+15); every lane ends up with the row's total.
// Four v_add_f32 with DPP operands (quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror,
// row_mirror) instead of four ds_bpermute round trips through the LDS crossbar.
__device__ __forceinline__ float row16_sum(float v)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// counter-based uniform in [0,1) of the dropout masks (models.py:84,88): element idx of the stream `seed`
__device__ __forceinline__ float hash_uniform(unsigned long long seed, unsigned long long idx)
{
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (float)(z >> 40) * (1.0f / 16777216.0f);   // 24 random bits -> [0,1)
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline long long cdivll(long long a, long long b) { return (a + b - 1) / b; }
