// f32 operands on the bf16 matrix pipe: x = x0 + x1 + x2, three round-to-nearest bf16 pieces.
#pragma once
#include "common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// (xe, xo) -> three dwords of packed bf16 pieces (even element in the low half): 11 operations.  Each piece is the
// round-to-nearest-even bf16 of what is left (v_cvt_pk_bf16_f32), each residual is exact in f32; after three pieces
// at most 2^-26 |x| is left.  (Pieces by truncation represent x exactly but are all of x's sign: the three dropped
// cross products then add up coherently -- measured as a 10x larger error of the per-channel sums of the convolution.)
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t bf3_pack_rne(float xe, float xo)
{
    const f32x2 v = {xe, xo};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ void bf3_split_pair(float xe, float xo, uint32_t &q0, uint32_t &q1, uint32_t &q2)
{
    q0 = bf3_pack_rne(xe, xo);
    const float re = xe - __builtin_bit_cast(float, q0 << 16);
    const float ro = xo - __builtin_bit_cast(float, q0 & 0xFFFF0000u);
    q1 = bf3_pack_rne(re, ro);
    const float se = re - __builtin_bit_cast(float, q1 << 16);
    const float so = ro - __builtin_bit_cast(float, q1 & 0xFFFF0000u);
    q2 = bf3_pack_rne(se, so);
}

__device__ __forceinline__ void bf3_split8(const float (&x)[8], u32x4 &p0, u32x4 &p1, u32x4 &p2)
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t a, b, c;
        bf3_split_pair(x[2 * i], x[2 * i + 1], a, b, c);
        p0[i] = a; p1[i] = b; p2[i] = c;
    }
}
