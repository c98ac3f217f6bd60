// Tools only: how v_mfma_f32_16x16x32_bf16 rounds C + sum(products) when the products are far below ulp(C), against
// v_mfma_f32_16x16x4_f32: C = 1.0 (and -1.0), ONE non-zero product x * 1 with x = +-2^-k.  Round-to-nearest keeps C for
// |x| < 2^-25; a truncating adder (two's complement: toward -inf) returns the float below C for every negative x.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/mfma_round_probe.hip -o build/mfma_round_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

__global__ void probe(const float* xs, float cval, float* out_bf, float* out_f32, int n) {
    const int lane = threadIdx.x;
    for (int t = 0; t < n; ++t) {
        bf8 a, b;
        for (int k = 0; k < 8; ++k) { a[k] = (__bf16)0.f; b[k] = (__bf16)0.f; }
        if (lane < 16) { a[0] = (__bf16)xs[t]; }          // A[row = lane][k = 0] = x for all 16 rows
        if (lane < 16) { b[0] = (__bf16)1.f; }            // B[k = 0][col = lane] = 1
        f4v c = {cval, cval, cval, cval};
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
        if (lane == 0) out_bf[t] = c[0];
        f4v d = {cval, cval, cval, cval};
        d = __builtin_amdgcn_mfma_f32_16x16x4f32(lane < 16 ? xs[t] : 0.f, lane < 16 ? 1.f : 0.f, d, 0, 0, 0);
        if (lane == 0) out_f32[t] = d[0];
    }
}

int main() {
    const int n = 16;
    float h[n];
    for (int i = 0; i < 8; ++i) { h[2 * i] = ldexpf(1.f, -24 - i); h[2 * i + 1] = -ldexpf(1.f, -24 - i); }
    float *xs, *ob, *of;
    (void)hipMalloc(&xs, sizeof(h)); (void)hipMalloc(&ob, sizeof(h)); (void)hipMalloc(&of, sizeof(h));
    (void)hipMemcpy(xs, h, sizeof(h), hipMemcpyHostToDevice);
    for (float cval : {1.0f, -1.0f, 1.5f}) {
        probe<<<1, 64>>>(xs, cval, ob, of, n);
        float rb[n], rf[n];
        (void)hipMemcpy(rb, ob, sizeof(rb), hipMemcpyDeviceToHost);
        (void)hipMemcpy(rf, of, sizeof(rf), hipMemcpyDeviceToHost);
        printf("C = %g\n", cval);
        for (int i = 0; i < n; ++i)
            printf("  x = %+.3e (2^%d): bf16 MFMA  C %+d ulp   f32 MFMA  C %+d ulp   (nearest: %+d)\n", h[i], -24 - i / 2,
                   (int)lrintf((rb[i] - cval) / ldexpf(1.f, -24 + (fabsf(cval) >= 1.f && rb[i] * (cval > 0 ? 1 : -1) >= fabsf(cval) ? 1 : 0))),
                   (int)lrintf((rf[i] - cval) / ldexpf(1.f, -24 + (fabsf(cval) >= 1.f && rf[i] * (cval > 0 ? 1 : -1) >= fabsf(cval) ? 1 : 0))),
                   (int)lrint(((double)(float)((double)cval + (double)h[i]) - cval) / ldexp(1.0, -24)));
    }
    return 0;
}
