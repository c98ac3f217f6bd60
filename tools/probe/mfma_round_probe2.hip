// Tools only: v_mfma_f32_16x16x32_bf16 with ALL 32 products small against C: C = 1.0, every product = s * 2^-k (k = 25..34),
// exact sum = 32 * s * 2^-k.  A dot-product unit that aligns the products to the largest exponent and drops the bits below a
// guard width shows (a) small terms vanishing and (b) a sign-dependent (toward -inf) offset.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

__global__ void probe(const float* xs, float cval, float* out, int n) {
    const int lane = threadIdx.x;
    for (int t = 0; t < n; ++t) {
        bf8 a, b;
        for (int k = 0; k < 8; ++k) { a[k] = (__bf16)xs[t]; b[k] = (__bf16)1.f; }
        f4v c = {cval, cval, cval, cval};
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
        if (lane == 0) out[t] = c[0];
    }
}

int main() {
    const int n = 20;
    float h[n];
    for (int i = 0; i < 10; ++i) { h[2 * i] = ldexpf(1.f, -25 - i); h[2 * i + 1] = -ldexpf(1.f, -25 - i); }
    float *xs, *ob;
    (void)hipMalloc(&xs, sizeof(h)); (void)hipMalloc(&ob, sizeof(h));
    (void)hipMemcpy(xs, h, sizeof(h), hipMemcpyHostToDevice);
    for (float cval : {1.0f, -1.0f, 0.0f}) {
        probe<<<1, 64>>>(xs, cval, ob, n);
        float r[n];
        (void)hipMemcpy(r, ob, sizeof(r), hipMemcpyDeviceToHost);
        printf("C = %g, 32 products of x each\n", cval);
        for (int i = 0; i < n; ++i) {
            const double exact = (double)cval + 32.0 * (double)h[i];
            printf("  x = %+.3e (2^%d): result - C = %+.6e   exact sum %+.6e   nearest f32 - C %+.6e\n", h[i], -25 - i / 2, (double)r[i] - cval,
                   32.0 * (double)h[i], (double)(float)exact - cval);
        }
    }
    return 0;
}
