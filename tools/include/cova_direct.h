/* cova_direct.h -- tools/lib/libcova_direct.so: the direct-form (implicit GEMM) 3x3 / 64->64 convolution kernels of
 * round 1 (tools/csrc/conv3x3_direct.hip).  Not part of the product library; kept as the measured reference point of
 * DESIGN.md section 4.1 and for tools/conv_bench.py.  Same conventions as include/cova_hip.h. */
#ifndef COVA_DIRECT_H
#define COVA_DIRECT_H
#ifdef __cplusplus
extern "C" {
#endif
int cova_direct_set_option(int key, int value);   /* 2 = cap on the persistent grids, 5 = ablation mask */
int cova_conv3x3_num_tiles(int B, int H, int W);
int cova_conv3x3_prep_weights(const float *w_oihw /*[64,64,3,3]*/, float *w_fwd /*[9,64,64]*/,
                              float *w_dgrad /*[9,64,64]*/, void *stream);
int cova_conv3x3_fwd(const float *in, const float *w_t, const float *addend, float *out,
                     float *stat_part, int B, int H, int W, void *stream);
int cova_conv3x3_dgrad_bnbwd(const float *dz, const float *w_dgrad, const float *addend,
                             const float *act, const float *z, const float *mean, const float *invstd,
                             float *dy, float *stat_part, int B, int H, int W, void *stream);
int cova_conv3x3_wgrad(const float *act, const float *dz, float *dw /*OIHW*/, float *ws /* >= 74,000 floats per CU */,
                       int B, int H, int W, void *stream);
#ifdef __cplusplus
}
#endif
#endif
