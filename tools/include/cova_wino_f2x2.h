/* C ABI of the F(2x2,3x3) Winograd kernels (tools/csrc/conv_wino_f2x2.hip): test-support library, NOT part of the product
 * (they were the 3x3 kernels of rounds 1-3; the product runs F(4x4,3x3): include/cova_hip.h).  tests/conftest.py loads
 * tools/lib/libcova_f2x2.so beside the product library so that the kernel tests can cross-check the two families. */
#ifndef COVA_WINO_F2X2_H
#define COVA_WINO_F2X2_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* 2 = cap on persistent grids (as cova_set_option 2), 5 = ablation mask of -DCOVA_ABLATE builds, 6 = tile geometry: 1 GeoA
 * (8x32 tiles, one block per CU; default) | 2 GeoB (8x16, two blocks per CU) */
int cova_wino_f2x2_set_option(int key, int value);

/* nn.Conv2d(64,64,3,1,1,bias=False) on NHWC [B,H,W,64] (torchvision BasicBlock / Bottleneck conv2; models.py:49-51), as
 * Winograd convolutions in exact f32 arithmetic (the direct implicit-GEMM kernels of round 1 live in tools/csrc).
 * F(2x2,3x3) form (2.25x fewer MFMAs than direct, same fp32 error): weights are transformed once per step into
 * u_fwd / u_dgrad [16,16,4,64]; with u_dgrad it is the data gradient; addend (nullable, NHWC) is added to the result
 * (residual-branch gradient).  With act / z / mean / invstd the data gradient is fused with the ReLU mask and the
 * BatchNorm-backward reduction of the layer in front of the conv: out = dy = (conv + addend) * (act > 0), stat_part =
 * (sum dy, sum dy*xhat); otherwise stat_part (nullable) = (sum y, sum y^2).  stat_part has one row per persistent
 * block: [cova_conv3x3_wino_num_partials][2][64]. */
/* rows of the statistics partials of cova_conv3x3_wino(_pro): one per persistent block (depends on the
 * device's CU count and on cova_wino_f2x2_set_option 2 / 6: query it right before allocating) */
int cova_conv3x3_wino_num_partials(int B, int H, int W);
int cova_conv3x3_prep_weights_wino(const float *w_oihw, float *u_fwd, float *u_dgrad, void *stream);
int cova_conv3x3_wino(const float *in, const float *u, const float *addend, const float *act,
                      const float *z, const float *mean, const float *invstd, float *out,
                      float *stat_part, int B, int H, int W, void *stream);
/* Inference form of cova_conv3x3_wino: out = f(scale[c]*conv(in) + shift[c] + addend), f = ReLU if relu --
 * the BatchNorm (running statistics, cova_bn_eval_params), residual add (addend, nullable) and ReLU that
 * follow each conv of a BasicBlock (models.py:49-51 -> torchvision BasicBlock.forward), evaluated in the
 * epilogue in cova_bn_act_fwd's operation order (bit-identical to conv + cova_bn_act_fwd). */
int cova_conv3x3_wino_bnact(const float *in, const float *u, const float *addend, const float *scale,
                            const float *shift, int relu, float *out, int B, int H, int W, void *stream);
/* input transformed on load: f(A[c]*in + B[c]*in2 + C[c]), f = ReLU if pro_relu; pro_abc [3,64]
 * (nullable = plain input); in2 nullable (B ignored).  Folds BatchNorm+ReLU (models.py:49-51 via
 * torchvision BasicBlock bn1/relu), or the BatchNorm-backward apply, into the consuming conv.
 * Epilogue mask: act > 0, or fma(mask_scale, z, mask_shift) > 0 when act == NULL. */
int cova_conv3x3_wino_pro(const float *in, const float *in2 /*nullable*/,
                          const float *pro_abc /*nullable*/, int pro_relu, const float *u,
                          const float *addend /*nullable*/, const float *act /*nullable*/,
                          const float *mask_scale /*nullable*/, const float *mask_shift /*nullable*/,
                          const float *z /*nullable*/, const float *mean /*nullable*/,
                          const float *invstd /*nullable*/, float *out, float *stat_part /*nullable*/,
                          int B, int H, int W, void *stream);
int cova_conv3x3_wgrad_wino(const float *act, const float *dz, float *dw /*OIHW*/, float *ws, int B,
                            int H, int W, void *stream);   /* weight gradient, Winograd F(2x2,3x3) form */
int cova_conv3x3_wgrad_wino_pro(const float *act, const float *act_abc /*nullable*/, int act_relu,
                                const float *dz, const float *dz2 /*nullable*/,
                                const float *dz_abc /*nullable*/, float *dw, float *ws, int B, int H,
                                int W, void *stream);
/* the same in two steps, so that one launch finishes the weight gradients of several convolutions: the per-block
 * partial sums only (own workspace per convolution) ... */
int cova_conv3x3_wgrad_wino_partial(const float *act, const float *act_abc, int act_relu, const float *dz,
                                    const float *dz2, const float *dz_abc, float *ws, int B, int H, int W,
                                    void *stream);
/* ... and the fp64 fold + final transform of up to four of them (pairs 1..3 nullable) into OIHW [64,64,3,3] */
int cova_conv3x3_wgrad_wino_finish(const float *ws0, float *dw0, const float *ws1, float *dw1, const float *ws2,
                                   float *dw2, const float *ws3, float *dw3, int B, int H, int W, void *stream);
int cova_conv3x3_wgrad_workspace_floats(int B, int H, int W);

#ifdef __cplusplus
}
#endif
#endif
