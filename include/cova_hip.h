/*
 * cova_hip.h -- C ABI of libcova_hip.so: the MI355X (gfx950) implementation of the CoVA
 * forward/backward hot path.
 *
 * The reference (kevalmorabia97/CoVA-Web-Object-Detection) is pure Python: its hot path is
 * `models.CoVA.forward` + autograd backward (models.py:94-122, train.py:47-60) and every kernel
 * it runs comes from torch / torchvision.  It has no FFI of its own, so this boundary is what a
 * native replacement of those torch/torchvision operator calls has to export (SURVEY.md
 * section 8b): one entry point per fused stage of the path, each citing the reference call it
 * replaces.  The reference-side binding is a ctypes stub; the host mirror of
 * the reference's nn.Module surface lives in cova-web-object-detection_amd/models.py.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless stated; the library never allocates:
 *     outputs and workspaces are caller-provided, sizes via the *_num_* / *_workspace_* queries;
 *   - `stream` is a hipStream_t (pass torch.cuda.current_stream().cuda_stream); all work is
 *     asynchronous on it, nothing synchronises, so calls are hipGraph-capturable;
 *   - return value: 0 on success, a hipError_t value, or COVA_ERR_BAD_ARG (10001);
 *   - all floating point data is IEEE fp32 (the reference computes in fp32), indices int64
 *     as in the reference, RoIPool argmax int32;
 *   - activations of the conv stack are NHWC ([B,H,W,64]; channel = fastest); the image is the
 *     reference's NCHW [B,3,H,W]; dense matrices are row-major with a leading dimension `ld*`
 *     counted in floats.
 */
#ifndef COVA_HIP_H
#define COVA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define COVA_ERR_BAD_ARG 10001

/* BatchNorm finalize as the TAIL of the launch that produced the statistics partials (cova_conv1_fwd_tail,
 * cova_conv3x3_wino4_full_tail): the last block to finish folds the partial rows and writes what
 * cova_bn_finalize_fwd (mode 1) or cova_bn_finalize_bwd_abc (mode 2) would write in a launch of its own -- same
 * fp64 arithmetic.  A HOST struct of device pointers, read at launch time.  `counter`: one device int, zero before the
 * first launch; the kernel leaves it zero.  64 channels. */
typedef struct cova_bn_tail {
    int mode;                                   /* 0 none | 1 forward statistics | 2 backward sums */
    int *counter;
    double count;                               /* elements per channel */
    const float *gamma, *beta;                  /* mode 1 */
    float *running_mean, *running_var;          /* mode 1, nullable: momentum update */
    long long *num_batches_tracked;             /* mode 1, nullable: += 1 */
    float momentum, eps;
    float *scale, *shift, *mean, *invstd;       /* mode 1: outputs | mode 2: inputs (mean, invstd, scale) */
    float *dgamma, *dbeta, *abc;                /* mode 2 outputs: dgamma / dbeta (nullable), abc [3][64] */
} cova_bn_tail;

/* ------------------------------------------------------------------ conv stack (models.py:49-51)
 * replaces: torchvision resnet18 children()[:-5] = nn.Conv2d(3,64,7,2,3), nn.BatchNorm2d(64),
 * nn.ReLU, nn.MaxPool2d(3,2,1), 2 x BasicBlock(64) as called at models.py:125 `self.convnet(images)`
 */
int cova_conv_out_size(int in_size, int kernel, int stride, int pad);
/* Test / A-B hooks, not part of the path's contract (six keys; everything else is refused):
 *   2  = cap on the persistent grids (tests force many tiles per block); 0 = none,
 *   7  = conv1 forward and weight gradient on the f32-MFMA kernels (1) instead of the bf16-split ones (0, default): bench.py's `ab` leg,
 *   9  = F(4x4,3x3) forward / data-gradient launches on the f32-MFMA main loop (1) instead of the bf16-split one (0, default): `ab` leg,
 *   14 = cova_bn1d_fwd / _bwd in the float4 form (1, default: taken when the operands are 16-byte aligned) or the 128-slice form (0),
 *   16 = cova_gat_fwd / _bwd with every 64-channel chunk of a neighbour row in flight (1, default; K <= 64, D <= 512) or chunk by chunk (0).
 *   22 = cova_sgemm with the operand tiles brought in by global -> LDS copies (1, default: taken when both operands allow 16-byte
 *        pieces) or staged through registers (0: the kernel every other shape takes).
 * 14, 16 and 22 select between two kernels that both run by default (by alignment / by shape): the tests use them to compare the forms.
 * The option state is a PER-PROCESS CONSTANT: it may be set until the library's first query or launch and is fixed from then on
 * (launches captured into a hipGraph, workspace sizes already queried and a trainer's buffers all depend on it) -- a later
 * cova_set_option that would CHANGE a value returns COVA_ERR_BAD_ARG (10001).  A process that sets COVA_ALLOW_OPTION_CHANGES=1
 * in its environment before the library is loaded keeps them mutable (the test suite and bench.py's `ab` legs, which re-query
 * every workspace size per step, do); options 7 and 9 change the result of the matching *_num_partials queries. */
int cova_set_option(int key, int value);

/* weight layout transforms (OIHW -> kernel layouts); run once per optimizer step */
int cova_conv1_prep_weights(const float *w_oihw /*[64,3,7,7]*/, float *w_k /*[154,64]*/, void *stream);

/* nn.Conv2d(3,64,7,stride 2,pad 3,bias=False): img NCHW -> out NHWC [B,H1,W1,64].
 * stat_part (nullable) [cova_conv1_num_partials][2][64]: per-block channel sum / sum of squares of
 * the output (feeds cova_bn_finalize_fwd: train-mode BatchNorm2d statistics).
 * Arithmetic (forward and cova_conv1_wgrad*): f32 in, f32 out, f32 accumulation; the products run on the bf16 matrix
 * pipe with every f32 operand taken as three bf16 pieces (x = x0 + x1 + x2 up to 2^-26 |x|) and the six products of
 * order <= 2 -- the error against a float64 convolution is that of the f32-MFMA kernels (cova_set_option 7 selects
 * those; tests/test_kernels_gpu.py::test_conv1_bf16_split_error_class), gfx950 having no faster f32 matrix path. */
int cova_conv1_num_tiles(int B, int H, int W);
/* rows of the statistics partials written by cova_conv1_fwd (per tile, or per persistent block) */
int cova_conv1_num_partials(int B, int H, int W);
int cova_conv1_fwd(const float *img, const float *w_k, float *out, float *stat_part, int B, int H,
                   int W, void *stream);
/* ... reading the [64,3,7,7] OIHW weight directly (no cova_conv1_prep_weights launch), and with the BatchNorm finalize
 * of its statistics as the launch's tail (tail: host pointer, mode 1; nullable) */
int cova_conv1_fwd_tail(const float *img, const float *w_oihw, float *out, float *stat_part, int B, int H, int W,
                        const cova_bn_tail *tail, void *stream);
/* gradient of conv1's weight (the image needs no gradient): dw OIHW [64,3,7,7] */
int cova_conv1_wgrad_workspace_floats(int B, int H, int W);
int cova_conv1_wgrad(const float *img, const float *dy /*NHWC*/, float *dw, float *ws, int B, int H,
                     int W, void *stream);
/* gradient w.r.t. the image -- what autograd gives the reference when `images.requires_grad` (models.py:94-122 through
 * nn.Conv2d(3,64,7,2,3)); not on the training hot path (train.py never asks for it): plain kernels.
 * cova_pool_bwd_dy1 writes out dy1 = abc[0]*route(dp, idx) + abc[1]*y1 + abc[2] (NHWC [B,H1,W1,64]), the operand
 * cova_conv1_wgrad_poolbwd forms on load; cova_conv1_dgrad: dimg NCHW [B,3,H,W] from dy1 and the OIHW weight */
int cova_pool_bwd_dy1(const float *dp, const uint8_t *idx, const float *y1, const float *abc, float *dy1, int B, int H1,
                      int W1, void *stream);
int cova_conv1_dgrad(const float *dy1, const float *w_oihw, float *dimg, int B, int H, int W, void *stream);
/* same with the BatchNorm+ReLU+MaxPool backward apply folded into the gradient operand:
 * dy1 = abc[0]*route(dp, idx) + abc[1]*y1 + abc[2]; dp [B,H2,W2,64] already ReLU-masked */
int cova_conv1_wgrad_poolbwd(const float *img, const float *y1, const float *dp, const uint8_t *idx,
                             const float *abc, float *dw, float *ws, int B, int H, int W,
                             void *stream);

/* nn.Conv2d(64,64,3,1,1,bias=False) on NHWC [B,H,W,64] (torchvision BasicBlock / Bottleneck conv2; models.py:49-51) runs as
 * Winograd F(4x4,3x3) convolutions in f32-class arithmetic (below).  The F(2x2,3x3) kernels of rounds 1-3 (forward, data
 * gradient, weight gradient) left the product library in round 6: tools/csrc/conv_wino_f2x2.hip + tools/include/cova_wino_f2x2.h,
 * built as a test-support library (the F(4x4) kernel tests cross-check against them); the direct implicit-GEMM kernels of
 * round 1 live in tools/csrc as well. */
/* Weight gradient in Winograd F(4x4,3x3) form (csrc/conv_wgrad4.hip; 1.78x fewer MFMAs than the F(2x2,3x3) form,
 * fp32 error 3.7e-6 of the gradient's scale): replaces autograd's conv2d weight gradient of the four layer1 3x3
 * convolutions (torchvision BasicBlock conv1 / conv2, models.py:49-51; loss.backward() at train.py:59).  Same two-step
 * contract as cova_conv3x3_wgrad_wino_partial / _finish: activation = relu?(A*act + C) on load (act_abc nullable),
 * gradient = A*dz + B*dz2 + C on load (dz_abc, dz2 nullable), per-block partials into ws
 * (cova_conv3x3_wgrad4_workspace_floats; each block applies G^T . G to its own sums in fp64 and writes [9][64][64]),
 * then the fp64 fold of up to four convolutions in one launch. */
int cova_conv3x3_wgrad4_num_partials(int B, int H, int W);
int cova_conv3x3_wgrad4_workspace_floats(int B, int H, int W);
int cova_conv3x3_wgrad4_partial(const float *act, const float *act_abc /*nullable*/, int act_relu, const float *dz,
                                const float *dz2 /*nullable*/, const float *dz_abc /*nullable*/,
                                float *dz_out /*nullable: also writes A*dz + B*dz2 + C, NHWC [B,H,W,64]*/, float *ws, int B,
                                int H, int W, void *stream);
int cova_conv3x3_wgrad4_finish(const float *ws0, float *dw0, const float *ws1, float *dw1, const float *ws2,
                               float *dw2, const float *ws3, float *dw3, int B, int H, int W, void *stream);
int cova_conv3x3_wgrad4(const float *act, const float *dz, float *dw /*OIHW*/, float *ws, int B, int H, int W,
                        void *stream);

/* F(4x4,3x3) form of the same convolution (csrc/conv_wino4.hip; 1.78x fewer MFMAs than F(2x2,3x3), fp32 error 2.9e-6 of
 * the output scale): u_fwd / u_dgrad cova_conv3x3_wino4_u_floats() floats each per convolution (the per-wave register
 * images written by the prep kernel: the f32 image, its three-bf16-piece image, and the piece image of -U -- tiles with odd
 * tx + ty are multiplied with the negated weights and un-negated in the epilogue, so that the bf16 MFMA's sign-asymmetric
 * accumulation (7e-8 of the mean magnitude toward -inf on every output, tools/w4s_bias.py) cancels in sums over a map);
 * stat_part (nullable) [cova_conv3x3_wino4_num_partials][2][64] = (sum y, sum y^2).
 * Arithmetic: f32 in, f32 out, f32 transforms and accumulation; the transform-domain products run on the bf16 matrix pipe
 * with both f32 operands taken as three round-to-nearest bf16 pieces and the six products of order <= 2 (as conv1, see
 * above): f32-class error (tests/test_kernels_gpu.py::test_conv3x3_winograd_f4x4_split_error_class).  The launches with a
 * SECOND input tensor (in2) and cova_set_option(9, 1) run the products on v_mfma_f32_16x16x4_f32. */
int cova_conv3x3_wino4_u_floats(void);
int cova_conv3x3_wino4_num_tiles(int B, int H, int W);
int cova_conv3x3_wino4_num_partials(int B, int H, int W);
int cova_conv3x3_wino4_prep(const float *w_oihw, float *u_fwd, float *u_dgrad, void *stream);
/* ... of up to four convolutions in ONE call (w1..w3 nullable): u_fwd / u_dgrad hold n x cova_conv3x3_wino4_u_floats() floats */
int cova_conv3x3_wino4_prep_multi(const float *w0, const float *w1, const float *w2, const float *w3, float *u_fwd,
                                  float *u_dgrad, void *stream);
int cova_conv3x3_wino4(const float *in, const float *u, float *out, float *stat_part /*nullable*/, int B, int H,
                       int W, void *stream);
/* ... on relu?(A[c]*in + C[c]) formed on load (pro_abc [3][64] = A | unused | C), zero padding stays zero */
int cova_conv3x3_wino4_pro(const float *in, const float *pro_abc, int pro_relu, const float *u, float *out,
                           float *stat_part /*nullable*/, int B, int H, int W, void *stream);
/* ... full form, the contract of cova_conv3x3_wino_pro: input f(A*in + B*in2 + C) on load (in2, pro_abc nullable);
 * epilogue (+ addend) x ReLU mask (act > 0, or fma(mask_scale, z, mask_shift) > 0 when act is NULL) with the
 * BatchNorm-backward sums (sum g, sum g*xhat(z)) in stat_part when z is given, plain statistics otherwise */
int cova_conv3x3_wino4_full(const float *in, const float *in2 /*nullable*/, const float *pro_abc /*nullable*/,
                            int pro_relu, const float *u, const float *addend /*nullable*/,
                            const float *act /*nullable*/, const float *mask_scale /*nullable*/,
                            const float *mask_shift /*nullable*/, const float *z /*nullable*/,
                            const float *mean /*nullable*/, const float *invstd /*nullable*/, float *out,
                            float *stat_part /*nullable*/, int B, int H, int W, void *stream);
/* ... with the BatchNorm finalize of stat_part as the launch's tail (tail: host pointer; mode 1 for plain statistics,
 * mode 2 for the BatchNorm-backward sums of the z epilogue).  act_bits (nullable, instead of act): the mask source as
 * one bit per element, [B*H*W][2] words as cova_bn_act_fwd_bits writes them */
int cova_conv3x3_wino4_full_tail(const float *in, const float *in2 /*nullable*/, const float *pro_abc /*nullable*/,
                                 int pro_relu, const float *u, const float *addend /*nullable*/,
                                 const float *act /*nullable*/, const uint32_t *act_bits /*nullable*/,
                                 const float *mask_scale /*nullable*/,
                                 const float *mask_shift /*nullable*/, const float *z /*nullable*/,
                                 const float *mean /*nullable*/, const float *invstd /*nullable*/, float *out,
                                 float *stat_part, int B, int H, int W, const cova_bn_tail *tail, void *stream);
/* ... inference form (replaces conv -> eval-mode BatchNorm -> (+ identity) -> ReLU of torchvision's BasicBlock.forward in
 * train.evaluate_model's no-grad forward, train.py:99-129; models.py:49-51):
 * out = f(scale[c]*conv(g(in)) + shift[c] + addend), f = ReLU if relu; g = relu?(A[c]*in + C[c]) on load when pro_abc
 * ([3][64] = A | unused | C) is given.  Same expression and operation order as cova_bn_act_fwd; no statistics */
int cova_conv3x3_wino4_bnact(const float *in, const float *pro_abc /*nullable*/, int pro_relu, const float *u,
                             const float *addend /*nullable*/, const float *scale, const float *shift, int relu,
                             float *out, int B, int H, int W, void *stream);

/* ---- ResNet-50-stem extension (BASELINE.json configs[2], [4]; the reference wires resnet18 only,
 * models.py:49): 1x1 convolutions of torchvision's Bottleneck (conv1, conv3, downsample[0]) on NHWC rows.
 * out[r,co] = sum_ci f(A[ci]*in[r,ci] + B[ci]*in2[r,ci] + C[ci]) * w[co,ci], (Cin,Cout) in {(64,64),(64,256),
 * (256,64)}; w [Cout,Cin] row-major (= OIHW), or with w_trans [Cin,Cout] (data gradient of the conv whose
 * weight it is).  Prologue / epilogue arguments as cova_conv3x3_wino_pro; stat_part
 * [cova_conv1x1_num_partials][2][Cout] = (sum y, sum y^2) when z == NULL, else (sum dy, sum dy*xhat);
 * z2/mean2/invstd2 + stat_part2: a second BatchNorm (the downsample branch) fed by the same dy.
 * z == NULL with act != NULL (Cout = 256): (acc + addend) * [act > 0] without sums (taken by cova_conv1x1_vprod);
 * the same with act_bits [R][8] words instead of act: bit (c & 31) of word c >> 5 = the decision for channel c, as
 * written by cova_conv1x1_materialize (1/32 of the mask source's bytes). */
int cova_conv1x1_num_partials(long long R, int Cin, int Cout);
int cova_conv1x1(const float *in, const float *in2 /*nullable*/, const float *pro_abc /*nullable [3,Cin]*/,
                 int pro_relu, const float *w, int w_trans, const float *addend /*nullable*/,
                 const float *act /*nullable*/, const uint32_t *act_bits /*nullable*/,
                 const float *mask_scale /*nullable*/,
                 const float *mask_shift /*nullable*/, const float *z /*nullable*/,
                 const float *mean /*nullable*/, const float *invstd /*nullable*/,
                 const float *z2 /*nullable*/, const float *mean2 /*nullable*/,
                 const float *invstd2 /*nullable*/, float *out, float *stat_part /*nullable*/,
                 float *stat_part2 /*nullable*/, long long R, int Cin, int Cout, void *stream);
/* cova_conv1x1 (256 -> 64, forward; stat_part as there, nullable) on relu(A*in + B*in2 + C), which is also
 * written to side [R,256]: the first consumer of a Bottleneck output materialises it -- and (side_bits nullable,
 * [R][8] words) its ReLU decisions as one bit per element */
int cova_conv1x1_materialize(const float *in, const float *in2, const float *pro_abc /*[3,256]*/, const float *w,
                             float *side, uint32_t *side_bits, float *out, float *stat_part /*nullable*/, long long R,
                             void *stream);
/* dw [Co,Ci] = sum_r (dz_abc[0]*dz + dz_abc[1]*dz2 + dz_abc[2])[r,co] * relu?(act_abc[0]*act + act_abc[2])[r,ci];
 * (Co,Ci) in {(64,64),(256,64),(64,256)}; ws >= cova_conv1x1_wgrad_workspace_floats */
int cova_conv1x1_wgrad_workspace_floats(long long R, int Co, int Ci);
int cova_conv1x1_wgrad(const float *dz, const float *dz2 /*nullable*/, const float *dz_abc /*nullable [3,Co]*/,
                       const float *act, const float *act_abc /*nullable [3,Ci]*/, int act_relu, float *dw,
                       float *ws, long long R, int Co, int Ci, void *stream);

/* Linear form of the backward of (1x1 conv 64->256, train-mode BatchNorm) -- Bottleneck conv3+bn3 and
 * downsample[0]+[1]: with z = a W^T, the BatchNorm sums, dW and dz*W are functions of P = v^T a, G = a^T a,
 * S = sum a, SU = sum v, so the 256-channel z is never read in the backward (csrc/conv1x1_lin.hip).
 *   cova_conv1x1_vprod      v [R,256], act [R,64] (a = relu?(act_abc[0]*act + act_abc[2]), act_abc nullable)
 *                           -> lin [cova_conv1x1_lin_floats] = P | G | S | SU
 *   cova_conv1x1_lin_bnsums lin, w [256,64], mean, invstd -> part [2][256] = (sum v, sum v*xhat(z)): one row of
 *                           partials for cova_bn_finalize_bwd_abc
 *   cova_conv1x1_lin_finish lin, abc [3][256] (dz = A*v + B*z + C), w -> dw [256,64], m [64,64] = W^T diag(B) W,
 *                           cvec [64] = C^T W, avec [3][256] = A | 0 | 0
 *   cova_conv1x1_lin_dgrad  out [R,64] = ((avec.v) W + a m + cvec (+ addend)) * [fma(mask_scale, z, mask_shift) > 0]
 *                           and (sum, sum*xhat(z)) partials [cova_conv1x1_lin_dgrad_num_partials][2][64] of the
 *                           64-channel BatchNorm in front (z, mean, invstd: that layer's) */
int cova_conv1x1_lin_floats(void);
int cova_conv1x1_vprod_workspace_floats(long long R);
int cova_conv1x1_vprod(const float *v, const float *act, const float *act_abc /*nullable [3,64]*/, int act_relu,
                       float *lin, float *ws, long long R, void *stream);
int cova_conv1x1_lin_bnsums(const float *lin, const float *w, const float *mean, const float *invstd,
                            float *part, void *stream);
int cova_conv1x1_lin_finish(const float *lin, const float *abc, const float *w, float *dw, float *m,
                            float *cvec, float *avec, void *stream);
int cova_conv1x1_lin_dgrad_num_partials(long long R);
int cova_conv1x1_lin_dgrad(const float *v, const float *avec, const float *w, const float *act,
                           const float *act_abc, int act_relu, const float *m, const float *cvec,
                           const float *addend /*nullable [R,64]*/, const float *mask_scale,
                           const float *mask_shift, const float *z, const float *mean, const float *invstd,
                           float *out, float *stat_part, long long R, void *stream);

/* ------------------------------------------------------------------ BatchNorm / ReLU / MaxPool
 * replaces: nn.BatchNorm2d / nn.BatchNorm1d (train: batch statistics + running-stat update with
 * momentum, unbiased running_var; eval: running statistics), nn.ReLU, the BasicBlock residual
 * add and nn.MaxPool2d(3,2,1) -- models.py:49-51, :68, :73, :86 and their autograd. */
/* deterministic pre-reduction of per-tile / per-chunk partial rows (fp64 inside a group) */
int cova_partials_fold(const float *partial, int nparts, int width, int group, float *out,
                       void *stream);
int cova_colreduce_rows_per_chunk(long long R, int C);
int cova_colreduce_num_chunks(long long R, int C);
int cova_colstats(const float *x, int ldx, long long R, int C, float *partial /*[chunks,2,C]*/,
                  void *stream);
int cova_bn_finalize_fwd(const float *partial, int nparts, int C, double count, const float *gamma,
                         const float *beta, float *running_mean /*nullable*/, float *running_var,
                         long long *num_batches_tracked /*nullable: += 1, like nn.BatchNorm in train mode*/,
                         float momentum, float eps, float *scale, float *shift, float *mean,
                         float *invstd, void *stream);
int cova_bn_eval_params(const float *gamma, const float *beta, const float *running_mean,
                        const float *running_var, float eps, int C, float *scale, float *shift,
                        float *mean /*nullable*/, float *invstd /*nullable*/, void *stream);
int cova_bn_act_fwd(const float *z, int ldz, const float *scale, const float *shift,
                    const float *res /*nullable*/, int ldres, float *out, int ldo, long long R, int C,
                    int relu, void *stream);
/* cova_bn_act_fwd with ReLU for a contiguous [R,64] map (res nullable), also writing the ReLU decisions as one bit per
 * element: bits [R][2] words, bit k of word w = (out[r][32 w + k] > 0) */
int cova_bn_act_fwd_bits(const float *z, const float *scale, const float *shift, const float *res, float *out,
                         uint32_t *bits, long long R, void *stream);
/* out = act(bn(z) + bn2(z2)): join of a Bottleneck whose identity branch is conv + BatchNorm */
int cova_bn_act2_fwd(const float *z, const float *scale, const float *shift, const float *z2,
                     const float *scale2, const float *shift2, float *out, long long R, int C, int relu,
                     void *stream);
int cova_bn_bwd_reduce(const float *dout, int ldd, const float *act /*nullable: relu mask source*/,
                       int lda, const float *z, int ldz, const float *mean, const float *invstd,
                       long long R, int C, float *partial /*[chunks,2,C]*/, void *stream);
int cova_bn_finalize_bwd(const float *partial, int nparts, int C, double count,
                         float *dgamma /*nullable*/, float *dbeta /*nullable*/, float *coef /*[2,C]*/,
                         void *stream);
/* same + the apply step in affine form: dz = abc[0]*dy + abc[1]*z + abc[2]; abc [3,C] */
int cova_bn_finalize_bwd_abc(const float *partial, int nparts, int C, double count,
                             float *dgamma /*nullable*/, float *dbeta /*nullable*/, const float *mean,
                             const float *invstd, const float *scale, float *abc, void *stream);
int cova_bn_bwd_apply(const float *dout, int ldd, const float *act, int lda, const float *z, int ldz,
                      const float *mean, const float *invstd, const float *scale, const float *coef,
                      float *dz, int lddz, float *dres /*nullable*/, int lddres, long long R, int C,
                      void *stream);
/* nn.BatchNorm1d in train mode over box rows x [R,C] (models.py:68 bbox_feat_encoder.1, :73 bn_additional_feat, :86
 * decoder.2) as ONE launch: column statistics, running-statistics update, scale/shift/mean/invstd, out = bn(x) (ReLU if
 * `relu`), and -- dropped != NULL -- nn.Dropout(p) of the result (models.py:88; mask [R,C] generated from `seed` and
 * stored, or taken as given), i.e. cova_colstats + cova_bn_finalize_fwd + cova_bn_act_fwd (+ cova_dropout_fwd). */
int cova_bn1d_fwd(const float *x, int ldx, int R, int C, const float *gamma, const float *beta,
                  float *running_mean /*nullable*/, float *running_var, long long *num_batches_tracked /*nullable*/,
                  float momentum, float eps, int relu, float *out, int ldo, float *dropped /*nullable*/, int ld_dropped,
                  uint8_t *mask, float p, unsigned long long seed, int mask_given, float *scale, float *shift,
                  float *mean, float *invstd, void *stream);
/* its backward in ONE launch: dy = dout (* drop_mask/(1-p) if drop_mask: the Dropout behind the layer) (* (act > 0) if
 * act: the ReLU behind it); dgamma, dbeta (nullable); dz [R,C]; dz_colsum (nullable) [C] = column sums of dz (the bias
 * gradient of the nn.Linear in front, models.py:85) -- i.e. (cova_dropout_bwd +) cova_bn_bwd_reduce +
 * cova_bn_finalize_bwd + cova_bn_bwd_apply (+ cova_colsum).  dz must not alias dout. */
int cova_bn1d_bwd(const float *dout, int ldg, const uint8_t *drop_mask /*nullable*/, float p,
                  const float *act /*nullable*/, int lda, const float *z, int ldz, const float *mean,
                  const float *invstd, const float *scale, int R, int C, float *dgamma, float *dbeta, float *dz,
                  int lddz, float *dz_colsum /*nullable*/, void *stream);
int cova_bn_relu_maxpool_fwd(const float *y /*[B,H1,W1,64]*/, const float *scale, const float *shift,
                             float *out /*[B,H2,W2,64]*/, uint8_t *idx,
                             float *ymax /*nullable [B,H2,W2,64]: raw y at the arg-max*/, int B, int H1,
                             int W1, void *stream);
int cova_bn_relu_maxpool_bwd_num_partials(int B, int H1, int W1);
int cova_bn_relu_maxpool_bwd_reduce(const float *dp, const uint8_t *idx, const float *y,
                                    const float *scale, const float *shift, const float *mean,
                                    const float *invstd, float *partial, int B, int H1, int W1,
                                    void *stream);
int cova_bn_relu_maxpool_bwd_apply(const float *dp, const uint8_t *idx, const float *y,
                                   const float *scale, const float *shift, const float *mean,
                                   const float *invstd, const float *coef, float *dz, int B, int H1,
                                   int W1, void *stream);

/* ------------------------------------------------------------------ RoIPool (models.py:58,125)
 * replaces: torchvision.ops.RoIPool(output_size, spatial_scale)(feat, rois) and its backward.
 * feat NHWC [B,H,W,C]; rois [N,5] = [batch_idx,x1,y1,x2,y2]; row n of the output (C*PH*PW
 * values, index c*PH*PW + ph*PW + pw exactly like `.view(N, n_visual_feat)` at models.py:125-127)
 * is written at out + n*ld_out so it can land directly in the concatenated feature matrix.
 * Memory safety: a box whose page index is outside [0,B) pools nothing (all bins 0, argmax -1). */
int cova_roipool_fwd(const float *feat, const float *rois, int n_rois, int B, int C, int H, int W,
                     int PH, int PW, float spatial_scale, float *out, int ld_out, int32_t *argmax,
                     void *stream);
/* backward: gfeat NHWC [B,H,W,C] (fully written: no zero-fill needed) = gout routed to the arg-max positions.
 * Deterministic -- no float atomics: every feature row has one owner wave that adds the boxes touching it in
 * ascending box order (the reference's scatter collides constantly: DOM parents contain their children). */
int cova_roipool_bwd(const float *gout, int ld_g, const float *rois, const int32_t *argmax,
                     int n_rois, int B, int C, int H, int W, int PH, int PW, float spatial_scale,
                     float *gfeat, void *ws /*cova_roipool_bwd_workspace_words x 4 bytes*/, void *stream);
int cova_roipool_bwd_workspace_words(int n_rois, int B, int C, int PH, int PW);
/* same for a map produced as relu(bn(z) + residual) and pooled by cova_roipool_fwd_bn: the routed gradient
 * is masked by that ReLU (pooled > 0: the pooled value is the map's value at the arg-max) and the producer's
 * BatchNorm-backward partial sums [cova_roipool_bwd_bn_num_partials][2][C] = (sum g', sum g' * xhat(zmax)) are
 * taken per pooled entry -- no map is read.  gfeat = the ReLU-masked gradient map. */
int cova_roipool_bwd_bn_num_partials(int n_rois);
int cova_roipool_bwd_bn(const float *gout, int ld_g, const float *pooled, int ld_p, const float *zmax,
                        const float *rois, const int32_t *argmax, int n_rois, int B, int C, int H, int W,
                        int PH, int PW, float spatial_scale, const float *mean, const float *invstd,
                        float *gfeat, float *partial, void *ws /*cova_roipool_bwd_workspace_words x 4 bytes*/,
                        void *stream);
/* ... with cova_bn_finalize_bwd_abc of those sums as the tail of the entry pass (tail: host pointer, mode 2; C = 64) */
int cova_roipool_bwd_bn_tail(const float *gout, int ld_g, const float *pooled, int ld_p, const float *zmax,
                             const float *rois, const int32_t *argmax, int n_rois, int B, int C, int H, int W,
                             int PH, int PW, float spatial_scale, const float *mean, const float *invstd,
                             float *gfeat, float *partial, void *ws, const cova_bn_tail *tail, void *stream);
/* RoIPool over relu(scale*z + shift + x) formed on the fly (last BasicBlock's bn2+residual+ReLU) */
int cova_roipool_fwd_bn(const float *z, const float *x, const float *scale, const float *shift,
                        const float *rois, int n_rois, int B, int C, int H, int W, int PH, int PW,
                        float spatial_scale, float *out, int ld_out, int32_t *argmax,
                        float *zmax /*nullable [N, C*PH*PW]: z at each arg-max, for cova_roipool_bwd_bn*/,
                        void *stream);

/* RoIAlign -- EXTENSION: north_star names it, the reference calls RoIPool (models.py:58), which stays the parity
 * operator.  torchvision.ops.RoIAlign semantics (bilinear samples, sampling_ratio^2 per bin, or ceil(roi/bin)^2
 * when sampling_ratio <= 0; `aligned` half-pixel shift); same tensor conventions as cova_roipool_fwd / _bwd;
 * backward deterministic (one owner wave per feature row), ws >= (2*B + 16) ints. */
int cova_roialign_fwd(const float *feat, const float *rois, int n_rois, int B, int C, int H, int W, int PH,
                      int PW, float spatial_scale, int sampling_ratio, int aligned, float *out, int ld_out,
                      void *stream);
int cova_roialign_bwd(const float *gout, int ld_g, const float *rois, int n_rois, int B, int C, int H, int W,
                      int PH, int PW, float spatial_scale, int sampling_ratio, int aligned, float *gfeat,
                      void *ws, void *stream);

/* ------------------------------------------------------------------ positional encoder
 * replaces: CoVA._get_bbox_features up to nn.Linear(5, Hd) (models.py:134-144):
 * raw = [x1, y1, x2-x1, y2-y1, (x2-x1)/(y2-y1)], z = raw W^T + b */
int cova_bbox_linear_fwd(const float *bboxes, const float *W /*[Hd,5]*/, const float *bias,
                         float *raw /*[N,5]*/, float *z /*[N,Hd]*/, int N, int Hd, void *stream);
int cova_bbox_linear_bwd(const float *dz, const float *raw, float *dW, float *db, int N, int Hd,
                         void *stream);

/* ------------------------------------------------------------------ dense layers
 * replaces: nn.Linear forward/backward GEMMs (models.py:85,161-162): C (+)= op(A) op(B) (+ bias) */
int cova_sgemm(int transA, int transB, int M, int N, int K, const float *A, int lda, const float *B,
               int ldb, float *C, int ldc, const float *bias /*nullable [N]*/, int accumulate,
               void *stream);
/* cova_sgemm (no bias, no accumulate) with cova_dropout_bwd applied to its result in the epilogue:
 * C = keep ? op(A) op(B) / (1 - p) : 0, keep [M,N] uint8 contiguous -- the decoder's first Dropout backward
 * (models.py:84) behind the data gradient of decoder.1 (models.py:85); bit-identical to the two launches */
int cova_sgemm_dropout_bwd(int transA, int transB, int M, int N, int K, const float *A, int lda, const float *B,
                           int ldb, float *C, int ldc, const uint8_t *keep /*[M,N]*/, float p, void *stream);

/* ------------------------------------------------------------------ graph attention (models.py:171-212)
 * replaces: GraphAttentionLayer.forward after the projections: gather, score, LeakyReLU, mask,
 * softmax, weighted sum.  Wh [N,2D] = h [W_i;W_j]^T.  K <= 256 (the K slots of a node are held by one wavefront in
 * ceil(K/64) passes over its lanes; models.py:171-177 takes any n_context).  Neighbour ids >= N are treated like the
 * -1 pad (the reference's h_i_padded[context_indices] raises an index error for them). */
int cova_gat_fwd(const float *Wh, int ldw, const float *att_w /*[2D]*/, const float *att_b /*[1]*/,
                 const int64_t *ctx /*[N,K]*/, int N, int K, int D, float slope, float *s /*[N]*/,
                 float *t /*[N]*/, float *attn /*[N,K]*/, float *hprime, int ldh, void *stream);
/* transposed neighbour index (CSR over destination nodes) of ctx: csr[0..N] row offsets, then the flat slots
 * i*K+k naming each node, ascending.  Lets the backward gather with a fixed summation order instead of
 * scattering with float atomics (torch's index_select backward): bit-identical reruns for ANY index table. */
int cova_gat_transpose_ints(int N, int K);
int cova_gat_transpose(const int64_t *ctx, int N, int K, int *csr /*[cova_gat_transpose_ints]*/, void *stream);
/* ... into a workspace kept from call to call (one per stream): the last 2N + 16 ints of csr must be zero on entry --
 * zero the buffer once when allocating it -- and are left zero: three launches, no memsets */
int cova_gat_transpose_reuse(const int64_t *ctx, int N, int K, int *csr, void *stream);
/* csr + du [N,K] scratch: deterministic gather form; csr == NULL: scatter form with float atomics */
int cova_gat_bwd(const float *g, int ldg, const float *Wh, int ldw, const float *s, const float *t,
                 const float *attn, const int64_t *ctx, const float *att_w, int N, int K, int D,
                 float slope, float *dWh /*[N,2D]*/, int lddw, float *ds /*[N]*/, float *dt /*[N]*/,
                 float *d_att_w /*[2D]*/, float *d_att_b /*[1]*/, const int *csr /*nullable*/,
                 float *du /*nullable [N,K]*/, void *stream);

/* ------------------------------------------------------------------ decoder tail, loss, optimizer
 * replaces: nn.Dropout (models.py:84,88), nn.Linear(T, n_classes) (models.py:89),
 * nn.CrossEntropyLoss(reduction="sum") + output.argmax(dim=1) (main.py:139, train.py:53,56),
 * torch.optim.Adam(lr, weight_decay).step() (main.py:133-135, train.py:60). */
int cova_dropout_fwd(const float *x, int ldx, float *out, int ldo, uint8_t *mask /*[R,C]*/,
                     long long R, int C, float p, unsigned long long seed, int mask_given,
                     void *stream);
int cova_dropout_bwd(const float *g, int ldg, const uint8_t *mask, float *dx, int ldx, long long R,
                     int C, float p, void *stream);
int cova_linear_small_fwd(const float *x, int ldx, const float *W /*[NC,Cin]*/, const float *b,
                          float *y /*[N,NC]*/, int N, int Cin, int NC, void *stream);
int cova_linear_small_bwd(const float *dy, const float *x, int ldx, const float *W, float *dx,
                          int lddx, float *dW, float *db, int N, int Cin, int NC, void *stream);
int cova_ce_sum(const float *logits, const int64_t *labels /*nullable*/, int N, int NC, float gscale,
                float *loss /*nullable [1]*/, float *dlogits /*nullable*/, int64_t *pred /*nullable*/,
                void *stream);
int cova_adam_step(float *p, const float *g, float *m, float *v, long long n, int step, double lr,
                   double beta1, double beta2, double eps, double weight_decay, void *stream);
int cova_colsum(const float *x, int ldx, int R, int C, float *out, void *stream);
/* evaluation decision (train.py:131-153): per page and class column, page-local indices of the k
 * highest-scoring boxes, best first; page_start [n_pages+1] are box offsets; out [n_pages,NC,k] */
int cova_page_class_topk(const float *logits, const int64_t *page_start, int n_pages, int NC, int k,
                         int64_t *out, void *stream);

/* ---- device-side input pipeline --------------------------------------
 * ToTensor of datasets.py:41-45,96-97: u8 [B,H,W,3] -> f32 [B,3,H,W], value/255 (bit-exact) */
int cova_images_u8_to_f32(const uint8_t *u8_nhwc, float *f32_nchw, int B, int H, int W, void *stream);
/* WebDataset.__getitem__ box part + custom_collate_fn (datasets.py:110-128,159-178):
 * rows [N,5] = x,y,w,h,label of B pages back to back, page_offsets int32 [B+1] (device) ->
 * bboxes [N,5] = page,x1,y1,x2,y2; labels [N] i64; ctx [N,2*context_size] i64 batch-global ids, -1 pads */
int cova_collate_boxes(const float *rows, const int *page_offsets, int B, int N, int context_size,
                       float *bboxes, long long *labels, long long *ctx /*nullable if context_size==0*/,
                       void *stream);
/* attention export rows (extract_attn_wts_and_visualize.py:104-135): out [N, 5+5K] =
 * x,y,w,h,label, K x (x,y,w,h) of the context boxes (0 for pads), K attention weights */
int cova_attn_export_rows(const float *bboxes, const long long *ctx, const float *attn,
                          const long long *labels, int N, int K, float *out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* COVA_HIP_H */
