// BatchNorm1d over the box rows ([R, C] matrices: the positional encoder's `bbox_feat_encoder.1`, models.py:68,
// CoVA++'s `bn_additional_feat`, :73, and the decoder's `decoder.2`, :86) -- train mode, ONE launch per application.
// The columns of a BatchNorm1d are independent, and the matrices are small (1 440 x 976 at configs[1]): a block owns
// 8 columns for ALL rows, so statistics, finalize and apply need no grid-wide step -- reduce (pass 1), per-column
// parameters (one thread per column, fp64, bn_tail.h's channel functions = what cova_bn_finalize_* compute), apply
// (pass 2; the block's columns x R rows come back from L2).  The neighbouring element-wise ops ride along:
//   forward : + ReLU (models.py:69,87) + the decoder's second Dropout (models.py:88) on the result;
//   backward: the Dropout backward of the incoming gradient, the ReLU mask, (sum dy, sum dy*xhat), dgamma / dbeta, dz,
//             and the column sums of dz (= the bias gradient of the Linear in front, models.py:85).
// Replaces, per application, colreduce + bn_finalize + bn_act (+ dropout) launches forward and (dropout_bwd +) colreduce
// + bn_finalize_bwd + bn_bwd_apply (+ colsum) backward.  SyncBN and eval-mode BatchNorm keep the separate kernels
// (bn.hip): a collective / a host decision sits between the sums and the parameters there.
#include "bn_tail.h"

namespace {

constexpr int COLS = 8, SLICES = 128, RUN = 32;      // block = 8 columns x 128 row slices (the launch is latency bound:
                                                     // rows in flight are what counts); RUN: fp32 runs of the dz column sums
constexpr int THREADS = COLS * SLICES;

// fixed-order fp64 total of the 16 slices' sums (deterministic: no atomics)
__device__ __forceinline__ void slices_total(double (*s_a)[COLS], double (*s_b)[COLS], int tx, int ty, double a, double b,
                                             double &ta, double &tb)
{
    s_a[ty][tx] = a;
    s_b[ty][tx] = b;
    __syncthreads();
    ta = 0.0;
    tb = 0.0;
    if (ty == 0)
        for (int j = 0; j < SLICES; ++j) {
            ta += s_a[j][tx];
            tb += s_b[j][tx];
        }
}

template <bool DROP>
__global__ __launch_bounds__(THREADS) void bn1d_fwd_kernel(
    const float *__restrict__ x, int ldx, int R, int C, const float *__restrict__ gamma, const float *__restrict__ beta,
    float *__restrict__ running_mean, float *__restrict__ running_var, long long *__restrict__ nbt, float momentum,
    float eps, int relu, float *__restrict__ out, int ldo, float *__restrict__ dropped, int ldd,
    uint8_t *__restrict__ mask, float p, unsigned long long seed, int mask_given, float *__restrict__ scale,
    float *__restrict__ shift, float *__restrict__ mean, float *__restrict__ invstd)
{
    __shared__ double s_a[SLICES][COLS], s_b[SLICES][COLS];
    __shared__ float s_sc[COLS], s_sh[COLS];
    const int tx = threadIdx.x & (COLS - 1), ty = threadIdx.x / COLS;
    const int c = blockIdx.x * COLS + tx;
    // sums and squares in fp64 from the first add on (the matrices are small and the fp64 rate is not the limit here):
    // the variance E[x^2] - mean^2 then carries no fp32 summation noise
    double a = 0.0, b = 0.0;
    if (c < C) {
#pragma unroll 4
        for (int r = ty; r < R; r += SLICES) {
            const double v = (double)x[(size_t)r * ldx + c];
            a += v;
            b = fma(v, v, b);
        }
    }
    double ta, tb;
    slices_total(s_a, s_b, tx, ty, a, b, ta, tb);
    if (ty == 0 && c < C) {
        bn_fwd_channel(ta, tb, (double)R, c, gamma, beta, running_mean, running_var, momentum, eps, scale, shift, mean,
                       invstd);
        s_sc[tx] = scale[c];
        s_sh[tx] = shift[c];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && nbt != nullptr) *nbt += 1;
    __syncthreads();
    if (c >= C) return;
    const float sc = s_sc[tx], sh = s_sh[tx], inv = DROP ? 1.f / (1.f - p) : 1.f;
#pragma unroll 4
    for (int r = ty; r < R; r += SLICES) {
        float y = fmaf(sc, x[(size_t)r * ldx + c], sh);             // same form as cova_bn_act_fwd
        if (relu) y = y > 0.f ? y : 0.f;
        out[(size_t)r * ldo + c] = y;
        if (DROP) {                                                 // cova_dropout_fwd on the result
            const size_t i = (size_t)r * C + c;
            uint8_t keep;
            if (mask_given) keep = mask[i];
            else { keep = hash_uniform(seed, (unsigned long long)i) >= p ? 1 : 0; mask[i] = keep; }
            dropped[(size_t)r * ldd + c] = keep ? y * inv : 0.f;
        }
    }
}

// dy = dout (* drop_mask / (1-p)) (* (act > 0)); dz = scale * (dy - mean(dy) - xhat * mean(dy * xhat))
template <bool DROP>
__global__ __launch_bounds__(THREADS) void bn1d_bwd_kernel(
    const float *__restrict__ dout, int ldg, const uint8_t *__restrict__ drop_mask, float p,
    const float *__restrict__ act, int lda, const float *__restrict__ z, int ldz, const float *__restrict__ mean,
    const float *__restrict__ invstd, const float *__restrict__ scale, int R, int C, float *__restrict__ dgamma,
    float *__restrict__ dbeta, float *__restrict__ dz, int lddz, float *__restrict__ dz_colsum)
{
    __shared__ double s_a[SLICES][COLS], s_b[SLICES][COLS];
    __shared__ float s_c1[COLS], s_c2[COLS];
    const int tx = threadIdx.x & (COLS - 1), ty = threadIdx.x / COLS;
    const int c = blockIdx.x * COLS + tx;
    const float inv = DROP ? 1.f / (1.f - p) : 1.f;
    float mu = 0.f, is = 0.f, sc = 0.f;
    if (c < C) { mu = mean[c]; is = invstd[c]; sc = scale[c]; }
    auto dy_at = [&](int r) {
        float g = dout[(size_t)r * ldg + c];
        if (DROP) g = drop_mask[(size_t)r * C + c] ? g * inv : 0.f;               // cova_dropout_bwd
        if (act != nullptr && !(act[(size_t)r * lda + c] > 0.f)) g = 0.f;          // ReLU mask
        return g;
    };
    double a = 0.0, b = 0.0;
    if (c < C) {
#pragma unroll 4
        for (int r = ty; r < R; r += SLICES) {
            const float g = dy_at(r);
            a += (double)g;
            b = fma((double)g, (double)((z[(size_t)r * ldz + c] - mu) * is), b);    // xhat in fp32, as the apply pass forms it
        }
    }
    double ta, tb;
    slices_total(s_a, s_b, tx, ty, a, b, ta, tb);
    if (ty == 0 && c < C) {
        if (dbeta) dbeta[c] = (float)ta;
        if (dgamma) dgamma[c] = (float)tb;
        s_c1[tx] = (float)(ta / (double)R);                          // cova_bn_finalize_bwd's coef rows
        s_c2[tx] = (float)(tb / (double)R);
    }
    __syncthreads();
    double cs = 0.0;
    if (c < C) {
        const float c1 = s_c1[tx], c2 = s_c2[tx];
        for (int r0 = ty; r0 < R; r0 += SLICES * RUN) {
            float s = 0.f;
            const int r1 = min(R, r0 + SLICES * RUN);
#pragma unroll 4
            for (int r = r0; r < r1; r += SLICES) {
                const float xh = (z[(size_t)r * ldz + c] - mu) * is;
                const float o = sc * (dy_at(r) - c1 - xh * c2);      // same form as cova_bn_bwd_apply
                dz[(size_t)r * lddz + c] = o;
                s += o;
            }
            cs += (double)s;
        }
    }
    if (dz_colsum != nullptr) {                                      // (uniform branch: every thread reaches the barrier)
        __syncthreads();                                             // s_a is read by the ty == 0 threads above
        double t1, t2;
        slices_total(s_a, s_b, tx, ty, cs, 0.0, t1, t2);
        if (ty == 0 && c < C) dz_colsum[c] = (float)t1;
    }
}

}  // namespace

COVA_API int cova_bn1d_fwd(const float *x, int ldx, int R, int C, const float *gamma, const float *beta,
                           float *running_mean, float *running_var, long long *num_batches_tracked, float momentum,
                           float eps, int relu, float *out, int ldo, float *dropped, int ld_dropped, uint8_t *mask,
                           float p, unsigned long long seed, int mask_given, float *scale, float *shift, float *mean,
                           float *invstd, void *stream)
{
    COVA_REQUIRE(x && gamma && beta && out && scale && shift && mean && invstd && R > 0 && C > 0);
    COVA_REQUIRE((running_mean == nullptr) == (running_var == nullptr));
    COVA_REQUIRE(dropped == nullptr || (mask != nullptr && p >= 0.f && p < 1.f));
    const dim3 grid(cdiv(C, COLS));
    if (dropped)
        hipLaunchKernelGGL(bn1d_fwd_kernel<true>, grid, dim3(THREADS), 0, (hipStream_t)stream, x, ldx, R, C, gamma, beta,
                           running_mean, running_var, num_batches_tracked, momentum, eps, relu, out, ldo, dropped,
                           ld_dropped, mask, p, seed, mask_given, scale, shift, mean, invstd);
    else
        hipLaunchKernelGGL(bn1d_fwd_kernel<false>, grid, dim3(THREADS), 0, (hipStream_t)stream, x, ldx, R, C, gamma, beta,
                           running_mean, running_var, num_batches_tracked, momentum, eps, relu, out, ldo, dropped,
                           ld_dropped, mask, p, seed, mask_given, scale, shift, mean, invstd);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

COVA_API int cova_bn1d_bwd(const float *dout, int ldg, const uint8_t *drop_mask, float p, const float *act, int lda,
                           const float *z, int ldz, const float *mean, const float *invstd, const float *scale, int R,
                           int C, float *dgamma, float *dbeta, float *dz, int lddz, float *dz_colsum, void *stream)
{
    COVA_REQUIRE(dout && z && mean && invstd && scale && dz && R > 0 && C > 0);
    COVA_REQUIRE(drop_mask == nullptr || (p >= 0.f && p < 1.f));
    const dim3 grid(cdiv(C, COLS));
    if (drop_mask)
        hipLaunchKernelGGL(bn1d_bwd_kernel<true>, grid, dim3(THREADS), 0, (hipStream_t)stream, dout, ldg, drop_mask, p, act,
                           lda, z, ldz, mean, invstd, scale, R, C, dgamma, dbeta, dz, lddz, dz_colsum);
    else
        hipLaunchKernelGGL(bn1d_bwd_kernel<false>, grid, dim3(THREADS), 0, (hipStream_t)stream, dout, ldg, drop_mask, p, act,
                           lda, z, ldz, mean, invstd, scale, R, C, dgamma, dbeta, dz, lddz, dz_colsum);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}
