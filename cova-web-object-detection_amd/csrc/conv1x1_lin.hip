// Backward of (1x1 convolution 64 -> 256, train-mode BatchNorm) in LINEAR form, for the ResNet-50-stem
// extension (torchvision Bottleneck conv3+bn3 and downsample[0]+[1] behind models.py:49-51).
//
// With a [R,64] the conv input, z = a W^T [R,256] its output and v [R,256] the (ReLU-masked) gradient
// w.r.t. the BatchNorm output, everything the backward needs is a function of FOUR small reductions over
// the pixel rows -- none of which reads z:
//     P  = v^T a     [256,64]      G = a^T a   [64,64]      S = sum_r a   [64]      SU = sum_r v   [256]
//   BatchNorm sums:    sum v = SU,   sum v*xhat(z) = invstd * (<W[c,:], P[c,:]> - mean*SU)       (z = a W^T)
//   => dgamma, dbeta and the apply coefficients  dz = A*v + B*z + C  (cova_bn_finalize_bwd_abc)
//   weight gradient:   dW = dz^T a = A.P + B.(W G) + C (x) S
//   data gradient:     dz W = (A.v) W + a (W^T diag(B) W) + C^T W          (cova_conv1x1_lin_dgrad)
// The direct form (cova_conv1x1_wgrad + cova_conv1x1 with the dz prologue) reads v and z twice each
// (4 x 256 channels per pixel) and takes the sums in the epilogue of the PREVIOUS data-gradient kernel
// (one more read of z); this form reads v twice and z never: 2 instead of 5 passes over 256-channel
// maps per Bottleneck, at +25 % MFMA work (the Gram matrix / the 64 extra K channels).  Same exact-f32
// MFMA arithmetic; the result differs from the direct form by summation order only (measured against
// fp64: 4e-7 relative, the direct form 1.4e-6).
#include "common.h"

int cova_internal_persistent_grid2(int ntiles, int blocks_per_cu);

namespace {

constexpr int CO = 256, CI = 64;
constexpr int LIN_P = 0, LIN_G = CO * CI, LIN_S = LIN_G + CI * CI, LIN_SU = LIN_S + CI, LIN_FLOATS = LIN_SU + CO;

struct VPArgs {
    const float *v, *act, *act_abc;      // act_abc [3][64] = A | unused | C, nullable
    float *ws;                           // [grid] x (P 256x64) | [4 grid] x (G 64x64) | [4 grid] x S | [grid] x SU
    long long R;
    int act_relu;
};

// GEMM M = 256 (v channels), N = 64 (a channels), K = pixels on v_mfma_f32_32x32x2_f32, operands straight
// from HBM (conv1x1_wgrad_kernel's layout: a k-step is a pixel pair, half-wave h takes pixel 2s+h, lane p
// supplies channels {2p, 2p+1}).  Wave u owns v channels 64u..64u+63; the Gram matrix of a (the same
// operand on both sides) is spread over the four waves by batch of 8 k-steps (batch & 3 == u), as are the sums of a.
template <bool PRO_ACT>
__global__ __launch_bounds__(256, 2) void conv1x1_vprod_kernel(const VPArgs a)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 31, h = lane >> 5;
    const int cob = wave * 64;
    float aA[2] = {1.f, 1.f}, aC[2] = {0.f, 0.f};
    if (PRO_ACT) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            aA[k] = a.act_abc[2 * p + k];
            aC[k] = a.act_abc[2 * CI + 2 * p + k];
        }
    }
    f32x16 acc[2][2], accg[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = accg[i][j][r] = 0.f;
    float su[2] = {0.f, 0.f}, sa[2] = {0.f, 0.f};

    const long long npairs = (a.R + 1) / 2;
    const long long per = (npairs + gridDim.x - 1) / gridDim.x;
    const long long s_lo = (long long)blockIdx.x * per;
    long long s_hi = s_lo + per;
    if (s_hi > npairs) s_hi = npairs;
    constexpr int U = 8;
    float2 g[2][U], x[2][U];
    const int lz = h * CO + cob + 2 * p, lx = h * CI + 2 * p;
    auto issue = [&](long long s0, float2 (&gg)[U], float2 (&xx)[U]) {
        const float *bz = a.v + (size_t)(2 * s0) * CO, *bx = a.act + (size_t)(2 * s0) * CI;
#pragma unroll
        for (int k = 0; k < U; ++k) {
            gg[k] = *reinterpret_cast<const float2 *>(bz + lz + k * 2 * CO);
            xx[k] = *reinterpret_cast<const float2 *>(bx + lx + k * 2 * CI);
        }
    };
    auto mac = [&](float d0, float d1, float x0, float x1, bool valid, bool gram) {
        if (PRO_ACT) {
            x0 = fmaf(aA[0], x0, aC[0]);
            x1 = fmaf(aA[1], x1, aC[1]);
            if (a.act_relu) { x0 = x0 > 0.f ? x0 : 0.f; x1 = x1 > 0.f ? x1 : 0.f; }
        }
        if (!valid) d0 = d1 = x0 = x1 = 0.f;
        acc[0][0] = mfma32(d0, x0, acc[0][0]);
        acc[0][1] = mfma32(d0, x1, acc[0][1]);
        acc[1][0] = mfma32(d1, x0, acc[1][0]);
        acc[1][1] = mfma32(d1, x1, acc[1][1]);
        su[0] += d0;
        su[1] += d1;
        if (gram) {                        // wave-uniform
            accg[0][0] = mfma32(x0, x0, accg[0][0]);
            accg[0][1] = mfma32(x0, x1, accg[0][1]);
            accg[1][0] = mfma32(x1, x0, accg[1][0]);
            accg[1][1] = mfma32(x1, x1, accg[1][1]);
            sa[0] += x0;
            sa[1] += x1;
        }
    };
    // The Gram matrix of a batch (8 pixel pairs) is taken by wave (batch & 3): one uniform branch per batch.
    auto consume = [&](const float2 (&gg)[U], const float2 (&xx)[U], long long batch) {
        const bool gram = (int)(batch & 3) == wave;          // wave-uniform
#pragma unroll
        for (int k = 0; k < U; ++k) mac(gg[k].x, gg[k].y, xx[k].x, xx[k].y, true, gram);
    };
    // full batches [s_lo + b*U, +U) inside the block's range AND inside the map.  Every load in the loop is
    // unconditional (the last batch re-requests itself): a load under a branch would make the compiler wait
    // for vmcnt(0) at the join, i.e. for the batch that was only just requested.
    long long full_hi = s_hi;
    if (2 * full_hi > a.R) full_hi = a.R / 2;
    const long long nfull = full_hi > s_lo ? (full_hi - s_lo) / U : 0;
    long long s0 = s_lo;
    if (nfull > 0) {
        issue(s_lo, g[0], x[0]);
        for (long long b = 0; b < nfull; b += 2) {
            const long long b1 = b + 1 < nfull ? b + 1 : nfull - 1;
            issue(s_lo + b1 * U, g[1], x[1]);
            __builtin_amdgcn_sched_barrier(0);             // keep the requests ahead of the MFMAs (not sunk to their uses)
            consume(g[0], x[0], b);
            if (b + 1 >= nfull) break;
            const long long b2 = b + 2 < nfull ? b + 2 : nfull - 1;
            issue(s_lo + b2 * U, g[0], x[0]);
            __builtin_amdgcn_sched_barrier(0);
            consume(g[1], x[1], b + 1);
        }
        s0 = s_lo + nfull * U;
    }
    for (; s0 < s_hi; s0 += U) {           // ragged tail: clamped, predicated
        for (int k = 0; k < U; ++k) {
            const long long sp = s0 + k;
            long long row = 2 * sp + h;
            const bool valid = sp < s_hi && row < a.R;
            if (row >= a.R) row = a.R - 1;
            const float2 d = *reinterpret_cast<const float2 *>(a.v + (size_t)row * CO + cob + 2 * p);
            const float2 xv = *reinterpret_cast<const float2 *>(a.act + (size_t)row * CI + 2 * p);
            mac(d.x, d.y, xv.x, xv.y, valid, ((int)(s0 / U) & 3) == wave);
        }
    }
    const size_t nb = gridDim.x, b = blockIdx.x;
    float *wsP = a.ws + b * (CO * CI);
    float *wsG = a.ws + nb * (CO * CI) + (b * 4 + wave) * (CI * CI);
    float *wsS = a.ws + nb * (CO * CI) + nb * 4 * (CI * CI) + (b * 4 + wave) * CI;
    float *wsU = a.ws + nb * (CO * CI) + nb * 4 * (CI * CI) + nb * 4 * CI + b * CO;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 2 * mfma32_row(r, lane) + i, col = 2 * p + j;
                wsP[(size_t)(cob + row) * CI + col] = acc[i][j][r];
                wsG[row * CI + col] = accg[i][j][r];
            }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float s1 = su[k] + __shfl_xor(su[k], 32, 64);
        const float s2 = sa[k] + __shfl_xor(sa[k], 32, 64);
        if (h == 0) {
            wsU[cob + 2 * p + k] = s1;
            wsS[2 * p + k] = s2;
        }
    }
}

// fp64 fixed-order fold of the per-block partials into lin = P | G | S | SU (block = 64 entries x 16 slices)
__global__ __launch_bounds__(1024) void vprod_reduce_kernel(const float *__restrict__ ws, int nb, float *__restrict__ lin)
{
    __shared__ double s_acc[16][64];
    const int tx = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + tx;
    double s = 0.0;
    if (i < LIN_FLOATS) {
        const float *base;
        int n, nparts, e;
        if (i < LIN_G) { base = ws; n = CO * CI; nparts = nb; e = i; }
        else if (i < LIN_S) { base = ws + (size_t)nb * (CO * CI); n = CI * CI; nparts = 4 * nb; e = i - LIN_G; }
        else if (i < LIN_SU) { base = ws + (size_t)nb * (CO * CI) + (size_t)nb * 4 * (CI * CI); n = CI; nparts = 4 * nb; e = i - LIN_S; }
        else { base = ws + (size_t)nb * (CO * CI) + (size_t)nb * 4 * (CI * CI) + (size_t)nb * 4 * CI; n = CO; nparts = nb; e = i - LIN_SU; }
        for (int q = slice; q < nparts; q += 16) s += (double)base[(size_t)q * n + e];
    }
    s_acc[slice][tx] = s;
    __syncthreads();
    if (slice == 0 && i < LIN_FLOATS) {
        double t = 0.0;
        for (int j = 0; j < 16; ++j) t += s_acc[j][tx];
        lin[i] = (float)t;
    }
}

// part [2][256] = (sum v, sum v*xhat(z)) of the BatchNorm behind the conv, from P and SU (one row of "partials")
__global__ __launch_bounds__(256) void lin_bnsums_kernel(const float *__restrict__ lin, const float *__restrict__ w,
                                                         const float *__restrict__ mean,
                                                         const float *__restrict__ invstd, float *__restrict__ part)
{
    const int c = threadIdx.x;
    double dot = 0.0;
    for (int k = 0; k < CI; ++k) dot += (double)w[c * CI + k] * (double)lin[LIN_P + c * CI + k];
    const double su = lin[LIN_SU + c];
    part[c] = (float)su;
    part[CO + c] = (float)((double)invstd[c] * (dot - (double)mean[c] * su));
}

// dW = A.P + B.(W G) + C (x) S;  M = W^T diag(B) W;  cvec = C^T W;  avec = A | 0 | 0
// blocks 0..255: row c of dW (64 threads);  blocks 256..319: row k of M;  block 320: cvec and avec
__global__ __launch_bounds__(64) void lin_finish_kernel(const float *__restrict__ lin, const float *__restrict__ abc,
                                                        const float *__restrict__ w, float *__restrict__ dw,
                                                        float *__restrict__ m, float *__restrict__ cvec,
                                                        float *__restrict__ avec)
{
    const int j = threadIdx.x, blk = blockIdx.x;
    if (blk < CO) {
        const int c = blk;
        double wg = 0.0;
        for (int k = 0; k < CI; ++k) wg += (double)w[c * CI + k] * (double)lin[LIN_G + k * CI + j];
        dw[c * CI + j] = (float)((double)abc[c] * (double)lin[LIN_P + c * CI + j] + (double)abc[CO + c] * wg +
                                 (double)abc[2 * CO + c] * (double)lin[LIN_S + j]);
    } else if (blk < CO + CI) {
        const int k = blk - CO;
        double t = 0.0;
        for (int c = 0; c < CO; ++c) t += (double)w[c * CI + k] * (double)abc[CO + c] * (double)w[c * CI + j];
        m[k * CI + j] = (float)t;
    } else {
        double t = 0.0;
        for (int c = 0; c < CO; ++c) t += (double)abc[2 * CO + c] * (double)w[c * CI + j];
        cvec[j] = (float)t;
        for (int c = j; c < CO; c += 64) {
            avec[c] = abc[c];
            avec[CO + c] = 0.f;
            avec[2 * CO + c] = 0.f;
        }
    }
}

inline int vprod_grid(long long R)
{
    const long long npairs = (R + 1) / 2, want = (npairs + 63) / 64;
    return cova_internal_persistent_grid2(want > (1 << 30) ? (1 << 30) : (int)want, 2);
}

}  // namespace

// ====================================================================================
// C ABI
// ====================================================================================
COVA_API int cova_conv1x1_lin_floats(void) { return LIN_FLOATS; }

COVA_API int cova_conv1x1_vprod_workspace_floats(long long R)
{
    (void)R;
    const long long nb = cova_internal_persistent_grid2(1 << 30, 2);
    return (int)(nb * (CO * CI + 4 * CI * CI + 4 * CI + CO));
}

// lin [cova_conv1x1_lin_floats] = P [256,64] | G [64,64] | S [64] | SU [256]  with  a = relu?(act_abc[0]*act +
// act_abc[2])  (act_abc nullable: a = act), v [R,256], act [R,64]; ws >= cova_conv1x1_vprod_workspace_floats
COVA_API int cova_conv1x1_vprod(const float *v, const float *act, const float *act_abc, int act_relu, float *lin,
                                float *ws, long long R, void *stream)
{
    COVA_REQUIRE(v && act && lin && ws && R > 0);
    const VPArgs a{v, act, act_abc, ws, R, act_relu};
    const int grid = vprod_grid(R);
    hipStream_t st = (hipStream_t)stream;
    if (act_abc) hipLaunchKernelGGL(conv1x1_vprod_kernel<true>, dim3(grid), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(conv1x1_vprod_kernel<false>, dim3(grid), dim3(256), 0, st, a);
    COVA_LAUNCH_CHECK();
    hipLaunchKernelGGL(vprod_reduce_kernel, dim3(cdiv(LIN_FLOATS, 64)), dim3(1024), 0, st, ws, grid, lin);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

// part [2][256] = (sum v, sum v*xhat) for cova_bn_finalize_bwd(_abc) with nparts = 1; w = the conv weight [256,64]
COVA_API int cova_conv1x1_lin_bnsums(const float *lin, const float *w, const float *mean, const float *invstd,
                                     float *part, void *stream)
{
    COVA_REQUIRE(lin && w && mean && invstd && part);
    hipLaunchKernelGGL(lin_bnsums_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, lin, w, mean, invstd, part);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

// abc [3][256] = A | B | C of dz = A*v + B*z + C  ->  dw [256,64], m [64,64], cvec [64], avec [3][256] = A | 0 | 0
// (the operands of cova_conv1x1_lin_dgrad)
COVA_API int cova_conv1x1_lin_finish(const float *lin, const float *abc, const float *w, float *dw, float *m,
                                     float *cvec, float *avec, void *stream)
{
    COVA_REQUIRE(lin && abc && w && dw && m && cvec && avec);
    hipLaunchKernelGGL(lin_finish_kernel, dim3(CO + CI + 1), dim3(64), 0, (hipStream_t)stream, lin, abc, w, dw, m,
                       cvec, avec);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}
