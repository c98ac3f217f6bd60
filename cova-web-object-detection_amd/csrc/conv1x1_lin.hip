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
#include "bf3.h"

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

// GEMM M = 256 (v channels), N = 64 (a channels), K = pixels, on the bf16 matrix pipe: every f32 operand as three
// round-to-nearest bf16 pieces, the six products of order <= 2 accumulated in f32 (bf3.h; DESIGN.md sections 11.8 / 12.8) --
// with v_mfma_f32_32x32x2_f32 the launch was bound by the matrix pipe (134 GFLOP = 0.86 ms of f32-MFMA time for 4.2 GB of
// operands); six bf16 products of K = 16 take 0.32 ms and leave it to HBM.
// A K-step = 16 pixels: half-wave h takes pixels 8h .. 8h+7, lane p loads channels {2p, 2p+1} of each as one float2 (operands
// straight from HBM, 256 contiguous bytes per half-wave and load) -- the eight pixels of a channel are the eight consecutive
// k of an MFMA operand, so accumulator (i, j) holds co = 64u + 2 row + i, ci = 2 col + j as before.  Wave u owns v channels
// 64u .. 64u+63; the Gram matrix of a (the same operand on both sides) is taken by wave (K-step & 3), as are the sums of a.
// The bf16 MFMA drops low product bits toward -infinity (tools/probe/mfma_round_probe.hip): odd blocks multiply with the
// NEGATED a on the B side and negate their partial sums back, so that the bias has no common sign over the blocks.
__device__ __forceinline__ f32x16 vp_mfma(u32x4 a, u32x4 b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// (one instantiation: without act_abc the prologue runs with A = 1, C = 0 -- exact; the variant without it made the register
// allocator spill the Gram accumulators)
__global__ __launch_bounds__(256, 2) void conv1x1_vprod_kernel(const VPArgs a)
{
    constexpr bool PRO_ACT = true;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = lane & 31, h = lane >> 5;
    const int cob = wave * 64;
    float aA[2] = {1.f, 1.f}, aC[2] = {0.f, 0.f};
    if (a.act_abc != nullptr) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            aA[k] = a.act_abc[2 * p + k];
            aC[k] = a.act_abc[2 * CI + 2 * p + k];
        }
    }
    // (the Gram matrix: wave u takes its quadrant (i, j) = (u >> 1, u & 1) of every K-step -- one accumulator instead of four:
    // with all four the register allocator spilled ~120 registers)
    f32x16 acc[2][2], accg;
#pragma unroll
    for (int r = 0; r < 16; ++r) accg[r] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float su[2] = {0.f, 0.f}, sa[2] = {0.f, 0.f};
    const bool relu = a.act_abc != nullptr && a.act_relu != 0;
    const int gi = wave >> 1, gj = wave & 1;                   // (wave-uniform)
    const bool neg = (blockIdx.x & 1) != 0;
    const float sgn = neg ? -1.f : 1.f;
    const uint32_t flip = neg ? 0x80008000u : 0u;

    // K-steps of this block: a contiguous range
    const long long nsteps = (a.R + 15) / 16;
    const long long per = (nsteps + gridDim.x - 1) / gridDim.x;
    const long long k_lo = (long long)blockIdx.x * per;
    long long k_hi = k_lo + per;
    if (k_hi > nsteps) k_hi = nsteps;
    float2 g[8], x[8];
    // lane offsets inside a K-step (loop invariant): a uniform base per K-step + these + t rows as the instruction's immediate
    const unsigned ov = (unsigned)((8 * h) * CO + cob + 2 * p), ox = (unsigned)((8 * h) * CI + 2 * p);
    auto issue = [&](long long ks) {                           // a K-step entirely inside the map
        const float *vb = a.v + (size_t)ks * 16 * CO, *xb = a.act + (size_t)ks * 16 * CI;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            g[t] = *reinterpret_cast<const float2 *>(vb + ov + t * CO);
            x[t] = *reinterpret_cast<const float2 *>(xb + ox + t * CI);
        }
    };
    // one K-step from the operands in g / x: prologue, validity, sums, pieces; then `next` (the following requests); MFMAs
    auto step = [&](long long ks, bool ragged, auto next) __attribute__((always_inline)) {
        const bool gram = (int)(ks & 3) == wave;               // wave-uniform: this wave takes the K-step's sums of a
        float gv[2][8], xv[2][8];
        const long long r0 = ks * 16 + 8 * h;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            float d0 = g[t].x, d1 = g[t].y, x0 = x[t].x, x1 = x[t].y;
            if (PRO_ACT) {
                x0 = fmaf(aA[0], x0, aC[0]);
                x1 = fmaf(aA[1], x1, aC[1]);
                if (relu) { x0 = x0 > 0.f ? x0 : 0.f; x1 = x1 > 0.f ? x1 : 0.f; }
            }
            if (ragged && r0 + t >= a.R) d0 = d1 = x0 = x1 = 0.f;
            su[0] += d0;
            su[1] += d1;
            if (gram) { sa[0] += x0; sa[1] += x1; }
            gv[0][t] = d0; gv[1][t] = d1;
            xv[0][t] = x0 * sgn; xv[1][t] = x1 * sgn;
        }
        u32x4 ap[2][3], bp[2][3];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            bf3_split8(gv[i], ap[i][0], ap[i][1], ap[i][2]);
            bf3_split8(xv[i], bp[i][0], bp[i][1], bp[i][2]);
        }
        next();
        __builtin_amdgcn_sched_barrier(0);                     // keep the requests ahead of the MFMAs (not sunk to their uses)
        // the six products of order <= 2, smallest first: a2 b0, a0 b2, a1 b1, a1 b0, a0 b1, a0 b0
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                f32x16 c = acc[i][j];
                c = vp_mfma(ap[i][2], bp[j][0], c);
                c = vp_mfma(ap[i][0], bp[j][2], c);
                c = vp_mfma(ap[i][1], bp[j][1], c);
                c = vp_mfma(ap[i][1], bp[j][0], c);
                c = vp_mfma(ap[i][0], bp[j][1], c);
                c = vp_mfma(ap[i][0], bp[j][0], c);
                acc[i][j] = c;
            }
        {                                                      // Gram quadrant; A side: the pieces of +a (= the B pieces with the block's sign undone)
            u32x4 ag[3];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
#pragma unroll
                for (int e = 0; e < 4; ++e) ag[pc][e] = (gi ? bp[1][pc][e] : bp[0][pc][e]) ^ flip;
            const u32x4 b0 = gj ? bp[1][0] : bp[0][0], b1 = gj ? bp[1][1] : bp[0][1], b2 = gj ? bp[1][2] : bp[0][2];
            f32x16 c = accg;
            c = vp_mfma(ag[2], b0, c);
            c = vp_mfma(ag[0], b2, c);
            c = vp_mfma(ag[1], b1, c);
            c = vp_mfma(ag[1], b0, c);
            c = vp_mfma(ag[0], b1, c);
            c = vp_mfma(ag[0], b0, c);
            accg = c;
        }
    };
    // K-steps entirely inside the map: the next one's operands are in flight during this one's MFMAs (the last one re-requests
    // itself: every load in the loop is unconditional -- a load under a branch makes the compiler wait for all of them at the join)
    long long k_full = a.R / 16;
    if (k_full > k_hi) k_full = k_hi;
    if (k_lo < k_full) {
        issue(k_lo);
#pragma unroll 1
        for (long long ks = k_lo; ks < k_full; ++ks) step(ks, false, [&]() { issue(ks + 1 < k_full ? ks + 1 : ks); });
    }
    if (k_full < k_hi && k_full >= k_lo) {                     // the map's last, ragged K-step (one block, once): clamped rows, zeroed
        const long long r0 = k_full * 16 + 8 * h;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            long long row = r0 + t;
            if (row >= a.R) row = a.R - 1;
            g[t] = *reinterpret_cast<const float2 *>(a.v + (size_t)row * CO + cob + 2 * p);
            x[t] = *reinterpret_cast<const float2 *>(a.act + (size_t)row * CI + 2 * p);
        }
        step(k_full, true, []() {});
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
                wsP[(size_t)(cob + row) * CI + col] = acc[i][j][r] * sgn;
                wsG[row * CI + col] = (i == gi && j == gj) ? accg[r] * sgn : 0.f;     // (the fold sums the four waves' matrices)
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
        for (int q0 = slice; q0 < nparts; q0 += 16 * 8) {        // eight partial rows in flight, added in the same order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = base[(size_t)(q0 + 16 * u < nparts ? q0 + 16 * u : q0) * n + e];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (q0 + 16 * u < nparts) s += (double)v[u];
        }
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
    hipLaunchKernelGGL(conv1x1_vprod_kernel, dim3(grid), dim3(256), 0, st, a);
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
