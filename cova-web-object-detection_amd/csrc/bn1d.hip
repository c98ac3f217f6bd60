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

// ---- float4 form (the default when the operands are 16-byte aligned; cova_set_option(14, 0) selects the kernels above) ----
// The kernels above walk a column block in 128 row slices: at configs[1] (1 440 rows) a thread makes 12 dependent round
// trips per pass and the backward takes 36 us for 11 MB.  Here a thread owns FOUR adjacent columns (one float4 per operand
// and row) of up to NR rows -- 512 row slices per block of 8 columns -- so that every load of the launch is requested at
// once and the second pass runs from registers (no re-read): one memory round trip, a shuffle + LDS tree (fixed order:
// deterministic), finalize, apply.  Same arithmetic per element as above (fp64 sums, bn_tail.h's channel function, the
// same fmaf forms); the sums are associated differently (tree instead of runs), which fp64 does not show in fp32.
constexpr int V4_COLS = 8, V4_SLICES = 512, V4_THREADS = 1024;

// fixed-order totals of eight columns x two sums over the block: lanes of a wave by shuffles, the 16 waves through LDS.
// v[j][w]: column j of the thread's quad q, sum w.  Threads 0..7 return (ta, tb) of column threadIdx.x of the block.
__device__ __forceinline__ void v4_totals(double (*s_red)[V4_COLS][2], double v[4][2], int q, int lane, int wave, double &ta,
                                          double &tb)
{
#pragma unroll
    for (int o = 2; o < 64; o <<= 1)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j][0] += __shfl_xor(v[j][0], o, 64);
            v[j][1] += __shfl_xor(v[j][1], o, 64);
        }
    if (lane < 2)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s_red[wave][q * 4 + j][0] = v[j][0];
            s_red[wave][q * 4 + j][1] = v[j][1];
        }
    __syncthreads();
    ta = 0.0;
    tb = 0.0;
    if (threadIdx.x < V4_COLS)
        for (int w = 0; w < V4_THREADS / 64; ++w) {
            ta += s_red[w][threadIdx.x][0];
            tb += s_red[w][threadIdx.x][1];
        }
}

template <int NR, bool DROP>
__global__ __launch_bounds__(V4_THREADS) void bn1d_fwd_v4_kernel(
    const float *__restrict__ x, int ldx, int R, int C, const float *__restrict__ gamma, const float *__restrict__ beta,
    float *__restrict__ running_mean, float *__restrict__ running_var, long long *__restrict__ nbt, float momentum,
    float eps, int relu, float *__restrict__ out, int ldo, float *__restrict__ dropped, int ldd,
    uint8_t *__restrict__ mask, float p, unsigned long long seed, int mask_given, float *__restrict__ scale,
    float *__restrict__ shift, float *__restrict__ mean, float *__restrict__ invstd)
{
    __shared__ double s_red[V4_THREADS / 64][V4_COLS][2];
    __shared__ float s_sc[V4_COLS], s_sh[V4_COLS];
    const int q = threadIdx.x & 1, slice = threadIdx.x >> 1, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * V4_COLS + q * 4;
    const bool colok = c < C;                                       // (C % 4 == 0: a quad is inside or outside)
    float4 xv[NR];
    double v[4][2] = {{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}};
#pragma unroll
    for (int u = 0; u < NR; ++u) {
        const int r = slice + u * V4_SLICES;
        xv[u] = (colok && r < R) ? *reinterpret_cast<const float4 *>(x + (size_t)r * ldx + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < NR; ++u) {                                   // (rows past R hold zeros: they add nothing)
        const double d[4] = {(double)xv[u].x, (double)xv[u].y, (double)xv[u].z, (double)xv[u].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j][0] += d[j];
            v[j][1] = fma(d[j], d[j], v[j][1]);
        }
    }
    double ta, tb;
    v4_totals(s_red, v, q, lane, wave, ta, tb);
    const int cf = blockIdx.x * V4_COLS + threadIdx.x;
    if (threadIdx.x < V4_COLS && cf < C) {
        bn_fwd_channel(ta, tb, (double)R, cf, gamma, beta, running_mean, running_var, momentum, eps, scale, shift, mean,
                       invstd);
        s_sc[threadIdx.x] = scale[cf];
        s_sh[threadIdx.x] = shift[cf];
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && nbt != nullptr) *nbt += 1;
    __syncthreads();
    if (!colok) return;
    const float sc[4] = {s_sc[q * 4], s_sc[q * 4 + 1], s_sc[q * 4 + 2], s_sc[q * 4 + 3]};
    const float sh[4] = {s_sh[q * 4], s_sh[q * 4 + 1], s_sh[q * 4 + 2], s_sh[q * 4 + 3]};
    const float inv = DROP ? 1.f / (1.f - p) : 1.f;
#pragma unroll
    for (int u = 0; u < NR; ++u) {
        const int r = slice + u * V4_SLICES;
        if (r >= R) break;
        const float xs[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
        float y[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            y[j] = fmaf(sc[j], xs[j], sh[j]);                       // same form as cova_bn_act_fwd
            if (relu) y[j] = y[j] > 0.f ? y[j] : 0.f;
        }
        *reinterpret_cast<float4 *>(out + (size_t)r * ldo + c) = make_float4(y[0], y[1], y[2], y[3]);
        if (DROP) {                                                 // cova_dropout_fwd on the result
            const size_t i = (size_t)r * C + c;
            uchar4 k4;
            if (mask_given) {
                k4 = *reinterpret_cast<const uchar4 *>(mask + i);
            } else {
                k4.x = hash_uniform(seed, (unsigned long long)i) >= p ? 1 : 0;
                k4.y = hash_uniform(seed, (unsigned long long)i + 1) >= p ? 1 : 0;
                k4.z = hash_uniform(seed, (unsigned long long)i + 2) >= p ? 1 : 0;
                k4.w = hash_uniform(seed, (unsigned long long)i + 3) >= p ? 1 : 0;
                *reinterpret_cast<uchar4 *>(mask + i) = k4;
            }
            *reinterpret_cast<float4 *>(dropped + (size_t)r * ldd + c) =
                make_float4(k4.x ? y[0] * inv : 0.f, k4.y ? y[1] * inv : 0.f, k4.z ? y[2] * inv : 0.f, k4.w ? y[3] * inv : 0.f);
        }
    }
}

template <int NR, bool DROP>
__global__ __launch_bounds__(V4_THREADS) void bn1d_bwd_v4_kernel(
    const float *__restrict__ dout, int ldg, const uint8_t *__restrict__ drop_mask, float p,
    const float *__restrict__ act, int lda, const float *__restrict__ z, int ldz, const float *__restrict__ mean,
    const float *__restrict__ invstd, const float *__restrict__ scale, int R, int C, float *__restrict__ dgamma,
    float *__restrict__ dbeta, float *__restrict__ dz, int lddz, float *__restrict__ dz_colsum)
{
    __shared__ double s_red[V4_THREADS / 64][V4_COLS][2];
    __shared__ float s_c1[V4_COLS], s_c2[V4_COLS];
    const int q = threadIdx.x & 1, slice = threadIdx.x >> 1, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * V4_COLS + q * 4;
    const bool colok = c < C;
    const float inv = DROP ? 1.f / (1.f - p) : 1.f;
    float mu[4] = {0.f, 0.f, 0.f, 0.f}, is[4] = {0.f, 0.f, 0.f, 0.f}, sc[4] = {0.f, 0.f, 0.f, 0.f};
    if (colok) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { mu[j] = mean[c + j]; is[j] = invstd[c + j]; sc[j] = scale[c + j]; }
    }
    // every operand of every row of the thread is requested before the first is used
    float4 gv[NR], av[NR], zv[NR];
    uchar4 mv[NR];
#pragma unroll
    for (int u = 0; u < NR; ++u) {
        const int r = slice + u * V4_SLICES;
        const bool ok = colok && r < R;
        const size_t rr = ok ? (size_t)r : 0;
        const int cc = ok ? c : 0;
        gv[u] = *reinterpret_cast<const float4 *>(dout + rr * ldg + cc);
        zv[u] = *reinterpret_cast<const float4 *>(z + rr * ldz + cc);
        if (act != nullptr) av[u] = *reinterpret_cast<const float4 *>(act + rr * lda + cc);
        if (DROP) mv[u] = *reinterpret_cast<const uchar4 *>(drop_mask + rr * C + cc);
        if (!ok) gv[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    double v[4][2] = {{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}};
    float dy[NR][4], xh[NR][4];
#pragma unroll
    for (int u = 0; u < NR; ++u) {
        const float g4[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
        const float z4[4] = {zv[u].x, zv[u].y, zv[u].z, zv[u].w};
        const float a4[4] = {av[u].x, av[u].y, av[u].z, av[u].w};
        const uint8_t m4[4] = {mv[u].x, mv[u].y, mv[u].z, mv[u].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float g = g4[j];
            if (DROP) g = m4[j] ? g * inv : 0.f;                                   // cova_dropout_bwd
            if (act != nullptr && !(a4[j] > 0.f)) g = 0.f;                          // ReLU mask
            dy[u][j] = g;
            xh[u][j] = (z4[j] - mu[j]) * is[j];                                     // xhat in fp32, as the apply pass forms it
            v[j][0] += (double)g;                                                   // (rows past R: g = 0)
            v[j][1] = fma((double)g, (double)xh[u][j], v[j][1]);
        }
    }
    double ta, tb;
    v4_totals(s_red, v, q, lane, wave, ta, tb);
    const int cf = blockIdx.x * V4_COLS + threadIdx.x;
    if (threadIdx.x < V4_COLS && cf < C) {
        if (dbeta) dbeta[cf] = (float)ta;
        if (dgamma) dgamma[cf] = (float)tb;
        s_c1[threadIdx.x] = (float)(ta / (double)R);                 // cova_bn_finalize_bwd's coef rows
        s_c2[threadIdx.x] = (float)(tb / (double)R);
    }
    __syncthreads();
    double cs[4][2] = {{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}};
    if (colok) {
        float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < NR; ++u) {
            const int r = slice + u * V4_SLICES;
            if (r >= R) break;
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] = sc[j] * (dy[u][j] - s_c1[q * 4 + j] - xh[u][j] * s_c2[q * 4 + j]);      // same form as cova_bn_bwd_apply
                s[j] += o[j];
            }
            *reinterpret_cast<float4 *>(dz + (size_t)r * lddz + c) = make_float4(o[0], o[1], o[2], o[3]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) cs[j][0] = (double)s[j];
    }
    if (dz_colsum != nullptr) {                                      // (uniform branch: every thread reaches the barriers)
        __syncthreads();                                             // s_red is read by threads 0..7 above
        double t1, t2;
        v4_totals(s_red, cs, q, lane, wave, t1, t2);
        if (threadIdx.x < V4_COLS && cf < C) dz_colsum[cf] = (float)t1;
    }
}

int g_bn1d_variant = 1;

inline bool al16(const void *p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

int cova_internal_set_bn1d_variant(int v) { g_bn1d_variant = v; return COVA_OK; }
int cova_internal_get_bn1d_variant() { return (int)g_bn1d_variant; }

COVA_API int cova_bn1d_fwd(const float *x, int ldx, int R, int C, const float *gamma, const float *beta,
                           float *running_mean, float *running_var, long long *num_batches_tracked, float momentum,
                           float eps, int relu, float *out, int ldo, float *dropped, int ld_dropped, uint8_t *mask,
                           float p, unsigned long long seed, int mask_given, float *scale, float *shift, float *mean,
                           float *invstd, void *stream)
{
    COVA_REQUIRE(x && gamma && beta && out && scale && shift && mean && invstd && R > 0 && C > 0);
    COVA_REQUIRE((running_mean == nullptr) == (running_var == nullptr));
    COVA_REQUIRE(dropped == nullptr || (mask != nullptr && p >= 0.f && p < 1.f));
    const bool v4 = g_bn1d_variant != 0 && C % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && al16(x) && al16(out) &&
                    (dropped == nullptr || (ld_dropped % 4 == 0 && al16(dropped) && ((uintptr_t)mask & 3) == 0)) &&
                    R <= 8 * V4_SLICES;
    if (v4) {
        const dim3 g4(cdiv(C, V4_COLS));
#define BN1D_FWD_V4(NR_, DROP_)                                                                                             \
        hipLaunchKernelGGL((bn1d_fwd_v4_kernel<NR_, DROP_>), g4, dim3(V4_THREADS), 0, (hipStream_t)stream, x, ldx, R, C, gamma, \
                           beta, running_mean, running_var, num_batches_tracked, momentum, eps, relu, out, ldo, dropped,   \
                           ld_dropped, mask, p, seed, mask_given, scale, shift, mean, invstd)
        if (R <= 3 * V4_SLICES) { if (dropped) BN1D_FWD_V4(3, true); else BN1D_FWD_V4(3, false); }
        else if (R <= 6 * V4_SLICES) { if (dropped) BN1D_FWD_V4(6, true); else BN1D_FWD_V4(6, false); }
        else { if (dropped) BN1D_FWD_V4(8, true); else BN1D_FWD_V4(8, false); }
#undef BN1D_FWD_V4
        COVA_LAUNCH_CHECK();
        return COVA_OK;
    }
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
    const bool v4 = g_bn1d_variant != 0 && C % 4 == 0 && ldg % 4 == 0 && ldz % 4 == 0 && lddz % 4 == 0 && al16(dout) &&
                    al16(z) && al16(dz) && (act == nullptr || (lda % 4 == 0 && al16(act))) &&
                    (drop_mask == nullptr || ((uintptr_t)drop_mask & 3) == 0) && R <= 6 * V4_SLICES;   // (8 rows: 128 registers, spills)
    if (v4) {
        const dim3 g4(cdiv(C, V4_COLS));
#define BN1D_BWD_V4(NR_, DROP_)                                                                                             \
        hipLaunchKernelGGL((bn1d_bwd_v4_kernel<NR_, DROP_>), g4, dim3(V4_THREADS), 0, (hipStream_t)stream, dout, ldg, drop_mask, \
                           p, act, lda, z, ldz, mean, invstd, scale, R, C, dgamma, dbeta, dz, lddz, dz_colsum)
        if (R <= 3 * V4_SLICES) { if (drop_mask) BN1D_BWD_V4(3, true); else BN1D_BWD_V4(3, false); }
        else { if (drop_mask) BN1D_BWD_V4(6, true); else BN1D_BWD_V4(6, false); }
#undef BN1D_BWD_V4
        COVA_LAUNCH_CHECK();
        return COVA_OK;
    }
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
