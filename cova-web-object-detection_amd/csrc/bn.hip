// BatchNorm (train-mode batch statistics / eval-mode running statistics), ReLU, residual add
// and the 3x3/s2 max-pool of the conv stack; also serves the two BatchNorm1d layers.
//
// Reference semantics: nn.BatchNorm2d / nn.BatchNorm1d as the reference instantiates them
// (models.py:49-51 via torchvision ResNet, models.py:68, :73, :86): biased variance for
// normalisation, unbiased variance into running_var, momentum 0.1, eps 1e-5; train-mode
// statistics couple the whole device batch (SURVEY.md section 8a row BN).
//
// Tensors are row-major [R, C] with a leading dimension (NHWC activations are R = B*H*W,
// C = 64).  Statistics are reduced in two stages: fp32 per-chunk partials (produced here or
// by the conv epilogues), then a finalize kernel that combines the partials in fp64.
// These kernels are HBM-bound: algorithmic bytes = 4 B per element read or written.
#include "common.h"
#include "bn_tail.h"

namespace {

__device__ __forceinline__ int pow2_cols(int C)
{
    int cb = 1;
    while (cb < C && cb < 256) cb <<= 1;
    return cb;
}
inline int pow2_cols_host(int C)
{
    int cb = 1;
    while (cb < C && cb < 256) cb <<= 1;
    return cb;
}

// MODE 0: partial[chunk][0][c] = sum x, [1][c] = sum x^2
// MODE 1: (BN backward) dy = dout * (act > 0 if act != null); [0] = sum dy, [1] = sum dy*xhat
template <int MODE>
__global__ __launch_bounds__(256) void colreduce_kernel(
    const float *__restrict__ x, int ldx, const float *__restrict__ act, int lda,
    const float *__restrict__ z, int ldz, const float *__restrict__ mean,
    const float *__restrict__ invstd, float *__restrict__ partial, long long R, int C,
    int rows_per_chunk)
{
    __shared__ float s_s[256], s_q[256];
    const int CB = pow2_cols(C);
    const int RPB = 256 / CB;
    const int tx = threadIdx.x % CB, ty = threadIdx.x / CB;
    const int col = blockIdx.y * CB + tx;
    const long long r0 = (long long)blockIdx.x * rows_per_chunk;
    long long r1 = r0 + rows_per_chunk;
    if (r1 > R) r1 = R;
    float s = 0.f, q = 0.f;
    if (col < C) {
        float mu = 0.f, is = 0.f;
        if (MODE == 1) { mu = mean[col]; is = invstd[col]; }
        for (long long r = r0 + ty; r < r1; r += RPB) {
            float v = x[r * ldx + col];
            if (MODE == 0) {
                s += v;
                q += v * v;
            } else {
                if (act != nullptr && !(act[r * lda + col] > 0.f)) v = 0.f;
                const float xh = (z[r * ldz + col] - mu) * is;
                s += v;
                q += v * xh;
            }
        }
    }
    s_s[threadIdx.x] = s;
    s_q[threadIdx.x] = q;
    __syncthreads();
    if (ty == 0 && col < C) {
        for (int j = 1; j < RPB; ++j) { s += s_s[j * CB + tx]; q += s_q[j * CB + tx]; }
        partial[((size_t)blockIdx.x * 2 + 0) * C + col] = s;
        partial[((size_t)blockIdx.x * 2 + 1) * C + col] = q;
    }
}

// Combine partials [nparts][2][C] in fp64.  Block = 64 channels x 16 slices.
__device__ __forceinline__ void combine_partials(const float *__restrict__ partial, int nparts,
                                                 int C, int c, int slice, double *s_a, double *s_b,
                                                 double &tot_a, double &tot_b)
{
    double a = 0.0, b = 0.0;
    if (c < C)
        for (int p0 = slice; p0 < nparts; p0 += 16 * 8) {        // eight rows in flight, added in the same order (see bn_tail.h)
            float va[8], vb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int p = p0 + 16 * u < nparts ? p0 + 16 * u : p0;
                va[u] = partial[((size_t)p * 2 + 0) * C + c];
                vb[u] = partial[((size_t)p * 2 + 1) * C + c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (p0 + 16 * u < nparts) {
                    a += (double)va[u];
                    b += (double)vb[u];
                }
        }
    s_a[slice * 64 + (threadIdx.x & 63)] = a;
    s_b[slice * 64 + (threadIdx.x & 63)] = b;
    __syncthreads();
    tot_a = 0.0;
    tot_b = 0.0;
    if (slice == 0)
        for (int j = 0; j < 16; ++j) {
            tot_a += s_a[j * 64 + (threadIdx.x & 63)];
            tot_b += s_b[j * 64 + (threadIdx.x & 63)];
        }
}

// C == 64, contiguous rows: float4 per thread, 16 rows per block pass (the NHWC activations).
template <int MODE>
__global__ __launch_bounds__(256) void colreduce64_kernel(
    const float *__restrict__ x, const float *__restrict__ act, const float *__restrict__ z,
    const float *__restrict__ mean, const float *__restrict__ invstd, float *__restrict__ partial,
    long long R, int rows_per_chunk)
{
    __shared__ float s_s[16][64], s_q[16][64];
    const int c4 = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const long long r0 = (long long)blockIdx.x * rows_per_chunk;
    long long r1 = r0 + rows_per_chunk;
    if (r1 > R) r1 = R;
    float4 mu = make_float4(0, 0, 0, 0), is = mu;
    if (MODE == 1) {
        mu = *reinterpret_cast<const float4 *>(mean + c4 * 4);
        is = *reinterpret_cast<const float4 *>(invstd + c4 * 4);
    }
    float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    for (long long r = r0 + ty; r < r1; r += 16) {
        const float4 v = *reinterpret_cast<const float4 *>(x + r * 64 + c4 * 4);
        float vv[4] = {v.x, v.y, v.z, v.w};
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { s[j] += vv[j]; q[j] += vv[j] * vv[j]; }
        } else {
            const float4 zz = *reinterpret_cast<const float4 *>(z + r * 64 + c4 * 4);
            if (act != nullptr) {
                const float4 a = *reinterpret_cast<const float4 *>(act + r * 64 + c4 * 4);
                if (!(a.x > 0.f)) vv[0] = 0.f;
                if (!(a.y > 0.f)) vv[1] = 0.f;
                if (!(a.z > 0.f)) vv[2] = 0.f;
                if (!(a.w > 0.f)) vv[3] = 0.f;
            }
            const float xh[4] = {(zz.x - mu.x) * is.x, (zz.y - mu.y) * is.y, (zz.z - mu.z) * is.z,
                                 (zz.w - mu.w) * is.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) { s[j] += vv[j]; q[j] += vv[j] * xh[j]; }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { s_s[ty][c4 * 4 + j] = s[j]; s_q[ty][c4 * 4 + j] = q[j]; }
    __syncthreads();
    if (threadIdx.x < 64) {
        float a = 0.f, b = 0.f;
        for (int r = 0; r < 16; ++r) { a += s_s[r][threadIdx.x]; b += s_q[r][threadIdx.x]; }
        partial[((size_t)blockIdx.x * 2 + 0) * 64 + threadIdx.x] = a;
        partial[((size_t)blockIdx.x * 2 + 1) * 64 + threadIdx.x] = b;
    }
}

// Fold groups of `group` consecutive partial rows ([nparts][width] fp32) into one row each
// (fp64 accumulation inside a group): a cheap, deterministic first reduction stage so that the
// single-block finalize kernels never walk more than a few hundred rows.
__global__ __launch_bounds__(256) void partials_fold_kernel(const float *__restrict__ partial,
                                                            int nparts, int width, int group,
                                                            float *__restrict__ out)
{
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= width) return;
    const int p0 = blockIdx.y * group;
    int p1 = p0 + group;
    if (p1 > nparts) p1 = nparts;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    int p = p0;
    for (; p + 3 < p1; p += 4) {
        a0 += (double)partial[(size_t)p * width + col];
        a1 += (double)partial[(size_t)(p + 1) * width + col];
        a2 += (double)partial[(size_t)(p + 2) * width + col];
        a3 += (double)partial[(size_t)(p + 3) * width + col];
    }
    for (; p < p1; ++p) a0 += (double)partial[(size_t)p * width + col];
    out[(size_t)blockIdx.y * width + col] = (float)((a0 + a1) + (a2 + a3));
}

__global__ __launch_bounds__(1024) void bn_finalize_fwd_kernel(
    const float *__restrict__ partial, int nparts, int C, double count,
    const float *__restrict__ gamma, const float *__restrict__ beta,
    float *__restrict__ running_mean, float *__restrict__ running_var, float momentum, float eps,
    float *__restrict__ scale, float *__restrict__ shift, float *__restrict__ mean,
    float *__restrict__ invstd, long long *__restrict__ num_batches_tracked)
{
    __shared__ double s_a[1024], s_b[1024];
    if (num_batches_tracked != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *num_batches_tracked += 1;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), slice = threadIdx.x >> 6;
    double sum, sumsq;
    combine_partials(partial, nparts, C, c, slice, s_a, s_b, sum, sumsq);
    if (slice == 0 && c < C)
        bn_fwd_channel(sum, sumsq, count, c, gamma, beta, running_mean, running_var, momentum, eps, scale, shift, mean,
                       invstd);
}

__global__ void bn_eval_params_kernel(const float *__restrict__ gamma, const float *__restrict__ beta,
                                      const float *__restrict__ running_mean,
                                      const float *__restrict__ running_var, float eps, int C,
                                      float *__restrict__ scale, float *__restrict__ shift,
                                      float *__restrict__ mean, float *__restrict__ invstd)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float is = 1.f / sqrtf(running_var[c] + eps);
    const float sc = gamma[c] * is;
    scale[c] = sc;
    shift[c] = beta[c] - running_mean[c] * sc;
    if (mean) mean[c] = running_mean[c];
    if (invstd) invstd[c] = is;
}

// dgamma = sum dy*xhat, dbeta = sum dy; coef[0][c] = dbeta/n, coef[1][c] = dgamma/n
__global__ __launch_bounds__(1024) void bn_finalize_bwd_kernel(
    const float *__restrict__ partial, int nparts, int C, double count,
    float *__restrict__ dgamma, float *__restrict__ dbeta, float *__restrict__ coef)
{
    __shared__ double s_a[1024], s_b[1024];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), slice = threadIdx.x >> 6;
    double s1, s2;
    combine_partials(partial, nparts, C, c, slice, s_a, s_b, s1, s2);
    if (slice == 0 && c < C) {
        if (dbeta) dbeta[c] = (float)s1;
        if (dgamma) dgamma[c] = (float)s2;
        coef[c] = (float)(s1 / count);
        coef[C + c] = (float)(s2 / count);
    }
}

// finalize_bwd that also emits the affine form of the apply step,
//   dz = scale*(dy - c1 - xhat*c2) = A*dy + B*z + C,   abc = [3][C] = A | B | C,
// for the convolutions that apply it on load (cova_conv3x3_wino_pro / _wgrad_wino_pro).
__global__ __launch_bounds__(1024) void bn_finalize_bwd_abc_kernel(
    const float *__restrict__ partial, int nparts, int C, double count,
    float *__restrict__ dgamma, float *__restrict__ dbeta, const float *__restrict__ mean,
    const float *__restrict__ invstd, const float *__restrict__ scale, float *__restrict__ abc)
{
    __shared__ double s_a[1024], s_b[1024];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), slice = threadIdx.x >> 6;
    double s1, s2;
    combine_partials(partial, nparts, C, c, slice, s_a, s_b, s1, s2);
    if (slice == 0 && c < C) bn_bwd_abc_channel(s1, s2, count, c, C, dgamma, dbeta, mean, invstd, scale, abc);
}

// out = act(z*scale + shift (+ res));  V = vector width (1 or 4)
template <int V>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(
    const float *__restrict__ z, int ldz, const float *__restrict__ scale,
    const float *__restrict__ shift, const float *__restrict__ res, int ldres,
    float *__restrict__ out, int ldo, long long R, int C, int relu)
{
    const int CV = C / V;
    const long long total = R * CV;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / CV;
        const int c = (int)(i - r * CV) * V;
        float v[V], sc[V], sh[V], rr[V];
        if (V == 4) {
            *reinterpret_cast<float4 *>(v) = *reinterpret_cast<const float4 *>(z + r * ldz + c);
            *reinterpret_cast<float4 *>(sc) = *reinterpret_cast<const float4 *>(scale + c);
            *reinterpret_cast<float4 *>(sh) = *reinterpret_cast<const float4 *>(shift + c);
            if (res) *reinterpret_cast<float4 *>(rr) = *reinterpret_cast<const float4 *>(res + r * ldres + c);
        } else {
            v[0] = z[r * ldz + c]; sc[0] = scale[c]; sh[0] = shift[c];
            if (res) rr[0] = res[r * ldres + c];
        }
#pragma unroll
        for (int j = 0; j < V; ++j) {
            float y = fmaf(sc[j], v[j], sh[j]);      // same form as the conv prologues
            if (res) y += rr[j];
            if (relu) y = y > 0.f ? y : 0.f;
            v[j] = y;
        }
        if (V == 4) *reinterpret_cast<float4 *>(out + r * ldo + c) = *reinterpret_cast<float4 *>(v);
        else out[r * ldo + c] = v[0];
    }
}

// bn_act_fwd for the NHWC maps of the 64-channel stack (contiguous [R][64], residual, ReLU) that also leaves the ReLU
// decisions as one bit per element: bits[r][w] bit k = (out[r][32 w + k] > 0).  The data gradient that needs this map
// only as its mask source (conv_wino4.hip, `act_bits`) then reads 1/32 of the bytes.  A thread owns 4 channels; the 8
// lanes of a 32-channel word combine their nibbles with three lane exchanges.
// U: elements a thread has in flight per trip (2 and 4 measured slower than 1 in round 5: the launch runs at the copy rate; the
// stride is a multiple of 8 lanes, so the 8 lanes of a word stay together)
template <int U>
__global__ __launch_bounds__(256) void bn_act_fwd_bits_kernel(
    const float *__restrict__ z, const float *__restrict__ scale, const float *__restrict__ shift,
    const float *__restrict__ res, float *__restrict__ out, uint32_t *__restrict__ bits, long long R)
{
    const long long total = R * 16;                      // multiple of 16: the 8 lanes of a word are active together
    const long long stride = (long long)gridDim.x * blockDim.x;
    const int c = (int)(threadIdx.x & 15) * 4;           // (blockDim.x and the stride are multiples of 16)
    const float4 sc = *reinterpret_cast<const float4 *>(scale + c), sh = *reinterpret_cast<const float4 *>(shift + c);
    for (long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x; i0 < total; i0 += stride * U) {
      float4 vv[U], rr[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long i = i0 + u * stride;
        const long long ic = i < total ? i : i0;
        vv[u] = *reinterpret_cast<const float4 *>(z + ic * 4);
        if (res != nullptr) rr[u] = *reinterpret_cast<const float4 *>(res + ic * 4);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long i = i0 + u * stride;
        if (i >= total) break;                           // (uniform over the 8 lanes of a word)
        const float4 v = vv[u];
        float y[4] = {fmaf(sc.x, v.x, sh.x), fmaf(sc.y, v.y, sh.y), fmaf(sc.z, v.z, sh.z), fmaf(sc.w, v.w, sh.w)};
        if (res != nullptr) {
            const float4 r4 = rr[u];
            y[0] += r4.x; y[1] += r4.y; y[2] += r4.z; y[3] += r4.w;
        }
        uint32_t word = 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            y[j] = y[j] > 0.f ? y[j] : 0.f;
            word |= (y[j] > 0.f ? 1u : 0u) << j;
        }
        *reinterpret_cast<float4 *>(out + i * 4) = make_float4(y[0], y[1], y[2], y[3]);
        word <<= 4 * (threadIdx.x & 7);
        word |= __shfl_xor(word, 1, 64);
        word |= __shfl_xor(word, 2, 64);
        word |= __shfl_xor(word, 4, 64);
        if ((threadIdx.x & 7) == 0) bits[i >> 3] = word;          // (i >> 3 = row * 2 + word of the row)
      }
    }
}

// out = act((z*scale + shift) + (z2*scale2 + shift2)): the join of a Bottleneck whose identity branch has
// its own conv + BatchNorm (torchvision Bottleneck.forward with `downsample`); contiguous [R, C], C % 4 == 0
__global__ __launch_bounds__(256) void bn_act2_fwd_kernel(
    const float *__restrict__ z, const float *__restrict__ scale, const float *__restrict__ shift,
    const float *__restrict__ z2, const float *__restrict__ scale2, const float *__restrict__ shift2,
    float *__restrict__ out, long long R, int C, int relu)
{
    const int CV = C / 4;
    const long long total = R * CV;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % CV) * 4;
        const float4 a = *reinterpret_cast<const float4 *>(z + i * 4);
        const float4 b = *reinterpret_cast<const float4 *>(z2 + i * 4);
        const float4 sa = *reinterpret_cast<const float4 *>(scale + c), ha = *reinterpret_cast<const float4 *>(shift + c);
        const float4 sb = *reinterpret_cast<const float4 *>(scale2 + c), hb = *reinterpret_cast<const float4 *>(shift2 + c);
        float4 y;
        y.x = fmaf(sa.x, a.x, ha.x) + fmaf(sb.x, b.x, hb.x);
        y.y = fmaf(sa.y, a.y, ha.y) + fmaf(sb.y, b.y, hb.y);
        y.z = fmaf(sa.z, a.z, ha.z) + fmaf(sb.z, b.z, hb.z);
        y.w = fmaf(sa.w, a.w, ha.w) + fmaf(sb.w, b.w, hb.w);
        if (relu) {
            y.x = y.x > 0.f ? y.x : 0.f; y.y = y.y > 0.f ? y.y : 0.f;
            y.z = y.z > 0.f ? y.z : 0.f; y.w = y.w > 0.f ? y.w : 0.f;
        }
        *reinterpret_cast<float4 *>(out + i * 4) = y;
    }
}

// dz = scale * (dy - c1 - xhat*c2), dy = dout * (act > 0) if act; optional dres = dy
template <int V>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const float *__restrict__ dout, int ldd, const float *__restrict__ act, int lda,
    const float *__restrict__ z, int ldz, const float *__restrict__ mean,
    const float *__restrict__ invstd, const float *__restrict__ scale,
    const float *__restrict__ coef, float *__restrict__ dz, int lddz, float *__restrict__ dres,
    int lddres, long long R, int C)
{
    const int CV = C / V;
    const long long total = R * CV;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / CV;
        const int c = (int)(i - r * CV) * V;
        float g[V], a[V], zz[V], o[V];
        if (V == 4) {
            *reinterpret_cast<float4 *>(g) = *reinterpret_cast<const float4 *>(dout + r * ldd + c);
            *reinterpret_cast<float4 *>(zz) = *reinterpret_cast<const float4 *>(z + r * ldz + c);
            if (act) *reinterpret_cast<float4 *>(a) = *reinterpret_cast<const float4 *>(act + r * lda + c);
        } else {
            g[0] = dout[r * ldd + c]; zz[0] = z[r * ldz + c];
            if (act) a[0] = act[r * lda + c];
        }
#pragma unroll
        for (int j = 0; j < V; ++j) {
            float dy = g[j];
            if (act && !(a[j] > 0.f)) dy = 0.f;
            g[j] = dy;
            const float xh = (zz[j] - mean[c + j]) * invstd[c + j];
            o[j] = scale[c + j] * (dy - coef[c + j] - xh * coef[C + c + j]);
        }
        if (V == 4) {
            *reinterpret_cast<float4 *>(dz + r * lddz + c) = *reinterpret_cast<float4 *>(o);
            if (dres) *reinterpret_cast<float4 *>(dres + r * lddres + c) = *reinterpret_cast<float4 *>(g);
        } else {
            dz[r * lddz + c] = o[0];
            if (dres) dres[r * lddres + c] = g[0];
        }
    }
}

// ------------------------------------------------------------------------------------
// conv1 tail: relu(bn(y)) followed by 3x3 / stride 2 / pad 1 max-pool, NHWC C = 64.
// idx[b,oy,ox,c] = window position (ky*3+kx) of the first maximum (row-major scan, strict >),
// which is where torch's max_pool2d backward routes the gradient.
// ------------------------------------------------------------------------------------
// A thread owns (output column ox, channel quad c4) and walks DOWN a strip of POOL_STRIP output rows, keeping the input
// row it shares with the next window (2 oy + 1) in registers: 6 float4 loads per output instead of 9, and a map row is
// fetched from HBM once per strip instead of once per output row that touches it (round 2 measured 1.56x read
// over-fetch with one thread per output: rows 2 oy - 1 / 2 oy + 1 were read by two blocks, usually on different XCDs).
// Neighbouring columns are neighbouring 16-lane groups of the same wave (L1 hits).
// STRIP / PRE (cova_set_option(13, v), tools/ew_bench.py): rows per strip; PRE: the two new input rows of output row
// oy + 1 are requested before output row oy is reduced and stored (12 instead of 6 loads of a thread in flight).
template <int POOL_STRIP, bool PRE, bool XCD = false, bool TILE2D = false>
__global__ __launch_bounds__(256) void bn_relu_maxpool_fwd_kernel(
    const float *__restrict__ y, const float *__restrict__ scale, const float *__restrict__ shift,
    float *__restrict__ out, uint8_t *__restrict__ idx, float *__restrict__ ymax, int B, int H1, int W1,
    int H2, int W2)
{
    // TILE2D (POOL_STRIP = 1): a block is 4 x 4 outputs x 16 channel quads (a wave = 4 adjacent columns of one output row)
    // instead of 16 adjacent columns of one row: 81 instead of 99 input pixels per block, the rest are L1 hits
    static_assert(!TILE2D || POOL_STRIP == 1, "TILE2D: one output row per thread");
    const int nstrips = (H2 + POOL_STRIP - 1) / POOL_STRIP;
    const int tiles_x = (W2 + 3) / 4, tiles_y = (H2 + 3) / 4;
    const long long total = TILE2D ? (long long)B * tiles_y * tiles_x * 256 : (long long)B * nstrips * W2 * 16;
    // XCD: consecutive blocks go to the eight XCDs in turn -- block b takes work block (b % 8) * (grid / 8) + b / 8, so that
    // an XCD walks ONE contiguous eighth of the maps and the input row two neighbouring strips share is met in its own L2
    long long wb = blockIdx.x;
    if (XCD) {
        const long long per = gridDim.x / 8;
        if (wb < per * 8) wb = (wb % 8) * per + wb / 8;
    }
    for (long long i = wb * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i & 15);
        long long p = i >> 4;
        int ox, strip, b;
        if (TILE2D) {
            const int px = (int)(p & 3), py = (int)((p >> 2) & 3);
            p >>= 4;                                               // tile index
            const int tx = (int)(p % tiles_x);
            p /= tiles_x;
            ox = 4 * tx + px;
            strip = 4 * (int)(p % tiles_y) + py;                   // (= the output row)
            b = (int)(p / tiles_y);
            if (ox >= W2 || strip >= H2) continue;
        } else {
            ox = (int)(p % W2);
            p /= W2;
            strip = (int)(p % nstrips);
            b = (int)(p / nstrips);
        }
        const float4 sc = *reinterpret_cast<const float4 *>(scale + c4 * 4);
        const float4 sh = *reinterpret_cast<const float4 *>(shift + c4 * 4);
        const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
        const float *yb = y + (size_t)b * H1 * W1 * 64 + c4 * 4;
        const int ix0 = 2 * ox - 1;
        const bool okx[3] = {ix0 >= 0, true, ix0 + 2 < W1};          // (2 ox < W1 always)
        auto load_row = [&](int iy, float4 (&r)[3]) {                // raw y of input row iy, columns ix0 .. ix0 + 2
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = min(max(ix0 + kx, 0), W1 - 1), iyc = min(max(iy, 0), H1 - 1);
                r[kx] = *reinterpret_cast<const float4 *>(yb + ((size_t)iyc * W1 + ix) * 64);
            }
        };
        const int oy0 = strip * POOL_STRIP, oy1 = min(oy0 + POOL_STRIP, H2);
        float4 top[3], mid[3], bot[3], nmid[3], nbot[3];
        load_row(2 * oy0 - 1, top);
        if (PRE) { load_row(2 * oy0, nmid); load_row(2 * oy0 + 1, nbot); }
        for (int oy = oy0; oy < oy1; ++oy) {
            if (PRE) {
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) { mid[kx] = nmid[kx]; bot[kx] = nbot[kx]; }
                if (oy + 1 < oy1) { load_row(2 * oy + 2, nmid); load_row(2 * oy + 3, nbot); }      // (rows clamped in load_row)
            } else {
                load_row(2 * oy, mid);
                load_row(2 * oy + 1, bot);
            }
            const bool oky[3] = {2 * oy - 1 >= 0, true, 2 * oy + 1 < H1};
            float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            float my[4] = {0.f, 0.f, 0.f, 0.f};          // raw conv output at the arg-max position
            int mi[4] = {0, 0, 0, 0};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float4 *row = ky == 0 ? top : ky == 1 ? mid : bot;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    if (!(oky[ky] && okx[kx])) continue;                   // padding: not part of the window
                    const float vv[4] = {row[kx].x, row[kx].y, row[kx].z, row[kx].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float t = fmaf(vv[j], scv[j], shv[j]);
                        const float a = t > 0.f ? t : 0.f;
                        if (a > m[j]) { m[j] = a; mi[j] = ky * 3 + kx; my[j] = vv[j]; }
                    }
                }
            }
            const size_t o = (((size_t)b * H2 + oy) * W2 + ox) * 64 + c4 * 4;
            *reinterpret_cast<float4 *>(out + o) = make_float4(m[0], m[1], m[2], m[3]);
            *reinterpret_cast<uchar4 *>(idx + o) = make_uchar4(mi[0], mi[1], mi[2], mi[3]);
            if (ymax != nullptr) *reinterpret_cast<float4 *>(ymax + o) = make_float4(my[0], my[1], my[2], my[3]);
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) top[kx] = bot[kx];              // row 2 oy + 1 is the next window's first row
        }
    }
}

// Gradient w.r.t. the bn output before relu at input pixel (b, Y, X): gather from the <= 4
// pooling windows that contain the pixel, then apply the relu mask.
__device__ __forceinline__ void pool_relu_gather(const float *__restrict__ dp,
                                                 const uint8_t *__restrict__ idx,
                                                 const float t[4], int b, int Y, int X, int H2,
                                                 int W2, int c4, float dy[4])
{
    dy[0] = dy[1] = dy[2] = dy[3] = 0.f;
    const int oy_lo = Y >> 1, oy_hi = (Y + 1) >> 1;     // equal when Y is even
    const int ox_lo = X >> 1, ox_hi = (X + 1) >> 1;
    for (int oy = oy_lo; oy <= oy_hi; ++oy) {
        if (oy >= H2) continue;
        const int ky = Y - (2 * oy - 1);
        for (int ox = ox_lo; ox <= ox_hi; ++ox) {
            if (ox >= W2) continue;
            const int kx = X - (2 * ox - 1);
            const int pos = ky * 3 + kx;
            const size_t o = (((size_t)b * H2 + oy) * W2 + ox) * 64 + c4 * 4;
            const uchar4 id = *reinterpret_cast<const uchar4 *>(idx + o);
            const float4 g = *reinterpret_cast<const float4 *>(dp + o);
            if (id.x == pos) dy[0] += g.x;
            if (id.y == pos) dy[1] += g.y;
            if (id.z == pos) dy[2] += g.z;
            if (id.w == pos) dy[3] += g.w;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (!(t[j] > 0.f)) dy[j] = 0.f;
}

// PASS 0: per-block partial sums of dy and dy*xhat ([block][2][64]);
// PASS 1: dz = scale*(dy - c1 - xhat*c2) written NHWC.
template <int PASS>
__global__ __launch_bounds__(256) void bn_relu_maxpool_bwd_kernel(
    const float *__restrict__ dp, const uint8_t *__restrict__ idx, const float *__restrict__ y,
    const float *__restrict__ scale, const float *__restrict__ shift,
    const float *__restrict__ mean, const float *__restrict__ invstd,
    const float *__restrict__ coef, float *__restrict__ dz, float *__restrict__ partial, int B,
    int H1, int W1, int H2, int W2)
{
    __shared__ float s_s[16][64], s_q[16][64];
    const int c4 = threadIdx.x & 15, prow = threadIdx.x >> 4;   // 16 pixels per block pass
    const float4 sc = *reinterpret_cast<const float4 *>(scale + c4 * 4);
    const float4 sh = *reinterpret_cast<const float4 *>(shift + c4 * 4);
    const float4 mu = *reinterpret_cast<const float4 *>(mean + c4 * 4);
    const float4 is = *reinterpret_cast<const float4 *>(invstd + c4 * 4);
    float c1[4] = {0, 0, 0, 0}, c2[4] = {0, 0, 0, 0};
    if (PASS == 1) {
        *reinterpret_cast<float4 *>(c1) = *reinterpret_cast<const float4 *>(coef + c4 * 4);
        *reinterpret_cast<float4 *>(c2) = *reinterpret_cast<const float4 *>(coef + 64 + c4 * 4);
    }
    const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
    const float muv[4] = {mu.x, mu.y, mu.z, mu.w}, isv[4] = {is.x, is.y, is.z, is.w};
    float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    const long long npix = (long long)B * H1 * W1;
    for (long long p = (long long)blockIdx.x * 16 + prow; p < npix; p += (long long)gridDim.x * 16) {
        const int X = (int)(p % W1);
        const long long pr = p / W1;
        const int Y = (int)(pr % H1);
        const int b = (int)(pr / H1);
        const float4 v = *reinterpret_cast<const float4 *>(y + (size_t)p * 64 + c4 * 4);
        const float vv[4] = {v.x, v.y, v.z, v.w};
        float t[4], dy[4], xh[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t[j] = vv[j] * scv[j] + shv[j];
            xh[j] = (vv[j] - muv[j]) * isv[j];
        }
        pool_relu_gather(dp, idx, t, b, Y, X, H2, W2, c4, dy);
        if (PASS == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { s[j] += dy[j]; q[j] += dy[j] * xh[j]; }
        } else {
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = scv[j] * (dy[j] - c1[j] - xh[j] * c2[j]);
            *reinterpret_cast<float4 *>(dz + (size_t)p * 64 + c4 * 4) =
                make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    if (PASS == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { s_s[prow][c4 * 4 + j] = s[j]; s_q[prow][c4 * 4 + j] = q[j]; }
        __syncthreads();
        if (threadIdx.x < 64) {
            float a = 0.f, bq = 0.f;
            for (int r = 0; r < 16; ++r) { a += s_s[r][threadIdx.x]; bq += s_q[r][threadIdx.x]; }
            partial[((size_t)blockIdx.x * 2 + 0) * 64 + threadIdx.x] = a;
            partial[((size_t)blockIdx.x * 2 + 1) * 64 + threadIdx.x] = bq;
        }
    }
}

// Window-based form of the PASS 0 reduction: every pooling window routes its gradient to exactly
// one input pixel (its argmax), so sum_pixels dy = sum_windows dp*[bn(y_argmax) > 0] and likewise
// for dy*xhat: one gather of y per (window, channel) instead of a sweep over the 4x larger input.
__global__ __launch_bounds__(256) void bn_relu_maxpool_bwd_reduce_win_kernel(
    const float *__restrict__ dp, const uint8_t *__restrict__ idx, const float *__restrict__ y,
    const float *__restrict__ scale, const float *__restrict__ shift,
    const float *__restrict__ mean, const float *__restrict__ invstd, float *__restrict__ partial,
    int B, int H1, int W1, int H2, int W2)
{
    __shared__ float s_s[16][64], s_q[16][64];
    const int c4 = threadIdx.x & 15, prow = threadIdx.x >> 4;
    const float4 sc = *reinterpret_cast<const float4 *>(scale + c4 * 4);
    const float4 sh = *reinterpret_cast<const float4 *>(shift + c4 * 4);
    const float4 mu = *reinterpret_cast<const float4 *>(mean + c4 * 4);
    const float4 is = *reinterpret_cast<const float4 *>(invstd + c4 * 4);
    const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
    const float muv[4] = {mu.x, mu.y, mu.z, mu.w}, isv[4] = {is.x, is.y, is.z, is.w};
    float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
    const long long nwin = (long long)B * H2 * W2;
    for (long long p = (long long)blockIdx.x * 16 + prow; p < nwin; p += (long long)gridDim.x * 16) {
        const int ox = (int)(p % W2);
        const long long pr = p / W2;
        const int oy = (int)(pr % H2);
        const int b = (int)(pr / H2);
        const uchar4 id = *reinterpret_cast<const uchar4 *>(idx + (size_t)p * 64 + c4 * 4);
        const float4 g = *reinterpret_cast<const float4 *>(dp + (size_t)p * 64 + c4 * 4);
        const int pos[4] = {id.x, id.y, id.z, id.w};
        const float gv[4] = {g.x, g.y, g.z, g.w};
        const float *yb = y + (size_t)b * H1 * W1 * 64 + c4 * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int ky = pos[j] / 3, kx = pos[j] - ky * 3;
            const int iy = 2 * oy - 1 + ky, ix = 2 * ox - 1 + kx;
            const float yv = yb[((size_t)iy * W1 + ix) * 64 + j];
            const float dy = (yv * scv[j] + shv[j] > 0.f) ? gv[j] : 0.f;
            s[j] += dy;
            q[j] += dy * ((yv - muv[j]) * isv[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { s_s[prow][c4 * 4 + j] = s[j]; s_q[prow][c4 * 4 + j] = q[j]; }
    __syncthreads();
    if (threadIdx.x < 64) {
        float a = 0.f, bq = 0.f;
        for (int r = 0; r < 16; ++r) { a += s_s[r][threadIdx.x]; bq += s_q[r][threadIdx.x]; }
        partial[((size_t)blockIdx.x * 2 + 0) * 64 + threadIdx.x] = a;
        partial[((size_t)blockIdx.x * 2 + 1) * 64 + threadIdx.x] = bq;
    }
}

// Quad form of the PASS 1 kernel: one thread per (pooling window, 4 channels) produces the 2x2 input
// pixels (2oy..2oy+1, 2ox..2ox+1).  Those four pixels only receive gradient from the windows
// (oy..oy+1, ox..ox+1), so 4 dp + 4 idx + 4 y loads (all independent) serve 4 outputs, instead of
// up to 9 dependent loads per output pixel.
__global__ __launch_bounds__(256) void bn_relu_maxpool_bwd_apply_quad_kernel(
    const float *__restrict__ dp, const uint8_t *__restrict__ idx, const float *__restrict__ y,
    const float *__restrict__ scale, const float *__restrict__ shift,
    const float *__restrict__ mean, const float *__restrict__ invstd,
    const float *__restrict__ coef, float *__restrict__ dz, int B, int H1, int W1, int H2, int W2)
{
    const int c4 = threadIdx.x & 15;
    const float4 sc4 = *reinterpret_cast<const float4 *>(scale + c4 * 4);
    const float4 sh4 = *reinterpret_cast<const float4 *>(shift + c4 * 4);
    const float4 mu4 = *reinterpret_cast<const float4 *>(mean + c4 * 4);
    const float4 is4 = *reinterpret_cast<const float4 *>(invstd + c4 * 4);
    const float4 k14 = *reinterpret_cast<const float4 *>(coef + c4 * 4);
    const float4 k24 = *reinterpret_cast<const float4 *>(coef + 64 + c4 * 4);
    const float sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
    const float mu[4] = {mu4.x, mu4.y, mu4.z, mu4.w}, is[4] = {is4.x, is4.y, is4.z, is4.w};
    const float k1[4] = {k14.x, k14.y, k14.z, k14.w}, k2[4] = {k24.x, k24.y, k24.z, k24.w};
    const long long nwin = (long long)B * H2 * W2;
    for (long long p = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4; p < nwin;
         p += ((long long)gridDim.x * blockDim.x) >> 4) {
        const int ox = (int)(p % W2);
        const long long pr = p / W2;
        const int oy = (int)(pr % H2);
        const int b = (int)(pr / H2);
        // the 2x2 windows (oy+i, ox+j)
        float g[2][2][4];
        int id[2][2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const bool ok = (oy + i < H2) && (ox + j < W2);
                float4 gv = make_float4(0.f, 0.f, 0.f, 0.f);
                uchar4 iv = make_uchar4(255, 255, 255, 255);
                if (ok) {
                    const size_t o = (((size_t)b * H2 + oy + i) * W2 + ox + j) * 64 + c4 * 4;
                    gv = *reinterpret_cast<const float4 *>(dp + o);
                    iv = *reinterpret_cast<const uchar4 *>(idx + o);
                }
                g[i][j][0] = gv.x; g[i][j][1] = gv.y; g[i][j][2] = gv.z; g[i][j][3] = gv.w;
                id[i][j][0] = iv.x; id[i][j][1] = iv.y; id[i][j][2] = iv.z; id[i][j][3] = iv.w;
            }
#pragma unroll
        for (int py = 0; py < 2; ++py) {
            const int Y = 2 * oy + py;
            if (Y >= H1) continue;
#pragma unroll
            for (int px = 0; px < 2; ++px) {
                const int X = 2 * ox + px;
                if (X >= W1) continue;
                const size_t o = (((size_t)b * H1 + Y) * W1 + X) * 64 + c4 * 4;
                const float4 yv4 = *reinterpret_cast<const float4 *>(y + o);
                const float yv[4] = {yv4.x, yv4.y, yv4.z, yv4.w};
                float out[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float dy = 0.f;
                    // window rows: py == 0 -> (oy, ky=1); py == 1 -> (oy, ky=2), (oy+1, ky=0)
#pragma unroll
                    for (int i = 0; i <= py; ++i) {
                        const int ky = py == 0 ? 1 : (i == 0 ? 2 : 0);
#pragma unroll
                        for (int j = 0; j <= px; ++j) {
                            const int kx = px == 0 ? 1 : (j == 0 ? 2 : 0);
                            if (id[i][j][c] == ky * 3 + kx) dy += g[i][j][c];
                        }
                    }
                    const float t = yv[c] * sc[c] + sh[c];
                    if (!(t > 0.f)) dy = 0.f;
                    const float xh = (yv[c] - mu[c]) * is[c];
                    out[c] = sc[c] * (dy - k1[c] - xh * k2[c]);
                }
                *reinterpret_cast<float4 *>(dz + o) = make_float4(out[0], out[1], out[2], out[3]);
            }
        }
    }
}

inline int ew_grid(long long total_threads)
{
    long long g = cdivll(total_threads, 256);
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (int)g;
}

inline bool vec4_ok(const void *p, int ld)
{
    return p == nullptr || (((uintptr_t)p & 15) == 0 && (ld & 3) == 0);
}

}  // namespace


// ====================================================================================
// C ABI
// ====================================================================================
COVA_API int cova_colreduce_rows_per_chunk(long long R, int C)
{
    const int rpb = 256 / pow2_cols_host(C);
    long long rows = cdivll(R, 2048);
    if (rows < 64) rows = 64;
    rows = cdivll(rows, rpb) * rpb;
    return (int)rows;
}

COVA_API int cova_colreduce_num_chunks(long long R, int C)
{
    return (int)cdivll(R, cova_colreduce_rows_per_chunk(R, C));
}

// out [cdiv(nparts, group)][width] = sums of `group` consecutive rows of partial [nparts][width]
COVA_API int cova_partials_fold(const float *partial, int nparts, int width, int group, float *out,
                                void *stream)
{
    COVA_REQUIRE(partial && out && nparts > 0 && width > 0 && group > 0);
    const dim3 grid(cdiv(width, 256), cdiv(nparts, group));
    hipLaunchKernelGGL(partials_fold_kernel, grid, dim3(256), 0, (hipStream_t)stream, partial, nparts,
                       width, group, out);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

// partial [num_chunks][2][C]: column sums and sums of squares of x [R, C] (ld = ldx)
COVA_API int cova_colstats(const float *x, int ldx, long long R, int C, float *partial, void *stream)
{
    COVA_REQUIRE(x && partial && R > 0 && C > 0);
    const int rpc = cova_colreduce_rows_per_chunk(R, C);
    const dim3 grid((unsigned)cdivll(R, rpc), (unsigned)cdiv(C, pow2_cols_host(C)));
    if (C == 64 && ldx == 64 && vec4_ok(x, ldx)) {
        hipLaunchKernelGGL(colreduce64_kernel<0>, dim3(grid.x), dim3(256), 0, (hipStream_t)stream, x,
                           (const float *)nullptr, (const float *)nullptr, (const float *)nullptr,
                           (const float *)nullptr, partial, R, rpc);
        COVA_LAUNCH_CHECK();
        return COVA_OK;
    }
    hipLaunchKernelGGL(colreduce_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, x, ldx,
                       (const float *)nullptr, 0, (const float *)nullptr, 0, (const float *)nullptr,
                       (const float *)nullptr, partial, R, C, rpc);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

// train-mode finalize: partial [nparts][2][C] -> scale/shift/mean/invstd (+ running stats update
// when running_mean != NULL)
COVA_API int cova_bn_finalize_fwd(const float *partial, int nparts, int C, double count,
                                  const float *gamma, const float *beta, float *running_mean,
                                  float *running_var, long long *num_batches_tracked, float momentum,
                                  float eps, float *scale, float *shift, float *mean, float *invstd,
                                  void *stream)
{
    COVA_REQUIRE(partial && gamma && beta && scale && shift && mean && invstd && nparts > 0 && C > 0);
    hipLaunchKernelGGL(bn_finalize_fwd_kernel, dim3(cdiv(C, 64)), dim3(1024), 0, (hipStream_t)stream,
                       partial, nparts, C, count, gamma, beta, running_mean, running_var, momentum,
                       eps, scale, shift, mean, invstd, num_batches_tracked);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

COVA_API int cova_bn_eval_params(const float *gamma, const float *beta, const float *running_mean,
                                 const float *running_var, float eps, int C, float *scale,
                                 float *shift, float *mean, float *invstd, void *stream)
{
    COVA_REQUIRE(gamma && beta && running_mean && running_var && scale && shift && C > 0);
    hipLaunchKernelGGL(bn_eval_params_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream,
                       gamma, beta, running_mean, running_var, eps, C, scale, shift, mean, invstd);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

// out = act(z*scale + shift (+ res))
COVA_API int cova_bn_act_fwd(const float *z, int ldz, const float *scale, const float *shift,
                             const float *res, int ldres, float *out, int ldo, long long R, int C,
                             int relu, void *stream)
{
    COVA_REQUIRE(z && scale && shift && out && R > 0 && C > 0);
    const bool v4 = (C % 4 == 0) && vec4_ok(z, ldz) && vec4_ok(res, ldres) && vec4_ok(out, ldo) &&
                    vec4_ok(scale, 0) && vec4_ok(shift, 0);
    if (v4)
        hipLaunchKernelGGL(bn_act_fwd_kernel<4>, dim3(ew_grid(R * (C / 4))), dim3(256), 0,
                           (hipStream_t)stream, z, ldz, scale, shift, res, ldres, out, ldo, R, C, relu);
    else
        hipLaunchKernelGGL(bn_act_fwd_kernel<1>, dim3(ew_grid(R * C)), dim3(256), 0,
                           (hipStream_t)stream, z, ldz, scale, shift, res, ldres, out, ldo, R, C, relu);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

// cova_bn_act_fwd (ReLU on) for a contiguous [R][64] map, also writing the ReLU decisions as bits [R][2] words
COVA_API int cova_bn_act_fwd_bits(const float *z, const float *scale, const float *shift, const float *res, float *out,
                                  uint32_t *bits, long long R, void *stream)
{
    COVA_REQUIRE(z && scale && shift && out && bits && R > 0);
    COVA_REQUIRE(vec4_ok(z, 64) && vec4_ok(res, 64) && vec4_ok(out, 64) && vec4_ok(scale, 0) && vec4_ok(shift, 0));
    hipLaunchKernelGGL(bn_act_fwd_bits_kernel<1>, dim3(ew_grid(R * 16)), dim3(256), 0, (hipStream_t)stream, z, scale, shift,
                       res, out, bits, R);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

// out = act(bn(z) + bn2(z2)), contiguous [R, C]
COVA_API int cova_bn_act2_fwd(const float *z, const float *scale, const float *shift, const float *z2,
                              const float *scale2, const float *shift2, float *out, long long R, int C,
                              int relu, void *stream)
{
    COVA_REQUIRE(z && scale && shift && z2 && scale2 && shift2 && out && R > 0 && C > 0 && C % 4 == 0);
    COVA_REQUIRE(vec4_ok(z, C) && vec4_ok(z2, C) && vec4_ok(out, C) && vec4_ok(scale, 0) && vec4_ok(shift, 0) &&
                 vec4_ok(scale2, 0) && vec4_ok(shift2, 0));
    hipLaunchKernelGGL(bn_act2_fwd_kernel, dim3(ew_grid(R * (C / 4))), dim3(256), 0, (hipStream_t)stream, z,
                       scale, shift, z2, scale2, shift2, out, R, C, relu);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

// BN backward, stage 1: partial [num_chunks][2][C] of (sum dy, sum dy*xhat), dy = dout*(act>0)
COVA_API int cova_bn_bwd_reduce(const float *dout, int ldd, const float *act, int lda,
                                const float *z, int ldz, const float *mean, const float *invstd,
                                long long R, int C, float *partial, void *stream)
{
    COVA_REQUIRE(dout && z && mean && invstd && partial && R > 0 && C > 0);
    const int rpc = cova_colreduce_rows_per_chunk(R, C);
    const dim3 grid((unsigned)cdivll(R, rpc), (unsigned)cdiv(C, pow2_cols_host(C)));
    if (C == 64 && ldd == 64 && ldz == 64 && (act == nullptr || lda == 64) && vec4_ok(dout, ldd) &&
        vec4_ok(z, ldz) && vec4_ok(act, lda)) {
        hipLaunchKernelGGL(colreduce64_kernel<1>, dim3(grid.x), dim3(256), 0, (hipStream_t)stream, dout,
                           act, z, mean, invstd, partial, R, rpc);
        COVA_LAUNCH_CHECK();
        return COVA_OK;
    }
    hipLaunchKernelGGL(colreduce_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, dout, ldd, act,
                       lda, z, ldz, mean, invstd, partial, R, C, rpc);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

// BN backward, stage 2: dgamma, dbeta (nullable) and coef [2][C] = (dbeta/n, dgamma/n)
COVA_API int cova_bn_finalize_bwd(const float *partial, int nparts, int C, double count,
                                  float *dgamma, float *dbeta, float *coef, void *stream)
{
    COVA_REQUIRE(partial && coef && nparts > 0 && C > 0);
    hipLaunchKernelGGL(bn_finalize_bwd_kernel, dim3(cdiv(C, 64)), dim3(1024), 0, (hipStream_t)stream,
                       partial, nparts, C, count, dgamma, dbeta, coef);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

// BN backward, stage 2 in affine form: dgamma, dbeta (nullable) and abc [3][C] with
// dz = abc[0]*dy + abc[1]*z + abc[2]  (applied on load by the *_pro convolutions)
COVA_API int cova_bn_finalize_bwd_abc(const float *partial, int nparts, int C, double count,
                                      float *dgamma, float *dbeta, const float *mean,
                                      const float *invstd, const float *scale, float *abc,
                                      void *stream)
{
    COVA_REQUIRE(partial && mean && invstd && scale && abc && nparts > 0 && C > 0);
    hipLaunchKernelGGL(bn_finalize_bwd_abc_kernel, dim3(cdiv(C, 64)), dim3(1024), 0,
                       (hipStream_t)stream, partial, nparts, C, count, dgamma, dbeta, mean, invstd,
                       scale, abc);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

// BN backward, stage 3: dz = scale*(dy - coef0 - xhat*coef1); optional dres = dy
COVA_API int cova_bn_bwd_apply(const float *dout, int ldd, const float *act, int lda, const float *z,
                               int ldz, const float *mean, const float *invstd, const float *scale,
                               const float *coef, float *dz, int lddz, float *dres, int lddres,
                               long long R, int C, void *stream)
{
    COVA_REQUIRE(dout && z && mean && invstd && scale && coef && dz && R > 0 && C > 0);
    const bool v4 = (C % 4 == 0) && vec4_ok(dout, ldd) && vec4_ok(act, lda) && vec4_ok(z, ldz) &&
                    vec4_ok(dz, lddz) && vec4_ok(dres, lddres);
    if (v4)
        hipLaunchKernelGGL(bn_bwd_apply_kernel<4>, dim3(ew_grid(R * (C / 4))), dim3(256), 0,
                           (hipStream_t)stream, dout, ldd, act, lda, z, ldz, mean, invstd, scale,
                           coef, dz, lddz, dres, lddres, R, C);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel<1>, dim3(ew_grid(R * C)), dim3(256), 0,
                           (hipStream_t)stream, dout, ldd, act, lda, z, ldz, mean, invstd, scale,
                           coef, dz, lddz, dres, lddres, R, C);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

// y NHWC [B,H1,W1,64] raw conv1 output -> out NHWC [B,H2,W2,64], idx uint8 same shape
// ymax (nullable): the raw y at each window's arg-max, [B,H2,W2,64]; with it the BatchNorm-backward
// sums of this layer can be taken in the epilogue of the convolution that produces the pooled
// gradient (mask = fma(scale, ymax, shift) > 0) instead of by cova_bn_relu_maxpool_bwd_reduce
COVA_API int cova_bn_relu_maxpool_fwd(const float *y, const float *scale, const float *shift,
                                      float *out, uint8_t *idx, float *ymax, int B, int H1, int W1,
                                      void *stream)
{
    COVA_REQUIRE(y && scale && shift && out && idx && B > 0 && H1 > 0 && W1 > 0);
    const int H2 = (H1 + 2 - 3) / 2 + 1, W2 = (W1 + 2 - 3) / 2 + 1;
    auto launch = [&](auto kern, int strip, bool capped) {
        const long long threads = (long long)B * cdiv(H2, strip) * W2 * 16;
        long long g = capped ? ew_grid(threads) : cdivll(threads, 256);
        if (g > 0x7fffffffll) g = 0x7fffffffll;                      // (the kernel strides over what is left)
        hipLaunchKernelGGL(kern, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, y, scale, shift, out, idx, ymax, B,
                           H1, W1, H2, W2);
    };
    // one output row per thread on XCD-contiguous work blocks: the fastest of 17 launch shapes measured in round 5
    launch(bn_relu_maxpool_fwd_kernel<1, false, true>, 1, false);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

COVA_API int cova_bn_relu_maxpool_bwd_num_partials(int B, int H1, int W1)
{
    long long g = cdivll((long long)B * H1 * W1, 16 * 64);
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

// stage 1: partial [num_partials][2][64]
COVA_API int cova_bn_relu_maxpool_bwd_reduce(const float *dp, const uint8_t *idx, const float *y,
                                             const float *scale, const float *shift,
                                             const float *mean, const float *invstd, float *partial,
                                             int B, int H1, int W1, void *stream)
{
    COVA_REQUIRE(dp && idx && y && scale && shift && mean && invstd && partial);
    const int H2 = (H1 + 2 - 3) / 2 + 1, W2 = (W1 + 2 - 3) / 2 + 1;
    const int grid = cova_bn_relu_maxpool_bwd_num_partials(B, H1, W1);
    hipLaunchKernelGGL(bn_relu_maxpool_bwd_reduce_win_kernel, dim3(grid), dim3(256), 0,
                       (hipStream_t)stream, dp, idx, y, scale, shift, mean, invstd, partial, B, H1, W1,
                       H2, W2);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

// stage 3 (after cova_bn_finalize_bwd): dz NHWC [B,H1,W1,64] = gradient w.r.t. conv1's output
COVA_API int cova_bn_relu_maxpool_bwd_apply(const float *dp, const uint8_t *idx, const float *y,
                                            const float *scale, const float *shift,
                                            const float *mean, const float *invstd,
                                            const float *coef, float *dz, int B, int H1, int W1,
                                            void *stream)
{
    COVA_REQUIRE(dp && idx && y && scale && shift && mean && invstd && coef && dz);
    const int H2 = (H1 + 2 - 3) / 2 + 1, W2 = (W1 + 2 - 3) / 2 + 1;
    const int grid = ew_grid((long long)B * H2 * W2 * 16);
    hipLaunchKernelGGL(bn_relu_maxpool_bwd_apply_quad_kernel, dim3(grid), dim3(256), 0,
                       (hipStream_t)stream, dp, idx, y, scale, shift, mean, invstd, coef, dz, B, H1, W1,
                       H2, W2);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}
