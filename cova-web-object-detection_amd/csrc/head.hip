// Decoder tail and training-step utilities: inverted dropout (models.py:84,88), the
// n_classes-wide last Linear (models.py:89) fwd/bwd, CrossEntropyLoss(reduction="sum") with its
// gradient and the per-box argmax (main.py:139, train.py:53,56), and the Adam update as
// configured at main.py:133-135 (L2 weight decay added to the gradient, bias-corrected).
#include "common.h"

namespace {

// out = x * keep / (1-p); keep is generated (and stored) unless given
__global__ __launch_bounds__(256) void dropout_fwd_kernel(const float *__restrict__ x, int ldx,
                                                          float *__restrict__ out, int ldo,
                                                          uint8_t *__restrict__ mask, long long R,
                                                          int C, float p, unsigned long long seed,
                                                          int given)
{
    const float inv = 1.f / (1.f - p);
    const long long total = R * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / C;
        const int c = (int)(i - r * C);
        uint8_t keep;
        if (given) keep = mask[i];
        else { keep = hash_uniform(seed, (unsigned long long)i) >= p ? 1 : 0; mask[i] = keep; }
        out[r * ldo + c] = keep ? x[r * ldx + c] * inv : 0.f;
    }
}

__global__ __launch_bounds__(256) void dropout_bwd_kernel(const float *__restrict__ g, int ldg,
                                                          const uint8_t *__restrict__ mask,
                                                          float *__restrict__ dx, int ldx,
                                                          long long R, int C, float p)
{
    const float inv = 1.f / (1.f - p);
    const long long total = R * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / C;
        const int c = (int)(i - r * C);
        dx[r * ldx + c] = mask[i] ? g[r * ldg + c] * inv : 0.f;
    }
}

constexpr int MAXNC = 16;

// y[n][k] = x[n] . W[k] + b[k]; one wave per row.  V = 4: a lane takes four adjacent input columns per step (one
// float4 of x and of every weight row: a 976-column row is four steps instead of sixteen dependent round trips).
template <int V>
__global__ __launch_bounds__(256) void linear_small_fwd_kernel(const float *__restrict__ x, int ldx,
                                                               const float *__restrict__ W,
                                                               const float *__restrict__ b,
                                                               float *__restrict__ y, int N, int Cin,
                                                               int NC)
{
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    float acc[MAXNC];
#pragma unroll
    for (int k = 0; k < MAXNC; ++k) acc[k] = 0.f;
    for (int c = lane * V; c < Cin; c += 64 * V) {
        float xv[V];
        if (V == 4) {
            const float4 t = *reinterpret_cast<const float4 *>(x + (size_t)n * ldx + c);
            xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w;
        } else {
            xv[0] = x[(size_t)n * ldx + c];
        }
#pragma unroll
        for (int k = 0; k < MAXNC; ++k)
            if (k < NC) {
                if (V == 4) {
                    const float4 w = *reinterpret_cast<const float4 *>(W + (size_t)k * Cin + c);
                    acc[k] += (xv[0] * w.x + xv[1] * w.y) + (xv[2] * w.z + xv[3] * w.w);
                } else {
                    acc[k] += xv[0] * W[(size_t)k * Cin + c];
                }
            }
    }
#pragma unroll
    for (int k = 0; k < MAXNC; ++k)
        if (k < NC) {
            const float t = wave_sum(acc[k]);
            if (lane == 0) y[(size_t)n * NC + k] = t + b[k];
        }
}

// dx[n][c] = sum_k dy[n][k] * W[k][c]: elements first, first + nthreads, ... (the blocks behind the weight-gradient
// blocks of linear_small_bwd_w_kernel: one launch for both gradients of the layer)
__device__ __forceinline__ void linear_small_bwd_x_body(const float *__restrict__ dy, const float *__restrict__ W,
                                                        float *__restrict__ dx, int ldx, int N, int Cin, int NC,
                                                        long long first, long long nthreads)
{
    const long long total = (long long)N * Cin;
    for (long long i = first; i < total; i += nthreads) {
        const int n = (int)(i / Cin), c = (int)(i - (long long)n * Cin);
        float a = 0.f;
        for (int k = 0; k < NC; ++k) a += dy[(size_t)n * NC + k] * W[(size_t)k * Cin + c];
        dx[(size_t)n * ldx + c] = a;
    }
}

// dW[k][c] = sum_n dy[n][k]*x[n][c]; block = 64 columns x 16 row slices; db by block 0
// V = 4: a thread owns 4 adjacent input columns (one float4 per row): 16 threads cover the block's 64 columns and a
// wave walks 4 row slices, the block 64 (V = 1: 16 slices, any alignment) -- see colsum_kernel.
// KN: compile-time bound of the class count (4: the reference's; MAXNC otherwise) -- the register budget of the row batches
template <int V, int KN>
__global__ __launch_bounds__(1024) void linear_small_bwd_w_kernel(const float *__restrict__ dy,
                                                                  const float *__restrict__ x, int ldx,
                                                                  float *__restrict__ dW,
                                                                  float *__restrict__ db, int N,
                                                                  int Cin, int NC, const float *__restrict__ Wt,
                                                                  float *__restrict__ dx, int lddx, int nwblocks)
{
    __shared__ float s[16][KN][64];
    if ((int)blockIdx.x >= nwblocks) {      // blocks behind the weight-gradient blocks: the input gradient
        linear_small_bwd_x_body(dy, Wt, dx, lddx, N, Cin, NC, (long long)(blockIdx.x - nwblocks) * 1024 + threadIdx.x,
                                (long long)(gridDim.x - nwblocks) * 1024);
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int TX = 64 / V, SPW = 64 / TX;               // column threads; row slices per wave
    const int tx = lane % TX, slice = wave * SPW + lane / TX;
    const int c = blockIdx.x * 64 + tx * V;
    float acc[KN][V];
#pragma unroll
    for (int k = 0; k < KN; ++k)
#pragma unroll
        for (int j = 0; j < V; ++j) acc[k][j] = 0.f;
    // Eight rows of the slice are requested before the first is used (unconditional loads at a clamped row, the value zeroed):
    // the launch has 16 blocks and ~22 rows per thread -- one row per round trip was 22 dependent memory latencies (27.8 us
    // at configs[1]).  Same rows in the same order per thread: bit-identical sums.
    constexpr int U = KN <= 4 ? 8 : 2, RSTEP = 16 * SPW;
    if (c < Cin)
        for (int n0 = slice; n0 < N; n0 += RSTEP * U) {
            float xv[U][4], dv[U][KN];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int n = n0 + u * RSTEP;
                const bool ok = n < N;
                const int nn = ok ? n : n0;
                if (V == 4) {
                    const float4 v = *reinterpret_cast<const float4 *>(x + (size_t)nn * ldx + c);
                    xv[u][0] = v.x; xv[u][1] = v.y; xv[u][2] = v.z; xv[u][3] = v.w;
                } else {
                    xv[u][0] = x[(size_t)nn * ldx + c];
                }
#pragma unroll
                for (int k = 0; k < KN; ++k) dv[u][k] = (k < NC) ? dy[(size_t)nn * NC + k] : 0.f;
                if (!ok) {              // padding slot: both factors zero (0 * x of a clamped row would be NaN where that x is inf)
#pragma unroll
                    for (int k = 0; k < KN; ++k) dv[u][k] = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) xv[u][j] = 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int k = 0; k < KN; ++k)
                    if (k < NC) {
#pragma unroll
                        for (int j = 0; j < V; ++j) acc[k][j] += dv[u][k] * xv[u][j];
                    }
        }
#pragma unroll
    for (int k = 0; k < KN; ++k)
#pragma unroll
        for (int j = 0; j < V; ++j) {
            float v = acc[k][j];
            if (V == 4) {                                   // the wave's 4 row slices
                v += __shfl_xor(v, 16, 64);
                v += __shfl_xor(v, 32, 64);
            }
            if (lane < TX) s[wave][k][tx * V + j] = v;
        }
    __syncthreads();
    if (wave == 0 && blockIdx.x * 64 + lane < Cin)
        for (int k = 0; k < NC; ++k) {
            float t = 0.f;
            for (int j = 0; j < 16; ++j) t += s[j][k][lane];
            dW[(size_t)k * Cin + blockIdx.x * 64 + lane] = t;
        }
    if (blockIdx.x == 0) {                  // db[k] = sum_n dy[n][k]: all 1024 threads, LDS tree
        __syncthreads();
        float pk[KN];
#pragma unroll
        for (int k = 0; k < KN; ++k) pk[k] = 0.f;
        for (int n = threadIdx.x; n < N; n += 1024)
#pragma unroll
            for (int k = 0; k < KN; ++k)
                if (k < NC) pk[k] += dy[(size_t)n * NC + k];
#pragma unroll
        for (int k = 0; k < KN; ++k) s[wave][k][lane] = pk[k];
        __syncthreads();
        if (wave == 0)
            for (int k = 0; k < NC; ++k) {
                float t = 0.f;
                for (int j = 0; j < 16; ++j) t += s[j][k][lane];
                s[0][k][lane] = t;
            }
        __syncthreads();
        if (threadIdx.x < NC) {
            float t = 0.f;
            for (int j = 0; j < 64; ++j) t += s[0][threadIdx.x][j];
            db[threadIdx.x] = t;
        }
    }
}

// loss = sum_n (logsumexp(l_n) - l_n[label]); dl = gscale*(softmax - onehot); pred = argmax
__global__ __launch_bounds__(1024) void ce_sum_kernel(const float *__restrict__ logits,
                                                      const int64_t *__restrict__ labels, int N,
                                                      int NC, float gscale, float *__restrict__ loss,
                                                      float *__restrict__ dlogits,
                                                      int64_t *__restrict__ pred)
{
    __shared__ double s_loss[1024];
    double local = 0.0;
    for (int n = threadIdx.x; n < N; n += 1024) {
        const float *l = logits + (size_t)n * NC;
        float m = l[0];
        int am = 0;
        for (int k = 1; k < NC; ++k)
            if (l[k] > m) { m = l[k]; am = k; }
        float se = 0.f;
        for (int k = 0; k < NC; ++k) se += expf(l[k] - m);
        const float lse = m + logf(se);
        if (pred) pred[n] = am;
        if (labels) {
            const int lab = (int)labels[n];
            local += (double)(lse - l[lab]);
            if (dlogits)
                for (int k = 0; k < NC; ++k)
                    dlogits[(size_t)n * NC + k] = gscale * (expf(l[k] - lse) - (k == lab ? 1.f : 0.f));
        }
    }
    s_loss[threadIdx.x] = local;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (threadIdx.x < o) s_loss[threadIdx.x] += s_loss[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0 && loss) loss[0] = (float)s_loss[0];
}

// torch.optim.Adam (non-amsgrad) single-tensor math on a flat parameter buffer
__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, const float *__restrict__ g,
                                                   float *__restrict__ m, float *__restrict__ v,
                                                   long long n, float lr, float beta1, float beta2,
                                                   float eps, float weight_decay, float bc1,
                                                   float bc2_sqrt)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (long long)gridDim.x * blockDim.x) {
        float pv = p[i];
        float gv = g[i] + weight_decay * pv;
        const float mv = beta1 * m[i] + (1.f - beta1) * gv;
        const float vv = beta2 * v[i] + (1.f - beta2) * gv * gv;
        m[i] = mv;
        v[i] = vv;
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        p[i] = pv - (lr / bc1) * (mv / denom);
    }
}

// column sums of x [R, C] -> out [C] (deterministic; small R); 64 columns x 16 row slices
// V = 4: a thread owns 4 adjacent columns (one float4 per row), so a block covers 64 columns with 16 threads and
// walks 64 row slices in parallel (the matrices here are ~1.4k x 1k: the launch is latency bound, rows in flight
// per block are what counts); V = 1: any alignment.
template <int V>
__global__ __launch_bounds__(1024) void colsum_kernel(const float *__restrict__ x, int ldx, int R,
                                                      int C, float *__restrict__ out)
{
    constexpr int TX = 64 / V, NS = 1024 / TX;
    __shared__ float s[NS][64];
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int c = blockIdx.x * 64 + tx * V;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < C)
        for (int r = ty; r < R; r += NS) {
            if (V == 4) {
                const float4 v = *reinterpret_cast<const float4 *>(x + (size_t)r * ldx + c);
                a[0] += v.x; a[1] += v.y; a[2] += v.z; a[3] += v.w;
            } else {
                a[0] += x[(size_t)r * ldx + c];
            }
        }
#pragma unroll
    for (int j = 0; j < V; ++j) s[ty][tx * V + j] = a[j];
    __syncthreads();
    if (threadIdx.x < 64 && blockIdx.x * 64 + threadIdx.x < C) {
        float t = 0.f;
        for (int j = 0; j < NS; ++j) t += s[j][threadIdx.x];
        out[blockIdx.x * 64 + threadIdx.x] = t;
    }
}

// Evaluation decision of train.py:131-154: for every page and every class column, the (page-local)
// indices of the k boxes with the highest score, best first.  One wave per (page, class); ties go
// to the lower box index.  page_start [n_pages+1] = box offsets of the pages in the flat batch.
__global__ __launch_bounds__(256) void page_class_topk_kernel(const float *__restrict__ logits,
                                                              const int64_t *__restrict__ page_start,
                                                              int n_pages, int NC, int k,
                                                              int64_t *__restrict__ out)
{
    const int task = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (task >= n_pages * NC) return;
    const int page = task / NC, c = task - page * NC;
    const int lo = (int)page_start[page], hi = (int)page_start[page + 1];
    float prev_v = INFINITY;
    int prev_i = -1;
    for (int j = 0; j < k; ++j) {
        // best (value, lowest index) strictly after (prev_v, prev_i) in the order (v desc, i asc)
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int n = lo + lane; n < hi; n += 64) {
            const float v = logits[(size_t)n * NC + c];
            const int i = n - lo;
            const bool after = (v < prev_v) || (v == prev_v && i > prev_i);
            if (after && (v > bv || (v == bv && i < bi))) { bv = v; bi = i; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (lane == 0) out[((size_t)page * NC + c) * k + j] = (bi == 0x7fffffff) ? -1 : bi;
        prev_v = bv;
        prev_i = bi;
    }
}

inline int ew_grid(long long total)
{
    long long g = cdivll(total, 256);
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

COVA_API int cova_dropout_fwd(const float *x, int ldx, float *out, int ldo, uint8_t *mask,
                              long long R, int C, float p, unsigned long long seed, int mask_given,
                              void *stream)
{
    COVA_REQUIRE(x && out && mask && R > 0 && C > 0 && p >= 0.f && p < 1.f);
    hipLaunchKernelGGL(dropout_fwd_kernel, dim3(ew_grid(R * C)), dim3(256), 0, (hipStream_t)stream, x,
                       ldx, out, ldo, mask, R, C, p, seed, mask_given);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

COVA_API int cova_dropout_bwd(const float *g, int ldg, const uint8_t *mask, float *dx, int ldx,
                              long long R, int C, float p, void *stream)
{
    COVA_REQUIRE(g && mask && dx && R > 0 && C > 0 && p >= 0.f && p < 1.f);
    hipLaunchKernelGGL(dropout_bwd_kernel, dim3(ew_grid(R * C)), dim3(256), 0, (hipStream_t)stream, g,
                       ldg, mask, dx, ldx, R, C, p);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

COVA_API int cova_linear_small_fwd(const float *x, int ldx, const float *W, const float *b, float *y,
                                   int N, int Cin, int NC, void *stream)
{
    COVA_REQUIRE(x && W && b && y && NC > 0 && NC <= MAXNC && Cin > 0);
    if (N == 0) return COVA_OK;
    const bool v4 = (Cin % 4 == 0) && (ldx % 4 == 0) && (((uintptr_t)x & 15) == 0) && (((uintptr_t)W & 15) == 0);
    hipLaunchKernelGGL((v4 ? linear_small_fwd_kernel<4> : linear_small_fwd_kernel<1>), dim3(cdiv(N, 4)), dim3(256), 0,
                       (hipStream_t)stream, x, ldx, W, b, y, N, Cin, NC);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

COVA_API int cova_linear_small_bwd(const float *dy, const float *x, int ldx, const float *W,
                                   float *dx, int lddx, float *dW, float *db, int N, int Cin, int NC,
                                   void *stream)
{
    COVA_REQUIRE(dy && x && W && dx && dW && db && NC > 0 && NC <= MAXNC && Cin > 0 && N > 0);
    hipStream_t st = (hipStream_t)stream;
    const bool v4 = (Cin % 4 == 0) && (ldx % 4 == 0) && (((uintptr_t)x & 15) == 0);
    // one launch: cdiv(Cin, 64) weight-gradient blocks, then the blocks of the input gradient (same arithmetic per element
    // as the two launches this replaces: bit-identical results)
    const int nw = cdiv(Cin, 64);
    long long nx = cdivll((long long)N * Cin, 1024);
    if (nx > 1024) nx = 1024;
    const dim3 wgrid(nw + (int)nx), wblock(1024);
    if (NC <= 4) hipLaunchKernelGGL((v4 ? linear_small_bwd_w_kernel<4, 4> : linear_small_bwd_w_kernel<1, 4>), wgrid, wblock, 0, st, dy, x, ldx, dW, db, N, Cin, NC, W, dx, lddx, nw);
    else hipLaunchKernelGGL((v4 ? linear_small_bwd_w_kernel<4, MAXNC> : linear_small_bwd_w_kernel<1, MAXNC>), wgrid, wblock, 0, st, dy, x, ldx, dW, db, N, Cin, NC, W, dx, lddx, nw);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

// labels / loss / dlogits / pred are each optional (NULL): pred-only = eval argmax
COVA_API int cova_ce_sum(const float *logits, const int64_t *labels, int N, int NC, float gscale,
                         float *loss, float *dlogits, int64_t *pred, void *stream)
{
    COVA_REQUIRE(logits && N > 0 && NC > 0);
    hipLaunchKernelGGL(ce_sum_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, logits, labels, N,
                       NC, gscale, loss, dlogits, pred);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

COVA_API int cova_adam_step(float *p, const float *g, float *m, float *v, long long n, int step,
                            double lr, double beta1, double beta2, double eps, double weight_decay,
                            void *stream)
{
    COVA_REQUIRE(p && g && m && v && n > 0 && step >= 1);
    const float bc1 = (float)(1.0 - pow(beta1, (double)step));
    const float bc2_sqrt = (float)sqrt(1.0 - pow(beta2, (double)step));
    hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n,
                       (float)lr, (float)beta1, (float)beta2, (float)eps, (float)weight_decay, bc1,
                       bc2_sqrt);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

COVA_API int cova_colsum(const float *x, int ldx, int R, int C, float *out, void *stream)
{
    COVA_REQUIRE(x && out && R > 0 && C > 0);
    const bool v4 = (C % 4 == 0) && (ldx % 4 == 0) && (((uintptr_t)x & 15) == 0);
    hipLaunchKernelGGL((v4 ? colsum_kernel<4> : colsum_kernel<1>), dim3(cdiv(C, 64)), dim3(1024), 0, (hipStream_t)stream, x, ldx, R,
                       C, out);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

// out [n_pages, NC, k] int64: page-local box indices of the k highest logits per class, best first
// (-1 when the page has fewer than k boxes).  replaces the per-page python loop + argsort of
// train.py:131-153.
COVA_API int cova_page_class_topk(const float *logits, const int64_t *page_start, int n_pages, int NC,
                                  int k, int64_t *out, void *stream)
{
    COVA_REQUIRE(logits && page_start && out && n_pages > 0 && NC > 0 && k > 0);
    hipLaunchKernelGGL(page_class_topk_kernel, dim3(cdiv(n_pages * NC, 4)), dim3(256), 0,
                       (hipStream_t)stream, logits, page_start, n_pages, NC, k, out);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

// ------------------------------------------------------------------------------------
// Attention export (extract_attn_wts_and_visualize.py:104-135): one row per box,
//   [x, y, w, h, label, K x (x, y, w, h) of the context boxes (zeros for -1 pads), K attention weights]
// The caller keeps the rows with label > 0 and writes them with np.savetxt(fmt="%.3f").
namespace {
__global__ __launch_bounds__(256) void attn_export_rows_kernel(
    const float *__restrict__ bboxes, const long long *__restrict__ ctx,
    const float *__restrict__ attn, const long long *__restrict__ labels, int N, int K,
    float *__restrict__ out)
{
    const int width = 5 + 5 * K;
    const long long total = (long long)N * width;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(t / width), c = (int)(t - (long long)i * width);
        float v;
        if (c < 4) {
            const float *b = bboxes + (size_t)i * 5 + 1;
            v = c < 2 ? b[c] : b[c] - b[c - 2];
        } else if (c == 4) {
            v = (float)labels[i];
        } else if (c < 5 + 4 * K) {
            const int k = (c - 5) >> 2, j = (c - 5) & 3;
            const long long n = ctx[(size_t)i * K + k];
            if (n < 0 || n >= N) {          // pads; out-of-range ids are treated like the GAT kernels treat them
                v = 0.f;
            } else {
                const float *b = bboxes + (size_t)n * 5 + 1;
                v = j < 2 ? b[j] : b[j] - b[j - 2];
            }
        } else {
            v = attn[(size_t)i * K + (c - 5 - 4 * K)];
        }
        out[t] = v;
    }
}
}  // namespace

// out [N, 5 + 5K]
COVA_API int cova_attn_export_rows(const float *bboxes, const long long *ctx, const float *attn,
                                   const long long *labels, int N, int K, float *out, void *stream)
{
    COVA_REQUIRE(bboxes && ctx && attn && labels && out && N >= 0 && K > 0);
    if (N == 0) return COVA_OK;
    long long g = ((long long)N * (5 + 5 * K) + 255) / 256;
    if (g > 8192) g = 8192;
    hipLaunchKernelGGL(attn_export_rows_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream,
                       bboxes, ctx, attn, labels, N, K, out);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}
