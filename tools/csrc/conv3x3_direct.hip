// Direct-form (implicit GEMM) 3x3 / 64->64 convolution kernels: forward / data gradient with the fused
// statistics and BatchNorm-backward epilogues, and the weight gradient.  They were the product path of round 1
// before the Winograd kernels (cova-web-object-detection_amd/csrc/conv_wino.hip, conv_wino4.hip) replaced them; they
// live here, in the TOOLS library (tools/lib/libcova_direct.so, tools/direct_lib.py), as the measured reference
// point of DESIGN.md section 4.1 (133 TFLOP/s = 0.85 of the f32-MFMA peak) and for tools/conv_bench.py.
#include "../../cova-web-object-detection_amd/csrc/common.h"

namespace {

// ------------------------------------------------------------------------------------
// shared epilogue: one wave owns a 32(pixel) x 64(channel) output strip held in two
// 32x32 accumulators (channel blocks 0 and 1).
// ------------------------------------------------------------------------------------
// BN-backward fusion (dgrad epilogue): the conv result is dL/d(post-ReLU activation); with
// bn.act != nullptr the stored value becomes dy = result * (act > 0) and the accumulated sums are
// (sum dy, sum dy * xhat), xhat = (z - mean) * invstd -- exactly what cova_bn_bwd_reduce computes,
// without its three extra passes over HBM.
struct BnBwdEpi {
    const float *act, *z, *mean, *invstd;
};

__device__ __forceinline__ void epilogue_store_stats(const f32x16 &acc0, const f32x16 &acc1,
                                                     float *__restrict__ out,
                                                     const float *__restrict__ addend,
                                                     size_t pix_row_base, int x0, int W,
                                                     bool row_valid, int lane, float &s0,
                                                     float &s1, float &q0, float &q1,
                                                     const BnBwdEpi bn = BnBwdEpi{nullptr, nullptr,
                                                                                  nullptr, nullptr})
{
    const int li = lane & 31;
    s0 = s1 = q0 = q1 = 0.f;
    float mu0 = 0.f, mu1 = 0.f, is0 = 0.f, is1 = 0.f;
    if (bn.act != nullptr) {
        mu0 = bn.mean[li]; mu1 = bn.mean[32 + li];
        is0 = bn.invstd[li]; is1 = bn.invstd[32 + li];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int px = mfma32_row(r, lane);
        const bool valid = row_valid && (x0 + px) < W;
        if (valid) {
            const size_t o = (pix_row_base + (size_t)px) * 64;
            float v0 = acc0[r], v1 = acc1[r];
            if (addend != nullptr) {
                v0 += addend[o + li];
                v1 += addend[o + 32 + li];
            }
            if (bn.act != nullptr) {
                if (!(bn.act[o + li] > 0.f)) v0 = 0.f;
                if (!(bn.act[o + 32 + li] > 0.f)) v1 = 0.f;
                const float xh0 = (bn.z[o + li] - mu0) * is0;
                const float xh1 = (bn.z[o + 32 + li] - mu1) * is1;
                out[o + li] = v0;
                out[o + 32 + li] = v1;
                s0 += v0; q0 += v0 * xh0;
                s1 += v1; q1 += v1 * xh1;
            } else {
                out[o + li] = v0;
                out[o + 32 + li] = v1;
                s0 += v0; q0 += v0 * v0;
                s1 += v1; q1 += v1 * v1;
            }
        }
    }
    s0 += __shfl_xor(s0, 32, 64);
    s1 += __shfl_xor(s1, 32, 64);
    q0 += __shfl_xor(q0, 32, 64);
    q1 += __shfl_xor(q1, 32, 64);
}

// Block-level reduction of the per-wave channel sums into stat_part[bid][2][64].
__device__ __forceinline__ void block_stats_reduce(float *s_red, float *__restrict__ stat_part,
                                                   int bid, int tid, int lane, int wave,
                                                   int nwaves, float s0, float s1, float q0,
                                                   float q1)
{
    if (lane < 32) {
        s_red[wave * 128 + lane] = s0;
        s_red[wave * 128 + 32 + lane] = s1;
        s_red[wave * 128 + 64 + lane] = q0;
        s_red[wave * 128 + 96 + lane] = q1;
    }
    __syncthreads();
    if (tid < 128) {
        float t = 0.f;
        for (int w = 0; w < nwaves; ++w) t += s_red[w * 128 + tid];
        stat_part[(size_t)bid * 128 + tid] = t;
    }
}

// ------------------------------------------------------------------------------------
// conv3x3, 64 -> 64, NHWC, stride 1, pad 1
// ------------------------------------------------------------------------------------
namespace c3 {
constexpr int TH = 8, TW = 32;            // output tile: 8 rows x 32 cols = 256 pixels
constexpr int PH = TH + 2, PW = TW + 2;   // input tile with halo
constexpr int PSTR = 68;                  // floats per staged pixel (64 + 4 pad: b128 reads
                                          // of 32 different pixels hit 16 distinct 16-B slots)
constexpr int IN_FLOATS = PH * PW * PSTR; // 23,120 floats = 92,480 B
constexpr int W_FLOATS = 64 * PSTR;       // one tap: 64 co rows x 68 = 17,408 B
constexpr int LDS_FLOATS = IN_FLOATS + 2 * W_FLOATS;   // 127,296 B
constexpr int THREADS = 512;              // 8 waves; wave w owns output row w of the tile
}  // namespace c3

// ------------------------------------------------------------------------------------
// conv3x3 v2: persistent blocks, software-pipelined.
//   * one block per CU walks tiles t = blockIdx.x, +gridDim.x, ...; while tile t computes, the
//     halo'd input of tile t+grid is already in flight into registers (11 float4 per thread) and
//     is written to LDS after the last tap -- global latency and most of the staging time
//     disappear behind the MFMAs;
//   * next-tap weights are fetched at the top of a tap and written to the other LDS buffer at
//     its end (buffer parity runs across tiles: 9 taps per tile is odd);
//   * LDS->register operand fragments are double buffered: step s+1 is read before the eight
//     MFMAs of step s issue.
// ------------------------------------------------------------------------------------
namespace c3 {
constexpr int NPRE = (PH * PW * 16 + THREADS - 1) / THREADS;     // 11 float4 per thread
constexpr int RED_FLOATS = 8 * 128;                               // stats scratch
}

template <bool STATS>
__global__ __launch_bounds__(c3::THREADS) void conv3x3_c64_v2_kernel(
    const float *__restrict__ in, const float *__restrict__ wt, const float *__restrict__ addend,
    float *__restrict__ out, float *__restrict__ stat_part, int H, int W, int tiles_x, int tiles_y,
    int ntiles, const BnBwdEpi bn, int abl_arg)
{
    const int abl = COVA_ABL(abl_arg);
    // abl: ablation mask for tools/conv_bench.py (0 in production): 1 no epilogue, 2 no LDS refill,
    // 4 no input prefetch loads, 8 no weight restaging, 16 no per-tap barrier
    using namespace c3;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS + RED_FLOATS];
    float *s_in = lds;
    float *s_w = lds + IN_FLOATS;
    float *s_red = lds + LDS_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh2 = lane >> 5;
    const int pco = tid >> 4, pc4 = tid & 15;
    // XCD-aware order: workgroup b runs on XCD b % 8 (observed dispatch order, speed only), so give
    // each XCD a contiguous run of tiles per round -- neighbouring tiles then share halo rows in
    // one L2 instead of fetching them through eight.
    int tile = blockIdx.x;
    if ((gridDim.x & 7) == 0) tile = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (tile >= ntiles) return;

    float4 pre[NPRE];
    // slot `it` of the halo'd input tile of tile t (tile coordinates passed in as scalars)
    auto issue_slot = [&](int it, const float *in_b, int ty, int tx) {
        const int idx = tid + it * THREADS;
        const int px = idx >> 4, c4 = idx & 15;
        const int r = px / PW, c = px - r * PW;
        const int gy = ty * TH + r - 1, gx = tx * TW + c - 1;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < PH * PW * 16 && gy >= 0 && gy < H && gx >= 0 && gx < W)
            v = *reinterpret_cast<const float4 *>(in_b + ((size_t)gy * W + gx) * 64 + c4 * 4);
        pre[it] = v;
    };
    auto issue_tile_loads = [&](int t) {
        const int tx = t % tiles_x;
        const int ty = (t / tiles_x) % tiles_y;
        const int b = t / (tiles_x * tiles_y);
        const float *in_b = in + (size_t)b * H * W * 64;
#pragma unroll
        for (int it = 0; it < NPRE; ++it) issue_slot(it, in_b, ty, tx);
    };
    auto write_tile_lds = [&]() {
#pragma unroll
        for (int it = 0; it < NPRE; ++it) {
            const int idx = tid + it * THREADS;
            if (idx < PH * PW * 16)
                *reinterpret_cast<float4 *>(s_in + (idx >> 4) * PSTR + (idx & 15) * 4) = pre[it];
        }
    };

    issue_tile_loads(tile);
    {
        const float4 w0 = *reinterpret_cast<const float4 *>(wt + pco * 64 + pc4 * 4);
        const float4 w1 = *reinterpret_cast<const float4 *>(wt + (pco + 32) * 64 + pc4 * 4);
        *reinterpret_cast<float4 *>(s_w + pco * PSTR + pc4 * 4) = w0;
        *reinterpret_cast<float4 *>(s_w + (pco + 32) * PSTR + pc4 * 4) = w1;
    }
    write_tile_lds();
    __syncthreads();

    int wbuf = 0;      // LDS buffer holding the current tap's weights
    for (; tile < ntiles; tile += gridDim.x) {
        const int tx = tile % tiles_x;
        const int ty = (tile / tiles_x) % tiles_y;
        const int b = tile / (tiles_x * tiles_y);
        const int y0 = ty * TH, x0 = tx * TW;
        const int next = tile + gridDim.x;
        const bool has_next = next < ntiles;
        // The next tile's input is fetched in slices spread over taps 0..5 (two slots per tap, issued
        // AFTER that tap's weight loads): HBM sees a steady stream instead of a 23 MB burst from all
        // CUs at once, and the end-of-tap wait for the weights (vmcnt is FIFO) never has to wait for
        // input slots younger than one tap.
        const int ntx = next % tiles_x, nty = (next / tiles_x) % tiles_y;
        const float *nin_b = in + (size_t)(next / (tiles_x * tiles_y)) * H * W * 64;

        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }

        const float *a_tile = s_in + (wave * PW + li) * PSTR + kh2 * 4;
        float4 a_cur = *reinterpret_cast<const float4 *>(a_tile);      // tap 0, step 0
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // next tap's weights (tap 8: tap 0 of the next tile) -> registers, early
            const int ntap = tap == 8 ? 0 : tap + 1;
            const float *wsrc = wt + (size_t)ntap * 4096;
            float4 wn0 = make_float4(0, 0, 0, 0), wn1 = wn0;
            if (!(abl & 8)) {
                wn0 = *reinterpret_cast<const float4 *>(wsrc + pco * 64 + pc4 * 4);
                wn1 = *reinterpret_cast<const float4 *>(wsrc + (pco + 32) * 64 + pc4 * 4);
            }
            if (has_next && !(abl & 4)) {
                if (2 * tap < NPRE) issue_slot(2 * tap, nin_b, nty, ntx);
                if (2 * tap + 1 < NPRE) issue_slot(2 * tap + 1, nin_b, nty, ntx);
            }
            __builtin_amdgcn_sched_barrier(0);
            const int kh = tap / 3, kw = tap - kh * 3;
            const float *a_base = a_tile + (kh * PW + kw) * PSTR;
            const float *b_base = s_w + wbuf * W_FLOATS + li * PSTR + kh2 * 4;
            float4 b0_cur = *reinterpret_cast<const float4 *>(b_base);
            float4 b1_cur = *reinterpret_cast<const float4 *>(b_base + 32 * PSTR);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                float4 a_nxt, b0_nxt, b1_nxt;
                if (s < 7) {
                    a_nxt = *reinterpret_cast<const float4 *>(a_base + (s + 1) * 8);
                    b0_nxt = *reinterpret_cast<const float4 *>(b_base + (s + 1) * 8);
                    b1_nxt = *reinterpret_cast<const float4 *>(b_base + 32 * PSTR + (s + 1) * 8);
                } else {
                    // first A fragment of the next tap (input tile: independent of the barrier)
                    const int t2 = tap == 8 ? 0 : tap + 1;
                    const int kh_n = t2 / 3, kw_n = t2 - kh_n * 3;
                    a_nxt = *reinterpret_cast<const float4 *>(a_tile + (kh_n * PW + kw_n) * PSTR);
                    b0_nxt = b0_cur;
                    b1_nxt = b1_cur;
                }
                acc0 = mfma32(a_cur.x, b0_cur.x, acc0);
                acc1 = mfma32(a_cur.x, b1_cur.x, acc1);
                acc0 = mfma32(a_cur.y, b0_cur.y, acc0);
                acc1 = mfma32(a_cur.y, b1_cur.y, acc1);
                acc0 = mfma32(a_cur.z, b0_cur.z, acc0);
                acc1 = mfma32(a_cur.z, b1_cur.z, acc1);
                acc0 = mfma32(a_cur.w, b0_cur.w, acc0);
                acc1 = mfma32(a_cur.w, b1_cur.w, acc1);
                a_cur = a_nxt;
                b0_cur = b0_nxt;
                b1_cur = b1_nxt;
            }
            if (!(abl & 8)) {
                float *wdst = s_w + (wbuf ^ 1) * W_FLOATS;
                *reinterpret_cast<float4 *>(wdst + pco * PSTR + pc4 * 4) = wn0;
                *reinterpret_cast<float4 *>(wdst + (pco + 32) * PSTR + pc4 * 4) = wn1;
                wbuf ^= 1;
            }
            if (!(abl & 16)) __syncthreads();
        }

        const int oy = y0 + wave;
        float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
        if (abl & 1) {
            asm volatile("" ::"v"(acc0), "v"(acc1));
        } else {
            epilogue_store_stats(acc0, acc1, out, addend, ((size_t)b * H + oy) * W + x0, x0, W,
                                 oy < H, lane, s0, s1, q0, q1, bn);
        }
        if (abl & 16) __syncthreads();
        // every wave has finished reading s_in (barrier after tap 8): refill it for the next tile
        if (has_next && !(abl & 2)) write_tile_lds();
        if (STATS) block_stats_reduce(s_red, stat_part, tile, tid, lane, wave, 8, s0, s1, q0, q1);
        __syncthreads();
    }
}

__global__ void prep_w3x3_kernel(const float *__restrict__ w, float *__restrict__ w_fwd,
                                 float *__restrict__ w_dgrad)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // over [tap][a][b]
    if (idx >= 9 * 64 * 64) return;
    const int tap = idx >> 12, a = (idx >> 6) & 63, bb = idx & 63;
    const int kh = tap / 3, kw = tap % 3;
    // forward: w_fwd[tap][co=a][ci=b] = w[co][ci][kh][kw]
    w_fwd[idx] = w[((a * 64 + bb) * 3 + kh) * 3 + kw];
    // dgrad: dx[p][ci] = sum_{tap',co} dz[p+off(tap')][co] * w[co][ci][2-kh'][2-kw']
    w_dgrad[idx] = w[((bb * 64 + a) * 3 + (2 - kh)) * 3 + (2 - kw)];
}

// ------------------------------------------------------------------------------------
// conv3x3 weight gradient: dW[tap][co][ci] = sum_p dz[p][co] * a[p+off(tap)][ci]
// GEMM M = co, N = ci (per tap), K = pixels; persistent blocks, partial results per block.
// 12 waves per block.  Wave w owns tap row kh = w/4 (3 taps), output-
// channel block w&1 and input-channel block (w>>1)&1 over ALL pixels of the tile: 3 accumulators
// (48 registers) instead of 9, no duplicated partials, and three waves per SIMD to hide LDS
// latency.  Tiles are 8x32 pixels; the next tile's halo'd activation tile and dz tile are
// prefetched into registers while the current tile computes and are written to LDS afterwards.
// ------------------------------------------------------------------------------------
namespace wg3v2 {
constexpr int TH = 8, TW = 32;
constexpr int PH = TH + 2, PW = TW + 2;
constexpr int A_FLOATS = PH * PW * 64;      // 21,760 floats = 87,040 B
constexpr int DZ_FLOATS = TH * TW * 64;     // 16,384 floats = 65,536 B
constexpr int THREADS = 768;
constexpr int NPRE_A = (PH * PW * 16 + THREADS - 1) / THREADS;   // 8 float4
constexpr int NPRE_D = (TH * TW * 16 + THREADS - 1) / THREADS;   // 6 float4
}  // namespace wg3v2

__global__ __launch_bounds__(wg3v2::THREADS) void conv3x3_wgrad_v2_kernel(
    const float *__restrict__ act, const float *__restrict__ dz, float *__restrict__ part,
    int H, int W, int tiles_x, int tiles_y, int ntiles, int abl_arg)
{
    const int abl = COVA_ABL(abl_arg);
    using namespace wg3v2;
    __shared__ __attribute__((aligned(16))) float lds[A_FLOATS + DZ_FLOATS];
    float *s_a = lds;
    float *s_dz = lds + A_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh2 = lane >> 5;
    const int cob = wave & 1, cib = (wave >> 1) & 1, trow = wave >> 2;

    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    float4 pa[NPRE_A], pd[NPRE_D];
    // slot s < NPRE_A: activation halo tile; s >= NPRE_A: dz tile (tile origin y0, x0 of batch b)
    auto issue_slot = [&](int s, const float *act_b, const float *dz_b, int y0, int x0) {
        if (s < NPRE_A) {
            const int idx = tid + s * THREADS;
            const int px = idx >> 4, c4 = idx & 15;
            const int r = px / PW, c = px - r * PW;
            const int gy = y0 + r - 1, gx = x0 + c - 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < PH * PW * 16 && gy >= 0 && gy < H && gx >= 0 && gx < W)
                v = *reinterpret_cast<const float4 *>(act_b + ((size_t)gy * W + gx) * 64 + c4 * 4);
            pa[s] = v;
        } else {
            const int it = s - NPRE_A;
            const int idx = tid + it * THREADS;
            const int px = idx >> 4, c4 = idx & 15;
            const int r = px / TW, c = px - r * TW;
            const int gy = y0 + r, gx = x0 + c;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < TH * TW * 16 && gy < H && gx < W)
                v = *reinterpret_cast<const float4 *>(dz_b + ((size_t)gy * W + gx) * 64 + c4 * 4);
            pd[it] = v;
        }
    };
    auto issue_loads = [&](int t) {
        const int tx = t % tiles_x;
        const int ty = (t / tiles_x) % tiles_y;
        const int b = t / (tiles_x * tiles_y);
#pragma unroll
        for (int s = 0; s < NPRE_A + NPRE_D; ++s)
            issue_slot(s, act + (size_t)b * H * W * 64, dz + (size_t)b * H * W * 64, ty * TH, tx * TW);
    };
    auto write_lds = [&]() {
#pragma unroll
        for (int it = 0; it < NPRE_A; ++it) {
            const int idx = tid + it * THREADS;
            if (idx < PH * PW * 16) *reinterpret_cast<float4 *>(s_a + idx * 4) = pa[it];
        }
#pragma unroll
        for (int it = 0; it < NPRE_D; ++it) {
            const int idx = tid + it * THREADS;
            if (idx < TH * TW * 16) *reinterpret_cast<float4 *>(s_dz + idx * 4) = pd[it];
        }
    };

    int tile = blockIdx.x;
    if (tile < ntiles) {
        issue_loads(tile);
        write_lds();
    }
    __syncthreads();
    for (; tile < ntiles; tile += gridDim.x) {
        const int next = tile + gridDim.x;
        const bool has_next = next < ntiles;
        // next tile's coordinates; its 14 prefetch slots are issued one per 8 k-pairs inside the
        // MFMA loop, so the address arithmetic and the loads overlap with the matrix pipe instead
        // of forming a load-issue phase at the head of every tile
        const int ntx = next % tiles_x, nty = (next / tiles_x) % tiles_y;
        const int nb = next / (tiles_x * tiles_y);
        const float *nact = act + (size_t)nb * H * W * 64;
        const float *ndz = dz + (size_t)nb * H * W * 64;
        // k-pair t covers tile pixels 2t, 2t+1 (row = p>>5, col = p&31); this lane: p = 2t + kh2
        const float *a_ptr = s_dz + kh2 * 64 + cob * 32 + li;
        const float *b_ptr = s_a + (trow * PW + kh2) * 64 + cib * 32 + li;
        float a_cur = a_ptr[0];
        float b_cur0 = b_ptr[0], b_cur1 = b_ptr[64], b_cur2 = b_ptr[128];
#pragma unroll
        for (int tt = 0; tt < TH * TW / 16; ++tt) {
            if (tt < NPRE_A + NPRE_D && has_next && !(abl & 4))
                issue_slot(tt, nact, ndz, nty * TH, ntx * TW);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = tt * 8 + u;
                // next k-pair (clamped on the last one; the extra read is unused)
                const int tn = t + 1 < TH * TW / 2 ? t + 1 : t;
                const int pn = 2 * tn;
                const int rown = pn >> 5, coln = pn & 31;
                const float a_nxt = a_ptr[(rown * TW + coln) * 64];
                const float *bn = b_ptr + (rown * PW + coln) * 64;
                const float b_nxt0 = bn[0], b_nxt1 = bn[64], b_nxt2 = bn[128];
                acc[0] = mfma32(a_cur, b_cur0, acc[0]);
                acc[1] = mfma32(a_cur, b_cur1, acc[1]);
                acc[2] = mfma32(a_cur, b_cur2, acc[2]);
                a_cur = a_nxt;
                b_cur0 = b_nxt0;
                b_cur1 = b_nxt1;
                b_cur2 = b_nxt2;
            }
        }
        __syncthreads();              // all waves done with this tile's LDS image
        if (has_next && !(abl & 2)) write_lds();
        __syncthreads();
    }
    // one partial per block: part[block][tap][co][ci]
    float *dst = part + (size_t)blockIdx.x * (9 * 4096);
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = cob * 32 + mfma32_row(r, lane);
            dst[(trow * 3 + kw) * 4096 + co * 64 + cib * 32 + li] = acc[kw][r];
        }
}

// sum partials and write dW in the reference's OIHW layout: dW[co][ci][kh][kw].
// block = 64 elements x 16 slices of the partial list, fp64 accumulation, fixed order.
__global__ __launch_bounds__(1024) void conv3x3_wgrad_reduce_kernel(const float *__restrict__ part,
                                                                    int nparts,
                                                                    float *__restrict__ dw)
{
    __shared__ double s_acc[16][64];
    const int tx = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + tx;                    // over [tap][co][ci] = 36864
    double s = 0.0;
    for (int p = slice; p < nparts; p += 16) s += (double)part[(size_t)p * (9 * 4096) + idx];
    s_acc[slice][tx] = s;
    __syncthreads();
    if (slice == 0) {
        double t = 0.0;
        for (int j = 0; j < 16; ++j) t += s_acc[j][tx];
        const int tap = idx >> 12, co = (idx >> 6) & 63, ci = idx & 63;
        dw[(co * 64 + ci) * 9 + tap] = (float)t;
    }
}

int g_ablate = 0;
int g_grid_cap = 0;
inline int persistent_grid(int ntiles, int blocks_per_cu = 1)
{
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
        cus = prop.multiProcessorCount;
    int g = ntiles < cus * blocks_per_cu ? ntiles : cus * blocks_per_cu;
    if (g_grid_cap > 0 && g > g_grid_cap) g = g_grid_cap;
    return g;
}

}  // namespace

// 2 = cap on the persistent grids, 5 = ablation mask (builds with -DCOVA_ABLATE)
COVA_API int cova_direct_set_option(int key, int value)
{
    if (key == 2) { g_grid_cap = value; return COVA_OK; }
    if (key == 5) { g_ablate = value; return COVA_OK; }
    return COVA_ERR_BAD_ARG;
}

COVA_API int cova_conv3x3_num_tiles(int B, int H, int W)
{
    return B * cdiv(H, c3::TH) * cdiv(W, c3::TW);
}

// rows of the statistics partials cova_conv1_fwd writes: one per persistent block

COVA_API int cova_conv3x3_prep_weights(const float *w_oihw, float *w_fwd, float *w_dgrad,
                                       void *stream)
{
    COVA_REQUIRE(w_oihw && w_fwd && w_dgrad);
    hipLaunchKernelGGL(prep_w3x3_kernel, dim3(cdiv(9 * 4096, 256)), dim3(256), 0,
                       (hipStream_t)stream, w_oihw, w_fwd, w_dgrad);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}


// in/out NHWC [B,H,W,64]; w_t [9][64][64]; addend (nullable) is added to the result;
// stat_part (nullable): [cova_conv3x3_num_tiles][2][64] per-tile channel sum / sum of squares.
static int launch_conv3x3(const float *in, const float *w_t, const float *addend, float *out,
                          float *stat_part, int B, int H, int W, const BnBwdEpi bn, void *stream)
{
    const int tiles_x = cdiv(W, c3::TW), tiles_y = cdiv(H, c3::TH);
    const int ntiles = B * tiles_x * tiles_y;
    const dim3 block(c3::THREADS), grid(persistent_grid(ntiles));
    if (stat_part)
        hipLaunchKernelGGL(conv3x3_c64_v2_kernel<true>, grid, block, 0, (hipStream_t)stream, in, w_t,
                           addend, out, stat_part, H, W, tiles_x, tiles_y, ntiles, bn, g_ablate);
    else
        hipLaunchKernelGGL(conv3x3_c64_v2_kernel<false>, grid, block, 0, (hipStream_t)stream, in, w_t,
                           addend, out, stat_part, H, W, tiles_x, tiles_y, ntiles, bn, g_ablate);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

// Data gradient of conv3x3 fused with the ReLU mask and the BatchNorm-backward reduction of the
// layer in FRONT of the conv (a = relu(bn(z)) was the conv's input):
//   out = dy = (conv(dz, w_dgrad) + addend) * (act > 0)
//   stat_part[tile] = (sum dy, sum dy * (z - mean) * invstd) per channel
// replaces cova_conv3x3_fwd(dgrad) + cova_bn_bwd_reduce; follow with cova_bn_finalize_bwd and
// cova_bn_bwd_apply(dout = dy, act = NULL).
COVA_API int cova_conv3x3_dgrad_bnbwd(const float *dz, const float *w_dgrad, const float *addend,
                                      const float *act, const float *z, const float *mean,
                                      const float *invstd, float *dy, float *stat_part, int B,
                                      int H, int W, void *stream)
{
    COVA_REQUIRE(dz && w_dgrad && act && z && mean && invstd && dy && stat_part && B > 0 && H > 0 &&
                 W > 0);
    return launch_conv3x3(dz, w_dgrad, addend, dy, stat_part, B, H, W, BnBwdEpi{act, z, mean, invstd},
                          stream);
}

COVA_API int cova_conv3x3_fwd(const float *in, const float *w_t, const float *addend, float *out,
                              float *stat_part, int B, int H, int W, void *stream)
{
    COVA_REQUIRE(in && w_t && out && B > 0 && H > 0 && W > 0);
    return launch_conv3x3(in, w_t, addend, out, stat_part, B, H, W,
                          BnBwdEpi{nullptr, nullptr, nullptr, nullptr}, stream);
}

// img NCHW [B,3,H,W]; w_k [154][64]; out NHWC [B,H1,W1,64]

// act, dz NHWC [B,H,W,64]; dw OIHW [64,64,3,3]; ws >= cova_conv3x3_wgrad_workspace_floats
COVA_API int cova_conv3x3_wgrad(const float *act, const float *dz, float *dw, float *ws, int B,
                                int H, int W, void *stream)
{
    COVA_REQUIRE(act && dz && dw && ws && B > 0 && H > 0 && W > 0);
    const int tiles_x = cdiv(W, wg3v2::TW), tiles_y = cdiv(H, wg3v2::TH);
    const int ntiles = B * tiles_x * tiles_y;
    const int grid = persistent_grid(ntiles);
    hipLaunchKernelGGL(conv3x3_wgrad_v2_kernel, dim3(grid), dim3(wg3v2::THREADS), 0,
                       (hipStream_t)stream, act, dz, ws, H, W, tiles_x, tiles_y, ntiles, g_ablate);
    COVA_LAUNCH_CHECK();
    hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3(9 * 4096 / 64), dim3(1024), 0,
                       (hipStream_t)stream, ws, grid, dw);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

