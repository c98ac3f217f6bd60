// Convolution kernels of the truncated ResNet-18 stack (reference models.py:49-51):
//   conv1   7x7 stride 2 pad 3, 3 -> 64   (NCHW image in, NHWC activation out)
//   conv3x3 3x3 stride 1 pad 1, 64 -> 64  (NHWC in / out) -- forward and, with the
//           transposed + tap-flipped weights, the data gradient
//   weight-gradient kernels for both, as split-K (over pixels) persistent kernels
//
// All are implicit GEMMs on v_mfma_f32_32x32x2_f32 (exact f32, 157 TF/s peak on gfx950):
// the reference computes in fp32, so does this path.  Activations are NHWC so that the
// 64 channels of a pixel are one 256-byte line: a wavefront's global loads are whole
// lines, LDS tiles keep channels contiguous, and a pixel's K-slice is one ds_read_b128.
//
// Roofline bookkeeping: 2*64*64*9 = 73,728 FLOP per output pixel
// for conv3x3 (fwd, dgrad and wgrad each), 2*64*147 = 18,816 FLOP per output pixel for conv1.
#include "common.h"
#include <stdlib.h>
#include "bf3.h"
#include "bn_tail.h"
#include <utility>
#include <type_traits>

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
// conv1: 7x7, stride 2, pad 3, 3 -> 64.  NCHW image -> NHWC activation.
// K = 147 (+1 zero row) = 74 MFMA k-pairs.
// ------------------------------------------------------------------------------------
namespace c1 {
constexpr int TH = 8, TW = 32;                 // output tile
constexpr int PR = 2 * TH + 5;                 // 21 input rows
constexpr int PCH = 36;                        // columns per parity plane (35 used + 1 pad)
constexpr int RSTR = 2 * PCH;                  // 72 floats per staged row (even | odd columns)
constexpr int CSTR = PR * RSTR;                // 1512 floats per channel plane
constexpr int IN_FLOATS = 3 * CSTR;            // 4536 floats
// K ordering: 77 k-pairs (p, h) -> tap, chosen so that the two half-waves of an MFMA (h = lane>>5)
// read LDS at a CONSTANT distance from each other within each of three sets; every operand read
// is then `per-lane base + immediate` (no per-tap address registers):
//   p =  0..48: (c=0, t=p)        | (c=1, t=p)          distance CSTR
//   p = 49..69: (c=2, kh=j/7, kw) | (c=2, kh+4, kw)      distance 4*RSTR   (j = p-49, kh = 0..2)
//   p = 70..76: (c=2, kh=3, kw)   | zero weight                            (kw = p-70)
constexpr int NPAIR = 77;
constexpr int KROWS = 2 * NPAIR;               // 154 weight rows (7 of them zero)
constexpr int W_FLOATS = KROWS * 64;           // 9856 floats
constexpr int THREADS = 512;
__host__ __device__ constexpr int pair_tap(int p, int h)
{
    return p < 49 ? h * 49 + p
                  : p < 70 ? 98 + ((p - 49) / 7 + 4 * h) * 7 + (p - 49) % 7
                           : (h == 0 ? 98 + 21 + (p - 70) : -1);
}
// LDS offset of tap t = kh*7+kw inside one channel plane, relative to the lane's pixel origin
__host__ __device__ constexpr int local_off(int t)
{
    return (t / 7) * RSTR + ((t % 7) & 1) * PCH + ((t % 7) >> 1);
}
// LDS offset of tap k = (c, kh, kw) relative to the lane's pixel origin (k < 0: any valid address)
__host__ __device__ constexpr int tap_off(int k)
{
    return k < 0 ? 0 : (k / 49) * CSTR + local_off(k % 49);
}
}  // namespace c1

// ------------------------------------------------------------------------------------
// conv1 v2: persistent blocks (2 per CU); the 154x64 weight matrix stays resident in LDS for the
// whole launch, the next tile's image patch is prefetched into registers during the MFMAs.
// ------------------------------------------------------------------------------------
namespace c1 {
constexpr int PATCH = 3 * PR * 69;                                  // 4347 loaded floats
constexpr int NPRE = (PATCH + THREADS - 1) / THREADS;               // 9 per thread
}

template <bool STATS>
__global__ __launch_bounds__(c1::THREADS, 4) void conv1_7x7_v2_kernel(
    const float *__restrict__ img, const float *__restrict__ wk, float *__restrict__ out,
    float *__restrict__ stat_part, int H, int W, int H1, int W1, int tiles_x, int tiles_y, int ntiles,
    int abl_arg, int w_oihw, const BnTail tail)
{
    const int abl = COVA_ABL(abl_arg);
    using namespace c1;
    __shared__ __attribute__((aligned(16))) float lds[IN_FLOATS + W_FLOATS + 8 * 128];
    float *s_in = lds;
    float *s_w = lds + IN_FLOATS;
    float *s_red = lds + IN_FLOATS + W_FLOATS;
    // STATS: s_red[wave][128] are per-wave running totals over all tiles of this block (same lanes every
    // tile, no barrier); the eight rows are added up once after the tile loop: one partial row per block
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (STATS) { s_red[tid] = 0.f; s_red[tid + 512] = 0.f; }
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);     // wave-uniform copy: scalar index math
    const int li = lane & 31, kh2 = lane >> 5;
    int tile = blockIdx.x;
    if (tile >= ntiles) return;

    float pre[NPRE];
    auto issue_slot = [&](int it, const float *img_b, int ty, int tx) {
        // slots 0..7 take columns 0..63 (= lane) of row segment wave + 8*it (63 segments = 3 channels x
        // 21 rows: channel, row and the image row address are wave-uniform, i.e. scalar work);
        // slot 8 takes the remaining columns 64..68 (threads 0..314)
        int seg, j;
        if (it < 8) {
            seg = wave_u + 8 * it;
            j = lane;
        } else {
            seg = tid / 5;
            j = 64 + tid - 5 * seg;
        }
        const int c = seg / PR, r = seg - c * PR;
        const int gy = 2 * ty * TH - 3 + r, gx = 2 * tx * TW - 3 + j;
        float v = 0.f;
        if (seg < 3 * PR && gy >= 0 && gy < H && gx >= 0 && gx < W)
            v = img_b[(unsigned)((c * H + gy) * W + gx)];          // 32-bit in-image offset (checked at launch)
        pre[it] = v;
    };
    auto issue_loads = [&](int t) {
        const int tx = t % tiles_x;
        const int ty = (t / tiles_x) % tiles_y;
        const int b = t / (tiles_x * tiles_y);
#pragma unroll
        for (int it = 0; it < NPRE; ++it) issue_slot(it, img + (size_t)b * 3 * H * W, ty, tx);
    };
    auto write_lds = [&]() {
#pragma unroll
        for (int it = 0; it < NPRE; ++it) {
            int seg, j;
            if (it < 8) {
                seg = wave_u + 8 * it;
                j = lane;
            } else {
                seg = tid / 5;
                j = 64 + tid - 5 * seg;
            }
            if (seg < 3 * PR) {
                const int c = seg / PR, r = seg - c * PR;
                s_in[c * CSTR + r * RSTR + (j & 1) * PCH + (j >> 1)] = pre[it];
            }
        }
    };
    issue_loads(tile);
    if (w_oihw) {                // wk is the OIHW weight itself [64][3][7][7]: the K-pair layout is formed here (no prep launch)
        for (int idx = tid; idx < W_FLOATS; idx += THREADS) {
            const int row = idx >> 6, co = idx & 63;
            const int tap = c1::pair_tap(row >> 1, row & 1);
            s_w[idx] = tap >= 0 ? wk[co * 147 + tap] : 0.f;
        }
    } else {
        for (int idx = tid; idx < W_FLOATS / 4; idx += THREADS)
            reinterpret_cast<float4 *>(s_w)[idx] = reinterpret_cast<const float4 *>(wk)[idx];
    }
    write_lds();
    __syncthreads();

    const float *a_org = s_in + (2 * wave) * RSTR + li;          // this lane's output pixel
    const float *a_set0 = a_org + kh2 * CSTR;                    // c=0 | c=1
    const float *a_set1 = a_org + 2 * CSTR + kh2 * 4 * RSTR;     // c=2, kh | kh+4
    const float *a_set2 = a_org + 2 * CSTR;                      // c=2, kh=3 | zero weight
    const float *b_base = s_w + kh2 * 64 + li;
    for (; tile < ntiles; tile += gridDim.x) {
        const int tx = tile % tiles_x;
        const int ty = (tile / tiles_x) % tiles_y;
        const int b = tile / (tiles_x * tiles_y);
        const int y0 = ty * TH, x0 = tx * TW;
        const int next = tile + gridDim.x;
        const bool has_next = next < ntiles;
        // the next tile's 9 patch slots are issued one every 5 k-pairs inside the MFMA loop
        const int ntx = next % tiles_x, nty = (next / tiles_x) % tiles_y;
        const float *nimg = img + (size_t)(next / (tiles_x * tiles_y)) * 3 * H * W;

        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
        // every LDS operand address is a per-lane base + a compile-time immediate (see pair_tap)
#pragma unroll
        for (int p = 0; p < 49; ++p) {
            if (p % 5 == 0 && p / 5 < NPRE && has_next && !(abl & 4)) issue_slot(p / 5, nimg, nty, ntx);
            const float a = a_set0[local_off(p)];
            acc0 = mfma32(a, b_base[p * 128], acc0);
            acc1 = mfma32(a, b_base[p * 128 + 32], acc1);
        }
#pragma unroll
        for (int j = 0; j < 21; ++j) {
            const float a = a_set1[local_off(j)];
            acc0 = mfma32(a, b_base[(49 + j) * 128], acc0);
            acc1 = mfma32(a, b_base[(49 + j) * 128 + 32], acc1);
        }
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const float a = a_set2[local_off(21 + j)];
            acc0 = mfma32(a, b_base[(70 + j) * 128], acc0);
            acc1 = mfma32(a, b_base[(70 + j) * 128 + 32], acc1);
        }
        const int oy = y0 + wave;
        float s0 = 0.f, s1 = 0.f, q0 = 0.f, q1 = 0.f;
        if (abl & 1)
            asm volatile("" ::"v"(acc0), "v"(acc1));
        else
            epilogue_store_stats(acc0, acc1, out, nullptr, ((size_t)b * H1 + oy) * W1 + x0, x0, W1,
                                 oy < H1, lane, s0, s1, q0, q1);
        __syncthreads();                 // every wave is done reading the patch
        if (has_next && !(abl & 2)) write_lds();
        if (STATS && lane < 32) {
            s_red[wave * 128 + lane] += s0;
            s_red[wave * 128 + 32 + lane] += s1;
            s_red[wave * 128 + 64 + lane] += q0;
            s_red[wave * 128 + 96 + lane] += q1;
        }
        __syncthreads();
    }
    if (STATS && tid < 128) {            // (the loop's last barrier orders the waves' final updates)
        float t = 0.f;
        for (int w = 0; w < 8; ++w) t += s_red[w * 128 + tid];
        bn_tail_store(stat_part + (size_t)blockIdx.x * 128 + tid, t);
    }
    // BatchNorm finalize by the last block to finish (mode 0: a separate cova_bn_finalize_fwd launch follows)
    if (STATS) bn_tail_run(tail, stat_part, (int)gridDim.x, reinterpret_cast<double *>(lds));
}

// ------------------------------------------------------------------------------------
// conv1 v3: the same implicit GEMM on the bf16 matrix pipe with f32-class error.  gfx950 has no reduced-precision
// fast path for f32 operands (no xf32), and v_mfma_f32_32x32x2_f32 runs at the VECTOR rate (1/16 of the bf16 MFMA rate)
// on the lanes the VALU uses: the v2 kernel above is bound by it (0.69 of that peak).  Here every f32 operand x is
// split EXACTLY into three bf16 pieces, x = x0 + x1 + x2 (8 + 8 + 8 significand bits, by truncation: each residual is
// exact in f32 and the third piece is exact in bf16), and a*b is accumulated in f32 as the six products of order
// <= 2:  a2 b0 + a0 b2 + a1 b1 + a1 b0 + a0 b1 + a0 b0  (the three dropped ones are <= 2^-24 |a b|: below the rounding
// of the f32 product itself).  Products of bf16 pieces are exact in f32, so the only roundings are the accumulator's
// -- 6 v_mfma_f32_32x32x16_bf16 (16 k-slots each, 32 cycles) replace 8 v_mfma_f32_32x32x2_f32 (2 taps each, 64 cycles),
// and the bf16 MFMAs leave the vector lanes to the other waves of the SIMD.  Measured against fp64 the error is that
// of the f32 kernel (tools/conv1_bench.py; tests/test_kernels_gpu.py::test_conv1_bf16_split_error_class).
//   * the image patch is split ONCE per element, when the prefetched registers are written to LDS, and stored as three
//     planes of packed bf16 PAIRS of horizontally adjacent columns (2m, 2m+1) of the patch: the stride-2 window of
//     output pixel x starts at patch column 2x, so its 7 taps of one kernel row are the four consecutive dwords
//     m = x .. x+3 of that row (the eighth half-dword meets a zero weight) -- the A operand of a K-step is 2
//     ds_read2_b32 per piece and NO vector arithmetic (a first version that split the taps per output pixel, 44
//     operations per K-step, had more vector instructions than the MFMAs can cover);
//   * K = 21 kernel rows (c, kh) of 8 slots, two per K-step (lane half g = l >> 5 takes row 2s + g): 11 K-steps;
//   * the WEIGHTS live in registers: wave = (channel block cb = w & 1, output rows 2q, 2q+1 of the tile, q = w >> 1)
//     holds the 11 x 3 B operands of its 32 channels (132 registers) for the whole launch, split once per block from
//     the OIHW weight.  LDS then holds the patch only (double-buffered, 34 KB): block = 256 threads = 4 rows x 32 px,
//     TWO blocks per CU whose phases (MFMA loop / LDS refill / output stores) drift apart -- a first 1024-thread
//     version with a 66 KB weight image in LDS and all 16 waves of the CU behind the same two barriers per tile ran
//     its phases back to back (0.83 ms: ablation 0.19 LDS reads + 0.33 MFMAs + 0.10 refill + 0.20 stores);
//   * one barrier per tile, over LDS traffic only: the next patch is written to the other buffer BEFORE this tile's
//     output stores are issued, so the stores drain beside the next tile's MFMAs.
// ------------------------------------------------------------------------------------
// Ablation builds of tools/c1b_abl_build.sh (0 in the product): 1 no output stores, 2 no MFMAs, 4 no patch prefetch / LDS refill
#ifndef C1B_ABL
#define C1B_ABL 0
#endif
#ifndef C1B_SB
#define C1B_SB 0             // what may cross the pipeline's scheduling fences (0: nothing; 6: VALU + SALU)
#endif

// -DC1B_TRACE (tools only): s_memtime stamps of all four waves of blocks 0 and gridDim.x/2 (the latter shares block 0's
// CU when the grid is 2 x CUs in dispatch order -- checked by the hardware id stamped in slot 7) over tiles 8..27
#ifdef C1B_TRACE
__device__ unsigned long long g_c1b_trace[2 * 4 * 20 * 8];
#define C1B_STAMP(slot)                                                                                       \
    do {                                                                                                      \
        if (tr_blk >= 0 && tr_it >= 0 && tr_it < 20 && lane == 0 && wave < 4)                                 \
            g_c1b_trace[((tr_blk * 4 + wave) * 20 + tr_it) * 8 + (slot)] = __builtin_amdgcn_s_memtime();      \
    } while (0)
#else
#define C1B_STAMP(slot) do { } while (0)
#endif

namespace c1b {
constexpr int TH = 8, TW = 32;
constexpr int PR = 2 * TH + 5;                 // 21 input rows
constexpr int NSEG = 3 * PR;                   // 63 (channel, row) segments
constexpr int SEGW = 36;                       // dwords per segment: pairs m = 0 .. 34 (+1)
constexpr int PLANE = NSEG * SEGW;             // 2268 dwords per piece plane
constexpr int BUF = 3 * PLANE;                 // one patch buffer: 27,216 B
constexpr int KSTEPS = 11;
constexpr int WB_VEC = KSTEPS * 2 * 3 * 64;    // uint4 entries of the weight image [K-step][channel block][piece][lane]: 67,584 B
constexpr int THREADS = 512;
constexpr int NPRE = (PLANE + THREADS - 1) / THREADS;        // 5 pair slots per thread
constexpr int RF0 = 7;                         // K-step at which the prefetched patch starts going to LDS
__host__ __device__ constexpr int row_off(int t) { return t < 21 ? ((t / 7) * PR + t % 7) * SEGW : 0; }
}  // namespace c1b

__device__ __forceinline__ f32x16 mfma32bf(u32x4 a, u32x4 b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// barrier over LDS traffic only (global stores and loads stay in flight across it)
__device__ __forceinline__ void c1b_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// compile-time loop: f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N-1>)
template <class F, int... I>
__device__ __forceinline__ void c1b_static_for_impl(F &&f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void c1b_static_for(F &&f)
{
    c1b_static_for_impl(f, std::make_integer_sequence<int, N>{});
}

template <bool STATS>
__global__ __launch_bounds__(c1b::THREADS) void conv1_7x7_bf3_kernel(
    const float *__restrict__ img, const float *__restrict__ wk, float *__restrict__ out,
    float *__restrict__ stat_part, int H, int W, int H1, int W1, int tiles_x, int tiles_y, int ntiles,
    int w_oihw, const BnTail tail)
{
    using namespace c1b;
    __shared__ __attribute__((aligned(16))) uint32_t lds[2 * BUF + 4 * WB_VEC + 8 * 128];
    uint32_t *s_pp = lds;                                        // [buffer][piece][segment][pair]
    u32x4 *s_wb = reinterpret_cast<u32x4 *>(lds + 2 * BUF);      // B operands [K-step][channel block][piece][lane]
    float *s_red = reinterpret_cast<float *>(lds + 2 * BUF + 4 * WB_VEC);     // [wave][sum 64 | sum of squares 64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (STATS) { s_red[tid] = 0.f; s_red[tid + 512] = 0.f; }
    const int li = lane & 31, kh2 = lane >> 5;
    const int cb = wave & 1, q = wave >> 1;
    int tile = blockIdx.x;

    // The next tile's patch is prefetched into registers during the MFMAs: slot it of thread tid = pair
    // (it * 512 + tid) of the [segment][36] grid, two loads.  Every load is UNCONDITIONAL at a clamped address (a
    // branch around a load makes the compiler wait for all outstanding loads at the join); whether a value is inside the
    // image is kept as one bit and applied when the slot is written to LDS.
    static_assert(NPRE == 5, "write_lds lists the ten prefetch registers");
    float pre[2 * NPRE];
    unsigned premask = 0u;
    unsigned po0 = 0u, po1 = 0u, pm = 0u;     // a slot's element offsets and in-image bits between its two halves
    auto slot_addr = [&](int it, int ty, int tx) {
        int t_ = tid;
        asm volatile("" : "+v"(t_));          // the slot's index math stays here (not hoisted out of the tile loop)
        const int item = it * THREADS + t_;
        const int itc = item < PLANE ? item : PLANE - 1;
        const int seg = itc / SEGW, m = itc - seg * SEGW;
        const int c = seg / PR, r = seg - c * PR;
        const int gy = 2 * ty * TH - 3 + r, gx = 2 * tx * TW - 3 + 2 * m;
        const bool oky = item < PLANE && gy >= 0 && gy < H;
        const bool ok0 = oky && gx >= 0 && gx < W, ok1 = oky && gx + 1 >= 0 && gx + 1 < W;
        const int cy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy);
        const int cx0 = gx < 0 ? 0 : (gx >= W ? W - 1 : gx), cx1 = gx + 1 < 0 ? 0 : (gx + 1 >= W ? W - 1 : gx + 1);
        const unsigned rowo = (unsigned)((c * H + cy) * W);              // 32-bit in-image offset (checked at launch)
        po0 = rowo + cx0;
        po1 = rowo + cx1;
        pm = (ok0 ? 1u : 0u) | (ok1 ? 2u : 0u);
    };
    auto slot_load = [&](int it, const float *img_b) {
        pre[2 * it] = img_b[po0];
        pre[2 * it + 1] = img_b[po1];
        premask = (premask & ~(3u << (2 * it))) | (pm << (2 * it));
    };
    auto issue_slot = [&](int it, const float *img_b, int ty, int tx) {
        slot_addr(it, ty, tx);
        slot_load(it, img_b);
    };
    uint32_t rq0 = 0u, rq1 = 0u, rq2 = 0u;     // a slot's three packed dwords between its two halves
    auto refill_split = [&](int it) {
        const float xe = ((premask >> (2 * it)) & 1u) ? pre[2 * it] : 0.f;
        const float xo = ((premask >> (2 * it + 1)) & 1u) ? pre[2 * it + 1] : 0.f;
        bf3_split_pair(xe, xo, rq0, rq1, rq2);
    };
    auto refill_write = [&](int it, uint32_t *dst) {
        int t_ = tid;
        asm volatile("" : "+v"(t_));
        const int item = it * THREADS + t_;
        if (item < PLANE) {
            dst[item] = rq0;
            dst[PLANE + item] = rq1;
            dst[2 * PLANE + item] = rq2;
        }
    };
    auto refill_slot = [&](int it, uint32_t *dst) {
        refill_split(it);
        refill_write(it, dst);
    };
    auto write_lds = [&](uint32_t *dst) {
#pragma unroll
        for (int it = 0; it < NPRE; ++it) refill_slot(it, dst);
    };
    if (tile < ntiles) {
        const int pt = tile;
        const int tx = pt % tiles_x, ty = (pt / tiles_x) % tiles_y, b = pt / (tiles_x * tiles_y);
#pragma unroll
        for (int it = 0; it < NPRE; ++it) issue_slot(it, img + (size_t)b * 3 * H * W, ty, tx);
    }
    // B-operand image: entry (K-step s, channel block cb, lane (n, g)) = the 7 weights of channel cb*32+n in kernel
    // row t = 2s + g = (c, kh) and a zero, as three packed-bf16 pieces.  wk is the OIHW weight, or (w_oihw == 0) the
    // [154][64] K-pair layout of cova_conv1_prep_weights.
    for (int e = tid; e < KSTEPS * 2 * 64; e += THREADS) {
        const int l = e & 63, cbe = (e >> 6) & 1, s = e >> 7;
        const int co = cbe * 32 + (l & 31), t = 2 * s + (l >> 5);
        float wv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = 0.f;
            if (t < 21 && j < 7) {
                const int tap = t * 7 + j;                       // c*49 + kh*7 + kw
                if (w_oihw) {
                    v = wk[co * 147 + tap];
                } else {                                         // inverse of c1::pair_tap
                    int row;
                    if (tap < 98) row = 2 * (tap % 49) + tap / 49;
                    else {
                        const int kh = (tap - 98) / 7, kw = (tap - 98) % 7;
                        row = kh < 3 ? 2 * (49 + kh * 7 + kw) : kh == 3 ? 2 * (70 + kw) : 2 * (49 + (kh - 4) * 7 + kw) + 1;
                    }
                    v = wk[row * 64 + co];
                }
            }
            wv[j] = v;
        }
        u32x4 q0, q1, q2;
        bf3_split8(wv, q0, q1, q2);
        s_wb[((s * 2 + cbe) * 3 + 0) * 64 + l] = q0;
        s_wb[((s * 2 + cbe) * 3 + 1) * 64 + l] = q1;
        s_wb[((s * 2 + cbe) * 3 + 2) * 64 + l] = q2;
    }
    if (tile < ntiles) write_lds(s_pp);
    __syncthreads();

    // Output of an INTERIOR tile (all 8 x 32 pixels inside the map) is not stored after its MFMAs but during the NEXT
    // tile's: the accumulators move to p0 / p1 and their 32 stores are issued three per K-step between the MFMA
    // groups.  The phase trace of a version that stored right away showed why: 7,700 of a tile's 17,200 cycles were the
    // 32 store instructions (the launch writes 1.68 GB: the write path runs at its rate while every wave of the chip is in
    // its store phase and idles through the MFMA phases) -- the matrix pipe was 49 % busy.
    f32x16 p0, p1;
    bool pend = false;
    float *pb00 = out, *pb01 = out, *pb10 = out, *pb11 = out;            // [row][pixels 0-15 | 16-31] bases of the pending tile
    const unsigned st_off = (unsigned)((4 * kh2) * 64 + li);            // lane: pixel 4*(l >> 5), channel l & 31
    double tot_s = 0.0, tot_q = 0.0;          // per-lane totals over the block's tiles: tile sums in f32, their sum in f64
    float sm = 0.f, sq = 0.f;
    auto pend_store = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if constexpr (k < 32) {
            constexpr int rr = k >> 4, r = k & 15;
            constexpr int imm = ((r & 3) + 8 * ((r >> 2) & 1)) * 64;
            const float v = rr == 0 ? p0[r] : p1[r];
            (rr == 0 ? ((r >> 3) ? pb01 : pb00) : ((r >> 3) ? pb11 : pb10))[st_off + imm] = v;
            sm += v;
            sq = fmaf(v, v, sq);
        }
    };
    const u32x4 *b_base = s_wb + cb * 3 * 64 + lane;

    int cur = 0;
#ifdef C1B_TRACE
    const int tr_blk = blockIdx.x == 0 ? 0 : (blockIdx.x == gridDim.x / 2 ? 1 : -1);
    int tr_it = -8;
#endif
    for (; tile < ntiles; tile += gridDim.x) {
        C1B_STAMP(0);
        const int pt = tile;
        const int tx = pt % tiles_x;
        const int ty = (pt / tiles_x) % tiles_y;
        const int b = pt / (tiles_x * tiles_y);
        const int y0 = ty * TH, x0 = tx * TW;
        const bool has_next = tile + (int)gridDim.x < ntiles;
        const int next = has_next ? tile + (int)gridDim.x : tile;                               // (last tile: re-reads its own patch, unused)
        const int ntx = next % tiles_x, nty = (next / tiles_x) % tiles_y;
        const float *nimg = img + (size_t)(next / (tiles_x * tiles_y)) * 3 * H * W;
        const uint32_t *a_org = s_pp + cur * BUF + (4 * q) * SEGW + li;  // output row 2q (+ 2*SEGW: row 2q+1), dwords li..li+3

        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
        // One stream per wave in which EVERY MFMA is followed by a few of the other instructions (a scheduling fence
        // after each chunk keeps the order): a wave that issues six MFMAs back to back and then its ~30 other
        // instructions leaves the matrix pipe idle for most of that lump, and the partner wave of the SIMD -- which has
        // the same shape -- does not fill it (phase trace of that version: the older wave of a SIMD finished its
        // K loop in 9,400 cycles, the younger 3,000 later, 8,450 of them MFMA time of the pair).
        // Operands are double-buffered in registers and fetched one K-step ahead: during the six MFMAs on the first
        // pieces (G1: a0 b2, a0 b1, a0 b0 of both rows) the next K-step's first pieces and B operands, during the six on
        // the second and third pieces (G2: a2 b0, a1 b1, a1 b0) the next K-step's second and third pieces.  The other
        // chunks carry the next tile's patch prefetch (K-steps 0-4), its split and LDS writes (K-steps 7-9), and the
        // pending tile's 32 output stores (three per K-step).
        u32x4 g1[2][2], g2a[2][2], g2b[2][2], bl[2][3];          // [K-step parity][row] / [K-step parity][piece]
        auto a_ptr = [&](int s_) { return a_org + (kh2 ? row_off(2 * s_ + 1) : row_off(2 * s_)); };
        auto load4 = [&](u32x4 &d, const uint32_t *p_) {
#pragma unroll
            for (int i = 0; i < 4; ++i) d[i] = p_[i];
        };
        auto load_b = [&](int s_, int pc) { bl[s_ & 1][pc] = b_base[(s_ * 2 * 3 + pc) * 64]; };
        load4(g1[0][0], a_ptr(0)); load4(g1[0][1], a_ptr(0) + 2 * SEGW);
        load_b(0, 0); load_b(0, 1); load_b(0, 2);
        load4(g2a[0][0], a_ptr(0) + PLANE); load4(g2a[0][1], a_ptr(0) + 2 * SEGW + PLANE);
        load4(g2b[0][0], a_ptr(0) + 2 * PLANE); load4(g2b[0][1], a_ptr(0) + 2 * SEGW + 2 * PLANE);
        uint32_t *rdst = s_pp + (cur ^ 1) * BUF;
        // (two instantiations -- with and without a pending tile's stores: straight-line code either way, so the
        // compiler's wait for a prefetched value counts exactly the loads and stores issued after it)
        auto kloop = [&](auto stc) {
        constexpr bool st = decltype(stc)::value && !(C1B_ABL & 1);
        c1b_static_for<KSTEPS>([&](auto sc) {
            constexpr int s = decltype(sc)::value, P = s & 1, N = P ^ 1;
            constexpr bool more = s + 1 < KSTEPS;
            const u32x4 b0 = bl[P][0], b1 = bl[P][1], b2 = bl[P][2];
            // slot of the patch refill handled in this K-step's G1 / G2 chunks 4, 5 (-1: none)
            constexpr int rs1 = (s >= RF0 && 2 * (s - RF0) < NPRE) ? 2 * (s - RF0) : -1;
            constexpr int rs2 = (s >= RF0 && 2 * (s - RF0) + 1 < NPRE) ? 2 * (s - RF0) + 1 : -1;
            // ---- G1: first pieces
            c1b_static_for<6>([&](auto ic) {
                constexpr int i = decltype(ic)::value, row = i & 1;
                if (!(C1B_ABL & 2)) {
                    const u32x4 bb = i < 2 ? b2 : (i < 4 ? b1 : b0);
                    if (row == 0) acc0 = mfma32bf(g1[P][0], bb, acc0); else acc1 = mfma32bf(g1[P][1], bb, acc1);
                } else {
                    asm volatile("" ::"v"(g1[P][row]), "v"(b0), "v"(b1), "v"(b2));
                }
                if constexpr (i == 0) { if (more) load4(g1[N][0], a_ptr(s + 1)); if (st) pend_store(std::integral_constant<int, 3 * s>{}); }
                if constexpr (i == 1) { if (more) load4(g1[N][1], a_ptr(s + 1) + 2 * SEGW); }
                if constexpr (i == 2) { if (more) { load_b(s + 1, 0); load_b(s + 1, 1); } if (st) pend_store(std::integral_constant<int, 3 * s + 1>{}); }
                if constexpr (i == 3) { if (more) load_b(s + 1, 2); }
                if constexpr (i == 4) {
                    if (s < NPRE && !(C1B_ABL & 4)) slot_addr(s, nty, ntx);          // the next tile's patch
                    if (rs1 >= 0 && !(C1B_ABL & 4)) refill_split(rs1);
                }
                if constexpr (i == 5) {
                    if (s < NPRE && !(C1B_ABL & 4)) slot_load(s, nimg);
                    if (rs1 >= 0 && !(C1B_ABL & 4)) refill_write(rs1, rdst);
                }
                __builtin_amdgcn_sched_barrier(C1B_SB);
            });
            // ---- G2: second and third pieces
            c1b_static_for<6>([&](auto ic) {
                constexpr int i = decltype(ic)::value, row = i & 1;
                if (!(C1B_ABL & 2)) {
                    if (i < 2) { if (row == 0) acc0 = mfma32bf(g2b[P][0], b0, acc0); else acc1 = mfma32bf(g2b[P][1], b0, acc1); }
                    else { const u32x4 bb = i < 4 ? b1 : b0;
                           if (row == 0) acc0 = mfma32bf(g2a[P][0], bb, acc0); else acc1 = mfma32bf(g2a[P][1], bb, acc1); }
                } else {
                    asm volatile("" ::"v"(g2a[P][row]), "v"(g2b[P][row]));
                }
                if constexpr (i == 0) { if (more) load4(g2a[N][0], a_ptr(s + 1) + PLANE); }
                if constexpr (i == 1) { if (more) load4(g2a[N][1], a_ptr(s + 1) + 2 * SEGW + PLANE); if (st) pend_store(std::integral_constant<int, 3 * s + 2>{}); }
                if constexpr (i == 2) { if (more) load4(g2b[N][0], a_ptr(s + 1) + 2 * PLANE); }
                if constexpr (i == 3) { if (more) load4(g2b[N][1], a_ptr(s + 1) + 2 * SEGW + 2 * PLANE); }
                if constexpr (i == 4) { if (rs2 >= 0 && !(C1B_ABL & 4)) refill_split(rs2); }
                if constexpr (i == 5) { if (rs2 >= 0 && !(C1B_ABL & 4)) refill_write(rs2, rdst); }
                __builtin_amdgcn_sched_barrier(C1B_SB);
            });
        });
        };
        if (pend) kloop(std::true_type{}); else kloop(std::false_type{});
        tot_s += (double)sm;             // (two v_add_f64 per tile)
        tot_q += (double)sq;
        sm = 0.f; sq = 0.f;
        pend = false;
        C1B_STAMP(1);
        C1B_STAMP(2);
        c1b_lds_barrier();               // the next patch is complete; every wave is done reading this one
        C1B_STAMP(3);
        cur ^= 1;
        // output rows y0 + 2q, + 1: D register r of lane l = pixel mfma32_row(r, l), channel cb*32 + (l & 31)
        if (y0 + TH <= H1 && x0 + TW <= W1) {                    // interior: stored during the next tile (or after the loop)
            p0 = acc0; p1 = acc1;
            pend = true;
            pb00 = out + (((size_t)b * H1 + y0 + 2 * q) * W1 + x0) * 64 + cb * 32;
            pb01 = pb00 + 16 * 64;
            pb10 = pb00 + (size_t)W1 * 64;
            pb11 = pb10 + 16 * 64;
        } else if (C1B_ABL & 1) {
            asm volatile("" ::"v"(acc0), "v"(acc1));
        } else {
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                const int oy = y0 + 2 * q + rr;
                const size_t rowb = ((size_t)b * H1 + oy) * W1 + x0;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int px = mfma32_row(r, lane);
                    if (oy < H1 && x0 + px < W1) {
                        const float v = rr == 0 ? acc0[r] : acc1[r];
                        out[(rowb + (size_t)px) * 64 + cb * 32 + li] = v;
                        sm += v;
                        sq = fmaf(v, v, sq);
                    }
                }
            }
        }
        C1B_STAMP(4);
#ifdef C1B_TRACE
        if (tr_blk >= 0 && tr_it >= 0 && tr_it < 20 && lane == 0) {
            unsigned hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            g_c1b_trace[((tr_blk * 4 + (wave & 3)) * 20 + tr_it) * 8 + 7] = hw;
        }
        ++tr_it;
#endif
    }
    if (pend && !(C1B_ABL & 1)) {        // the block's last tile
        c1b_static_for<32>(pend_store);
    }
    tot_s += (double)sm;
    tot_q += (double)sq;
    if (STATS) {
        tot_s += __shfl_xor(tot_s, 32, 64);
        tot_q += __shfl_xor(tot_q, 32, 64);
        if (lane < 32) {
            s_red[wave * 128 + cb * 32 + li] = (float)tot_s;
            s_red[wave * 128 + 64 + cb * 32 + li] = (float)tot_q;
        }
    }
    __syncthreads();
    if (STATS && tid < 128) {
        float t = 0.f;
        for (int w = 0; w < 8; ++w) t += s_red[w * 128 + tid];
        bn_tail_store(stat_part + (size_t)blockIdx.x * 128 + tid, t);
    }
    if (STATS) bn_tail_run(tail, stat_part, (int)gridDim.x, reinterpret_cast<double *>(lds));
}

// ------------------------------------------------------------------------------------
// weight layout transforms (tiny; run once per step because the weights change every step)
// ------------------------------------------------------------------------------------
__global__ void prep_w7x7_kernel(const float *__restrict__ w, float *__restrict__ wk)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // over [154][64]
    if (idx >= c1::W_FLOATS) return;
    const int row = idx >> 6, co = idx & 63;
    const int tap = c1::pair_tap(row >> 1, row & 1);         // w[co][c][kh][kw], tap = c*49+kh*7+kw
    wk[idx] = tap >= 0 ? w[co * 147 + tap] : 0.f;
}

// ------------------------------------------------------------------------------------
// conv1 weight gradient: dW[co][k] = sum_p dy[p][co] * x[patch(p, k)],  k = (c,kh,kw) < 147
// M = co (2 blocks), N = taps padded to 160 (5 blocks), K = pixels.  Each wave holds all
// 10 accumulators and owns one output row of the 8x32 tile.
// ------------------------------------------------------------------------------------
namespace wg1 {
constexpr int TH = 8, TW = 32;
constexpr int THREADS = 512;
constexpr int DY_FLOATS = TH * TW * 64;     // 16,384 floats = 65,536 B
}  // namespace wg1

// conv1 weight gradient v2: wave w owns output-channel block w&1 and tile rows 2q, 2q+1
// (q = w>>1): 5 accumulators (80 registers) over 64 pixels, four partials per block; the next
// tile's dy tile and image patch are prefetched into registers during the MFMAs.
namespace wg1 {
constexpr int NPRE_D = (DY_FLOATS / 4 + THREADS - 1) / THREADS;    // 8 float4
}

// POOL: the gradient operand is not read but rebuilt while staging (the BatchNorm+ReLU+MaxPool
// backward "apply" pass folded in):  dy1 = A*g + B*y1 + C  per channel, where g routes the pooled
// gradient dp (already ReLU-masked) to each window's arg-max (idx).  The tile is staged as B*y1 + C
// and each thread then gathers, into its own pixels, A*dp of the pooling windows (of the 5x17 that can reach the
// tile) whose arg-max they are: no atomics, and dy1 (1.68 GB at the bench size) is never written.
struct PoolBwd {
    const float *dp;          // [B,H2,W2,64] pooled gradient, ReLU mask already applied
    const uint8_t *idx;       // [B,H2,W2,64] arg-max position ky*3+kx inside the 3x3/s2 window
    const float *abc;         // [3][64] A | B | C
    int H2, W2;
};

template <bool POOL>
__global__ __launch_bounds__(wg1::THREADS) void conv1_wgrad_v2_kernel(
    const float *__restrict__ img, const float *__restrict__ dy, float *__restrict__ part,
    int H, int W, int H1, int W1, int tiles_x, int tiles_y, int ntiles, const PoolBwd pool)
{
    using namespace c1;
    constexpr int NPOOL = POOL ? 3 : 0;           // (window, 4-channel group) items per thread: 5*17*16 = 1360
    constexpr int WIN_W = wg1::TW / 2 + 1, WIN_ITEMS = (wg1::TH / 2 + 1) * WIN_W * 16;
    constexpr int NWIN = (wg1::TH / 2 + 1) * WIN_W;                  // 85 windows can reach a tile
    __shared__ __attribute__((aligned(16))) float lds[IN_FLOATS + wg1::DY_FLOATS + (POOL ? NWIN * 80 : 0)];
    float *s_in = lds;
    float *s_dy = lds + IN_FLOATS;
    float *s_dpw = lds + IN_FLOATS + wg1::DY_FLOATS;                 // POOL: [window][64] pooled gradient
    uint32_t *s_ixw = reinterpret_cast<uint32_t *>(s_dpw + NWIN * 64);   // POOL: [window][16] arg-max codes x4
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);     // wave-uniform copy: scalar index math
    const int li = lane & 31, kh2 = lane >> 5;
    const int cob = wave & 1, q = wave >> 1;

    int toff[5];
#pragma unroll
    for (int tb = 0; tb < 5; ++tb) {
        const int k = tb * 32 + li;
        toff[tb] = (k < 147) ? ((k / 49) * CSTR + ((k % 49) / 7) * RSTR + ((k % 7) & 1) * PCH +
                                ((k % 7) >> 1))
                             : 0;
    }
    f32x16 acc[5];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    float pre[NPRE];
    float4 pd[wg1::NPRE_D];
    uint32_t pd_in = 0;                           // POOL: bit it = pixel of slot it lies inside the map
    float4 pdp[POOL ? 3 : 1];
    uint32_t pix[POOL ? 3 : 1];
    // this thread's 4 channels of A | B | C; fetched where used (kept out of the MFMA loop's registers)
    auto coef = [&](int which) {
        int o = which * 64 + (tid & 15) * 4;
        asm volatile("" : "+v"(o));
        return *reinterpret_cast<const float4 *>(pool.abc + o);
    };
    // slot s < NPRE: image patch element; then NPRE_D float4 of the dy (POOL: y1) tile; then NPOOL
    // (window, channel group) items of the pooled gradient
    auto issue_slot = [&](int s, const float *img_b, const float *dy_b, int y0, int x0, int b) {
        if (s >= NPRE + wg1::NPRE_D) {
            if (POOL) {
                const int k = s - NPRE - wg1::NPRE_D;
                int t_ = tid;
                asm volatile("" : "+v"(t_));      // index math stays here (hoisted, it costs more registers than it saves)
                const int item = t_ + k * THREADS;
                const int wr = item / (WIN_W * 16), wc = (item >> 4) % WIN_W;
                const int ph = (y0 >> 1) + wr, pw = (x0 >> 1) + wc;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                uint32_t code = 0;
#if defined(POOL_ABL) && (POOL_ABL & 2)
                if (false) {
#else
                if (item < WIN_ITEMS && ph < pool.H2 && pw < pool.W2) {
#endif
                    const size_t o = (((size_t)b * pool.H2 + ph) * pool.W2 + pw) * 64 + (t_ & 15) * 4;
                    v = *reinterpret_cast<const float4 *>(pool.dp + o);
                    code = *reinterpret_cast<const uint32_t *>(pool.idx + o);
                }
                pdp[k] = v;
                pix[k] = code;
            }
        } else if (s < NPRE) {
            int seg, j;                           // same slot -> patch element mapping as conv1_7x7_v2_kernel
            if (s < 8) {
                seg = wave_u + 8 * s;
                j = lane;
            } else {
                seg = tid / 5;
                j = 64 + tid - 5 * seg;
            }
            const int c = seg / PR, r = seg - c * PR;
            const int gy = 2 * y0 - 3 + r, gx = 2 * x0 - 3 + j;
            float v = 0.f;
            if (seg < 3 * PR && gy >= 0 && gy < H && gx >= 0 && gx < W)
                v = img_b[(unsigned)((c * H + gy) * W + gx)];          // 32-bit in-image offset (checked at launch)
            pre[s] = v;
        } else {
            const int it = s - NPRE;
            const int idx = tid + it * THREADS;
            const int px = idx >> 4, c4 = idx & 15;
            const int r = px / wg1::TW, c = px - r * wg1::TW;
            const int gy = y0 + r, gx = x0 + c;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const bool in = gy < H1 && gx < W1;
            if (in) v = *reinterpret_cast<const float4 *>(dy_b + (unsigned)(((gy * W1 + gx) << 6) + c4 * 4));
            pd[it] = v;
            pd_in = (pd_in & ~(1u << it)) | ((in ? 1u : 0u) << it);
        }
    };
    auto issue_loads = [&](int t) {
        const int tx = t % tiles_x;
        const int ty = (t / tiles_x) % tiles_y;
        const int b = t / (tiles_x * tiles_y);
#pragma unroll
        for (int s = 0; s < NPRE + wg1::NPRE_D + NPOOL; ++s)
            issue_slot(s, img + (size_t)b * 3 * H * W, dy + (size_t)b * H1 * W1 * 64, ty * wg1::TH,
                       tx * wg1::TW, b);
    };
    auto write_lds = [&]() {
#pragma unroll
        for (int it = 0; it < NPRE; ++it) {
            int seg, j;
            if (it < 8) {
                seg = wave_u + 8 * it;
                j = lane;
            } else {
                seg = tid / 5;
                j = 64 + tid - 5 * seg;
            }
            if (seg < 3 * PR) {
                const int c = seg / PR, r = seg - c * PR;
                s_in[c * CSTR + r * RSTR + (j & 1) * PCH + (j >> 1)] = pre[it];
            }
        }
        float4 cB = make_float4(0.f, 0.f, 0.f, 0.f), cC = cB;
        if (POOL) {
            cB = coef(1);
            cC = coef(2);
        }
#pragma unroll
        for (int it = 0; it < wg1::NPRE_D; ++it) {
            float4 v = pd[it];
            if (POOL) {                                                   // B*y1 + C, 0 outside the map
                const bool in = (pd_in >> it) & 1u;
                v.x = in ? fmaf(cB.x, v.x, cC.x) : 0.f;
                v.y = in ? fmaf(cB.y, v.y, cC.y) : 0.f;
                v.z = in ? fmaf(cB.z, v.z, cC.z) : 0.f;
                v.w = in ? fmaf(cB.w, v.w, cC.w) : 0.f;
            }
            *reinterpret_cast<float4 *>(s_dy + (tid + it * THREADS) * 4) = v;
        }
#pragma unroll
        for (int k = 0; k < NPOOL; ++k) {
            int t_ = tid;
            asm volatile("" : "+v"(t_));
            const int item = t_ + k * THREADS;                             // = window * 16 + channel group
            if (item < WIN_ITEMS) {
                *reinterpret_cast<float4 *>(s_dpw + item * 4) = pdp[k];
                s_ixw[item] = pix[k];
            }
        }
    };
    // POOL: finish dy1 in place.  Thread -> pixel (row it, column cc), channels 4*c4..+3; the column
    // permutation gives every wave four columns of one parity, so the set of pooling windows that
    // can route into a pixel (1, 2 or 4: 3x3 windows, stride 2) is wave-uniform and rows are
    // compile-time.  No atomics: each thread gathers into its own pixels.
    auto scatter_pool = [&]() {
#if defined(POOL_ABL) && (POOL_ABL & 1)
        return;
#endif
        const int qx = tid >> 4, c4 = tid & 15;
        const float4 cA = coef(0);
        const int cc = (((qx & 3) << 1) | ((qx >> 2) & 1)) + 8 * (qx >> 3);
        const bool codd = (wave & 1) != 0;                                  // == cc & 1
        const int wc0 = codd ? (cc - 1) >> 1 : cc >> 1;                    // first candidate window column
        const int kx0 = codd ? 2 : 1;                                      // its kx; the second (odd only): wc0+1, kx 0
        // four batches of two rows (2b, 2b+1); all LDS reads of a batch are issued before the first use.
        // Odd columns have two candidate window columns, even ones a single one (wave-uniform branch).
        auto batch = [&](const int bq, const int ne) {
            uint32_t word[3][2];
            float4 dval[3][2], cur[2];
            // candidates of rows 2b, 2b+1: (row, window row - b, ky)
            constexpr int crow[3] = {0, 1, 1}, cwr[3] = {0, 0, 1}, cky[3] = {1, 2, 0};
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    if (e >= ne) continue;
                    const int win = (bq + cwr[k]) * WIN_W + wc0 + e;
                    word[k][e] = s_ixw[win * 16 + c4];
                    dval[k][e] = *reinterpret_cast<const float4 *>(s_dpw + win * 64 + c4 * 4);
                }
#pragma unroll
            for (int rr = 0; rr < 2; ++rr)
                cur[rr] = *reinterpret_cast<const float4 *>(s_dy + ((2 * bq + rr) * wg1::TW + cc) * 64 + c4 * 4);
            float g[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    if (e >= ne) continue;
                    const uint32_t code = (uint32_t)(cky[k] * 3 + (e == 0 ? kx0 : 0));
                    const float dv[4] = {dval[k][e].x, dval[k][e].y, dval[k][e].z, dval[k][e].w};
#pragma unroll
                    for (int jx = 0; jx < 4; ++jx)
                        g[crow[k]][jx] += (((word[k][e] >> (8 * jx)) & 255u) == code) ? dv[jx] : 0.f;
                }
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
                float4 v = cur[rr];
                v.x = fmaf(cA.x, g[rr][0], v.x);
                v.y = fmaf(cA.y, g[rr][1], v.y);
                v.z = fmaf(cA.z, g[rr][2], v.z);
                v.w = fmaf(cA.w, g[rr][3], v.w);
                *reinterpret_cast<float4 *>(s_dy + ((2 * bq + rr) * wg1::TW + cc) * 64 + c4 * 4) = v;
            }
            __builtin_amdgcn_sched_barrier(0);       // one batch in flight at a time (registers)
        };
        if (codd) {
#pragma unroll
            for (int bq = 0; bq < 4; ++bq) batch(bq, 2);
        } else {
#pragma unroll
            for (int bq = 0; bq < 4; ++bq) batch(bq, 1);
        }
    };

    int tile = blockIdx.x;
    if (tile < ntiles) {
        issue_loads(tile);
        write_lds();
        if (POOL) {
            __syncthreads();
            scatter_pool();
        }
    }
    __syncthreads();
    for (; tile < ntiles; tile += gridDim.x) {
        const int next = tile + gridDim.x;
        const bool has_next = next < ntiles;
        // next tile's 17 prefetch slots: one per k-pair inside the MFMA loop
        const int ntx = next % tiles_x, nty = (next / tiles_x) % tiles_y;
        const int nb = next / (tiles_x * tiles_y);
        const float *nimg = img + (size_t)nb * 3 * H * W;
        const float *ndy = dy + (size_t)nb * H1 * W1 * 64;
        // this wave: tile rows 2q, 2q+1; k-pair t -> pixels p = 2t + kh2 (row 2q + (p>>5), col p&31)
        const float *a_ptr = s_dy + (2 * q * wg1::TW + kh2) * 64 + cob * 32 + li;
        const float *b_ptr = s_in + (4 * q) * RSTR + kh2;
        float a_op[2], b_op[2][5];                 // operands of k-pair t in slot t & 1 (no copies)
        a_op[0] = a_ptr[0];
#pragma unroll
        for (int tb = 0; tb < 5; ++tb) b_op[0][tb] = b_ptr[toff[tb]];
#pragma unroll
        for (int t = 0; t < 32; ++t) {
            if (t < NPRE + wg1::NPRE_D + NPOOL && has_next)
                issue_slot(t, nimg, ndy, nty * wg1::TH, ntx * wg1::TW, nb);
            if (t < 31) {
                const int pn = 2 * (t + 1);
                const int rown = pn >> 5, coln = pn & 31;
                a_op[(t + 1) & 1] = a_ptr[(rown * wg1::TW + coln) * 64];
                const float *bn = b_ptr + (2 * rown) * RSTR + coln;
#pragma unroll
                for (int tb = 0; tb < 5; ++tb) b_op[(t + 1) & 1][tb] = bn[toff[tb]];
            }
#pragma unroll
            for (int tb = 0; tb < 5; ++tb) acc[tb] = mfma32(a_op[t & 1], b_op[t & 1][tb], acc[tb]);
        }
        __syncthreads();
        if (has_next) write_lds();
        if (POOL) {
            __syncthreads();
            if (has_next) scatter_pool();
        }
        __syncthreads();
    }
    // partial layout: part[(block*4 + q)][co 64][k 160]
    float *dst = part + ((size_t)(blockIdx.x * 4 + q)) * (64 * 160);
#pragma unroll
    for (int tb = 0; tb < 5; ++tb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = cob * 32 + mfma32_row(r, lane);
            dst[co * 160 + tb * 32 + li] = acc[tb][r];
        }
}

// ------------------------------------------------------------------------------------
// conv1 weight gradient on the bf16 matrix pipe (operands split exactly into three bf16 pieces, six products, f32
// accumulation: see conv1_7x7_bf3_kernel).  M = output channel (2 blocks), N = tap (5 blocks of 32), K = pixel:
// 30 v_mfma_f32_32x32x16_bf16 (960 cycles) per 16 pixels replace 40 v_mfma_f32_32x32x2_f32 (2,560 cycles).
//   * K-slot PAIRS are vertically adjacent output pixels (rows 2q, 2q+1 of the tile, same column): the thread that
//     finishes dy1 of a column (BatchNorm + ReLU + MaxPool backward, below) holds both, splits them and writes packed
//     dwords [pair][piece][channel] -- the A operand of a lane (channel, 8 pixels) is 4 dwords at a constant stride;
//   * the image patch is stored as packed pairs of rows (r, r+2) -- the two input rows a tap reads for such a pair --
//     [c][r][piece][column parity][36]: the B operand of a lane (tap, 8 pixels = 4 consecutive columns x 2 rows) is 4
//     consecutive dwords; every element is split once, by the thread that prefetched it;
//   * wave = (channel block, row pair q): 5 accumulators over the 64 pixels of its rows = 4 K-steps of 30 MFMAs,
//     every MFMA followed by one LDS operand read for the next group or one slot of the next tile's prefetch
//     (operands double-buffered in registers);
//   * LDS: 96 KB of packed dy1 + 48 KB shared in time by the pooled-gradient windows (while dy1 is finished) and the
//     patch planes (during the MFMAs): four barriers per tile.
// ------------------------------------------------------------------------------------
// (The phase-structured kernel this comment was written for -- conv1_wgrad_bf3_kernel, 1.12 ms -- left the library in
// round 6; the role-split kernel below keeps its arithmetic and operand layouts: DESIGN.md 11.8.)

// ------------------------------------------------------------------------------------
// conv1 weight gradient, ROLE-SPLIT form of conv1_wgrad_bf3_kernel (same arithmetic, same operand layouts, a quarter of
// its tile): of the two waves of every SIMD one multiplies and the other one stages.
//   * tile = 4 x 16 output pixels.  Waves 0-3 (one per SIMD) = (channel block, row pair): 2 K-steps x 30 MFMAs per tile on
//     the tile's operands in LDS; between the MFMAs they also bring in the NEXT tile's image patch (prefetch, split,
//     packed row pairs -> the other patch buffer: ~3 instructions per MFMA).
//   * Waves 4-7 finish dy1 of the next tile meanwhile -- thread = (column, 4 channels), all 4 rows: y1 and the 3 or 6
//     pooling windows that can route into the column straight from global memory into registers (one tile ahead),
//     A * route(dp) + B * y1 + C, split, packed row pairs -> the other dy1 buffer.  No windows in LDS, no phases.
//   * everything double-buffered (2 x 24 KB dy1, 2 x 15.5 KB patch): ONE barrier per tile.
// The phase trace of conv1_wgrad_bf3_kernel had shown 16,000 of a tile's 27,000 cycles in staging phases with the matrix
// pipe idle and 11,200 in an MFMA loop of 7,700 cycles of pipe work that the two waves of a SIMD do not share evenly.
// ------------------------------------------------------------------------------------
#ifndef WG1R_ABL
#define WG1R_ABL 0             // tools: 1 no dy1 operand loads, 2 no patch loads, 4 no dy1 arithmetic / LDS writes, 8 no MFMAs
#endif

namespace wg1r {
constexpr int TH = 4, TW = 16, THREADS = 512;
constexpr int D_DW = 32 * 3 * 64;               // dy1 planes [pair 2 x 16][piece][channel]: 6,144 dwords
constexpr int PROW = 3 * 40;                    // dwords per (c, r) row of the patch planes: [piece][parity][20]
constexpr int P_DW = 3 * 11 * PROW;             // 3,960 dwords
constexpr int PATCH_ITEMS = 3 * 11 * 37;        // 1221 (c, r, column) pairs of rows (r, r+2)
constexpr int NPRE_P = (PATCH_ITEMS + 255) / 256;                // 5 slots per MFMA-wave thread
}  // namespace wg1r

// -DC1B_TRACE: stamps of the eight waves of block 0, tiles 40..59: 0 tile start, 1 work done (before the barrier), 2 after the
// barrier, 3 (staging waves) dy1 finished / loads not yet issued
#ifdef C1B_TRACE
__device__ unsigned long long g_wg1b_trace[8 * 20 * 8];
#define WG1R_STAMP(slot)                                                                                      \
    do {                                                                                                      \
        if (blockIdx.x == 0 && tr_it >= 0 && tr_it < 20 && lane == 0)                                         \
            g_wg1b_trace[(wave * 20 + tr_it) * 8 + (slot)] = __builtin_amdgcn_s_memtime();                    \
    } while (0)
#define WG1R_TRIT() ++tr_it
#else
#define WG1R_STAMP(slot) do { } while (0)
#define WG1R_TRIT() do { } while (0)
#endif

template <bool POOL>
__global__ __launch_bounds__(wg1r::THREADS) void conv1_wgrad_rs_kernel(
    const float *__restrict__ img, const float *__restrict__ dy, float *__restrict__ part,
    int H, int W, int H1, int W1, int tiles_x, int tiles_y, int ntiles, const PoolBwd pool)
{
    using namespace wg1r;
    __shared__ __attribute__((aligned(16))) uint32_t lds[2 * D_DW + 2 * P_DW];
    uint32_t *s_d = lds, *s_p = lds + 2 * D_DW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool mm = wave < 4;                                        // multiplying | staging wave
    struct TileC { int b, y0, x0; };
    auto coords = [&](int t) {
        TileC c;
        const int tt = t < ntiles ? t : ntiles - 1;    // (past the end: the last tile again, unused)
        const int r_ = tt / tiles_x;
        c.x0 = (tt - r_ * tiles_x) * TW;
        c.b = r_ / tiles_y;
        c.y0 = (r_ - c.b * tiles_y) * TH;
        return c;
    };
    const int t0 = blockIdx.x, tstep = gridDim.x;
#ifdef C1B_TRACE
    int tr_it = -40;
#endif

    if (mm) {
        // =================================================================== multiplying waves
        const int li = lane & 31, kh2 = lane >> 5;
        const int cob = wave & 1, q = wave >> 1;
        const int mtid = tid;                                        // 0 .. 255
        int toff[5];
#pragma unroll
        for (int tb = 0; tb < 5; ++tb) {
            const int k = tb * 32 + li;
            const int kk = k < 147 ? k : 0;                          // (columns 147..159 of the result are never read)
            toff[tb] = ((kk / 49) * 11 + 4 * q + (kk % 49) / 7) * PROW + ((kk % 7) & 1) * 20 + ((kk % 7) >> 1) + 4 * kh2;
        }
        const int a_base = ((q * 16 + 4 * kh2) * 3) * 64 + cob * 32 + li;
        f32x16 acc[5];
#pragma unroll
        for (int t = 0; t < 5; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        // patch slots: slot n of this thread = item n * 256 + mtid of the [c 3][r 11][column 37] grid -- the same element of
        // every tile's patch, so what does not depend on the tile is computed once: the element's offset inside the image
        // relative to the patch origin, its LDS destination, and (c, r, j) packed for the edge path.  (Recomputed per
        // tile -- two divisions, clamps, validity -- the ten loads of a tile cost ~325 vector instructions in the MFMA
        // waves' stream: 0.45 ms of the launch, ablation.)
        int prel[NPRE_P], pldo[NPRE_P], pcrj[NPRE_P];
#pragma unroll
        for (int n = 0; n < NPRE_P; ++n) {
            const int item = n * 256 + mtid;
            const int itc = item < PATCH_ITEMS ? item : PATCH_ITEMS - 1;
            const int seg = itc / 37, j = itc - seg * 37;
            const int c = seg / 11, r = seg - c * 11;
            prel[n] = (c * H + r) * W + j;
            pldo[n] = seg * PROW + (j & 1) * 20 + (j >> 1);
            pcrj[n] = c | (r << 2) | (j << 6) | (item < PATCH_ITEMS ? (1 << 12) : 0);
        }
        float preS[2][2 * NPRE_P];               // two register sets: a tile's patch is requested a whole tile before it is written
        unsigned premaskS[2] = {0u, 0u};
        // interior: the 13 x 37 patch of the tile lies inside the image (no clamps, no validity bits)
        auto patch_interior = [&](const TileC &tc) {
            return 2 * tc.y0 - 3 >= 0 && 2 * tc.y0 + 9 < H && 2 * tc.x0 - 3 >= 0 && 2 * tc.x0 + 33 < W;
        };
        auto patch_load = [&](auto setc, int n, const TileC &tc, bool fast) {
            if (WG1R_ABL & 2) return;
            auto &pre = preS[decltype(setc)::value];
            unsigned &premask = premaskS[decltype(setc)::value];
            const float *img_b = img + (size_t)tc.b * 3 * H * W;
            if (fast) {
                const float *org = img_b + ((2 * tc.y0 - 3) * W + 2 * tc.x0 - 3);       // wave-uniform
                pre[2 * n] = org[prel[n]];
                pre[2 * n + 1] = (org + 2 * W)[prel[n]];
                premask |= 3u << (2 * n);
                return;
            }
            const int c = pcrj[n] & 3, r = (pcrj[n] >> 2) & 15, j = (pcrj[n] >> 6) & 63;
            const int gy = 2 * tc.y0 - 3 + r, gx = 2 * tc.x0 - 3 + j;
            const bool okx = ((pcrj[n] >> 12) & 1) && gx >= 0 && gx < W;
            const bool ok0 = okx && gy >= 0 && gy < H, ok1 = okx && gy + 2 >= 0 && gy + 2 < H;
            const int cx = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
            const int cy0 = gy < 0 ? 0 : (gy >= H ? H - 1 : gy), cy1 = gy + 2 < 0 ? 0 : (gy + 2 >= H ? H - 1 : gy + 2);
            pre[2 * n] = img_b[(unsigned)((c * H + cy0) * W + cx)];
            pre[2 * n + 1] = img_b[(unsigned)((c * H + cy1) * W + cx)];
            premask = (premask & ~(3u << (2 * n))) | ((ok0 ? 1u : 0u) << (2 * n)) | ((ok1 ? 2u : 0u) << (2 * n));
        };
        auto patch_write = [&](auto setc, int n, uint32_t *dst) {
            auto &pre = preS[decltype(setc)::value];
            const unsigned premask = premaskS[decltype(setc)::value];
            if ((pcrj[n] >> 12) & 1) {
                const float x0v = ((premask >> (2 * n)) & 1u) ? pre[2 * n] : 0.f;
                const float x1v = ((premask >> (2 * n + 1)) & 1u) ? pre[2 * n + 1] : 0.f;
                uint32_t q0, q1, q2;
                bf3_split_pair(x0v, x1v, q0, q1, q2);
                uint32_t *d_ = dst + pldo[n];
                d_[0] = q0;
                d_[40] = q1;
                d_[80] = q2;
            }
        };
        TileC c1 = coords(t0);
        if (t0 < ntiles) {                       // prologue: the first tile's patch -> buffer 0; the second tile's in registers
#pragma unroll
            for (int n = 0; n < NPRE_P; ++n) patch_load(std::integral_constant<int, 1>{}, n, c1, false);
#pragma unroll
            for (int n = 0; n < NPRE_P; ++n) patch_write(std::integral_constant<int, 1>{}, n, s_p);
            c1 = coords(t0 + tstep);
#pragma unroll
            for (int n = 0; n < NPRE_P; ++n) patch_load(std::integral_constant<int, 0>{}, n, c1, false);
        }
        __syncthreads();
        int cur = 0;
        // tile n (parity par): set par holds the patch of tile n + 1 (requested during tile n - 1) and is written to the other
        // LDS buffer; the patch of tile n + 2 is requested into set par ^ 1 first thing
        auto tile_body = [&](int tile, auto parc) {
            constexpr int par = decltype(parc)::value;
            WG1R_STAMP(0);
            const TileC c2 = coords(tile + 2 * tstep);
            const bool pfast = patch_interior(c2);
            const uint32_t *sd = s_d + cur * D_DW, *sx = s_p + cur * P_DW;
            uint32_t *sxn = s_p + (cur ^ 1) * P_DW;
            u32x4 A[2][3], Bq[2][3];                                 // [K-step][piece] | [unit of the pair][piece]
            auto load_a = [&](int ks, int pc) {
#pragma unroll
                for (int i = 0; i < 4; ++i) A[ks][pc][i] = sd[a_base + ((8 * ks + i) * 3 + pc) * 64];
            };
            auto load_bu = [&](int u, int unit, int pc) {            // unit = 5 ks + tb
                const uint32_t *p_ = sx + toff[unit % 5] + pc * 40 + 8 * (unit / 5);
#pragma unroll
                for (int i = 0; i < 4; ++i) Bq[u][pc][i] = p_[i];
            };
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) { load_a(0, pc); load_bu(0, 0, pc); load_bu(1, 1, pc); }
            // 5 pairs of (K-step, tap block) units, 12 MFMAs each (product order a0 b1, a1 b1, a0 b2, a0 b0, a1 b0, a2 b0: a
            // unit's B pieces 1 / 2 / 0 are dead after MFMAs 3 / 5 / 11 of its pair and replaced right there); the six free
            // slots of a pair carry K-step 1's A operand (pair 0), the next tile's patch into the other buffer (split + LDS
            // writes from one register set) and the prefetch of the tile after it into the other set
            c1b_static_for<5>([&](auto gc) {
                constexpr int gp = decltype(gc)::value;
                constexpr bool more = gp + 1 < 5;
                c1b_static_for<12>([&](auto mc) {
                    constexpr int m = decltype(mc)::value, pr = m >> 1, u = m & 1, unit = 2 * gp + u;
                    constexpr int ks = unit / 5, tb = unit % 5;
                    constexpr int pa = (pr == 1 || pr == 4) ? 1 : (pr == 5 ? 2 : 0), pb = pr < 2 ? 1 : (pr == 2 ? 2 : 0);
                    if (!(WG1R_ABL & 8)) acc[tb] = mfma32bf(A[ks][pa], Bq[u][pb], acc[tb]);
                    constexpr int nunit = 2 * (gp + 1) + u;
                    if constexpr (m == 2 || m == 3) { if (more) load_bu(u, nunit, 1); }
                    else if constexpr (m == 4 || m == 5) { if (more) load_bu(u, nunit, 2); }
                    else if constexpr (m == 10 || m == 11) { if (more) load_bu(u, nunit, 0); }
                    else {
                        constexpr int f = gp * 6 + (m < 2 ? m : m - 4);          // 0 .. 29
                        if constexpr (f < 3) load_a(1, f);
                        else if constexpr (f < 3 + NPRE_P) patch_load(std::integral_constant<int, par ^ 1>{}, f - 3, c2, pfast);
                        else if constexpr (f >= 12 && f < 12 + NPRE_P) patch_write(std::integral_constant<int, par>{}, f - 12, sxn);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
            WG1R_STAMP(1);
            c1b_lds_barrier();
            WG1R_STAMP(2);
            cur ^= 1;
#ifdef C1B_TRACE
            ++tr_it;
#endif
        };
        for (int tile = t0; tile < ntiles; tile += 2 * tstep) {
            tile_body(tile, std::integral_constant<int, 0>{});
            if (tile + tstep < ntiles) tile_body(tile + tstep, std::integral_constant<int, 1>{});
        }
        // partial layout: part[(block*2 + q)][co 64][k 160]
        float *dst = part + ((size_t)(blockIdx.x * 2 + q)) * (64 * 160);
#pragma unroll
        for (int tb = 0; tb < 5; ++tb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cor = cob * 32 + mfma32_row(r, lane);
                dst[cor * 160 + tb * 32 + li] = acc[tb][r];
            }
        return;
    }
    // ======================================================================= staging waves
    {
        const int st = tid - 256;                                    // 0 .. 255
        const int pw = wave - 4, c4 = st & 15;
        // wave = (column group cg: columns 8 cg .. 8 cg + 7, row pair hp: rows 2 hp, 2 hp + 1 of the tile); lanes = (4 columns, 16
        // channel quads).  It finishes its row pair for the group's four EVEN columns, then for the four ODD ones: an odd column
        // gathers from twice as many pool windows, and with whole waves of one column parity (rounds 4-5: wave = 4 columns x 4 rows)
        // the two odd-column waves were the launch's critical path (4 100 against 3 000 cycles per tile) -- now the four staging
        // waves do the same work.  Same arithmetic per element, same LDS layout: bit-identical results.
        const int hp = pw & 1;
        const int cbase = 2 * ((pw >> 1) * 4 + ((st >> 4) & 3));     // this lane's even column; its odd one: + 1
        f32x4 cA = {0.f, 0.f, 0.f, 0.f}, cB = cA, cC = cA;
        if (POOL) {
            cA = *reinterpret_cast<const f32x4 *>(pool.abc + c4 * 4);
            cB = *reinterpret_cast<const f32x4 *>(pool.abc + 64 + c4 * 4);
            cC = *reinterpret_cast<const f32x4 *>(pool.abc + 128 + c4 * 4);
        }
        struct Ops {                             // what a thread loads for its two columns of a tile ([0] even, [1] odd)
            f32x4 yv[2][2];                      // dy (POOL: y1) of rows 2 hp, 2 hp + 1, 4 channels
            f32x4 dpv[2][2][2];                  // POOL: pooled gradient of windows (P0 + hp, P0 + hp + 1) x (W0 [, W0 + 1: odd column])
            uint32_t cdv[2][2][2];               //       their four arg-max codes
            unsigned ok;                         // bits 2 par + rr: row inside the map (with the column); bits 4 + 4 par + 2 k + e: window inside
        };
        Ops opa, opb, opc;                       // three sets: a tile's operands are requested two tiles before they are used
        // (a clamp-free form of these loads for interior tiles -- wave-uniform row bases + one constant lane offset, 22 loads
        // back to back -- measured SLOWER on the same box: 1.14 against 1.06 ms per launch)
        auto load_tile = [&](Ops &o_, const TileC &tc) {
            if (WG1R_ABL & 1) return;
            unsigned ok = 0u;
            const float *yb = dy + (size_t)tc.b * H1 * W1 * 64;
#pragma unroll
            for (int par = 0; par < 2; ++par) {
                const int gx = tc.x0 + cbase + par, cx = gx < W1 ? gx : W1 - 1;
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    const int gy = tc.y0 + 2 * hp + rr, cy = gy < H1 ? gy : H1 - 1;
                    o_.yv[par][rr] = *reinterpret_cast<const f32x4 *>(yb + (unsigned)(((cy * W1 + cx) << 6) + c4 * 4));
                    ok |= (gy < H1 && gx < W1) ? (1u << (2 * par + rr)) : 0u;
                }
                if (POOL) {
                    const float *db = pool.dp + (size_t)tc.b * pool.H2 * pool.W2 * 64;
                    const uint8_t *ib = pool.idx + (size_t)tc.b * pool.H2 * pool.W2 * 64;
                    const int P0 = (tc.y0 >> 1) + hp, W0 = par ? (gx - 1) >> 1 : gx >> 1;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int ph = P0 + k, cph = ph < pool.H2 ? ph : pool.H2 - 1;
#pragma unroll
                        for (int e = 0; e <= par; ++e) {
                            const int pwc = W0 + e, cpw = pwc < pool.W2 ? pwc : pool.W2 - 1;
                            const unsigned o = (unsigned)((cph * pool.W2 + cpw) * 64 + c4 * 4);
                            o_.dpv[par][k][e] = *reinterpret_cast<const f32x4 *>(db + o);
                            o_.cdv[par][k][e] = *reinterpret_cast<const uint32_t *>(ib + o);
                            ok |= (ph < pool.H2 && pwc < pool.W2) ? (1u << (4 + 4 * par + 2 * k + e)) : 0u;
                        }
                    }
                }
            }
            o_.ok = ok;
        };
        // rows 2 hp, 2 hp + 1 of the even, then of the odd column -> dy1, split by the row pair -> dst
        auto finish = [&](const Ops &o_, uint32_t *dst) {
            if (WG1R_ABL & 4) return;
            const unsigned ok = o_.ok;
#pragma unroll
            for (int par = 0; par < 2; ++par) {
                const int kx0 = par ? 2 : 1;                         // kx of window column W0; W0 + 1 (odd columns): kx 0
                float d[2][4];
#pragma unroll
                for (int rr = 0; rr < 2; ++rr) {
                    float g[4] = {0.f, 0.f, 0.f, 0.f};
                    if (POOL) {
                        // even row: window row hp with ky 1; odd row: window row hp with ky 2, hp + 1 with ky 0
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            if (rr == 0 && k == 1) continue;
                            const uint32_t ky = rr == 0 ? 1u : (k == 0 ? 2u : 0u);
#pragma unroll
                            for (int e = 0; e <= par; ++e) {
                                const uint32_t code = ky * 3u + (e == 0 ? (uint32_t)kx0 : 0u);
                                const bool win = (ok >> (4 + 4 * par + 2 * k + e)) & 1u;
#pragma unroll
                                for (int jx = 0; jx < 4; ++jx)
                                    g[jx] += (win && ((o_.cdv[par][k][e] >> (8 * jx)) & 255u) == code) ? o_.dpv[par][k][e][jx] : 0.f;
                            }
                        }
                    }
                    const bool in = (ok >> (2 * par + rr)) & 1u;
#pragma unroll
                    for (int jx = 0; jx < 4; ++jx) {
                        float v = o_.yv[par][rr][jx];
                        if (POOL) v = fmaf(cA[jx], g[jx], fmaf(cB[jx], v, cC[jx]));
                        d[rr][jx] = in ? v : 0.f;
                    }
                }
                u32x4 w0, w1, w2;
#pragma unroll
                for (int jx = 0; jx < 4; ++jx) {
                    uint32_t u0, u1, u2;
                    bf3_split_pair(d[0][jx], d[1][jx], u0, u1, u2);
                    w0[jx] = u0; w1[jx] = u1; w2[jx] = u2;
                }
                u32x4 *o4 = reinterpret_cast<u32x4 *>(dst + ((hp * 16 + cbase + par) * 3) * 64 + c4 * 4);
                o4[0] = w0;
                o4[16] = w1;
                o4[32] = w2;
            }
        };
        if (t0 < ntiles) {                       // prologue: the first tile's dy1 -> buffer 0; the next two tiles' operands in registers
            load_tile(opa, coords(t0));
            finish(opa, s_d);
            load_tile(opb, coords(t0 + tstep));
            load_tile(opc, coords(t0 + 2 * tstep));
        }
        __syncthreads();
        // tiles in triples (three register sets rotate): request the tile three ahead, then finish the next tile's dy1 from
        // operands requested two tiles ago (a one-tile lead, ~2,500 cycles, was less than the memory latency under load)
        int nb = 1;                              // dy1 buffer of the tile being finished
        for (int tile = t0; tile < ntiles; tile += 3 * tstep) {
#define WG1R_PSTEP(A_, B_, K_)                                                                   \
            WG1R_STAMP(0);                                                                       \
            finish(A_, s_d + nb * D_DW);                                                         \
            WG1R_STAMP(3);                                                                       \
            load_tile(B_, coords(tile + (K_) * tstep));                                          \
            WG1R_STAMP(1);                                                                       \
            c1b_lds_barrier();                                                                   \
            WG1R_STAMP(2);                                                                       \
            nb ^= 1;                                                                             \
            WG1R_TRIT();
            WG1R_PSTEP(opb, opa, 3)              // finishes tile + tstep, requests tile + 3 tstep
            if (tile + tstep >= ntiles) break;   // (block-uniform: the multiplying waves leave their loop at the same tile)
            WG1R_PSTEP(opc, opb, 4)
            if (tile + 2 * tstep >= ntiles) break;
            WG1R_PSTEP(opa, opc, 5)
        }
    }
}

__global__ __launch_bounds__(1024) void conv1_wgrad_reduce_kernel(const float *__restrict__ part,
                                                                  int nparts, float *__restrict__ dw)
{
    __shared__ double s_acc[16][64];
    const int tx = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + tx;                    // over [co][160]
    double s = 0.0;
    for (int p0 = slice; p0 < nparts; p0 += 16 * 8) {        // eight rows in flight, added in the same order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int p = p0 + 16 * u;
            v[u] = part[(size_t)(p < nparts ? p : p0) * (64 * 160) + idx];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (p0 + 16 * u < nparts) s += (double)v[u];
    }
    s_acc[slice][tx] = s;
    __syncthreads();
    if (slice == 0) {
        double t = 0.0;
        for (int j = 0; j < 16; ++j) t += s_acc[j][tx];
        const int co = idx / 160, k = idx - co * 160;
        if (k < 147) dw[co * 147 + k] = (float)t;           // OIHW: co*147 + c*49 + kh*7 + kw
    }
}

extern int g_grid_cap;
bool g_options_frozen = false;       // cova_set_option: the option state is fixed from the library's first query or launch on
inline int persistent_grid(int ntiles, int blocks_per_cu = 1)
{
    g_options_frozen = true;             // every size query and every launch of the big kernels comes through here
    // compute units of the CURRENT device (the caller launches under the device guard of its tensors; the host
    // queries run under the same guard): cached per device id, not once per process
    static int cus_of[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int cus = cus_of[dev];
    if (cus == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
        cus_of[dev] = cus;
    }
    int g = ntiles < cus * blocks_per_cu ? ntiles : cus * blocks_per_cu;
    if (g_grid_cap > 0 && g > g_grid_cap) g = g_grid_cap;
    return g;
}

int g_ablate = 0;            // conv3x3 v2 ablation mask (tools only)
int g_grid_cap = 0;          // > 0: cap on persistent grids (tests force many tiles per block)
int g_conv1_f32 = 0;         // 1: conv1 forward on the f32 MFMA kernel (v2) instead of the bf16-split one (A/B, tests)

}  // namespace

// shared with conv_wino.hip (same shared object; hidden visibility)
int cova_internal_persistent_grid(int ntiles) { return persistent_grid(ntiles); }
int cova_internal_persistent_grid2(int ntiles, int blocks_per_cu) { return persistent_grid(ntiles, blocks_per_cu); }

// ====================================================================================
// C ABI
// ====================================================================================
// test / tool hooks (not part of the path's contract): 2 = cap on persistent grids (tests force many tiles per
// block), 5 = ablation mask of builds made with -DCOVA_ABLATE (tools/conv_bench.py), 6 = Winograd tile geometry
int cova_internal_set_wino4_f32(int v);
int cova_internal_set_bn1d_variant(int v);
int cova_internal_set_gat_wide(int v);
// The option state is a per-process constant: mutable until the first query or launch, fixed afterwards (include/cova_hip.h) --
// unless the process opted in with COVA_ALLOW_OPTION_CHANGES=1 (tests, bench.py's A/B legs).
int cova_internal_get_bn1d_variant();
int cova_internal_get_gat_wide();
int cova_internal_get_wino4_f32();
int cova_internal_get_sgemm_dma();
int cova_internal_set_sgemm_dma(int v);
COVA_API int cova_set_option(int key, int value)
{
    static const bool allow = [] { const char *e = getenv("COVA_ALLOW_OPTION_CHANGES"); return e != nullptr && e[0] == '1'; }();
    int cur;
    switch (key) {
    case 2: cur = g_grid_cap; break;
    case 7: cur = g_conv1_f32; value = value != 0; break;
    case 9: cur = cova_internal_get_wino4_f32(); value = value != 0; break;
    case 14: cur = cova_internal_get_bn1d_variant(); break;
    case 16: cur = cova_internal_get_gat_wide(); value = value != 0; break;
    case 22: cur = cova_internal_get_sgemm_dma(); value = value != 0; break;
    default: return COVA_ERR_BAD_ARG;
    }
    if (cur == value) return COVA_OK;
    if (g_options_frozen && !allow) return COVA_ERR_BAD_ARG;
    switch (key) {
    case 2: g_grid_cap = value; return COVA_OK;
    case 7: g_conv1_f32 = value; return COVA_OK;
    case 9: return cova_internal_set_wino4_f32(value);
    case 14: return cova_internal_set_bn1d_variant(value);
    case 22: return cova_internal_set_sgemm_dma(value);
    default: return cova_internal_set_gat_wide(value);
    }
}

#ifdef C1B_TRACE
COVA_API int cova_wg1b_trace_read(unsigned long long *host)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wg1b_trace), sizeof(unsigned long long) * 8 * 20 * 8);
}
COVA_API int cova_c1b_trace_read(unsigned long long *host)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_c1b_trace), sizeof(unsigned long long) * 2 * 4 * 20 * 8);
}
#endif

COVA_API int cova_conv_out_size(int in_size, int kernel, int stride, int pad)
{
    return (in_size + 2 * pad - kernel) / stride + 1;
}

COVA_API int cova_conv1_num_tiles(int B, int H, int W);
static int conv1_fwd_tiles(int B, int H, int W)
{
    const int H1 = cova_conv_out_size(H, 7, 2, 3), W1 = cova_conv_out_size(W, 7, 2, 3);
    return g_conv1_f32 ? B * cdiv(H1, c1::TH) * cdiv(W1, c1::TW) : B * cdiv(H1, c1b::TH) * cdiv(W1, c1b::TW);
}

COVA_API int cova_conv1_num_partials(int B, int H, int W)
{
    return persistent_grid(conv1_fwd_tiles(B, H, W), g_conv1_f32 ? 2 : 1);
}

COVA_API int cova_conv1_num_tiles(int B, int H, int W)
{
    const int H1 = cova_conv_out_size(H, 7, 2, 3), W1 = cova_conv_out_size(W, 7, 2, 3);
    return B * cdiv(H1, c1::TH) * cdiv(W1, c1::TW);
}

COVA_API int cova_conv1_prep_weights(const float *w_oihw, float *w_k, void *stream)
{
    COVA_REQUIRE(w_oihw && w_k);
    hipLaunchKernelGGL(prep_w7x7_kernel, dim3(cdiv(c1::W_FLOATS, 256)), dim3(256), 0,
                       (hipStream_t)stream, w_oihw, w_k);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

static int conv1_fwd_launch(const float *img, const float *w_k, int w_oihw, float *out, float *stat_part, int B, int H,
                           int W, const cova_bn_tail *tail, void *stream)
{
    COVA_REQUIRE(img && w_k && out && B > 0 && H > 0 && W > 0);
    BnTail t{};
    if (tail != nullptr && tail->mode != 0) {
        t = *tail;
        COVA_REQUIRE(t.mode == 1 && stat_part && t.counter && t.count > 0 && t.gamma && t.beta && t.scale && t.shift &&
                     t.mean && t.invstd);
    }
    const int H1 = cova_conv_out_size(H, 7, 2, 3), W1 = cova_conv_out_size(W, 7, 2, 3);
    if (!g_conv1_f32) {
        const int tiles_x = cdiv(W1, c1b::TW), tiles_y = cdiv(H1, c1b::TH);
        const int ntiles = B * tiles_x * tiles_y;
        const dim3 pgrid(persistent_grid(ntiles, 1)), block(c1b::THREADS);
        if (stat_part)
            hipLaunchKernelGGL(conv1_7x7_bf3_kernel<true>, pgrid, block, 0, (hipStream_t)stream, img, w_k, out, stat_part,
                               H, W, H1, W1, tiles_x, tiles_y, ntiles, w_oihw, t);
        else
            hipLaunchKernelGGL(conv1_7x7_bf3_kernel<false>, pgrid, block, 0, (hipStream_t)stream, img, w_k, out,
                               stat_part, H, W, H1, W1, tiles_x, tiles_y, ntiles, w_oihw, t);
        COVA_LAUNCH_CHECK();
        return COVA_OK;
    }
    const int tiles_x = cdiv(W1, c1::TW), tiles_y = cdiv(H1, c1::TH);
    const dim3 block(c1::THREADS);
    const int ntiles = B * tiles_x * tiles_y;
    const dim3 pgrid(persistent_grid(ntiles, 2));
    if (stat_part)
        hipLaunchKernelGGL(conv1_7x7_v2_kernel<true>, pgrid, block, 0, (hipStream_t)stream, img,
                           w_k, out, stat_part, H, W, H1, W1, tiles_x, tiles_y, ntiles, g_ablate, w_oihw, t);
    else
        hipLaunchKernelGGL(conv1_7x7_v2_kernel<false>, pgrid, block, 0, (hipStream_t)stream, img,
                           w_k, out, stat_part, H, W, H1, W1, tiles_x, tiles_y, ntiles, g_ablate, w_oihw, t);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

COVA_API int cova_conv1_fwd(const float *img, const float *w_k, float *out, float *stat_part,
                            int B, int H, int W, void *stream)
{
    return conv1_fwd_launch(img, w_k, 0, out, stat_part, B, H, W, nullptr, stream);
}

// w_oihw: the [64,3,7,7] weight as the reference stores it (every block forms the kernel's K-pair layout itself while it
// stages the weights: no cova_conv1_prep_weights launch); tail (nullable): the BatchNorm finalize of the statistics
COVA_API int cova_conv1_fwd_tail(const float *img, const float *w_oihw, float *out, float *stat_part, int B, int H,
                                 int W, const cova_bn_tail *tail, void *stream)
{
    return conv1_fwd_launch(img, w_oihw, 1, out, stat_part, B, H, W, tail, stream);
}

COVA_API int cova_conv1_wgrad_workspace_floats(int B, int H, int W)
{
    return persistent_grid(cova_conv1_num_tiles(B, H, W)) * 8 * 64 * 160;
}

// conv1 weight gradient with the BatchNorm+ReLU+MaxPool backward apply folded into its operand:
//   dy1 = abc[0]*route(dp, idx) + abc[1]*y1 + abc[2]   (what cova_bn_relu_maxpool_bwd_apply would write)
// y1 NHWC [B,H1,W1,64] = conv1 output; dp NHWC [B,H2,W2,64] pooled gradient with the ReLU mask already
// applied; idx = arg-max codes of cova_bn_relu_maxpool_fwd; abc [3,64] from cova_bn_finalize_bwd_abc.
COVA_API int cova_conv1_wgrad_poolbwd(const float *img, const float *y1, const float *dp,
                                      const uint8_t *idx, const float *abc, float *dw, float *ws,
                                      int B, int H, int W, void *stream)
{
    COVA_REQUIRE(img && y1 && dp && idx && abc && dw && ws && B > 0 && H > 0 && W > 0);
    COVA_REQUIRE((long long)H * W * 64 < (1ll << 31));            // 32-bit in-image offsets in the kernel
    const int H1 = cova_conv_out_size(H, 7, 2, 3), W1 = cova_conv_out_size(W, 7, 2, 3);
    const int H2 = cova_conv_out_size(H1, 3, 2, 1), W2 = cova_conv_out_size(W1, 3, 2, 1);
    const int tiles_x = cdiv(W1, wg1::TW), tiles_y = cdiv(H1, wg1::TH);
    const int grid = persistent_grid(B * tiles_x * tiles_y);
    if (!g_conv1_f32) {
        const int rtx = cdiv(W1, wg1r::TW), rty = cdiv(H1, wg1r::TH);
        const int rgrid = persistent_grid(B * rtx * rty);
        hipLaunchKernelGGL(conv1_wgrad_rs_kernel<true>, dim3(rgrid), dim3(wg1r::THREADS), 0, (hipStream_t)stream, img, y1,
                           ws, H, W, H1, W1, rtx, rty, B * rtx * rty, PoolBwd{dp, idx, abc, H2, W2});
        COVA_LAUNCH_CHECK();
        hipLaunchKernelGGL(conv1_wgrad_reduce_kernel, dim3(64 * 160 / 64), dim3(1024), 0, (hipStream_t)stream, ws,
                           rgrid * 2, dw);
        COVA_LAUNCH_CHECK();
        return COVA_OK;
    }
    hipLaunchKernelGGL(conv1_wgrad_v2_kernel<true>, dim3(grid), dim3(wg1::THREADS), 0,
                       (hipStream_t)stream, img, y1, ws, H, W, H1, W1, tiles_x, tiles_y,
                       B * tiles_x * tiles_y, PoolBwd{dp, idx, abc, H2, W2});
    COVA_LAUNCH_CHECK();
    hipLaunchKernelGGL(conv1_wgrad_reduce_kernel, dim3(64 * 160 / 64), dim3(1024), 0,
                       (hipStream_t)stream, ws, grid * 4, dw);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

// img NCHW [B,3,H,W]; dy NHWC [B,H1,W1,64]; dw OIHW [64,3,7,7]
COVA_API int cova_conv1_wgrad(const float *img, const float *dy, float *dw, float *ws, int B, int H,
                              int W, void *stream)
{
    COVA_REQUIRE(img && dy && dw && ws && B > 0 && H > 0 && W > 0);
    COVA_REQUIRE((long long)H * W * 64 < (1ll << 31));            // 32-bit in-image offsets in the kernel
    const int H1 = cova_conv_out_size(H, 7, 2, 3), W1 = cova_conv_out_size(W, 7, 2, 3);
    const int tiles_x = cdiv(W1, wg1::TW), tiles_y = cdiv(H1, wg1::TH);
    const int grid = persistent_grid(B * tiles_x * tiles_y);
    if (!g_conv1_f32) {
        const int rtx = cdiv(W1, wg1r::TW), rty = cdiv(H1, wg1r::TH);
        const int rgrid = persistent_grid(B * rtx * rty);
        hipLaunchKernelGGL(conv1_wgrad_rs_kernel<false>, dim3(rgrid), dim3(wg1r::THREADS), 0, (hipStream_t)stream, img, dy,
                           ws, H, W, H1, W1, rtx, rty, B * rtx * rty, PoolBwd{nullptr, nullptr, nullptr, 0, 0});
        COVA_LAUNCH_CHECK();
        hipLaunchKernelGGL(conv1_wgrad_reduce_kernel, dim3(64 * 160 / 64), dim3(1024), 0, (hipStream_t)stream, ws,
                           rgrid * 2, dw);
        COVA_LAUNCH_CHECK();
        return COVA_OK;
    }
    hipLaunchKernelGGL(conv1_wgrad_v2_kernel<false>, dim3(grid), dim3(wg1::THREADS), 0,
                       (hipStream_t)stream, img, dy, ws, H, W, H1, W1, tiles_x, tiles_y,
                       B * tiles_x * tiles_y, PoolBwd{nullptr, nullptr, nullptr, 0, 0});
    COVA_LAUNCH_CHECK();
    hipLaunchKernelGGL(conv1_wgrad_reduce_kernel, dim3(64 * 160 / 64), dim3(1024), 0,
                       (hipStream_t)stream, ws, grid * 4, dw);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}
