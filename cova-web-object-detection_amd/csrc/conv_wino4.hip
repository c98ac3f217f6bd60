// Winograd F(4x4, 3x3) form of the 3x3 / 64->64 convolution (forward and data gradient): 36 multiplications per
// 4x4 output tile and channel pair instead of 144 (F(2x2,3x3): 64) -- 1.78x fewer MFMAs than conv_wino.hip.
// fp32 error on this layer (post-ReLU input, 64 channels, against fp64): max 2.9e-6 of the output scale
// (F(2x2,3x3) 1.9e-7, direct 3.0e-7), inside the 1e-4 forward / 2e-4 gradient gates of the parity tests.
//
//   Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A          B^T 6x6, G 6x3, A^T 4x6 (Lavin & Gray 2015)
//
// Mapping (round 3; the round-2 kernel kept both MFMA operands in LDS and was bound by LDS bandwidth and by two block
// barriers per 4-channel chunk):
//   * block = 512 threads, output tile 8 x 32 px = 2 x 8 Winograd tiles (16 tiles = one MFMA column block);
//   * the 36 transform positions are SPLIT between the two waves of a SIMD: wave w = (channel group cog = w & 3: 16
//     output channels, position half ph = w >> 2: transform rows i = 3 ph .. 3 ph + 2) holds 18 accumulators of
//     v_mfma_f32_16x16x4_f32 = 72 registers.  The output transform is separable, A^T M A = sum over the two row halves,
//     so each wave transforms its half in registers and the pair swaps half of the 16 partial outputs through LDS once
//     per tile (8 KB per wave);
//   * with 72 accumulator registers there is room to keep the transformed WEIGHTS out of LDS altogether: a wave's
//     operand U[16 co][8 ci][18 positions] is 36 registers per 8-channel chunk, fetched straight from L2
//     (9 global_load_dwordx4 per wave and chunk, 1 KB contiguous each: the prep kernel writes the register image), each
//     quad re-requested for the next chunk right after its last MFMA (a two-chunk lead, 72 registers, measured no
//     faster);
//   * the transformed input V = B^T d B is formed once per block through LDS in two stages (column stage, row stage:
//     768 + 768 six-point transforms per 8-channel chunk, three per thread), from planes that arrive as
//     global -> LDS copies (global_load_lds_dwordx4: no registers), 16 channels = 64 contiguous bytes per pixel and
//     lane quad (16 bytes per pixel and lane made every wave instruction touch 64 cache lines: -10 % on the launch);
//   * ONE block barrier per 8-channel chunk: column stage (chunk g+2), row stage (chunk g+1) and the 36 MFMAs of chunk g
//     run in the same iteration on double-buffered intermediates.
// LDS traffic per 8-channel chunk and 16 tiles: 74 KB of V operand reads + 73 KB of stage traffic + 11 KB of copies
// (round 2: 147 + 73 + 47 KB per 4 channels x 32 tiles).
#include "common.h"
#include "bf3.h"
#include "bn_tail.h"
#include <type_traits>

// Ablation builds of tools/wino4_bench.py (COVA_EXTRA_FLAGS=-DW4_ABL=<mask>; 0 in the product): 1 no column stage,
// 2 no row stage, 4 no MFMAs, 8 no plane copies, 16 no weight loads (stale registers), 32 no tile epilogue,
// 64 no wait for the plane copies, 128 weight loads always from chunk 0, 256 no output stores, 512 no exchange barriers,
// 2048 plane copies always from the same rows of the map (L2-resident source)
#ifndef W4_ABL
#define W4_ABL 0
#endif

// -DW4_TRACE (tools only): s_memtime stamps of waves 0 and 4 (the two position halves on one SIMD) of block 0 over the
// sixteen iterations + two tile epilogues that start at chunk 32 of its stream
#ifdef W4_TRACE
__device__ unsigned long long g_w4_trace[2 * 20 * 8];
#define W4_STAMP(row, slot)                                                                                  \
    do {                                                                                                     \
        const int r__ = (row);                                                                               \
        if (blockIdx.x == 0 && (wave == 0 || wave == 4) && r__ >= 0 && r__ < 20 && lane == 0)                \
            g_w4_trace[((wave >> 2) * 20 + r__) * 8 + (slot)] = __builtin_amdgcn_s_memtime();                \
    } while (0)
#else
#define W4_STAMP(row, slot) do { } while (0)
#endif

namespace {

namespace w4 {
constexpr int TH = 8, TW = 32, PH = TH + 2, PW = TW + 2, NPIX = PH * PW;      // 340 halo pixels
constexpr int THREADS = 512;
constexpr int SLOT_FLOATS = NPIX * 16;         // planes of one 16-channel group (two chunks), pixel-major [px][16]: 21,760 B
constexpr int TMP_CI = 16 * 36 + 4;            // column-stage result [ci][tile][c][i]; +4: the channels of an
constexpr int TMP_FLOATS = 8 * TMP_CI;         // instruction start 4 banks apart; 18,560 B
constexpr int VROW = 36;                       // V row (ci, tile): [positions 0-15 of half 0 | 0-15 of half 1 | 16-17 of half 0 |
                                               // 16-17 of half 1] (half 0 = transform rows i 0,1,2; half 1 = i 5,3,4): both
                                               // halves read 4 x b128 + 1 x b64 at 16-byte-aligned offsets; 9 x 16 B is an odd
                                               // slot stride: conflict-free ds_read_b128
constexpr int V_FLOATS = 8 * 16 * VROW;        // 18,432 B
constexpr int U_FLOATS = 8 * 8 * 9 * 64 * 4;   // [chunk 8][wave 8][quad 9][lane 64][4]: 147,456 floats
constexpr int U_SPLIT_DWORDS = 2 * 8 * 18 * 3 * 256;     // the bf16-piece image of the split main loop (conv_wino4_split.h): 221,184
constexpr int U_TOTAL = U_FLOATS + 2 * U_SPLIT_DWORDS;    // one convolution, one direction: [f32 image | piece image | piece image of -U] = 589,824 floats
}  // namespace w4

__device__ __forceinline__ f32x4 mfma16x4(float a, float b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// B^T x for one 6-vector (12 operations)
__device__ __forceinline__ void bt6(const float (&d)[6], float (&o)[6])
{
    const float a = fmaf(-4.f, d[2], d[4]), b = fmaf(-4.f, d[1], d[3]);
    const float c = d[4] - d[2], e = 2.f * (d[3] - d[1]);
    o[0] = fmaf(4.f, d[0], fmaf(-5.f, d[2], d[4]));
    o[1] = a + b;
    o[2] = a - b;
    o[3] = c + e;
    o[4] = c - e;
    o[5] = fmaf(4.f, d[1], fmaf(-5.f, d[3], d[5]));
}

// A^T m for one 6-vector -> 4 outputs
__device__ __forceinline__ void at6(const float (&m)[6], float (&o)[4])
{
    const float s1 = m[1] + m[2], d1 = m[1] - m[2], s2 = m[3] + m[4], d2 = m[3] - m[4];
    o[0] = m[0] + s1 + s2;
    o[1] = fmaf(2.f, d2, d1);
    o[2] = fmaf(4.f, s2, s1);
    o[3] = fmaf(8.f, d2, d1) + m[5];
}

// Epilogue of the data-gradient launches (same contract as conv_wino.hip's BnBwdEpiW): ReLU mask + BatchNorm-backward
// sums.  The mask is act > 0, or, when the activation was never materialised (act == nullptr),
// fma(msc, z, msh) > 0 -- the same expression the affine-on-load prologues evaluate.
struct W4Epi {
    const float *addend, *act, *z, *mean, *invstd, *msc, *msh;
    const uint32_t *act_bits;       // BN == 2: the mask as one bit per element ([pixel][2] words) instead of act
    int inf_relu;                   // BN == 3 (inference epilogue  out = f(msc*y + msh + addend)): f = ReLU
    int inference;                  // 1: the inference epilogue (set by cova_conv3x3_wino4_bnact only)
};

struct W4Args {
    const float *in, *in2, *u;
    float *out, *stat_part;
    int H, W, tiles_x, tiles_y, ntiles;
    const float *pro_abc;       // PRO: the conv input is f(A[c]*in + B[c]*in2 + C[c]) ([3][64] = A | B | C), zero outside
    int pro_relu;               // the image; f = ReLU or identity
    W4Epi epi;
    BnTail tail;                // BatchNorm finalize of stat_part by the last block (mode 0: none)
};

// 256 zero bytes in global memory: halo pixels outside the image are fetched from here
__device__ __attribute__((aligned(256))) float g_w4_zero_page[64];

typedef __attribute__((address_space(3))) void lds_void;

// Sum over the 16 lanes of a wave that share (lane & 3); every lane ends up with its class's total.  Two v_add_f32 with
// DPP operands inside the 16-lane rows (row_ror:4, row_ror:8), two cross-row exchanges.
__device__ __forceinline__ float quad_class_sum(float v)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

// global -> LDS copy of 16 bytes per lane (lane i lands at lds_base + 16 i; lds_base wave-uniform) as inline asm:
// the compiler then neither knows the copy (no conservative vmcnt(0) in front of every later LDS read, which the
// builtin gets unless each buffer is its own __shared__ object) nor waits for it -- every wait on these copies is
// an explicit s_waitcnt in the kernel.
// Address = wave-uniform base (scalar register pair) + this lane's unsigned 32-bit byte offset: no 64-bit per-lane
// pointers (which the compiler otherwise precomputes per pixel and keeps -- or spills -- across the whole loop).
__device__ __forceinline__ void copy16_to_lds(const float *base, unsigned off_bytes, unsigned lds_base_bytes)
{
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_base_bytes), "v"(off_bytes), "s"(base) : "memory", "m0");
}

// Barrier that orders LDS accesses only: this wave's ds_writes are complete (lgkmcnt(0)) while global -> LDS copies
// and weight loads stay in flight.  __syncthreads() is a release fence over LDS, and a pending copy IS an LDS store:
// it waits vmcnt(0).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Software pipeline, one iteration = one 8-channel chunk g of the block's chunk stream (8 chunks per tile, the stream
// runs across tiles):
//   even g: request the planes of the 16-channel group of chunks g+4, g+5 as global -> LDS copies (two slots per tensor) |
//   column stage(g+2): planes -> tmp[g & 1] | row stage(g+1): tmp[(g+1) & 1] -> V[(g+1) & 1] | 36 MFMAs of chunk g from
//   V[g & 1] and the weight registers of chunk g, each weight quad re-requested for chunk g+1 right after its last use |
//   odd g: wait for the copies issued at g-1 | barrier.
// Every wave interleaves its three stage items with its MFMAs (reads, 16 MFMAs, transforms, 20 MFMAs; the prologue
// variants one item at a time).  vmcnt retires in order, so the wait in front of a weight quad's first use also waits
// for older plane copies (measured: giving the copies two full iterations through exact wait counts changes nothing).
// BN: 0 plain (+ statistics if STATS), 1 = ReLU mask from z + BatchNorm-backward sums, 2 = mask from act + sums,
//     3 = inference: out = f(msc*y + msh (+ addend)), the BatchNorm (running statistics) + residual + ReLU that follow
//         the conv in a BasicBlock (same expression and order as cova_bn_act_fwd), no statistics.
template <bool STATS, int PRO, bool ADD, int BN>
__global__ __launch_bounds__(w4::THREADS, 1) void conv3x3_c64_wino4_kernel(const W4Args a)
{
    using namespace w4;
    // Arguments that are needed once per tile (epilogue operands, output) or once per launch (BatchNorm tail) are read
    // from the kernel-argument segment WHERE they are used, through a pointer the compiler cannot see through: loaded
    // at entry like the rest they would pin ~50 scalar registers for the whole kernel, and the scalar spills that follow
    // cost vector registers in the main loop.
    typedef const W4Args __attribute__((address_space(4))) *KArgs;
    auto late_args = [&]() -> KArgs {
        unsigned long long kp = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        return (KArgs)kp;
    };
    // Plane copies bring 16 channels = 64 contiguous bytes of every pixel (four adjacent lanes per pixel), one copy per TWO
    // chunks, two slots per tensor.  (Round 3's first version copied 8-channel chunks as 2 x 16 bytes per pixel: 16 bytes out
    // of a 256-byte pixel per lane make every wave instruction touch 64 different cache lines for 1 KB of data -- measured
    // 0.049 of the plain launch's 0.47 ms against contiguous reads.)
    constexpr int NSLOT = 2;
    constexpr int NT = PRO == 2 ? 2 : 1;          // tensors staged
    constexpr int IN_FLOATS = NSLOT * SLOT_FLOATS;
    // tmp[0] and V[1] are adjacent: both are idle at a tile boundary and carry the pair exchange of the output transform
    __shared__ __attribute__((aligned(16))) float lds[NT * IN_FLOATS + 2 * TMP_FLOATS + 2 * V_FLOATS];
    __shared__ __attribute__((aligned(16))) float s_epi[BN ? 256 : 4];   // mean | invstd | mask scale | mask shift of the epilogue
    __shared__ float s_pro[PRO ? 192 : 1];        // A | B | C of the affine-on-load prologue
    __shared__ __attribute__((aligned(16))) float s_red[STATS ? 8 * 32 : 4];   // per wave: 16 channels x (sum | second kind), running totals
    float *s_in = lds;
    float *s_in2 = lds + IN_FLOATS;               // (PRO == 2)
    float *s_tmp1 = lds + NT * IN_FLOATS;
    float *s_tmp0 = s_tmp1 + TMP_FLOATS;
    float *s_v1 = s_tmp0 + TMP_FLOATS;
    float *s_v0 = s_v1 + V_FLOATS;
    float *s_x = s_tmp0;                          // exchange area: tmp[0] + V[1] = 41 KB >= 32 KB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cog = wave & 3, ph = wave >> 2;
    const int l15 = lane & 15, kq = lane >> 4;
    const int H = a.H, W = a.W;
    if (PRO && tid < 192) s_pro[tid] = late_args()->pro_abc[tid];
    if (BN && tid >= 256) {
        const int c = tid & 63, kind = (tid >> 6) & 3;
        const KArgs la = late_args();
        const float *src = kind == 0 ? la->epi.mean : kind == 1 ? la->epi.invstd : kind == 2 ? la->epi.msc : la->epi.msh;
        const bool used = BN == 3 ? kind >= 2 : (kind < 2 || BN == 1);       // (unused pointers may be NULL)
        s_epi[kind * 64 + c] = used ? src[c] : 0.f;
    }
    if (PRO || BN) __syncthreads();

    // ---- the tiles of this block (XCD-aware order: consecutive blocks of an XCD take consecutive tiles)
    int first = blockIdx.x;
    if ((gridDim.x & 7) == 0) first = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (first >= a.ntiles) return;
    const int nk = (a.ntiles - first + (int)gridDim.x - 1) / (int)gridDim.x;      // tiles of this block
    const int G = nk * 8;                                                          // chunks of this block
    auto tile_of = [&](int k) { return first + (k < nk ? k : nk - 1) * (int)gridDim.x; };   // past the end: a valid tile

    // ---- plane copies.  A group = 16 channels x 340 pixels = 22 wave instructions of 16 pixels x 64 B: instruction j covers
    // pixels 16 j .. 16 j + 15 (lane = pixel * 4 + 16-byte piece); wave w issues j = w, w + 8, w + 16 (j < 22).
    // Source of this thread's pixels: (tile base: wave-uniform, scalar registers) + (the pixel's fixed byte offset from
    // the tile's halo origin: one register per pixel) when the pixel lies inside the image, the zero page otherwise --
    // two copy instructions under complementary lane masks (the second one is skipped by interior tiles).  The in-image
    // flags are per tile (lane masks).
    const unsigned in_base = (unsigned)(size_t)(lds_void *)s_in, in2_base = (unsigned)(size_t)(lds_void *)s_in2;
    const float *tbase = a.in, *tbase2 = a.in;         // element (ty*TH - 1, tx*TW - 1, channel 0) of the current tile's image
    unsigned rel16[3] = {0, 0, 0};
    bool in16[3] = {false, false, false};
    const int q16 = lane & 3;
    auto px16 = [&](int i) { return 16 * (wave + 8 * i) + (lane >> 2); };
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int px = px16(i), r = px / PW, c = px - r * PW;
        rel16[i] = (unsigned)((r * W + c) * 64 + 4 * q16) * 4u;
    }
    auto plane_src = [&](int k) {
        const int tile_ = tile_of(k);
        const int tx = tile_ % a.tiles_x, ty = (tile_ / a.tiles_x) % a.tiles_y, b = tile_ / (a.tiles_x * a.tiles_y);
        long long org = ((long long)b * H * W + (long long)(ty * TH - 1) * W + (tx * TW - 1)) * 64;
        if (W4_ABL & 2048) org = ((long long)(1 + (blockIdx.x & 7)) * 16 * W + 33) * 64;      // copies always from the same few rows (L2 hits)
        tbase = a.in + org;
        if (PRO == 2) tbase2 = a.in2 + org;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            // (row, column) of the pixel recomputed here, once per tile, behind an opaque move: hoisted out of the loop
            // they would be six more registers live (or spilled) across it
            int px = px16(i);
            asm volatile("" : "+v"(px));
            const int r = (px * 241) >> 13, c = px - r * PW;          // px / 34 for px < 352
            const int gy = ty * TH + r - 1, gx = tx * TW + c - 1;
            in16[i] = gy >= 0 && gy < H && gx >= 0 && gx < W;
        }
    };
    auto copy_group = [&](int gi, int slot) {        // 16-channel group gi (0..3) of the tile plane_src() was called for
        if (W4_ABL & 8) return;
        const unsigned sb = (unsigned)slot * (SLOT_FLOATS * 4);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int j = wave + 8 * i;
            if (j < 22 && px16(i) < NPIX) {
                if (in16[i]) {
                    copy16_to_lds(tbase + 16 * gi, rel16[i], in_base + sb + j * 1024);
                    if (PRO == 2) copy16_to_lds(tbase2 + 16 * gi, rel16[i], in2_base + sb + j * 1024);
                } else {                                            // (the zero page holds 64 floats)
                    copy16_to_lds(g_w4_zero_page, (unsigned)q16 * 16u, in_base + sb + j * 1024);
                    if (PRO == 2) copy16_to_lds(g_w4_zero_page, (unsigned)q16 * 16u, in2_base + sb + j * 1024);
                }
            }
        }
    };
    // Copies are issued at even iterations only; the group issued at the top of iteration g-1 must have landed at the end
    // of the odd iteration g: behind it in the (in-order) queue are the 9 + 9 weight loads of the two iterations.
    auto wait_planes = [&](int g) {
        if (W4_ABL & 64) return;
        if (g & 1) __builtin_amdgcn_s_waitcnt(0x0F70 | (18 & 15) | ((18 >> 4) << 14));
    };

    // ---- stage items of this thread: three per iteration, read in one batch (one LDS round trip per iteration).
    // column stage, 768 items it = ((kg*16 + tile)*6 + c)*4 + cl (the channel fastest, as the planes are [px][4]);
    // row stage, 768 items it = ((ci*16 + tile)*6 + i (the channel slowest, as V rows are (ci, tile)).
    // Waves 0-3: column items tid, 512 + tid and row item tid; waves 4-7: column item tid, row items tid, 256 + tid.
    const bool lo = wave < 4;                       // (wave-uniform: scalar branches)
    int col_src[2], col_dst[2], col_ch[2], row_src[2], row_dst[2], row_dst2[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        {      // it = ((tile*6 + c)*2 + kg)*4 + cl: 32 lanes spread over 16 banks of the [px][16] planes
            const int it = tid + e * THREADS;
            const int cl = it & 3, kg = (it >> 2) & 1, tc = it >> 3, t = tc / 6, c = tc - t * 6;
            col_src[e] = ((4 * (t >> 3)) * PW + 4 * (t & 7) + c) * 16 + kg * 4 + cl;              // + r * PW * 16 + 8 (chunk & 1)
            col_dst[e] = (kg * 4 + cl) * TMP_CI + t * 36 + c * 6;                                 // + i (6 contiguous)
            col_ch[e] = kg * 4 + cl;
        }
        {
            const int it = tid + e * 256;
            const int i = it % 6, rt = it / 6, t = rt & 15, ci = rt >> 4;
            row_src[e] = ci * TMP_CI + t * 36 + i;                                                // + c * 6
            // transform row i -> (half, row within the half): half 1 in the order i = 5, 3, 4; position p = il*6 + j of a
            // half sits at float 16 half + p (p < 16) or 32 + 2 half + (p - 16): j 0..3 contiguous, j 4..5 possibly apart
            const int half = i < 3 ? 0 : 1, il = i < 3 ? i : (i - 2) % 3;
            row_dst[e] = (ci * 16 + t) * VROW + 16 * half + 6 * il;
            row_dst2[e] = (ci * 16 + t) * VROW + (il < 2 ? 16 * half + 6 * il + 4 : 32 + 2 * half);
        }
    }
    // PRO: rows of an item's column that lie inside the image (bit r), for the tile the planes belong to
    unsigned vm[2] = {0u, 0u};
    auto row_masks = [&](int k) {
        const int tile_ = tile_of(k);
        const int tx_ = tile_ % a.tiles_x, ty_ = (tile_ / a.tiles_x) % a.tiles_y;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            // image column / first row of column item e (recomputed from the thread index: once per tile)
            const int it = tid + e * THREADS, tc = it >> 3, tt = tc / 6, c = tc - tt * 6, t = tt & 15;
            const int gx = tx_ * TW + 4 * (t & 7) + c - 1, gy = ty_ * TH + 4 * (t >> 3) - 1;
            unsigned m = 0;
#pragma unroll
            for (int r = 0; r < 6; ++r) m |= (gy + r >= 0 && gy + r < H) ? (1u << r) : 0u;
            vm[e] = (gx < 0 || gx >= W) ? 0u : m;
        }
    };
    // One stage item in flight: its six inputs (plus the six of the second tensor), where the result goes, and whether
    // the affine prologue applies (column items).  Reads and transform are separate steps so that MFMAs sit between them.
    struct Item { float d[6], w[6]; float *q, *q2; };       // q: outputs 0..3, q2: outputs 4..5
    // (slot, coff): where the chunk's planes are -- the slot of its 16-channel group, coff = 8 * (chunk & 1)
    auto read_col = [&](int e, int slot, int coff, float *tmp_w, Item &it) {
        constexpr int RS = PW * 16;
        const float *p = s_in + slot * SLOT_FLOATS + coff + col_src[e];
#pragma unroll
        for (int r = 0; r < 6; ++r) it.d[r] = p[r * RS];
        if (PRO == 2) {
            const float *p2 = s_in2 + slot * SLOT_FLOATS + coff + col_src[e];
#pragma unroll
            for (int r = 0; r < 6; ++r) it.w[r] = p2[r * RS];
        }
        it.q = tmp_w + col_dst[e];
        it.q2 = it.q + 4;
    };
    auto read_row = [&](int e, const float *tmp_r, float *v_w, Item &it) {
        const float *p = tmp_r + row_src[e];
#pragma unroll
        for (int c = 0; c < 6; ++c) it.d[c] = p[c * 6];
        it.q = v_w + row_dst[e];
        it.q2 = v_w + row_dst2[e];
    };
    // affine prologue of column item e (chunk s of its tile: channel 8 s + ci); zero padding stays zero
    auto pro_apply = [&](int e, int s, Item &it) {
        const int ch = 8 * s + col_ch[e];
        const float A = s_pro[ch], C = s_pro[128 + ch];
        const float Bc = PRO == 2 ? s_pro[64 + ch] : 0.f;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            float v = fmaf(A, it.d[r], PRO == 2 ? fmaf(Bc, it.w[r], C) : C);      // same expression as conv_wino.hip's prologue
            if (a.pro_relu) v = fmaxf(v, 0.f);
            it.d[r] = ((vm[e] >> r) & 1u) ? v : 0.f;
        }
    };
    // tmp[ci][tile][c][i] = sum_r Bt[i][r] d[r][c]  /  V[(ci, tile)][i][j] = sum_c tmp[c][i] Bt[j][c]
    auto transform_store = [&](const Item &it) {
        float o[6];
        bt6(it.d, o);
        *reinterpret_cast<float2 *>(it.q) = make_float2(o[0], o[1]);
        *reinterpret_cast<float2 *>(it.q + 2) = make_float2(o[2], o[3]);
        *reinterpret_cast<float2 *>(it.q2) = make_float2(o[4], o[5]);
    };
    // the three items of an iteration: k = 0 column item 0 | k = 1 column item 1 (waves 0-3) or row item 0 (waves 4-7) |
    // k = 2 row item 0 (waves 0-3) or row item 1.  `lo` is wave-uniform: scalar branches (the empty asm keeps the
    // compiler from turning the arms into per-lane address selects)
    auto item_read = [&](const int k, Item &it, int slot, int coff, float *tmp_w, const float *tmp_r, float *v_w) {
        if (W4_ABL & 3) return;
        if (k == 0) {
            read_col(0, slot, coff, tmp_w, it);
        } else if (k == 1) {
            if (lo) { asm volatile(""); read_col(1, slot, coff, tmp_w, it); }
            else { asm volatile(""); read_row(0, tmp_r, v_w, it); }
        } else {
            if (lo) { asm volatile(""); read_row(0, tmp_r, v_w, it); }
            else { asm volatile(""); read_row(1, tmp_r, v_w, it); }
        }
    };
    auto item_finish = [&](const int k, Item &it, int s) {
        if (W4_ABL & 3) return;
        if (PRO) {
            if (k == 0) pro_apply(0, s, it);
            else if (k == 1 && lo) pro_apply(1, s, it);
        }
        transform_store(it);
    };

    // ---- weights: 9 float4 per lane and chunk (the prep kernel wrote the register image: wave-contiguous 1 KB rows);
    // wave-uniform base (scalar registers) + this lane's fixed 32-bit byte offset + immediate
    const char *ubase = reinterpret_cast<const char *>(a.u) + (size_t)wave * (9 * 1024);
    const unsigned ulane = (unsigned)lane * 16u;
    auto u_ptr = [&](int s) { return ubase + (size_t)((W4_ABL & 128) ? 0 : (s & 7)) * (8 * 9 * 1024); };
    // Lead of the weights: every quad is re-requested for the NEXT chunk right after its last MFMA (36 registers: E =
    // quads 0-4, L = quads 5-8).  Measured against it: quads 0-4 two chunks ahead (56 registers) and all quads two chunks
    // ahead (72) -- no faster; the plane copies, forced to completion by the first wait on a younger load (vmcnt is
    // in-order), have one to two iterations of flight either way.
    float Ea[20], L[16];
    auto load_quads = [&](float *U, int s, const int q0, const int q1, const int base) {
        const char *p = u_ptr(s);
#pragma unroll
        for (int q = q0; q < q1; ++q) {
            const float4 t = *reinterpret_cast<const float4 *>(p + q * 1024 + ulane);
            U[4 * q - base] = t.x; U[4 * q + 1 - base] = t.y; U[4 * q + 2 - base] = t.z; U[4 * q + 3 - base] = t.w;
        }
    };

    // ---- prime the pipeline: the planes of groups 0 and 1 (chunks 0..3) copied, the first weights requested
    plane_src(0);
    copy_group(0, 0);
    copy_group(1, 1);
    load_quads(Ea, 0, 0, 5, 0);
    load_quads(L, 0, 5, 9, 20);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (PRO) row_masks(0);
    {   // column stage of chunks 0 and 1, row stage of chunk 0
        Item it;
        for (int c = 0; c < 2; ++c) {
            float *tw = c ? s_tmp1 : s_tmp0;
            const int sl = 0, coff = 8 * c;
            read_col(0, sl, coff, tw, it);
            if (PRO) pro_apply(0, c, it);
            transform_store(it);
            if (lo) {
                read_col(1, sl, coff, tw, it);
                if (PRO) pro_apply(1, c, it);
                transform_store(it);
            }
        }
        __syncthreads();
        read_row(0, s_tmp0, s_v0, it);
        transform_store(it);
        if (!lo) { read_row(1, s_tmp0, s_v0, it); transform_store(it); }
    }
    __syncthreads();

    if (STATS && tid < 256) s_red[tid] = 0.f;       // (ordered before its first use by the barriers of the first tile)
    f32x4 acc[18];
#pragma unroll
    for (int p = 0; p < 18; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
    // V operand rows of this lane: (ci = kg*4 + kq, tile l15), floats ph*20 .. ph*20 + 17
    const int v_off = (kq * 16 + l15) * VROW;

    // one iteration; PAR = g & 1 (compile time: buffers and weight registers are static)
    auto iteration = [&](const int g, auto par, float (&E)[20]) {
        constexpr int PAR = decltype(par)::value;
        float *tmp_w = PAR ? s_tmp1 : s_tmp0;               // column stage (chunk g+2) writes tmp[g & 1]
        const float *tmp_r = PAR ? s_tmp0 : s_tmp1;         // row stage (chunk g+1) reads tmp[(g+1) & 1]
        float *v_w = PAR ? s_v0 : s_v1;                     //   and writes V[(g+1) & 1]
        const float *v_r = (PAR ? s_v1 : s_v0) + v_off;     // MFMAs of chunk g read V[g & 1]
        // group (g+4) >> 1 (chunks g+4, g+5) -> the slot of group (g >> 1), consumed by the last two column stages
        // (issued behind the first 16 MFMAs instead -- three LDS-DMA instructions stall the wave ~450 cycles, tools/w4_trace.py --
        //  the launch takes exactly as long: 0.492 vs 0.493 ms; the stall moves, the iteration does not shrink)
        if (!(g & 1)) {
            if (((g + 4) & 7) == 0) plane_src((g + 4) >> 3);
            copy_group(((g + 4) & 7) >> 1, (g >> 1) & 1);
        }
        const int slot2 = ((g + 2) >> 1) & 1, coff2 = 8 * (g & 1);      // the planes of chunk g + 2 (this iteration's column stage)
        if (PRO && ((g + 2) & 7) == 0) row_masks((g + 2) >> 3);
        const int s2 = (g + 2) & 7;
        // wave-uniform weight bases of chunk g + 1 in scalar registers (global_load with an SGPR base + this lane's offset);
        // one base per 4 KB of immediate-offset range
        // (the opaque offsets keep the compiler from re-associating the bases into one base + offsets beyond the
        //  immediate range; the pointers themselves stay derived from the kernel argument: global, not flat, loads)
        unsigned o4k = 4096, o8k = 8192;
        asm volatile("" : "+s"(o4k), "+s"(o8k));
        const char *ue = u_ptr(g + 1), *ue1 = ue + o4k, *ul1 = u_ptr(g + 1) + o4k, *ul2 = u_ptr(g + 1) + o8k;
        unsigned ul = ulane;
        asm volatile("" : "+v"(ul));         // keeps (uniform base) + (lane offset) apart: LICM would fold the lane offset into
                                             // a 64-bit VGPR base and every load would need vector address arithmetic
        float v[2][18];
        auto read_v = [&](int kg) {
            if (W4_ABL & 4) return;
            const float *vr = v_r + kg * 64 * VROW;
            const float *vh = vr + 16 * ph;           // positions 0-15 of this wave's half, then 16-17 behind both halves
            const float4 v0 = *reinterpret_cast<const float4 *>(vh), v1 = *reinterpret_cast<const float4 *>(vh + 4);
            const float4 v2 = *reinterpret_cast<const float4 *>(vh + 8), v3 = *reinterpret_cast<const float4 *>(vh + 12);
            const float2 v4 = *reinterpret_cast<const float2 *>(vr + 32 + 2 * ph);
            const float t[18] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w,
                                 v3.x, v3.y, v3.z, v3.w, v4.x, v4.y};
#pragma unroll
            for (int p = 0; p < 18; ++p) v[kg][p] = t[p];
        };
        // MFMAs idx0 .. idx1-1 of the chunk (idx = kg*18 + p); a weight quad is re-requested right after its last use, so
        // the nine loads of a chunk are spread between the MFMAs.
        auto mfma_range = [&](const int idx0, const int idx1) {
            if (W4_ABL & 4) return;
#pragma unroll
            for (int idx = idx0; idx < idx1; ++idx) {
                const int kg = idx / 18, p = idx - 18 * kg;
                acc[p] = mfma16x4(idx < 20 ? E[idx] : L[idx - 20], v[kg][p], acc[p]);      // D[co][tile] += U[co][ci] * V[ci][tile]
                if ((idx & 3) == 3 && !(W4_ABL & 16)) {
                    const int q = idx >> 2;
                    const char *bp = q < 4 ? ue : q == 4 ? ue1 : q < 8 ? ul1 : ul2;
                    const float4 t = *reinterpret_cast<const float4 *>(bp + (q & 3) * 1024 + ul);
                    float *U = q < 5 ? &E[4 * q] : &L[4 * q - 20];
                    U[0] = t.x; U[1] = t.y; U[2] = t.z; U[3] = t.w;
                }
            }
        };
        // Order: the prologue variants keep ONE stage item in flight (register pressure), its reads issued four MFMAs before
        // its transform, the second V group requested before the last item's transform; the plain variant reads all three
        // items up front (measured equal or faster there).  The compiler's schedule inside these groups measured equal
        // to one pinned with scheduling barriers.
        W4_STAMP(g - 32, 0);
        if constexpr (PRO != 0) {
        Item it;
        read_v(0);
        mfma_range(0, 4);
        item_read(0, it, slot2, coff2, tmp_w, tmp_r, v_w);
        mfma_range(4, 8);
        item_finish(0, it, s2);
        item_read(1, it, slot2, coff2, tmp_w, tmp_r, v_w);
        mfma_range(8, 12);
        item_finish(1, it, s2);
        item_read(2, it, slot2, coff2, tmp_w, tmp_r, v_w);
        mfma_range(12, 16);
        read_v(1);
        item_finish(2, it, s2);
        mfma_range(16, 36);
        } else {                                    // all three items' reads up front
        Item it0, it1, it2;
        item_read(0, it0, slot2, coff2, tmp_w, tmp_r, v_w);
        item_read(1, it1, slot2, coff2, tmp_w, tmp_r, v_w);
        item_read(2, it2, slot2, coff2, tmp_w, tmp_r, v_w);
        read_v(0);
        W4_STAMP(g - 32, 1);
        mfma_range(0, 16);
        W4_STAMP(g - 32, 2);
        read_v(1);
        mfma_range(16, 18);
        item_finish(0, it0, s2);
        item_finish(1, it1, s2);
        item_finish(2, it2, s2);
        W4_STAMP(g - 32, 3);
        mfma_range(18, 36);
        }
        W4_STAMP(g - 32, 4);
        wait_planes(g);
        W4_STAMP(g - 32, 5);
        lds_barrier();
        W4_STAMP(g - 32, 6);
    };

    constexpr bool W4_SIGNED_TILES = false;
    constexpr bool w4_neg = false;
#include "conv_wino4_epi.h"

#pragma unroll 1
    for (int g = 0; g < G; g += 2) {
        iteration(g, std::integral_constant<int, 0>{}, Ea);
        iteration(g + 1, std::integral_constant<int, 1>{}, Ea);
        if ((g & 7) == 6) tile_epilogue(g >> 3);
    }
#include "conv_wino4_tail.h"
}

// Register image of the transformed weights: U[chunk s 8][wave w 8][quad q 9][lane 64][e 4].  Wave w = (cog = w & 3,
// ph = w >> 2), lane = (kq = lane >> 4, l15 = lane & 15); element n = 4 q + e = kg*18 + p is the MFMA A operand of
// position (i = 3 ph + p / 6, j = p % 6) for output channel cog*16 + l15 and input channel 8 s + 4 kg + kq:
// (G g G^T)[i][j].   fwd: g = w[co][ci][:, :];   dgrad: output channel = ci, input channel = co, g rotated by 180 degrees
struct PrepW { const float *w[4]; };
// element idx of the bf16-piece image of one convolution (conv_wino4_split.h): the same launch writes both images
__device__ void prep_wino4s_element(const float *__restrict__ w, uint16_t *__restrict__ u_fwd, uint16_t *__restrict__ u_dgrad, int idx);
__global__ void prep_wino4_kernel(const PrepW pw, float *__restrict__ u_fwd, float *__restrict__ u_dgrad)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= w4::U_FLOATS) return;
    const float *__restrict__ w = pw.w[blockIdx.y];          // convolution blockIdx.y of the launch
    u_fwd += (size_t)blockIdx.y * w4::U_TOTAL;
    u_dgrad += (size_t)blockIdx.y * w4::U_TOTAL;
    const int e = idx & 3, lane = (idx >> 2) & 63, rest = idx >> 8;
    const int q = rest % 9, ws = rest / 9, wv = ws & 7, s = ws >> 3;
    const int n = 4 * q + e, kg = n / 18, p = n - 18 * kg;
    const int cog = wv & 3, ph = wv >> 2, kq = lane >> 4, l15 = lane & 15;
    const int o = cog * 16 + l15, c = 8 * s + 4 * kg + kq;
    const int il = p / 6, j = p % 6;
    const int i = ph == 0 ? il : (il == 0 ? 5 : 2 + il);      // second half in the order i = 5, 3, 4 (see the tile epilogue)
    const float G[6][3] = {{0.25f, 0.f, 0.f},
                           {-1.f / 6, -1.f / 6, -1.f / 6},
                           {-1.f / 6, 1.f / 6, -1.f / 6},
                           {1.f / 24, 1.f / 12, 1.f / 6},
                           {1.f / 24, -1.f / 12, 1.f / 6},
                           {0.f, 0.f, 1.f}};
    double uf = 0.0, ud = 0.0;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const double coef = (double)G[i][r] * (double)G[j][t];
            uf += coef * (double)w[((o * 64 + c) * 3 + r) * 3 + t];
            ud += coef * (double)w[((c * 64 + o) * 3 + (2 - r)) * 3 + (2 - t)];
        }
    u_fwd[idx] = (float)uf;
    u_dgrad[idx] = (float)ud;
    prep_wino4s_element(w, reinterpret_cast<uint16_t *>(u_fwd + w4::U_FLOATS), reinterpret_cast<uint16_t *>(u_dgrad + w4::U_FLOATS), idx);
}

#include "conv_wino4_split.h"

}  // namespace

int cova_internal_persistent_grid2(int ntiles, int blocks_per_cu);

// ====================================================================================
// C ABI
// ====================================================================================
#ifdef W4_TRACE
COVA_API int cova_w4_trace_read(unsigned long long *host)
{
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_w4_trace), sizeof(unsigned long long) * 320);
}
#endif

COVA_API int cova_conv3x3_wino4_num_tiles(int B, int H, int W)
{
    return B * cdiv(W, w4::TW) * cdiv(H, w4::TH);
}

// Blocks of a launch: one per CU, or fewer on small maps -- even, and at most two per tile row of odd ty (the split main loop
// walks the even and the odd tile rows with its even and odd blocks: conv_wino4_split.h) -- so that no block is idle
static int w4_grid(int B, int H, int W)
{
    const int tiles_x = cdiv(W, w4::TW), tiles_y = cdiv(H, w4::TH);
    int g = cova_internal_persistent_grid2(B * tiles_x * tiles_y, 1);
    const int len_odd = B * tiles_x * (tiles_y / 2);
    if (len_odd > 0) {
        if (g > 2 * len_odd) g = 2 * len_odd;
        g &= ~1;
        if (g < 2) g = 2;
    }
    return g;
}

COVA_API int cova_conv3x3_wino4_num_partials(int B, int H, int W) { return w4_grid(B, H, W); }

// 1: every launch on the f32 main loop (A/B, tests); 0 (default): the split main loop wherever it exists (one input tensor)
int g_w4_f32 = 0;
int cova_internal_set_wino4_f32(int v) { g_w4_f32 = v != 0; return COVA_OK; }
int cova_internal_get_wino4_f32() { return (int)g_w4_f32; }

// floats per convolution and direction of the transformed-weight buffers: [f32 register image | bf16-piece register image]
COVA_API int cova_conv3x3_wino4_u_floats(void) { return w4::U_TOTAL; }

// u_fwd / u_dgrad: cova_conv3x3_wino4_u_floats() floats per convolution (both register images, see the prep kernels)
COVA_API int cova_conv3x3_wino4_prep_multi(const float *w0, const float *w1, const float *w2, const float *w3,
                                           float *u_fwd, float *u_dgrad, void *stream)
{
    COVA_REQUIRE(w0 && u_fwd && u_dgrad);
    const PrepW pw{{w0, w1, w2, w3}};
    const int n = w1 == nullptr ? 1 : w2 == nullptr ? 2 : w3 == nullptr ? 3 : 4;
    hipLaunchKernelGGL(prep_wino4_kernel, dim3(cdiv(w4::U_FLOATS, 256), n), dim3(256), 0, (hipStream_t)stream, pw, u_fwd,
                       u_dgrad);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

COVA_API int cova_conv3x3_wino4_prep(const float *w_oihw, float *u_fwd, float *u_dgrad, void *stream)
{
    return cova_conv3x3_wino4_prep_multi(w_oihw, nullptr, nullptr, nullptr, u_fwd, u_dgrad, stream);
}

namespace {

template <bool STATS, int PRO, bool ADD, int BN>
void launch_w4(const W4Args &a, int grid, hipStream_t st)
{
    if constexpr (PRO != 2) {
        if (!g_w4_f32) {             // split main loop: the piece image sits behind the f32 image
            W4Args b = a;
            b.u = a.u + w4::U_FLOATS;
            hipLaunchKernelGGL((conv3x3_c64_wino4s_kernel<STATS, PRO, ADD, BN>), dim3(grid), dim3(w4::THREADS), 0, st, b);
            return;
        }
    }
    hipLaunchKernelGGL((conv3x3_c64_wino4_kernel<STATS, PRO, ADD, BN>), dim3(grid), dim3(w4::THREADS), 0, st, a);
}

template <int PRO>
void launch_w4_pro(const W4Args &a, int grid, hipStream_t st)
{
    const bool add = a.epi.addend != nullptr;
    const int bn = a.epi.z == nullptr ? 0 : (a.epi.act == nullptr && a.epi.act_bits == nullptr ? 1 : 2);
    if (a.epi.inference) {                                     // inference epilogue (no statistics)
        if (add) launch_w4<false, PRO, true, 3>(a, grid, st); else launch_w4<false, PRO, false, 3>(a, grid, st);
    } else if (bn == 0) {
        if (a.stat_part) { if (add) launch_w4<true, PRO, true, 0>(a, grid, st); else launch_w4<true, PRO, false, 0>(a, grid, st); }
        else             { if (add) launch_w4<false, PRO, true, 0>(a, grid, st); else launch_w4<false, PRO, false, 0>(a, grid, st); }
    } else if (bn == 1) {
        if (add) launch_w4<true, PRO, true, 1>(a, grid, st); else launch_w4<true, PRO, false, 1>(a, grid, st);
    } else {
        if (add) launch_w4<true, PRO, true, 2>(a, grid, st); else launch_w4<true, PRO, false, 2>(a, grid, st);
    }
}

int run_w4(const float *in, const float *in2, const float *pro_abc, int pro_relu, const float *u, const W4Epi &epi,
           float *out, float *stat_part, int B, int H, int W, void *stream, const cova_bn_tail *tail = nullptr)
{
    BnTail t{};
    if (tail != nullptr && tail->mode != 0) {
        t = *tail;
        COVA_REQUIRE(stat_part && t.counter && t.count > 0 && (t.mode == 1 || t.mode == 2));
        COVA_REQUIRE(t.mode != 1 || (epi.z == nullptr && t.gamma && t.beta && t.scale && t.shift && t.mean && t.invstd));
        COVA_REQUIRE(t.mode != 2 || (epi.z != nullptr && t.mean && t.invstd && t.scale && t.abc));
    }
    COVA_REQUIRE(in && u && out && B > 0 && H > 0 && W > 0);
    COVA_REQUIRE((long long)H * W * 64 < (1ll << 31));          // 32-bit in-image offsets in the epilogue
    COVA_REQUIRE(epi.z == nullptr || ((epi.act || epi.act_bits || (epi.msc && epi.msh)) && epi.mean && epi.invstd && stat_part));
    COVA_REQUIRE(!epi.inference || (epi.z == nullptr && epi.msc != nullptr && epi.msh != nullptr && stat_part == nullptr));
    COVA_REQUIRE(epi.inference || epi.z != nullptr || (epi.msc == nullptr && epi.msh == nullptr));     // scale / shift need z, or the inference form
    COVA_REQUIRE(!(epi.act && epi.act_bits));
    const int tiles_x = cdiv(W, w4::TW), tiles_y = cdiv(H, w4::TH), ntiles = B * tiles_x * tiles_y;
    const W4Args a{in, pro_abc ? in2 : nullptr, u, out, stat_part, H, W, tiles_x, tiles_y, ntiles, pro_abc, pro_relu, epi, t};
    const int grid = w4_grid(B, H, W);
    if (!pro_abc) launch_w4_pro<0>(a, grid, (hipStream_t)stream);
    else if (!in2) launch_w4_pro<1>(a, grid, (hipStream_t)stream);
    else launch_w4_pro<2>(a, grid, (hipStream_t)stream);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

}  // namespace

// out NHWC [B,H,W,64] = conv3x3(in) with the F(4x4,3x3) weights `u`; stat_part (nullable):
// [cova_conv3x3_wino4_num_partials][2][64] = (sum y, sum y^2)
COVA_API int cova_conv3x3_wino4(const float *in, const float *u, float *out, float *stat_part, int B, int H, int W,
                                void *stream)
{
    return run_w4(in, nullptr, nullptr, 0, u, W4Epi{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, out,
                  stat_part, B, H, W, stream);
}

// The same convolution on relu?(A[c]*in + C[c]) formed on load (pro_abc [3][64] = A | unused | C; zero padding stays
// zero): a BatchNorm(+ReLU) folded into the consuming conv, as cova_conv3x3_wino_pro without a second tensor.
COVA_API int cova_conv3x3_wino4_pro(const float *in, const float *pro_abc, int pro_relu, const float *u, float *out,
                                    float *stat_part, int B, int H, int W, void *stream)
{
    COVA_REQUIRE(pro_abc);
    return run_w4(in, nullptr, pro_abc, pro_relu, u, W4Epi{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr},
                  out, stat_part, B, H, W, stream);
}

// Full form, same contract as cova_conv3x3_wino_pro (conv_wino.hip): input f(A*in + B*in2 + C) on load (pro_abc / in2
// nullable), epilogue  (+ addend) (x ReLU mask from act, or from fma(mask_scale, z, mask_shift) when act is NULL)
// with the BatchNorm-backward sums (sum g, sum g*xhat(z)) in stat_part when z is given, else plain statistics.
COVA_API int cova_conv3x3_wino4_full(const float *in, const float *in2, const float *pro_abc, int pro_relu,
                                     const float *u, const float *addend, const float *act, const float *mask_scale,
                                     const float *mask_shift, const float *z, const float *mean, const float *invstd,
                                     float *out, float *stat_part, int B, int H, int W, void *stream)
{
    return run_w4(in, in2, pro_abc, pro_relu, u,
                  W4Epi{addend, z ? act : nullptr, z, mean, invstd, z ? mask_scale : nullptr, z ? mask_shift : nullptr}, out,
                  stat_part, B, H, W, stream);
}

// Inference form (eval-mode forward, train.py:99-129): out = f(scale[c]*conv(g(in)) + shift[c] + addend), f = ReLU if relu;
// g = relu?(A[c]*in + C[c]) on load when pro_abc is given (the producer's BatchNorm + ReLU), identity otherwise.
// The BatchNorm (running statistics), residual add and ReLU that follow the conv in a BasicBlock (torchvision resnet.py
// BasicBlock.forward; models.py:49-51) sit in the epilogue: same expression and operation order as cova_bn_act_fwd.
COVA_API int cova_conv3x3_wino4_bnact(const float *in, const float *pro_abc, int pro_relu, const float *u,
                                      const float *addend, const float *scale, const float *shift, int relu, float *out,
                                      int B, int H, int W, void *stream)
{
    COVA_REQUIRE(scale && shift);
    return run_w4(in, nullptr, pro_abc, pro_relu, u,
                  W4Epi{addend, nullptr, nullptr, nullptr, nullptr, scale, shift, nullptr, relu, 1}, out, nullptr, B, H, W, stream);
}

// act_bits (instead of act): the mask source as one bit per element, [B*H*W][2] words as cova_bn_act_fwd_bits writes them
COVA_API int cova_conv3x3_wino4_full_tail(const float *in, const float *in2, const float *pro_abc, int pro_relu,
                                          const float *u, const float *addend, const float *act,
                                          const uint32_t *act_bits, const float *mask_scale, const float *mask_shift,
                                          const float *z, const float *mean, const float *invstd, float *out,
                                          float *stat_part, int B, int H, int W, const cova_bn_tail *tail, void *stream)
{
    return run_w4(in, in2, pro_abc, pro_relu, u,
                  W4Epi{addend, z ? act : nullptr, z, mean, invstd, z ? mask_scale : nullptr, z ? mask_shift : nullptr,
                        z ? act_bits : nullptr},
                  out, stat_part, B, H, W, stream, tail);
}
