// Weight gradient of the 3x3 / 64->64 convolution in Winograd F(4x4,3x3) form (forward / data gradient: conv_wino4.hip):
//
//   dW = G^T [ sum over 4x4 output tiles of  (A dY A^T) .* (B^T d B) ] G        A 6x4 (= the forward's A^T transposed),
//                                                                                B^T 6x6, G 6x3 (Lavin & Gray 2015)
//
// i.e. for each of the 36 transform positions one GEMM  Q[pos][co][ci] = sum_tiles Wd[pos][co][tile] * V[pos][tile][ci]
// with K = number of 4x4 tiles: 36 * 64 * 64 MACs per 16 pixels = 2.25 per (pixel, co, ci) instead of 9 (direct) or 4
// (the F(2x2,3x3) kernel of conv_wino.hip: 1.78x more MFMAs).  fp32 error against fp64 on this layer: 3.7e-6 of the
// gradient's scale (F(2x2): 5.9e-7, direct fp32: 2.9e-6; numpy study in DESIGN.md section 11).
//
// Mapping:
//   * the 36 x 64 x 64 accumulators are 576 KB, more than a CU's register file: a block owns ONE HALF of the positions
//     (transform rows a = 0,1,2 or a = 5,3,4 -- in that order, so that both halves run the same second stage), all 64 x 64
//     channel pairs: 18 x 16 MFMA blocks of v_mfma_f32_16x16x4_f32 = 36 accumulators (144 registers) per wave, wave =
//     (9 positions) x (2 co blocks) x (2 ci blocks).  Splitting by position ROWS costs no redundant arithmetic: the
//     first transform stage yields exactly the three rows of the half (6 operations per column instead of 12);
//   * the two halves of a tile walk run as two blocks on the same XCD (block b and b + 8): the second reader of a pixel
//     row finds it in L2;
//   * a K-step = four horizontally adjacent tiles (16 x 4 output pixels) = the MFMA's K = 4.  Each wave transforms ONE
//     operand tile per K-step with lane = channel: waves 0-3 the input patches (6 x 6 pixels -> 18 values of V), waves
//     4-7 the gradient tiles (4 x 4 -> 18 values of Wd), both stages in registers, results to LDS as
//     [position][tile][channel] rows;
//   * the pixels arrive as global -> LDS copies (global_load_lds_dwordx4, no registers: with 144 accumulator registers a
//     36-register patch in flight made the allocator split live ranges across the edge / interior paths and wait for the
//     prefetch at every join): lane = (pixel of a 2 x 2 block, channel quad), 1 KB per wave instruction, landing in the
//     wave's PRIVATE 9 KB of LDS as [pixel][64 channels] -- the copy is also the transposition to lane = channel, and a
//     wave-private buffer needs no barrier, only the issuing wave's own vmcnt;
//   * a block walks DOWN a 16-pixel-wide strip of the map: the two bottom rows of a patch are the two top rows of the
//     next one and stay in the wave's buffer (a ring of three row pairs): 24 of a patch's 36 pixels are requested;
//   * the next patch is requested once the first transform stage has read the current one (every copy is unconditional:
//     edge patches clamp the address, the consumer zeroes padding);
//   * the two waves of a SIMD (input role / gradient role) run an iteration in opposite order (transform then MFMAs /
//     MFMAs then transform); operands are double buffered, one block barrier per K-step (LDS-only).
// Side output: the gradient operand A*dz + B*dz2 + C written once (gradient-role waves of the half-0 blocks), for the data-
// gradient launch of the same convolution.
// Operand prologues as the F(2x2) kernel's: activation = relu?(A*act + C) on load, gradient = A*dz + B*dz2 + C on load
// (the BatchNorm+ReLU of the producer / the BatchNorm-backward apply, never materialised); zero padding stays zero.
#include "common.h"
#include <type_traits>

// Ablation builds of tools/wgrad4_bench.py (tools/wg4_abl_build.sh, -DWG4_ABL=<mask>; 0 in the product): 1 no MFMAs,
// 2 no pixel copies, 8 no transform (reads + arithmetic + operand writes), 16 no operand reads of the
// MFMA phase (stale registers), 32 copies always from the walker's first patch (L2-resident source), 64 both roles
// in the same order (MFMAs first), 128 no row-pair ring (every patch requested whole)
#ifndef WG4_ABL
#define WG4_ABL 0
#endif

int cova_internal_persistent_grid2(int ntiles, int blocks_per_cu);

// 157,696 B of static LDS (operand buffers 83,968 + wave-private pixel buffers 73,728): only gfx950's 160 KB per CU holds it
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "conv_wgrad4.hip is written for gfx950 (160 KB of LDS per workgroup)"
#endif

// -DWG4_TRACE (tools only): s_memtime stamps of waves 0 and 4 of block 0 at the phase boundaries of iterations 100..103
#ifdef WG4_TRACE
__device__ unsigned long long g_wg4_trace[2 * 4 * 8];
#define WG4_STAMP(slot)                                                                                         \
    do {                                                                                                        \
        if (blockIdx.x == 0 && (wave == 0 || wave == 4) && wg4_k >= 100 && wg4_k < 104 && lane == 0)           \
            g_wg4_trace[((wave >> 2) * 4 + (wg4_k - 100)) * 8 + (slot)] = __builtin_amdgcn_s_memtime();         \
    } while (0)
#else
#define WG4_STAMP(slot) do { } while (0)
#endif

namespace {

namespace wg4 {
constexpr int THREADS = 512;
constexpr int ROW = 10;                              // floats per operand row (position group, tile, channel): 9 positions + 1 pad --
                                                     // 8-byte aligned, 10 l mod 64 distinct bank pairs for l = 0..15
constexpr int TILE_FLOATS = 128 * ROW + 32;          // the two tiles of a 32-lane read group sit 32 banks apart: conflict-free ds_read_b64
constexpr int BUF_FLOATS = 2 * 4 * TILE_FLOATS;      // [position group 2][tile 4][channel 128 = co | ci][10]: 41,984 B
constexpr int RAW_FLOATS = 36 * 64;                  // a wave's pixel buffer: 36 pixels x 64 channels = 9,216 B
constexpr int PART_FLOATS = 9 * 4096;                // one block's partial, output-transformed: [r * 3 + t][co 64][ci 64]
}  // namespace wg4

struct Wg4Args {
    const float *act, *dz, *dz2;     // NHWC [B,H,W,64]; dz2 nullable
    float *part;                     // [grid][9][64][64]
    const float *act_abc, *dz_abc;   // [3][64] = A | B | C, or nullptr (plain operand)
    float *dz_out;                   // nullable: the gradient operand as formed on load (A*dz + B*dz2 + C), NHWC [B,H,W,64]
    int act_relu;
    int H, W, tiles_y, strips, nseg_y, seg_rows, nsegs;
};

typedef __attribute__((address_space(3))) void wg4_lds_void;

__device__ __forceinline__ f32x4 wg4_mfma(float a, float b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// barrier over LDS traffic only: global -> LDS copies (the next K-step's pixels) stay in flight across it
__device__ __forceinline__ void wg4_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// global -> LDS copy of 16 bytes per lane (lane i lands at lds_base + 16 i; lds_base wave-uniform), as inline asm: the
// compiler neither tracks nor waits for it -- the one wait is the explicit vmcnt(0) in front of the transform's reads
__device__ __forceinline__ void wg4_copy16(const char *base, unsigned off_bytes, unsigned lds_base_bytes)
{
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_base_bytes), "v"(off_bytes), "s"(base) : "memory", "m0");
}

// B^T x for one 6-vector (second stage of the input transform; 12 operations)
__device__ __forceinline__ void wg4_bt6(const float (&d)[6], float (&o)[6])
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

// A x for one 4-vector -> 6 (second stage of the gradient transform; 9 operations)
__device__ __forceinline__ void wg4_a6(const float (&g)[4], float (&o)[6])
{
    const float p = g[0] + g[2], q = g[1] + g[3];
    const float u = fmaf(4.f, g[2], g[0]), w = 2.f * fmaf(4.f, g[3], g[1]);
    o[0] = g[0];
    o[1] = p + q;
    o[2] = p - q;
    o[3] = u + w;
    o[4] = u - w;
    o[5] = g[3];
}

// The three rows of a half, first stage.  Half 0: transform rows a = 0, 1, 2; half 1: a = 5, 3, 4 (in this order).
template <int HALF>
__device__ __forceinline__ void wg4_bt_half(float d0, float d1, float d2, float d3, float d4, float d5, float (&o)[3])
{
    if (HALF == 0) {
        const float a = fmaf(-4.f, d2, d4), b = fmaf(-4.f, d1, d3);
        o[0] = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
        o[1] = a + b;
        o[2] = a - b;
    } else {
        const float c = d4 - d2, e = 2.f * (d3 - d1);
        o[0] = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
        o[1] = c + e;
        o[2] = c - e;
    }
}

template <int HALF>
__device__ __forceinline__ void wg4_a_half(float g0, float g1, float g2, float g3, float (&o)[3])
{
    if (HALF == 0) {
        const float p = g0 + g2, q = g1 + g3;
        o[0] = g0;
        o[1] = p + q;
        o[2] = p - q;
    } else {
        const float u = fmaf(4.f, g2, g0), w = 2.f * fmaf(4.f, g3, g1);
        o[0] = g3;
        o[1] = u + w;
        o[2] = u - w;
    }
}

// position of the K-step stream: which segment (image, 16-pixel strip, run of tile rows) and which row of it
struct Wg4It {
    int seg, rr, nrows, b, strip, ty;
};

// PROD: 0 plain gradient operand, 1 = A*dz + C, 2 = A*dz + B*dz2 + C on load
template <bool PROA, int PROD>
__global__ __launch_bounds__(wg4::THREADS, 1) void conv3x3_wgrad4_kernel(const Wg4Args a)
{
    using namespace wg4;
    __shared__ __attribute__((aligned(16))) float s_op[2 * BUF_FLOATS];
    __shared__ __attribute__((aligned(1024))) float s_raw[8 * RAW_FLOATS];      // wave-private pixel buffers
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = a.H, W = a.W;
    // block -> (position half, walker): the two halves of a walker sit 8 blocks apart (same XCD) when the grid allows
    const int G = (int)gridDim.x, NW = G >> 1;
    int half, walker;
    if ((G & 15) == 0) {
        const int m = (int)blockIdx.x >> 3, x = (int)blockIdx.x & 7;
        half = m & 1;
        walker = (m >> 1) * 8 + x;
    } else {
        half = (int)blockIdx.x & 1;
        walker = (int)blockIdx.x >> 1;
    }
    const bool vrole = wave < 4;                  // waves 0-3: input patches; waves 4-7: gradient tiles
    const int tcol = wave & 3;                    // the tile of the K-step this wave transforms
    // MFMA role: 9 positions x 2 co blocks x 2 ci blocks
    const int pg = wave & 1, cp = (wave >> 1) & 1, np = wave >> 2;
    const int l15 = lane & 15, kq = lane >> 4;

    int n_k = 0;                                  // K-steps of this block
    for (int s = walker; s < a.nsegs; s += NW) n_k += min(a.seg_rows, a.tiles_y - (s % a.nseg_y) * a.seg_rows);

    auto seg_setup = [&](Wg4It &it) __attribute__((always_inline)) {
        const int s = it.seg < a.nsegs ? it.seg : walker;        // past the end: a valid segment (its pixels are never used)
        const int sy = s % a.nseg_y, st = (s / a.nseg_y) % a.strips;
        it.b = s / (a.nseg_y * a.strips);
        it.strip = st;
        it.ty = sy * a.seg_rows;
        it.nrows = min(a.seg_rows, a.tiles_y - it.ty);
        it.rr = 0;
    };
    auto advance = [&](Wg4It &it) __attribute__((always_inline)) {
        if (++it.rr < it.nrows) {
            ++it.ty;
        } else {
            it.seg += NW;
            seg_setup(it);
        }
    };

    f32x4 acc[9][4];            // [position of the group][co block i * 2 + ci block j]
#pragma unroll
    for (int q = 0; q < 9; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[q][e] = f32x4{0.f, 0.f, 0.f, 0.f};

    // operand elements of this lane: A = Wd[position][co = 16 (2 cp + i) + l15][tile kq], B = V[position][tile kq][ci = 16 (2 np + j) + l15]
    const int a_off = (pg * 4 + kq) * TILE_FLOATS + (32 * cp + l15) * ROW;
    const int b_off = (pg * 4 + kq) * TILE_FLOATS + (64 + 32 * np + l15) * ROW;
    auto mfma_phase = [&](const float *buf) __attribute__((always_inline)) {
        if (WG4_ABL & 1) return;
        const float *pa = (WG4_ABL & 16) ? s_op + a_off : buf + a_off, *pb = (WG4_ABL & 16) ? s_op + b_off : buf + b_off;
        auto pair = [&](const float *p, float (&d)[2]) __attribute__((always_inline)) {
            const float2 v = *reinterpret_cast<const float2 *>(p);
            d[0] = v.x; d[1] = v.y;
        };
#pragma unroll
        for (int q0 = 0; q0 < 8; q0 += 2) {
            float a0[2], a1[2], b0[2], b1[2];
            pair(pa + q0, a0);
            pair(pa + 16 * ROW + q0, a1);
            pair(pb + q0, b0);
            pair(pb + 16 * ROW + q0, b1);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                acc[q0 + q][0] = wg4_mfma(a0[q], b0[q], acc[q0 + q][0]);
                acc[q0 + q][1] = wg4_mfma(a0[q], b1[q], acc[q0 + q][1]);
                acc[q0 + q][2] = wg4_mfma(a1[q], b0[q], acc[q0 + q][2]);
                acc[q0 + q][3] = wg4_mfma(a1[q], b1[q], acc[q0 + q][3]);
            }
        }
        const float a0 = pa[8], a1 = pa[16 * ROW + 8], b0 = pb[8], b1 = pb[16 * ROW + 8];
        acc[8][0] = wg4_mfma(a0, b0, acc[8][0]);
        acc[8][1] = wg4_mfma(a0, b1, acc[8][1]);
        acc[8][2] = wg4_mfma(a1, b0, acc[8][2]);
        acc[8][3] = wg4_mfma(a1, b1, acc[8][3]);
    };

    // The whole pipeline of one (role, position half): specialised bodies, chosen once per wave (the role and the half are
    // wave- / block-uniform).
    //   pixels of patch k + 2 requested | patch k + 1 transformed into the other operand buffer | MFMAs of patch k
    auto run = [&](auto role, auto hsel) __attribute__((always_inline)) {
        constexpr bool VROLE = decltype(role)::value;
        constexpr int HALF = decltype(hsel)::value;
        constexpr int NR = VROLE ? 6 : 4;                    // patch rows = columns
        constexpr int NB = NR / 2;                           // 2 x 2 pixel blocks per patch row / column
        constexpr bool TWO = !VROLE && PROD == 2;
        // per-channel prologue constants (lane = channel); ReLU as max(., 0) or max(., -inf): one instruction either way
        const float relu_floor = a.act_relu ? 0.f : -INFINITY;
        float pA = 1.f, pB = 0.f, pC = 0.f;
        if (VROLE) {
            if (PROA) { pA = a.act_abc[lane]; pC = a.act_abc[128 + lane]; }
        } else {
            if (PROD) { pA = a.dz_abc[lane]; pC = a.dz_abc[128 + lane]; }
            if (PROD == 2) pB = a.dz_abc[64 + lane];
        }
        // Copy group j = (br, bc) = the 2 x 2 pixel block at rows 2 br, columns 2 bc of the patch; lane = (pixel of the
        // block s = lane >> 4 = (dr, dc), channel quad lane & 15): address = (scalar base of the row pair) + 512 bc +
        // ONE lane offset; it lands in slot 4 j + s of the wave's pixel buffer ([slot][64 channels]; the second
        // gradient tensor behind the first: + 16 slots).
        const int sub = lane >> 4, c4 = lane & 15;
        const unsigned laneoff = (unsigned)((((sub >> 1) * W + (sub & 1)) * 64 + c4 * 4) * 4);
        float *raw = s_raw + wave * RAW_FLOATS;
        const unsigned raw_lds = (unsigned)(size_t)(wg4_lds_void *)raw;
        const float *raw_r = raw + lane;                     // slot s, this lane's channel: raw_r[64 s]
        // pixel validity of a patch: bit r of the row mask / bit c of the column mask (wave-uniform; a few scalar operations:
        // the valid rows / columns of a patch are a contiguous range)
        auto range_mask = [&](int g0, int n) __attribute__((always_inline)) {        // bits i with 0 <= g0 + i < n, i < NR
            const int lo = max(-g0, 0), hi = min(n - g0, NR);
            return hi > lo ? (((1u << hi) - 1u) & ~((1u << lo) - 1u)) : 0u;
        };
        auto masks = [&](const Wg4It &it, unsigned &rm, unsigned &cm) __attribute__((always_inline)) {
            rm = range_mask(4 * it.ty - (VROLE ? 1 : 0), H);
            cm = range_mask(16 * it.strip + 4 * tcol - (VROLE ? 1 : 0), W);
        };
        constexpr unsigned FULL = (1u << NR) - 1u;
        // every pixel of the patch of `it` requested (unconditionally: an edge patch clamps its addresses, the consumer
        // zeroes what lies outside the image; the K-steps past the end re-request a valid patch)
        // Input role: the buffer is a RING of the patch's three row pairs -- walking down a strip, the bottom pair of a patch
        // is the top pair of the next one and stays where it is; only the two new pairs are requested (24 of 36 pixels).
        // Row pair p of a patch with ring phase phi lives in pair slot (p + phi) % 3 (12 pixel slots each).
        // one copy: block (br, bc) of the patch of `it` (and of the second gradient tensor)
        auto piece = [&](const Wg4It &it_, bool edge, int phi, int br, int bc) __attribute__((always_inline)) {
            if (WG4_ABL & 2) return;
            Wg4It it = it_;
            if (WG4_ABL & 32) { it.seg = walker; seg_setup(it); }
            const int gy0 = 4 * it.ty - (VROLE ? 1 : 0), gx0 = 16 * it.strip + 4 * tcol - (VROLE ? 1 : 0);
            // (64-bit base of the walk's page, wave-uniform; 32-bit byte offsets inside the page: any batch size)
            const size_t pgb = (size_t)it.b * (size_t)H * (size_t)W * 256u;
            const char *src = reinterpret_cast<const char *>(VROLE ? a.act : a.dz) + pgb;
            const char *src2 = TWO ? reinterpret_cast<const char *>(a.dz2) + pgb : nullptr;
            int ps = br + phi;                                   // pair slot (the gradient role has no ring: phi = 0)
            if (VROLE && ps >= 3) ps -= 3;
            const unsigned dst = raw_lds + (VROLE ? 3072u * (unsigned)ps : 2048u * (unsigned)br) + 1024u * (unsigned)bc;
            if (!edge) {                                         // scalar base of the row pair + the block column
                const unsigned off = (unsigned)(((gy0 + 2 * br) * W + gx0) * 256 + 512 * bc);     // (a page < 4 GB: host check)
                wg4_copy16(src + off, laneoff, dst);
                if (TWO) wg4_copy16(src2 + off, laneoff, dst + 4096);
            } else {
                const int r = 2 * br + (sub >> 1), c = 2 * bc + (sub & 1);
                const int gy = min(max(gy0 + r, 0), H - 1), gx = min(max(gx0 + c, 0), W - 1);
                const unsigned eo = (unsigned)((gy * W + gx) * 256 + c4 * 16);
                wg4_copy16(src, eo, dst);
                if (TWO) wg4_copy16(src2, eo, dst + 4096);
            }
        };
        // a continued walk of the input role keeps row pair 0 (wave-uniform)
        auto first_pair = [&](const Wg4It &it) __attribute__((always_inline)) {
            return (VROLE && it.rr != 0 && !(WG4_ABL & 128)) ? 1 : 0;
        };
        auto fetch = [&](const Wg4It &it, bool edge, int phi) __attribute__((always_inline)) {
            const int br0 = first_pair(it);
#pragma unroll
            for (int br = 0; br < NB; ++br) {
                if (br < br0) continue;
#pragma unroll
                for (int bc = 0; bc < NB; ++bc) piece(it, edge, phi, br, bc);
            }
        };
        // side output (gradient role): the operand as formed on load, written once -- the data-gradient launch of the same
        // convolution then reads ONE tensor with no prologue instead of forming it again per tap.  The two position halves
        // of a walk (blocks b, b + 8: same tiles, same order) take TURNS, tile by tile: with all the stores in half 0 that
        // block fell behind its partner, the pair stopped meeting in L2 and the launch fetched every pixel twice (FETCH_SIZE
        // of the step's launches 2.5 GB against 1.3 GB stand-alone without the side output: 3 GB moved in 0.59 ms)
        const bool emit = !VROLE && PROD != 0 && a.dz_out != nullptr;
        int emit_k = 0;                                      // tiles transformed so far by this wave
        unsigned emit_lane = (unsigned)lane * 4u;
        asm volatile("" : "+v"(emit_lane));
        unsigned cur_org = 0;                                // element offset of the current tile's pixel (0, 0) inside its page
        char *cur_out = reinterpret_cast<char *>(a.dz_out);  // ... and that page of the side output
        auto origin = [&](const Wg4It &it) __attribute__((always_inline)) {
            cur_out = reinterpret_cast<char *>(a.dz_out) + (size_t)it.b * (size_t)H * (size_t)W * 256u;
            return (unsigned)(((4 * it.ty) * W + 16 * it.strip + 4 * tcol) * 64);
        };
        int phi = 0;                                         // ring phase of the patch in the buffer
        int wg4_k = -1;                                      // (trace builds)
        (void)wg4_k;
        // (The two position halves of a walk -- blocks b, b ^ 8: same XCD, same pixels in the same order -- run free.  A pacing
        // through progress words was built and measured in round 5: same HBM traffic, 3 % slower)
        auto transform = [&](float *buf, const Wg4It &nxt, unsigned crm, unsigned ccm, unsigned nrm, unsigned ncm) __attribute__((always_inline)) {
            // this lane's channel of pair slot (p + phi) % 3, p = 0..2 (input role; gradient role: two fixed pairs)
            const float *pair_r[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                int ps = p + phi;
                if (ps >= 3) ps -= 3;
                pair_r[p] = raw_r + (VROLE ? 768 * ps : 512 * p);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's copies of the patch have landed
            const int nphi = (!VROLE || (WG4_ABL & 128)) ? 0 : (nxt.rr == 0 ? phi : (phi == 0 ? 2 : phi - 1));      // (phi + 2) % 3 down the strip
            const bool nedge = nrm != FULL || ncm != FULL;
            if (WG4_ABL & 8) { fetch(nxt, nedge, nphi); phi = nphi; return; }
            WG4_STAMP(1);
            float t[3][NR];
            const bool emit_now = emit && ((emit_k & 1) == HALF);
            ++emit_k;
            auto stage1 = [&](auto cedge_t) __attribute__((always_inline)) {
            constexpr bool cedge = decltype(cedge_t)::value;
#pragma unroll
            for (int c = 0; c < NR; ++c) {
                float d[NR];
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int slot = 4 * (c >> 1) + 2 * (r & 1) + (c & 1);          // within the row pair's slots
                    float v = pair_r[r >> 1][64 * slot];
                    if (VROLE && PROA) v = fmaxf(fmaf(pA, v, pC), relu_floor);     // same expression as the forward prologue
                    if (TWO) v = fmaf(pA, v, fmaf(pB, pair_r[r >> 1][64 * (16 + slot)], pC));   // dz = A*dy + B*z + C
                    if (!VROLE && PROD == 1) v = fmaf(pA, v, pC);
                    if (cedge && !(((crm >> r) & (ccm >> c) & 1u) != 0u)) v = 0.f;              // zero padding stays zero
                    d[r] = v;
                    if (!VROLE && PROD != 0) {
                        // (scalar base of the pixel + ONE lane offset: global_store with an SGPR pair, no per-lane pointers)
                        if (emit_now && (!cedge || (((crm >> r) & (ccm >> c) & 1u) != 0u))) {
                            unsigned ob = (cur_org + (unsigned)((r * W + c) * 64)) * 4u;
                            asm volatile("" : "+s"(ob));
                            *reinterpret_cast<float *>(cur_out + ob + emit_lane) = v;
                        }
                    }
                }
                float o[3];
                if constexpr (VROLE) wg4_bt_half<HALF>(d[0], d[1], d[2], d[3], d[4], d[5], o);
                else wg4_a_half<HALF>(d[0], d[1], d[2], d[3], o);
                t[0][c] = o[0]; t[1][c] = o[1]; t[2][c] = o[2];
            }
            };
            if (crm != FULL || ccm != FULL) stage1(std::true_type{});      // (wave-uniform: nothing but registers lives across)
            else stage1(std::false_type{});
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the buffer has been read: it may be overwritten
            WG4_STAMP(2);
            fetch(nxt, nedge, nphi);          // the buffer is free: the next patch is requested
            phi = nphi;
            WG4_STAMP(3);
            // second stage row by row, stored as soon as a row is complete: position p = 6 al + b sits at float p of the
            // group-0 row (p < 9), at float p - 9 of the group-1 row otherwise
            float *r0 = buf + tcol * TILE_FLOATS + ((VROLE ? 64 : 0) + lane) * ROW;
            float *r1 = r0 + 4 * TILE_FLOATS;
#pragma unroll
            for (int al = 0; al < 3; ++al) {
                float o[6];
                if constexpr (VROLE) wg4_bt6(t[al], o);
                else wg4_a6(t[al], o);
                if (al == 0) {
                    *reinterpret_cast<float2 *>(r0) = make_float2(o[0], o[1]);
                    *reinterpret_cast<float2 *>(r0 + 2) = make_float2(o[2], o[3]);
                    *reinterpret_cast<float2 *>(r0 + 4) = make_float2(o[4], o[5]);
                } else if (al == 1) {
                    *reinterpret_cast<float2 *>(r0 + 6) = make_float2(o[0], o[1]);
                    r0[8] = o[2];
                    *reinterpret_cast<float2 *>(r1) = make_float2(o[3], o[4]);
                    r1[2] = o[5];
                } else {
                    r1[3] = o[0];
                    *reinterpret_cast<float2 *>(r1 + 4) = make_float2(o[1], o[2]);
                    *reinterpret_cast<float2 *>(r1 + 6) = make_float2(o[3], o[4]);
                    r1[8] = o[5];
                }
            }
        };

        Wg4It nxt;
        nxt.seg = walker;
        seg_setup(nxt);
        unsigned crm, ccm, nrm, ncm;
        masks(nxt, nrm, ncm);
        fetch(nxt, true, 0);                                  // patch 0 (through the clamping path: once per block)
        crm = nrm; ccm = ncm;
        cur_org = origin(nxt);
        advance(nxt);
        masks(nxt, nrm, ncm);
        transform(s_op, nxt, crm, ccm, nrm, ncm);             // patch 0 -> buffer 0; requests patch 1
        crm = nrm; ccm = ncm;
        cur_org = origin(nxt);
        advance(nxt);
        wg4_lds_barrier();
        // The two waves of a SIMD (wave w: input role, wave w + 4: gradient role) run the iteration in OPPOSITE order --
        // transform then MFMAs / MFMAs then transform: one wave's pixel wait and transform arithmetic sit beside the
        // other's matrix phase instead of all eight waves idling the matrix pipe together.  (Within an iteration every wave
        // only reads buffer k & 1 and only writes the other one: the order inside it is free.)
#pragma unroll 1
        for (int k = 0; k < n_k; ++k) {
            wg4_k = k;
            WG4_STAMP(0);
            if (!VROLE || (WG4_ABL & 64)) { mfma_phase(s_op + (k & 1) * BUF_FLOATS); WG4_STAMP(5); }
            const bool more = k + 1 < n_k;
            if (more) {
                masks(nxt, nrm, ncm);
                transform(s_op + ((k + 1) & 1) * BUF_FLOATS, nxt, crm, ccm, nrm, ncm);      // patch k + 1
            }
            WG4_STAMP(4);
            if (VROLE && !(WG4_ABL & 64)) { mfma_phase(s_op + (k & 1) * BUF_FLOATS); WG4_STAMP(5); }
            if (more) {
                crm = nrm; ccm = ncm;
                cur_org = origin(nxt);
                advance(nxt);
            }
            wg4_lds_barrier();
            WG4_STAMP(6);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the last request, never consumed, has landed before the LDS is released)
    };

    if (n_k > 0) {
        using T = std::true_type;
        using Fl = std::false_type;
        using H0 = std::integral_constant<int, 0>;
        using H1 = std::integral_constant<int, 1>;
        if (vrole) {
            if (half == 0) run(T{}, H0{});
            else run(T{}, H1{});
        } else {
            if (half == 0) run(Fl{}, H0{});
            else run(Fl{}, H1{});
        }
    }
    // ---- this block's partial, already through the output transform (linear, so the sum over blocks commutes with it):
    //        part[block][r * 3 + t][co][ci] = sum over the half's positions (a, b) of G[a][r] * G[b][t] * Q[a][b][co][ci]
    // in fp64 on the f32 accumulators (rows first: c[t] = sum_b G[b][t] Q[a][b], then T[r][t] += G[a][r] c[t]; zero
    // coefficients dropped at compile time: 33 / 39 fused multiply-adds per entry for the two position groups), rounded to
    // f32 once.  The two position groups of a (co, ci) block are the waves w and w ^ 1: group 1 hands its nine values per
    // entry over through the operand buffers (idle now), group 0 adds and stores -- 9 x 4096 floats per block instead of
    // 18 x 4096: half the partial traffic of the launch and a quarter for the fold kernel, which no longer transforms.
    // (the transform of a wave's entries is specialised by its position half and group; the exchange around the block barrier is
    // ONE piece of code every wave runs -- a barrier reached from four different instantiations works on gfx950, where s_barrier
    // counts arrivals whatever their program counter, but is not something to rely on)
    auto fold_values = [&](auto hsel, auto psel, auto esel, float (&tv)[4][9]) __attribute__((always_inline)) {
        constexpr int HALF = decltype(hsel)::value, PG = decltype(psel)::value, e = decltype(esel)::value;
        constexpr double G[6][3] = {{0.25, 0.0, 0.0},         {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                    {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0.0, 0.0, 1.0}};
        {
            {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double T[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) T[k] = 0.0;
#pragma unroll
                for (int al = 0; al < 3; ++al) {
                    constexpr int p0 = 9 * PG, p1 = 9 * PG + 9;
                    if (6 * al + 6 <= p0 || 6 * al >= p1) continue;              // the row is not in this group
                    const int arow = HALF == 0 ? al : (al == 0 ? 5 : 2 + al);     // half 1 holds rows 5, 3, 4
                    double c[3] = {0.0, 0.0, 0.0};
#pragma unroll
                    for (int b = 0; b < 6; ++b) {
                        const int pos = 6 * al + b;
                        if (pos < p0 || pos >= p1) continue;
                        const double qv = (double)acc[pos - p0][e][r];
#pragma unroll
                        for (int t = 0; t < 3; ++t)
                            if (G[b][t] != 0.0) c[t] = fma(G[b][t], qv, c[t]);
                    }
#pragma unroll
                    for (int r3 = 0; r3 < 3; ++r3)
#pragma unroll
                        for (int t = 0; t < 3; ++t)
                            if (G[arow][r3] != 0.0) T[r3 * 3 + t] = fma(G[arow][r3], c[t], T[r3 * 3 + t]);
                }
#pragma unroll
                for (int k = 0; k < 9; ++k) tv[r][k] = (float)T[k];
            }
            }
        }
    };
    float *dst = a.part + (size_t)blockIdx.x * PART_FLOATS;
    auto fold_round = [&](auto esel) __attribute__((always_inline)) {
        constexpr int e = decltype(esel)::value;
        using H0 = std::integral_constant<int, 0>;
        using H1 = std::integral_constant<int, 1>;
        float *xch = s_op + (e & 1) * (4 * 36 * 64) + (wave >> 1) * (36 * 64) + lane;       // [entry r][k 9][lane 64]
        float tv[4][9];
        if (half == 0) { if (pg == 0) fold_values(H0{}, H0{}, esel, tv); else fold_values(H0{}, H1{}, esel, tv); }
        else           { if (pg == 0) fold_values(H1{}, H0{}, esel, tv); else fold_values(H1{}, H1{}, esel, tv); }
        if (pg == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int k = 0; k < 9; ++k) xch[(r * 9 + k) * 64] = tv[r][k];
        }
        __syncthreads();              // (area e & 1 is written again in round e + 2: the barrier of round e + 1 lies between)
        if (pg == 0) {
            const int i = e >> 1, j = e & 1;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int k = 0; k < 9; ++k)
                    dst[k * 4096 + (16 * (2 * cp + i) + 4 * kq + r) * 64 + 16 * (2 * np + j) + l15] =
                        tv[r][k] + xch[(r * 9 + k) * 64];
        }
    };
    __syncthreads();                      // every wave is out of its last operand reads (blocks without K-steps included)
    fold_round(std::integral_constant<int, 0>{});
    fold_round(std::integral_constant<int, 1>{});
    fold_round(std::integral_constant<int, 2>{});
    fold_round(std::integral_constant<int, 3>{});
}

// Fold of the per-block partials (already transformed: part[block][r * 3 + t][co][ci]), up to four convolutions in one launch:
//   dW[co][ci][r][t] = sum over the blocks of part[block][r * 3 + t][co][ci]       (fp64, fixed order; OIHW)
// block = 64 (co, ci) pairs x 16 slices of blocks: a wave is ONE slice over 64 adjacent pairs (256 contiguous bytes per load;
// round 4's 16 pairs x 16 slices read 64-byte pieces: 50 us for the 151 MB of a step, 3 TB/s), the sums are taken per slice in
// block order and over the slices in slice order exactly as before (identical bits)
struct Wg4FinishJobs { const float *part[4]; float *dw[4]; };
__global__ __launch_bounds__(1024) void wgrad4_finish_kernel(const Wg4FinishJobs jobs, int grid)
{
    __shared__ double s_acc[9][16][64];          // [r * 3 + t][slice][pair]
    const float *__restrict__ part = jobs.part[blockIdx.y];
    float *__restrict__ dw = jobs.dw[blockIdx.y];
    const int tx = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + tx;        // (co, ci) pair
    double acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = 0.0;
    // eight blocks' rows (72 loads) are requested before the first is added: two dependent round trips per thread for the 256
    // blocks of a launch; same adds in the same order
    for (int blk0 = slice; blk0 < grid; blk0 += 16 * 8) {
        float v[8][9];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int blk = blk0 + 16 * u;
            const float *row = part + (size_t)(blk < grid ? blk : blk0) * wg4::PART_FLOATS + idx;
#pragma unroll
            for (int k = 0; k < 9; ++k) v[u][k] = row[k * 4096];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (blk0 + 16 * u < grid) {
#pragma unroll
                for (int k = 0; k < 9; ++k) acc[k] += (double)v[u][k];
            }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) s_acc[k][slice][tx] = acc[k];
    __syncthreads();
    if (slice < 9) {
        double t = 0.0;
        for (int j = 0; j < 16; ++j) t += s_acc[slice][j][tx];
        dw[idx * 9 + slice] = (float)t;
    }
}

struct Wg4Geo { int tiles_y, strips, nseg_y, seg_rows, nsegs, grid; };

Wg4Geo wg4_geometry(int B, int H, int W)
{
    Wg4Geo g;
    g.tiles_y = cdiv(H, 4);
    g.strips = cdiv(W, 16);
    g.nseg_y = (g.tiles_y + 10) / 20 > 0 ? (g.tiles_y + 10) / 20 : 1;         // runs of ~20 tile rows
    g.seg_rows = cdiv(g.tiles_y, g.nseg_y);
    g.nseg_y = cdiv(g.tiles_y, g.seg_rows);
    g.nsegs = B * g.strips * g.nseg_y;
    const int cus = cova_internal_persistent_grid2(1 << 30, 1);                // (honours the tests' grid cap)
    int walkers = cus / 2 > 0 ? cus / 2 : 1;
    if (walkers > g.nsegs) walkers = g.nsegs;
    g.grid = 2 * walkers;
    return g;
}

int launch_wgrad4(const float *act, const float *act_abc, int act_relu, const float *dz, const float *dz2,
                  const float *dz_abc, float *dz_out, float *ws, int B, int H, int W, hipStream_t st)
{
    const Wg4Geo g = wg4_geometry(B, H, W);
    const Wg4Args a{act, dz, dz_abc ? dz2 : nullptr, ws, act_abc, dz_abc, dz_abc ? dz_out : nullptr, act_relu, H, W,
                    g.tiles_y, g.strips, g.nseg_y, g.seg_rows, g.nsegs};
    const dim3 grid(g.grid), blk(wg4::THREADS);
    const int prod = !dz_abc ? 0 : (dz2 ? 2 : 1);
#define COVA_WG4(PA, PD) hipLaunchKernelGGL((conv3x3_wgrad4_kernel<PA, PD>), grid, blk, 0, st, a)
    if (act_abc) { if (prod == 2) COVA_WG4(true, 2); else if (prod == 1) COVA_WG4(true, 1); else COVA_WG4(true, 0); }
    else         { if (prod == 2) COVA_WG4(false, 2); else if (prod == 1) COVA_WG4(false, 1); else COVA_WG4(false, 0); }
#undef COVA_WG4
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

}  // namespace

// ====================================================================================
// C ABI
// ====================================================================================
COVA_API int cova_conv3x3_wgrad4_num_partials(int B, int H, int W) { return wg4_geometry(B, H, W).grid; }

COVA_API int cova_conv3x3_wgrad4_workspace_floats(int B, int H, int W)
{
    const int grid = wg4_geometry(B, H, W).grid;
    return grid * wg4::PART_FLOATS;          // <= 256 blocks x 36,864 floats
}

// Per-block partial sums of the weight gradient in the F(4x4,3x3) domain; operands transformed on load as
// cova_conv3x3_wgrad_wino_partial: activation = relu?(A*act + C) (act_abc [3,64], B row ignored; NULL = plain),
// gradient = A*dz + B*dz2 + C (dz_abc [3,64], dz2 nullable; NULL = plain).  ws: cova_conv3x3_wgrad4_workspace_floats.
// dz_out (nullable, needs dz_abc): also writes the gradient operand as formed on load, NHWC [B,H,W,64] -- every pixel once.
COVA_API int cova_conv3x3_wgrad4_partial(const float *act, const float *act_abc, int act_relu, const float *dz,
                                         const float *dz2, const float *dz_abc, float *dz_out, float *ws, int B, int H,
                                         int W, void *stream)
{
    COVA_REQUIRE(act && dz && ws && B > 0 && H > 0 && W > 0);
    COVA_REQUIRE((long long)H * W * 256 < (1ll << 32));              // 32-bit byte offsets of the pixel rows inside a page
    COVA_REQUIRE(dz_out == nullptr || dz_abc != nullptr);
    return launch_wgrad4(act, act_abc, act_relu, dz, dz2, dz_abc, dz_out, ws, B, H, W, (hipStream_t)stream);
}

// Fold + final transform of up to four convolutions' partials (pairs 1..3 nullable) into their OIHW gradients
COVA_API int cova_conv3x3_wgrad4_finish(const float *ws0, float *dw0, const float *ws1, float *dw1, const float *ws2,
                                        float *dw2, const float *ws3, float *dw3, int B, int H, int W, void *stream)
{
    COVA_REQUIRE(ws0 && dw0 && B > 0 && H > 0 && W > 0);
    COVA_REQUIRE((ws1 == nullptr) == (dw1 == nullptr) && (ws2 == nullptr) == (dw2 == nullptr) &&
                 (ws3 == nullptr) == (dw3 == nullptr));
    const float *ws[4] = {ws0, ws1, ws2, ws3};
    float *dw[4] = {dw0, dw1, dw2, dw3};
    Wg4FinishJobs jobs{};
    int n = 0;
    for (int i = 0; i < 4; ++i)
        if (ws[i]) { jobs.part[n] = ws[i]; jobs.dw[n] = dw[i]; ++n; }
    const int grid = wg4_geometry(B, H, W).grid;
    hipLaunchKernelGGL(wgrad4_finish_kernel, dim3(4096 / 64, n), dim3(1024), 0, (hipStream_t)stream, jobs, grid);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

#ifdef WG4_TRACE
COVA_API int cova_wg4_trace_read(unsigned long long *host64)
{
    return (int)hipMemcpyFromSymbol(host64, HIP_SYMBOL(g_wg4_trace), sizeof(unsigned long long) * 64);
}
#endif

// act, dz NHWC [B,H,W,64] -> dw OIHW [64,64,3,3] (plain operands, both steps)
COVA_API int cova_conv3x3_wgrad4(const float *act, const float *dz, float *dw, float *ws, int B, int H, int W, void *stream)
{
    COVA_REQUIRE(act && dz && dw && ws && B > 0 && H > 0 && W > 0);
    const int rc = launch_wgrad4(act, nullptr, 0, dz, nullptr, nullptr, nullptr, ws, B, H, W, (hipStream_t)stream);
    if (rc != COVA_OK) return rc;
    return cova_conv3x3_wgrad4_finish(ws, dw, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, B, H, W, stream);
}
