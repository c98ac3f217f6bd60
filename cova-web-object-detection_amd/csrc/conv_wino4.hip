// Winograd F(4x4, 3x3) form of the 3x3 / 64->64 convolution (forward and data gradient): 36 multiplications per
// 4x4 output tile and channel pair instead of 144 (F(2x2,3x3): 64) -- 1.78x fewer MFMAs than conv_wino.hip.
// fp32 error on this layer (post-ReLU input, 64 channels, against fp64): max 2.9e-6 of the output scale
// (F(2x2,3x3) 1.9e-7, direct 3.0e-7), inside the 1e-4 forward / 2e-4 gradient gates of the parity tests.
//
//   Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A          B^T 6x6, G 6x3, A^T 4x6 (Lavin & Gray 2015)
//
// Why this is a different kernel and not a parameter of conv_wino.hip: with 36 transform positions a wave can own
// 16 output channels x 16 tiles at most (36 accumulators of v_mfma_f32_16x16x4_f32 = 144 registers), so four waves
// share every tile and forming B^T d B per wave in registers would repeat the 144-FMA patch transform four times.
// The transformed input is therefore formed ONCE per block and staged through LDS:
//   block = 512 threads, output tile 16 x 32 px = 4 x 8 Winograd tiles; wave w = (channel group w & 3, tile group w >> 2);
//   per 4-channel chunk:  planes (18 x 34 px halo tile, 4 channels) -> column stage -> row stage -> V[36][4][32] in LDS,
//   then 36 MFMAs per wave with both operands from LDS (positions contiguous: one ds_read_b128 = 4 positions).
#include "common.h"

// Ablation builds of tools/wino4_bench.py (COVA_EXTRA_FLAGS=-DW4_ABL=<mask>; 0 in the product): 1 no column stage,
// 2 no row stage, 4 no MFMAs, 8 no global -> LDS copies (16: no plane copies, 32: no weight copies)
#ifndef W4_ABL
#define W4_ABL 0
#endif

namespace {

namespace w4 {
constexpr int TH = 16, TW = 32, PH = TH + 2, PW = TW + 2, NPIX = PH * PW;      // 612 halo pixels
constexpr int THREADS = 512;
constexpr int ROW = 36;                        // floats per operand row = the 36 positions (9 ds_read_b128; 9 x 16 B is an
                                               // odd slot stride: conflict-free without padding)
constexpr int U_FLOATS = 4 * 64 * ROW;         // rows (k, co): 9,216 floats = 36.9 KB
constexpr int V_FLOATS = 4 * 32 * ROW;         // rows (k, tile): 4,608 floats = 18.4 KB
constexpr int IN_FLOATS = 4 * NPIX;            // one chunk of planes, pixel-major [px][4]
constexpr int TMP_CI = 32 * 36 + 8;            // column-stage result [ci][tile][c][i]; +8: the four channels of an
constexpr int TMP_FLOATS = 4 * TMP_CI;         // instruction start 8 banks apart
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

struct W4Args {
    const float *in, *u;
    float *out, *stat_part;
    int H, W, tiles_x, tiles_y, ntiles;
    const float *pro_abc;       // PRO: the conv input is relu?(A[c]*in + C[c]) ([3][64] = A | unused | C), zero outside the image
    int pro_relu;
};

// 256 zero bytes in global memory: halo pixels outside the image are fetched from here
__device__ __attribute__((aligned(256))) float g_w4_zero_page[64];

typedef __attribute__((address_space(3))) void lds_void;

// global -> LDS copy of 16 bytes per lane (lane i lands at lds_base + 16 i; lds_base wave-uniform) as inline asm:
// the compiler then neither knows the copy (no conservative vmcnt(0) in front of every later LDS read, which the
// builtin gets unless each buffer is its own __shared__ object) nor waits for it -- every wait on these copies is
// an explicit s_waitcnt in the kernel.
__device__ __forceinline__ void copy16_to_lds(const float *gptr, unsigned lds_base_bytes)
{
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_base_bytes), "v"(gptr) : "memory", "m0");
}

// Barrier that orders LDS accesses only: this wave's ds_writes are complete (lgkmcnt(0)) while global -> LDS copies
// stay in flight.  __syncthreads() is a release fence over LDS, and a pending copy IS an LDS store: it waits vmcnt(0).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Software pipeline (one iteration = one 4-channel chunk s of a tile; the chunk counter runs across tiles):
//   request weights(s+1) and planes(s+3) as global -> LDS copies (global_load_lds_dwordx4: no registers; the weight
//   chunk in global memory IS the LDS image) | column stage(s+1) | LDS barrier | row stage(s+1) -> V[(s+1)&1] ,
//   36 MFMAs of chunk s from U[s&1], V[s&1] | wait for weights(s+1) -- the planes copy stays in flight | barrier
// so a wave's transform work runs under the MFMAs of the other wave on its SIMD, the weights (L2) have one chunk to
// land and the planes (HBM) two.  768 stage items on 512 threads: waves 0-3 take two, waves 4-7 one -- the two waves
// of a SIMD (w, w+4) together always three.  LDS: 3 x 9.8 (planes) + 18.6 (column stage) + 2 x 36.9 (weights)
// + 2 x 18.4 (V) = 158.6 KB.
template <bool STATS, bool PRO>
__global__ __launch_bounds__(w4::THREADS, 1) void conv3x3_c64_wino4_kernel(const W4Args a)
{
    using namespace w4;
    __shared__ __attribute__((aligned(16))) float s_in[3 * IN_FLOATS], s_tmp[TMP_FLOATS];
    __shared__ __attribute__((aligned(16))) float s_u[2 * U_FLOATS], s_v[2 * V_FLOATS];
    float *s_red = s_tmp;                         // (after the tile loop)
    __shared__ float s_pro[PRO ? 128 : 1];        // A | C of the affine-on-load prologue
    if (PRO) {
        if (threadIdx.x < 64) s_pro[threadIdx.x] = a.pro_abc[threadIdx.x];
        else if (threadIdx.x < 128) s_pro[threadIdx.x] = a.pro_abc[64 + threadIdx.x];
        __syncthreads();
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cog = wave & 3, grp = wave >> 2;
    const int l15 = lane & 15, kq = lane >> 4;
    const int H = a.H, W = a.W;
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};

    // ---- copies.  planes: pixel-major [px][4 channels]; thread px (and px + 512 for threads < 100)
    const int px1 = tid + THREADS;
    const int pr0 = tid / PW, pc0 = tid - pr0 * PW, pr1 = px1 / PW, pc1 = px1 - pr1 * PW;
    const unsigned in_base = (unsigned)(size_t)(lds_void *)s_in, u_base = (unsigned)(size_t)(lds_void *)s_u;
    // this thread's source pixels of a tile, channel 0 (out-of-image pixels: the zero page); computed once per tile,
    // a chunk's copy only adds 16 s bytes
    struct PlaneSrc { const float *p0, *p1; int step; };          // step = 4 floats per chunk, 0 for the zero page
    auto plane_src = [&](int tile_) {
        if (tile_ >= a.ntiles) tile_ = a.ntiles - 1;             // past the end: a valid address, result unused
        const int tx = tile_ % a.tiles_x, ty = (tile_ / a.tiles_x) % a.tiles_y, b = tile_ / (a.tiles_x * a.tiles_y);
        const float *base = a.in + (size_t)b * H * W * 64;
        const int gy0 = ty * TH + pr0 - 1, gx0 = tx * TW + pc0 - 1;
        const int gy1 = ty * TH + pr1 - 1, gx1 = tx * TW + pc1 - 1;
        const bool in0 = gy0 >= 0 && gy0 < H && gx0 >= 0 && gx0 < W;
        const bool in1 = gy1 >= 0 && gy1 < H && gx1 >= 0 && gx1 < W;
        PlaneSrc r;
        // (the zero page holds 64 floats: the per-chunk offset 4 s <= 60 stays inside it)
        r.p0 = in0 ? base + ((size_t)gy0 * W + gx0) * 64 : g_w4_zero_page;
        r.p1 = in1 ? base + ((size_t)gy1 * W + gx1) * 64 : g_w4_zero_page;
        r.step = 4;
        return r;
    };
    auto copy_planes = [&](const PlaneSrc &src, int s, unsigned slot_bytes) {
        // lane i of a wave lands at (wave-uniform LDS base) + 16 i
        copy16_to_lds(src.p0 + src.step * s, slot_bytes + wave * 1024);
        if (px1 < NPIX) copy16_to_lds(src.p1 + src.step * s, slot_bytes + (8 + wave) * 1024);
    };
    auto copy_u = [&](int s, unsigned dst_bytes) {               // 2304 float4: 4.5 per thread, linear
        const float *ug = a.u + (size_t)s * U_FLOATS + tid * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) copy16_to_lds(ug + j * THREADS * 4, dst_bytes + (j * 8 + wave) * 1024);
        if (wave < 4) copy16_to_lds(ug + 4 * THREADS * 4, dst_bytes + (32 + wave) * 1024);
    };
    // Wait on the copies (vmcnt retires in order; per chunk a wave issues its weight instructions, then P plane
    // instructions, P = 2 on waves 0-1 and 1 elsewhere): weights(s+1) -- and planes(s+2), requested one iteration
    // earlier -- have landed when at most P instructions are outstanding; the planes(s+3) copy stays in flight.
    auto wait_copies = [&]() {
        if (wave < 2) __builtin_amdgcn_s_waitcnt(0x0072);
        else __builtin_amdgcn_s_waitcnt(0x0071);
    };
    // ---- stage items of this thread (e = 0, and e = 1 for threads < 256).
    // column stage: it = (tile*6 + c)*4 + ci -- the channel fastest, as the planes are [px][4];
    // row stage:    it = (ci*32 + tile)*6 + i -- the channel slowest, as V rows are (ci, tile)
    int col_src[2], col_dst[2], row_src[2], row_dst[2], col_x[2], col_y[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int it = tid + e * THREADS;
        {
            const int ci = it & 3, tc = it >> 2, t = tc / 6, c = tc - t * 6;
            col_src[e] = ((4 * (t >> 3)) * PW + 4 * (t & 7) + c) * 4 + ci;      // + r * PW * 4
            col_dst[e] = ci * TMP_CI + t * 36 + c * 6;                           // + i (6 contiguous)
            col_x[e] = 4 * (t & 7) + c - 1;                                      // image column / first row of the item,
            col_y[e] = 4 * (t >> 3) - 1;                                         // relative to the tile origin
        }
        {
            const int i = it % 6, pt = it / 6, t = pt & 31, ci = pt >> 5;
            row_src[e] = ci * TMP_CI + t * 36 + i;                               // + c * 6
            row_dst[e] = (ci * 32 + t) * ROW + 6 * i;                            // + j (6 contiguous)
        }
    }
    // PRO: rows of an item's column that lie inside the image (bit r), for the tile the planes belong to
    auto row_mask = [&](int tile_, int e) -> unsigned {
        if (tile_ >= a.ntiles) return 0u;
        const int tx_ = tile_ % a.tiles_x, ty_ = (tile_ / a.tiles_x) % a.tiles_y;
        const int gx = tx_ * TW + col_x[e], gy = ty_ * TH + col_y[e];
        if (gx < 0 || gx >= W) return 0u;
        unsigned m = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r) m |= (gy + r >= 0 && gy + r < H) ? (1u << r) : 0u;
        return m;
    };
    // tmp[ci][tile][c][i] = sum_r Bt[i][r] d[r][c];  ch0 = first channel of the chunk, vm = row masks (PRO)
    auto column_stage = [&](const float *slot, int ch0, const unsigned (&vm)[2]) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (e == 1 && tid >= 256) break;
            const float *p = slot + col_src[e];
            float d[6], o[6];
#pragma unroll
            for (int r = 0; r < 6; ++r) d[r] = p[r * PW * 4];
            if (PRO) {
                const int ch = ch0 + ((tid + e * THREADS) & 3);
                const float A = s_pro[ch], C = s_pro[64 + ch];
#pragma unroll
                for (int r = 0; r < 6; ++r) {
                    float v = fmaf(A, d[r], C);
                    if (a.pro_relu) v = fmaxf(v, 0.f);
                    d[r] = ((vm[e] >> r) & 1u) ? v : 0.f;
                }
            }
            bt6(d, o);
            float *q = s_tmp + col_dst[e];
            *reinterpret_cast<float2 *>(q) = make_float2(o[0], o[1]);
            *reinterpret_cast<float2 *>(q + 2) = make_float2(o[2], o[3]);
            *reinterpret_cast<float2 *>(q + 4) = make_float2(o[4], o[5]);
        }
    };
    auto row_stage = [&](float *vdst) {               // V[(ci, tile)][6i + j] = sum_c tmp[c][i] Bt[j][c]
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            if (e == 1 && tid >= 256) break;
            const float *p = s_tmp + row_src[e];
            float d[6], o[6];
#pragma unroll
            for (int c = 0; c < 6; ++c) d[c] = p[c * 6];
            bt6(d, o);
            float *q = vdst + row_dst[e];
            *reinterpret_cast<float2 *>(q) = make_float2(o[0], o[1]);
            *reinterpret_cast<float2 *>(q + 2) = make_float2(o[2], o[3]);
            *reinterpret_cast<float2 *>(q + 4) = make_float2(o[4], o[5]);
        }
    };

    int tile = blockIdx.x;
    if (tile >= a.ntiles) return;
    // ---- prime the pipeline: V(0), U(0), planes(1), planes(2) in LDS
    PlaneSrc src_cur = plane_src(tile), src_nxt = src_cur;
    copy_planes(src_cur, 0, in_base);
    copy_u(0, u_base);
    copy_planes(src_cur, 1, in_base + IN_FLOATS * 4);
    copy_planes(src_cur, 2, in_base + 2 * IN_FLOATS * 4);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    unsigned vm_cur[2] = {0u, 0u}, vm_nxt[2] = {0u, 0u};
    if (PRO) {
        vm_cur[0] = row_mask(tile, 0);
        vm_cur[1] = row_mask(tile, 1);
    }
    column_stage(s_in, 0, vm_cur);
    __syncthreads();
    row_stage(s_v);
    __syncthreads();

    int slot = 0;                                     // planes ring: slot of the current chunk = (chunk counter) mod 3
    for (; tile < a.ntiles; tile += gridDim.x) {
        const int tx = tile % a.tiles_x, ty = (tile / a.tiles_x) % a.tiles_y, b = tile / (a.tiles_x * a.tiles_y);
        const int y0 = ty * TH, x0 = tx * TW;
        const size_t img = (size_t)b * H * W * 64;
        f32x4 acc[36];
#pragma unroll
        for (int p = 0; p < 36; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
        src_nxt = plane_src(tile + (int)gridDim.x);
        if (PRO) {                                    // chunk 0 of the NEXT tile is transformed in this tile's last iteration
            vm_nxt[0] = row_mask(tile + (int)gridDim.x, 0);
            vm_nxt[1] = row_mask(tile + (int)gridDim.x, 1);
        }

#pragma unroll 1
        for (int s = 0; s < 16; ++s) {
            const int cur = s & 1, nxt = cur ^ 1;
            const int slot_nxt = slot == 2 ? 0 : slot + 1;
            // slot `slot` held planes(s), consumed by the previous iteration's column stage -> planes(s+3)
            if (!(W4_ABL & 8)) {
                if (!(W4_ABL & 32)) copy_u((s + 1) & 15, u_base + nxt * U_FLOATS * 4);
                if (!(W4_ABL & 16))
                    copy_planes(s + 3 < 16 ? src_cur : src_nxt, (s + 3) & 15, in_base + slot * IN_FLOATS * 4);
            }
            // position quads [q0, q1) of chunk s: D[co][tile] += U[co][ci] * V[ci][tile]
            const float *ua = s_u + cur * U_FLOATS + (kq * 64 + cog * 16 + l15) * ROW;
            const float *vb = s_v + cur * V_FLOATS + (kq * 32 + grp * 16 + l15) * ROW;
            auto mfmas = [&](const int q0, const int q1) {
                if (W4_ABL & 4) return;
#pragma unroll
                for (int q = q0; q < q1; ++q) {
                    const float4 u4 = *reinterpret_cast<const float4 *>(ua + 4 * q);
                    const float4 v4 = *reinterpret_cast<const float4 *>(vb + 4 * q);
                    acc[4 * q + 0] = mfma16x4(u4.x, v4.x, acc[4 * q + 0]);
                    acc[4 * q + 1] = mfma16x4(u4.y, v4.y, acc[4 * q + 1]);
                    acc[4 * q + 2] = mfma16x4(u4.z, v4.z, acc[4 * q + 2]);
                    acc[4 * q + 3] = mfma16x4(u4.w, v4.w, acc[4 * q + 3]);
                }
            };
            // The two waves of a SIMD (w and w + 4: tile groups 0 and 1) run the iteration in opposite order, so that
            // one of them is always in its MFMAs while the other does transform work -- the barriers would otherwise
            // line all eight waves up in the same phase and leave the matrix pipe idle during both stages.
            // (the MFMA code exists once; only the stage calls, which do not touch the accumulators, sit under the
            //  wave-uniform branches -- accumulator updates in both arms of a branch made the allocator spill them)
            const int ch_nxt = 4 * ((s + 1) & 15);
            if (grp == 0 && !(W4_ABL & 1)) column_stage(s_in + slot_nxt * IN_FLOATS, ch_nxt, s == 15 ? vm_nxt : vm_cur);   // planes(s+1)
            mfmas(0, 5);
            if (grp != 0 && !(W4_ABL & 1)) column_stage(s_in + slot_nxt * IN_FLOATS, ch_nxt, s == 15 ? vm_nxt : vm_cur);
            lds_barrier();
            if (grp == 0 && !(W4_ABL & 2)) row_stage(s_v + nxt * V_FLOATS);
            mfmas(5, 9);
            if (grp != 0 && !(W4_ABL & 2)) row_stage(s_v + nxt * V_FLOATS);
            wait_copies();                      // weights(s+1) and planes(s+2) have landed; planes(s+3) stays in flight
            asm volatile("s_barrier" ::: "memory");
            slot = slot_nxt;
        }
        src_cur = src_nxt;
        if (PRO) { vm_cur[0] = vm_nxt[0]; vm_cur[1] = vm_nxt[1]; }
        // ---- output transform Y = A^T M A in registers: lane = (tile grp*16 + l15, channels cog*16 + kq*4 .. +3)
        const int t = grp * 16 + l15;
        const int oy0 = y0 + 4 * (t >> 3), ox0 = x0 + 4 * (t & 7);
        const int co0 = cog * 16 + kq * 4;
        float y[4][4][4];                   // [row][col][channel]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float tm[4][6];                 // A^T M: [out row][j]
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                float m[6], o[4];
#pragma unroll
                for (int i = 0; i < 6; ++i) m[i] = acc[6 * i + j][r];
                at6(m, o);
#pragma unroll
                for (int i = 0; i < 4; ++i) tm[i][j] = o[i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float o[4];
                at6(tm[i], o);
#pragma unroll
                for (int j = 0; j < 4; ++j) y[i][j][r] = o[j];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int oy = oy0 + i, ox = ox0 + j;
                if (oy < H && ox < W) {
                    *reinterpret_cast<float4 *>(a.out + img + ((size_t)oy * W + ox) * 64 + co0) =
                        make_float4(y[i][j][0], y[i][j][1], y[i][j][2], y[i][j][3]);
                    if (STATS) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            ssum[r] += y[i][j][r];
                            ssq[r] += y[i][j][r] * y[i][j][r];
                        }
                    }
                }
            }
    }
    if (STATS) {        // one partial row [sum 64 | sum of squares 64] per block
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            ssum[r] = row16_sum(ssum[r]);
            ssq[r] = row16_sum(ssq[r]);
        }
        __syncthreads();
        if (l15 == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s_red[grp * 128 + cog * 16 + kq * 4 + r] = ssum[r];
                s_red[grp * 128 + 64 + cog * 16 + kq * 4 + r] = ssq[r];
            }
        }
        __syncthreads();
        if (tid < 128) a.stat_part[(size_t)blockIdx.x * 128 + tid] = s_red[tid] + s_red[128 + tid];
    }
}

// U[s][k][co][pos = i*6 + j] = (G g G^T)[i][j] for input channel 4s+k.
//  fwd:   g = w[co][ci][:, :];   dgrad: output channel = ci, input channel = co, g = w[co][ci] rotated by 180 degrees
__global__ void prep_wino4_kernel(const float *__restrict__ w, float *__restrict__ u_fwd, float *__restrict__ u_dgrad)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;    // over [s 16][row = k*64 + o][pos 36]: the LDS image
    if (idx >= 16 * 256 * 36) return;
    const int pos = idx % 36, row = (idx / 36) & 255, s = idx / (36 * 256);
    const int o = row & 63, k = row >> 6, c = 4 * s + k;
    const int i = pos / 6, j = pos - 6 * i;
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
}

}  // namespace

int cova_internal_persistent_grid2(int ntiles, int blocks_per_cu);

// ====================================================================================
// C ABI (experimental: the F(2x2,3x3) kernels of conv_wino.hip stay the engine's default)
// ====================================================================================
COVA_API int cova_conv3x3_wino4_num_tiles(int B, int H, int W)
{
    return B * cdiv(W, w4::TW) * cdiv(H, w4::TH);
}

COVA_API int cova_conv3x3_wino4_num_partials(int B, int H, int W)
{
    return cova_internal_persistent_grid2(cova_conv3x3_wino4_num_tiles(B, H, W), 1);
}

// u_fwd / u_dgrad: [16 chunks][4 ci][64 co][36 positions] floats each (147,456): a chunk is the LDS image
COVA_API int cova_conv3x3_wino4_prep(const float *w_oihw, float *u_fwd, float *u_dgrad, void *stream)
{
    COVA_REQUIRE(w_oihw && u_fwd && u_dgrad);
    hipLaunchKernelGGL(prep_wino4_kernel, dim3(cdiv(16 * 256 * 36, 256)), dim3(256), 0, (hipStream_t)stream,
                       w_oihw, u_fwd, u_dgrad);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

// out NHWC [B,H,W,64] = conv3x3(in) with the F(4x4,3x3) weights `u`; stat_part (nullable):
// [cova_conv3x3_wino4_num_partials][2][64] = (sum y, sum y^2)
COVA_API int cova_conv3x3_wino4(const float *in, const float *u, float *out, float *stat_part, int B, int H, int W,
                                void *stream)
{
    COVA_REQUIRE(in && u && out && B > 0 && H > 0 && W > 0);
    const int tiles_x = cdiv(W, w4::TW), tiles_y = cdiv(H, w4::TH), ntiles = B * tiles_x * tiles_y;
    const W4Args a{in, u, out, stat_part, H, W, tiles_x, tiles_y, ntiles, nullptr, 0};
    const int grid = cova_internal_persistent_grid2(ntiles, 1);
    if (stat_part)
        hipLaunchKernelGGL((conv3x3_c64_wino4_kernel<true, false>), dim3(grid), dim3(w4::THREADS), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((conv3x3_c64_wino4_kernel<false, false>), dim3(grid), dim3(w4::THREADS), 0, (hipStream_t)stream, a);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

// The same convolution on relu?(A[c]*in + C[c]) formed on load (pro_abc [3][64] = A | unused | C; zero padding stays
// zero): a BatchNorm(+ReLU) folded into the consuming conv, as cova_conv3x3_wino_pro without a second tensor.
COVA_API int cova_conv3x3_wino4_pro(const float *in, const float *pro_abc, int pro_relu, const float *u, float *out,
                                    float *stat_part, int B, int H, int W, void *stream)
{
    COVA_REQUIRE(in && pro_abc && u && out && B > 0 && H > 0 && W > 0);
    const int tiles_x = cdiv(W, w4::TW), tiles_y = cdiv(H, w4::TH), ntiles = B * tiles_x * tiles_y;
    const W4Args a{in, u, out, stat_part, H, W, tiles_x, tiles_y, ntiles, pro_abc, pro_relu};
    const int grid = cova_internal_persistent_grid2(ntiles, 1);
    if (stat_part)
        hipLaunchKernelGGL((conv3x3_c64_wino4_kernel<true, true>), dim3(grid), dim3(w4::THREADS), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((conv3x3_c64_wino4_kernel<false, true>), dim3(grid), dim3(w4::THREADS), 0, (hipStream_t)stream, a);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}
