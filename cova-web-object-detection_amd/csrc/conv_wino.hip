// Winograd F(2x2, 3x3) form of the 3x3 / 64->64 convolution (forward and data gradient).
//
//   Y = A^T [ sum_ci (G g G^T) .* (B^T d B) ] A      (Lavin & Gray 2015)
//
// 16 multiplications per 2x2 output tile and channel pair instead of 36: the MFMA work of the
// dominant kernel drops by 2.25x, which is the only way past the direct-convolution ceiling in
// exact-f32 arithmetic.  fp32 error of F(2x2,3x3) equals the direct form's (transform
// coefficients are +-1, +-1/2): 3.3e-7 vs 3.2e-7 against fp64 on this layer.
//
// Mapping (one persistent 512-thread block per CU, output tile 8 x 32 px = 4 x 16 Winograd tiles):
//   * wave w owns Winograd tile row (w & 3) (16 tiles) and 32 output channels (w >> 2), for ALL
//     16 transform positions: 2 x 16 accumulators of v_mfma_f32_16x16x4_f32 (128 VGPRs), so the
//     output transform A^T M A happens in registers and results go straight to HBM;
//   * the MFMA A operand V = B^T d B is formed on the fly: lane (tile = l & 15, ci = 4s + (l >> 4))
//     reads its 4 x 4 input patch from LDS (16 ds_read_b32 with immediate offsets) and needs 32
//     adds for the 16 positions -- no transformed-input buffer exists anywhere;
//   * the input tile lives in LDS channel-major [ci][341 px] (odd pixel stride => the stride-2 tile
//     walk and the 4 channels of a k-step hit 32 distinct banks);
//   * the transformed weights U[chunk s][ci 4][co 64][pos 16] (prepared once per step) stream through
//     a double-buffered 2 x 20 KB LDS ring, one 4-channel chunk per barrier; a lane's 16 positions
//     are contiguous (4 ds_read_b128 per channel block), rows padded to 20 floats (conflict free).
// Per tile: 16 chunks x (16 A reads + 32 adds + 32 B reads + 32 MFMAs) per wave.
#include "common.h"

namespace {

struct BnBwdEpiW {
    const float *act, *z, *mean, *invstd;
};

namespace wn {
constexpr int TH = 8, TW = 32, PH = TH + 2, PW = TW + 2;
constexpr int NPIX = PH * PW;                 // 340
constexpr int PIXS = 341;                     // odd channel-plane stride
constexpr int IN_FLOATS = 64 * PIXS;          // 21,824 floats = 87.3 KB
constexpr int UROW = 20;                      // one (ci, co) row = 16 positions + 4 pad floats
constexpr int UCH = 4 * 64 * UROW;            // 5,120 floats = 20 KB per chunk in LDS
constexpr int UCH_G = 16 * 4 * 64;            // 4,096 floats per chunk in global memory
constexpr int THREADS = 512;
constexpr int RED_FLOATS = 8 * 64;
constexpr int LDS_FLOATS = IN_FLOATS + 2 * UCH + RED_FLOATS;   // 130.3 KB
}  // namespace wn

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c)
{
    // v_mfma_f32_16x16x4_f32: lane l supplies A[i = l&15][k = l>>4], B[k = l>>4][j = l&15];
    // D register r of lane l is D[row = (l>>4)*4 + r][col = l&15]
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Input transform of one 4x4 patch: V = B^T d B (32 adds), d read from LDS with immediate offsets.
__device__ __forceinline__ void wino_input_transform(const float *__restrict__ a, float (&V)[16])
{
    using namespace wn;
    float d[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) d[r][c] = a[r * PW + c];
    float t[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        t[0][c] = d[0][c] - d[2][c];
        t[1][c] = d[1][c] + d[2][c];
        t[2][c] = d[2][c] - d[1][c];
        t[3][c] = d[1][c] - d[3][c];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        V[r * 4 + 0] = t[r][0] - t[r][2];
        V[r * 4 + 1] = t[r][1] + t[r][2];
        V[r * 4 + 2] = t[r][2] - t[r][1];
        V[r * 4 + 3] = t[r][1] - t[r][3];
    }
}

template <bool STATS>
__global__ __launch_bounds__(wn::THREADS) void conv3x3_c64_wino_kernel(
    const float *__restrict__ in, const float *__restrict__ ug, const float *__restrict__ addend,
    float *__restrict__ out, float *__restrict__ stat_part, int H, int W, int tiles_x, int tiles_y,
    int ntiles, const BnBwdEpiW bn, int abl_arg)
{
    const int abl = COVA_ABL(abl_arg);
    // abl (tools/conv_bench.py only, 0 in production): 1 no epilogue, 2 no refill stores,
    // 4 no refill loads, 8 no weight restaging, 16 no per-chunk barrier, 32 no input transform
    using namespace wn;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    float *s_in = lds;
    float *s_u = lds + IN_FLOATS;
    float *s_red = lds + IN_FLOATS + 2 * UCH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tb = wave & 3, cbp = wave >> 2;
    const int ti = lane & 15, kq = lane >> 4;

    int tile = blockIdx.x;       // XCD-aware order (see conv.hip)
    if ((gridDim.x & 7) == 0) tile = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (tile >= ntiles) return;

    // transformed-weight chunk s: 1024 float4 in global memory, 2 per thread
    auto u_lds_off = [&](int f) {            // float4 index -> LDS float offset inside a chunk
        const int flat = f * 4;                 // global chunk layout [k 4][co 64][pos 16]
        return (flat >> 4) * UROW + (flat & 15);
    };
    const int uo0 = u_lds_off(tid), uo1 = u_lds_off(tid + THREADS);

    // ---- first tile: whole halo'd input tile, channel-major in LDS
    {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
        const float *in_b = in + (size_t)b * H * W * 64;
#pragma unroll 1
        for (int idx = tid; idx < NPIX * 16; idx += THREADS) {
            const int px = idx >> 4, c = (idx & 15) * 4;
            const int r = px / PW, cc = px - r * PW;
            const int gy = ty * TH + r - 1, gx = tx * TW + cc - 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy >= 0 && gy < H && gx >= 0 && gx < W)
                v = *reinterpret_cast<const float4 *>(in_b + ((size_t)gy * W + gx) * 64 + c);
            s_in[(c + 0) * PIXS + px] = v.x;
            s_in[(c + 1) * PIXS + px] = v.y;
            s_in[(c + 2) * PIXS + px] = v.z;
            s_in[(c + 3) * PIXS + px] = v.w;
        }
        const float4 u0 = reinterpret_cast<const float4 *>(ug)[tid];
        const float4 u1 = reinterpret_cast<const float4 *>(ug)[tid + THREADS];
        *reinterpret_cast<float4 *>(s_u + uo0) = u0;
        *reinterpret_cast<float4 *>(s_u + uo1) = u1;
    }
    __syncthreads();

    // Streaming refill: chunk s of a tile reads only channel planes 4s..4s+3, so once its barrier
    // has passed those planes are dead and are overwritten with the NEXT tile's data two chunks
    // later (loaded into a 2-deep register ring in between).  The tile buffer is therefore refilled
    // during the MFMAs, 4 planes per chunk: no tile-sized prefetch registers, no refill phase
    // between tiles, and HBM sees a perfectly steady stream.  Thread px (< 340) owns pixel px.
    const int rpx = tid < NPIX ? tid : 0;
    const int rrow = rpx / PW, rcol = rpx - rrow * PW;
    float4 ring1 = make_float4(0.f, 0.f, 0.f, 0.f), ring2 = ring1;   // loaded 1 / 2 steps ago
    int ring1_plane = -1, ring2_plane = -1;                            // -1: nothing to write

    int ubuf = 0;
    const float *a_lane = s_in + kq * PIXS + (2 * tb) * PW + 2 * ti;      // this lane's patch origin
    const float *b_lane = s_u + (kq * 64 + cbp * 32 + ti) * UROW;   // row (ci kq, co cbp*32 + ti)
    float V[16];
    wino_input_transform(a_lane, V);                                       // chunk 0 of the first tile
    for (; tile < ntiles; tile += gridDim.x) {
        const int tx = tile % tiles_x;
        const int ty = (tile / tiles_x) % tiles_y;
        const int b = tile / (tiles_x * tiles_y);
        const int y0 = ty * TH, x0 = tx * TW;
        const int next = tile + gridDim.x;
        const bool has_next = next < ntiles;
        const int ntx = next % tiles_x, nty = (next / tiles_x) % tiles_y;
        const float *nin_b = in + (size_t)(next / (tiles_x * tiles_y)) * H * W * 64;
        const int ngy = nty * TH + rrow - 1, ngx = ntx * TW + rcol - 1;
        const bool nload = has_next && tid < NPIX && ngy >= 0 && ngy < H && ngx >= 0 && ngx < W;
        const float *nsrc = nin_b + ((size_t)ngy * W + ngx) * 64;

        f32x4 acc[2][16];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int p = 0; p < 16; ++p) acc[c][p] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
        for (int s = 0; s < 16; ++s) {
            // (1) refill: planes consumed two chunks ago <- data of the tile after theirs
            if (ring2_plane >= 0 && tid < NPIX && !(abl & 2)) {
                float *dst = s_in + (size_t)ring2_plane * PIXS + rpx;
                dst[0] = ring2.x;
                dst[PIXS] = ring2.y;
                dst[2 * PIXS] = ring2.z;
                dst[3 * PIXS] = ring2.w;
            }
            ring2 = ring1;
            ring2_plane = ring1_plane;
            // (2) fetch this chunk's planes for the next tile (zero outside the image)
            ring1 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (nload && !(abl & 4)) ring1 = *reinterpret_cast<const float4 *>(nsrc + 4 * s);
            ring1_plane = has_next ? 4 * s : -1;
            // (3) next chunk of transformed weights (chunk 0 again for the next tile)
            const int ns = (s + 1) & 15;
            float4 un0 = make_float4(0, 0, 0, 0), un1 = un0;
            if (!(abl & 8)) {
                un0 = reinterpret_cast<const float4 *>(ug + (size_t)ns * UCH_G)[tid];
                un1 = reinterpret_cast<const float4 *>(ug + (size_t)ns * UCH_G)[tid + THREADS];
            }
            __builtin_amdgcn_sched_barrier(0);

            // (4) 32 MFMAs of this chunk
            // the 16 positions of a (ci, co) pair are contiguous: 4 ds_read_b128 per channel block
            const float *bp = b_lane + ubuf * UCH;
#pragma unroll
            for (int p4 = 0; p4 < 4; ++p4) {
                const float4 u0 = *reinterpret_cast<const float4 *>(bp + p4 * 4);
                const float4 u1 = *reinterpret_cast<const float4 *>(bp + 16 * UROW + p4 * 4);
                // D[co][tile] += U[co][ci] * V[ci][tile]
                acc[0][p4 * 4 + 0] = mfma16(u0.x, V[p4 * 4 + 0], acc[0][p4 * 4 + 0]);
                acc[1][p4 * 4 + 0] = mfma16(u1.x, V[p4 * 4 + 0], acc[1][p4 * 4 + 0]);
                acc[0][p4 * 4 + 1] = mfma16(u0.y, V[p4 * 4 + 1], acc[0][p4 * 4 + 1]);
                acc[1][p4 * 4 + 1] = mfma16(u1.y, V[p4 * 4 + 1], acc[1][p4 * 4 + 1]);
                acc[0][p4 * 4 + 2] = mfma16(u0.z, V[p4 * 4 + 2], acc[0][p4 * 4 + 2]);
                acc[1][p4 * 4 + 2] = mfma16(u1.z, V[p4 * 4 + 2], acc[1][p4 * 4 + 2]);
                acc[0][p4 * 4 + 3] = mfma16(u0.w, V[p4 * 4 + 3], acc[0][p4 * 4 + 3]);
                acc[1][p4 * 4 + 3] = mfma16(u1.w, V[p4 * 4 + 3], acc[1][p4 * 4 + 3]);
            }
            // (5) input transform of the NEXT chunk (for s = 15: chunk 0 of the next tile, whose
            //     planes were refilled at chunk 2 of this tile) -- overlaps the MFMAs in flight
            if (!(abl & 32)) wino_input_transform(a_lane + (4 * ns) * PIXS, V);
            // (6) publish the next weight chunk
            if (!(abl & 8)) {
                float *ud = s_u + (ubuf ^ 1) * UCH;
                *reinterpret_cast<float4 *>(ud + uo0) = un0;
                *reinterpret_cast<float4 *>(ud + uo1) = un1;
                ubuf ^= 1;
            }
            if (!(abl & 16)) __syncthreads();
        }

        // ---- output transform Y = A^T M A in registers, straight to HBM.
        // D layout (rows = co, cols = tiles): this lane holds, for c2 = 0..1 and q = 0..3, channel
        // cbp*32 + c2*16 + kq*4 + q of Winograd tile (tb, ti) -- four consecutive channels per
        // pixel, so every epilogue access is a float4.
        float ssum[2][4], ssq[2][4];
        if (abl & 16) __syncthreads();
        if (abl & 1) {
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int p = 0; p < 16; ++p) asm volatile("" ::"v"(acc[c2][p]));
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int q = 0; q < 4; ++q) { ssum[c2][q] = 0.f; ssq[c2][q] = 0.f; }
        } else
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            const int co = cbp * 32 + c2 * 16 + kq * 4;
            float4 mu4 = make_float4(0.f, 0.f, 0.f, 0.f), is4 = mu4;
            if (bn.z != nullptr) {
                mu4 = *reinterpret_cast<const float4 *>(bn.mean + co);
                is4 = *reinterpret_cast<const float4 *>(bn.invstd + co);
            }
            const float mu[4] = {mu4.x, mu4.y, mu4.z, mu4.w}, is[4] = {is4.x, is4.y, is4.z, is4.w};
            float y[2][2][4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float s0[4], s1[4];
#pragma unroll
                for (int bcol = 0; bcol < 4; ++bcol) {
                    const float m0 = acc[c2][0 * 4 + bcol][q], m1 = acc[c2][1 * 4 + bcol][q];
                    const float m2 = acc[c2][2 * 4 + bcol][q], m3 = acc[c2][3 * 4 + bcol][q];
                    s0[bcol] = m0 + m1 + m2;
                    s1[bcol] = m1 - m2 - m3;
                }
                y[0][0][q] = s0[0] + s0[1] + s0[2];
                y[0][1][q] = s0[1] - s0[2] - s0[3];
                y[1][0][q] = s1[0] + s1[1] + s1[2];
                y[1][1][q] = s1[1] - s1[2] - s1[3];
                ssum[c2][q] = 0.f;
                ssq[c2][q] = 0.f;
            }
#pragma unroll
            for (int yy = 0; yy < 2; ++yy) {
                const int oy = y0 + 2 * tb + yy;
#pragma unroll
                for (int xx = 0; xx < 2; ++xx) {
                    const int ox = x0 + 2 * ti + xx;
                    if (oy < H && ox < W) {
                        const size_t o = (((size_t)b * H + oy) * W + ox) * 64 + co;
                        float v[4] = {y[yy][xx][0], y[yy][xx][1], y[yy][xx][2], y[yy][xx][3]};
                        if (addend != nullptr) {
                            const float4 ad = *reinterpret_cast<const float4 *>(addend + o);
                            v[0] += ad.x; v[1] += ad.y; v[2] += ad.z; v[3] += ad.w;
                        }
                        if (bn.z != nullptr) {
                            const float4 a4 = *reinterpret_cast<const float4 *>(bn.act + o);
                            const float4 z4 = *reinterpret_cast<const float4 *>(bn.z + o);
                            const float av[4] = {a4.x, a4.y, a4.z, a4.w}, zv[4] = {z4.x, z4.y, z4.z, z4.w};
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                if (!(av[q] > 0.f)) v[q] = 0.f;
                                ssum[c2][q] += v[q];
                                ssq[c2][q] += v[q] * ((zv[q] - mu[q]) * is[q]);
                            }
                        } else {
#pragma unroll
                            for (int q = 0; q < 4; ++q) { ssum[c2][q] += v[q]; ssq[c2][q] += v[q] * v[q]; }
                        }
                        *reinterpret_cast<float4 *>(out + o) = make_float4(v[0], v[1], v[2], v[3]);
                    }
                }
            }
        }
        if (STATS) {
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) {
                        ssum[c2][q] += __shfl_xor(ssum[c2][q], o, 64);
                        ssq[c2][q] += __shfl_xor(ssq[c2][q], o, 64);
                    }
                }
            if (ti == 0) {         // s_red[wave][0..31] sums, [32..63] sums of squares (32 channels)
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        s_red[wave * 64 + c2 * 16 + kq * 4 + q] = ssum[c2][q];
                        s_red[wave * 64 + 32 + c2 * 16 + kq * 4 + q] = ssq[c2][q];
                    }
            }
            __syncthreads();
            if (tid < 128) {
                const int which = tid >> 6, ch = tid & 63;        // 0 = sum, 1 = sumsq
                const int half = ch >> 5, idx = ch & 31;
                float tsum = 0.f;
#pragma unroll
                for (int w4 = 0; w4 < 4; ++w4) tsum += s_red[(half * 4 + w4) * 64 + which * 32 + idx];
                stat_part[(size_t)tile * 128 + tid] = tsum;
            }
        }
        __syncthreads();
    }
}

// U[s][k][co][pos = a*4+b] = (G g G^T)[a][b] for input channel 4s+k.
//  fwd:   g = w[co][ci][:, :]
//  dgrad: output channel = ci, input channel = co, g = w[co][ci] rotated by 180 degrees
__global__ void prep_wino_kernel(const float *__restrict__ w, float *__restrict__ u_fwd,
                                 float *__restrict__ u_dgrad)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;    // over [s 16][k 4][o 64][pos 16]
    if (idx >= 16 * 16 * 4 * 64) return;
    const int pos = idx & 15, o = (idx >> 4) & 63, k = (idx >> 10) & 3, s = idx >> 12;
    const int c = 4 * s + k, a = pos >> 2, bb = pos & 3;
    const float G[4][3] = {{1.f, 0.f, 0.f}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0.f, 0.f, 1.f}};
    float uf = 0.f, ud = 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const float coef = G[a][r] * G[bb][t];
            uf += coef * w[((o * 64 + c) * 3 + r) * 3 + t];
            ud += coef * w[((c * 64 + o) * 3 + (2 - r)) * 3 + (2 - t)];
        }
    u_fwd[idx] = uf;
    u_dgrad[idx] = ud;
}

}  // namespace

extern int cova_internal_persistent_grid(int ntiles);
extern int cova_internal_ablate();

// u_fwd / u_dgrad: [16 chunks][4 ci][64 co][16 positions] floats each (65,536)
COVA_API int cova_conv3x3_prep_weights_wino(const float *w_oihw, float *u_fwd, float *u_dgrad,
                                            void *stream)
{
    COVA_REQUIRE(w_oihw && u_fwd && u_dgrad);
    hipLaunchKernelGGL(prep_wino_kernel, dim3(65536 / 256), dim3(256), 0, (hipStream_t)stream, w_oihw,
                       u_fwd, u_dgrad);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

// Same contract as cova_conv3x3_fwd / cova_conv3x3_dgrad_bnbwd (stat_part is indexed by the
// 8x32 tiles of cova_conv3x3_num_tiles), with Winograd-transformed weights `u`.
// act/z/mean/invstd may all be NULL (plain conv, statistics = sum / sum of squares).
COVA_API int cova_conv3x3_wino(const float *in, const float *u, const float *addend,
                               const float *act, const float *z, const float *mean,
                               const float *invstd, float *out, float *stat_part, int B, int H, int W,
                               void *stream)
{
    COVA_REQUIRE(in && u && out && B > 0 && H > 0 && W > 0);
    COVA_REQUIRE((z == nullptr) || (act && mean && invstd && stat_part));
    const int tiles_x = cdiv(W, wn::TW), tiles_y = cdiv(H, wn::TH);
    const int ntiles = B * tiles_x * tiles_y;
    const dim3 grid(cova_internal_persistent_grid(ntiles)), block(wn::THREADS);
    const BnBwdEpiW bn{act, z, mean, invstd};
    if (stat_part)
        hipLaunchKernelGGL(conv3x3_c64_wino_kernel<true>, grid, block, 0, (hipStream_t)stream, in, u,
                           addend, out, stat_part, H, W, tiles_x, tiles_y, ntiles, bn,
                           cova_internal_ablate());
    else
        hipLaunchKernelGGL(conv3x3_c64_wino_kernel<false>, grid, block, 0, (hipStream_t)stream, in, u,
                           addend, out, stat_part, H, W, tiles_x, tiles_y, ntiles, bn,
                           cova_internal_ablate());
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}
