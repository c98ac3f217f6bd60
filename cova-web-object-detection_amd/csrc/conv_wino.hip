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

// ====================================================================================
// Winograd weight gradient:  dW = G^T [ sum_tiles (A dY A^T) .* (B^T d B) ] G
// For each of the 16 transform positions a GEMM  Q[pos][co][ci] = sum_tiles Wd[pos][tile][co] *
// V[pos][tile][ci]  (K = number of 2x2 tiles = pixels/4): 4*64*64 MACs per pixel instead of 9*64*64.
// Persistent 512-thread blocks, 8x32-pixel tiles (64 Winograd tiles).  Wave w owns transform row
// a = w>>1 and the two columns b = 2*(w&1), +1 for the whole 64x64 (co, ci) plane: 8 accumulators
// of v_mfma_f32_32x32x2_f32.  Both MFMA operands are formed on the fly from the NHWC tiles in LDS
// (lanes over channels: conflict-free ds_read_b32): per k-pair 8 reads of dY (shared by the two
// columns) + 12 reads of the input patch (3 columns x 2 rows x 2 channel blocks) + 32 FMAs for
// 8 MFMAs.  The tiles are refilled band by band (2 input rows + 2 dY rows per tile row, one
// barrier each) with the next tile's data fetched during the MFMAs.
// ====================================================================================
namespace {

namespace wgw {
constexpr int TH = 8, TW = 32, PH = 10, PW = 34;
constexpr int D_FLOATS = PH * PW * 64;       // 21,760 floats
constexpr int DY_FLOATS = TH * TW * 64;      // 16,384 floats
constexpr int THREADS = 512;
}  // namespace wgw

__global__ __launch_bounds__(wgw::THREADS) void conv3x3_wgrad_wino_kernel(
    const float *__restrict__ act, const float *__restrict__ dz, float *__restrict__ part, int H, int W,
    int tiles_x, int tiles_y, int ntiles)
{
    using namespace wgw;
    __shared__ __attribute__((aligned(16))) float lds[D_FLOATS + DY_FLOATS];
    float *s_d = lds;
    float *s_dy = lds + D_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh2 = lane >> 5;
    // per-wave (uniform) transform coefficients
    const int a = wave >> 1, pair = wave & 1;
    // B^T row a = sr1 * e[r1] + sr2 * e[r2]
    const int r1 = a == 0 ? 0 : 1, r2 = a == 3 ? 3 : 2;
    const float sr1 = a == 2 ? -1.f : 1.f, sr2 = (a == 0 || a == 3) ? -1.f : 1.f;
    // the two columns b0 = 2*pair, b1 = b0+1 read input columns c0, c0+1, c0+2
    const int c0 = pair;                         // pair 0: cols 0,1,2 ; pair 1: cols 1,2,3
    const float vc[2][3] = {{pair == 0 ? 1.f : -1.f, pair == 0 ? 0.f : 1.f, pair == 0 ? -1.f : 0.f},
                            {pair == 0 ? 0.f : 1.f, pair == 0 ? 1.f : 0.f, pair == 0 ? 1.f : -1.f}};
    // A (4x2) rows: (1,0) (1,1) (1,-1) (0,-1)
    const float cy0 = a == 3 ? 0.f : 1.f, cy1 = a == 0 ? 0.f : (a == 1 ? 1.f : -1.f);
    const float cx[2][2] = {{pair == 0 ? 1.f : 1.f, pair == 0 ? 0.f : -1.f},       // b0 = 0 | 2
                            {pair == 0 ? 1.f : 0.f, pair == 0 ? 1.f : -1.f}};      // b1 = 1 | 3

    int tile = blockIdx.x;
    if ((gridDim.x & 7) == 0) tile = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);

    f32x16 acc[2][2][2];        // [column e][co block][ci block]
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[e][i][j][r] = 0.f;

    // fetch helpers: float4 slot `idx` of input rows [row0, row0+nrows) / dy rows of tile (ty, tx, b)
    auto load_d = [&](const float *act_b, int ty, int tx, int row0, int idx) {
        const int px = idx >> 4, c4 = idx & 15;
        const int r = row0 + px / PW, c = px % PW;
        const int gy = ty * TH + r - 1, gx = tx * TW + c - 1;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gy >= 0 && gy < H && gx >= 0 && gx < W)
            v = *reinterpret_cast<const float4 *>(act_b + ((size_t)gy * W + gx) * 64 + c4 * 4);
        return v;
    };
    auto load_dy = [&](const float *dz_b, int ty, int tx, int row0, int idx) {
        const int px = idx >> 4, c4 = idx & 15;
        const int r = row0 + px / TW, c = px % TW;
        const int gy = ty * TH + r, gx = tx * TW + c;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gy < H && gx < W)
            v = *reinterpret_cast<const float4 *>(dz_b + ((size_t)gy * W + gx) * 64 + c4 * 4);
        return v;
    };

    if (tile < ntiles) {      // first tile: everything
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
        const float *act_b = act + (size_t)b * H * W * 64, *dz_b = dz + (size_t)b * H * W * 64;
#pragma unroll 1
        for (int idx = tid; idx < PH * PW * 16; idx += THREADS)
            *reinterpret_cast<float4 *>(s_d + idx * 4) = load_d(act_b, ty, tx, 0, idx);
#pragma unroll 1
        for (int idx = tid; idx < TH * TW * 16; idx += THREADS)
            *reinterpret_cast<float4 *>(s_dy + idx * 4) = load_dy(dz_b, ty, tx, 0, idx);
    }
    __syncthreads();

    for (; tile < ntiles; tile += gridDim.x) {
        const int next = tile + gridDim.x;
        const bool has_next = next < ntiles;
        const int ntx = next % tiles_x, nty = (next / tiles_x) % tiles_y;
        const int nb = next / (tiles_x * tiles_y);
        const float *nact = act + (size_t)nb * H * W * 64, *ndz = dz + (size_t)nb * H * W * 64;
#pragma unroll
        for (int tr = 0; tr < 4; ++tr) {
            // band tr of the NEXT tile: input rows 2tr, 2tr+1 (+ rows 8, 9 with the last band) and dY
            // rows 2tr, 2tr+1; fetched now, written after this tile row's barrier
            constexpr int kMaxD = 5, kDy = 2;
            const int nd_rows = tr == 3 ? 4 : 2;
            float4 rd[kMaxD], rdy[kDy];
            if (has_next) {
#pragma unroll
                for (int k = 0; k < kMaxD; ++k) {
                    const int idx = tid + k * THREADS;
                    if (idx < nd_rows * PW * 16) rd[k] = load_d(nact, nty, ntx, 2 * tr, idx);
                }
#pragma unroll
                for (int k = 0; k < kDy; ++k) rdy[k] = load_dy(ndz, nty, ntx, 2 * tr, tid + k * THREADS);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- 8 k-pairs: tiles (tr, 2t + kh2)
            const float *dyb = s_dy + ((2 * tr) * TW + 2 * kh2) * 64 + li;
            const float *db = s_d + ((2 * tr) * PW + 2 * kh2 + c0) * 64 + li;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                float wd[2][2], vv[2][2];       // [column e][channel block]
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    const float *p = dyb + (4 * t) * 64 + blk * 32;          // tile col 2t+kh2 -> px 4t+2kh2
                    const float y00 = p[0], y01 = p[64], y10 = p[TW * 64], y11 = p[TW * 64 + 64];
                    const float top0 = cx[0][0] * y00 + cx[0][1] * y01, bot0 = cx[0][0] * y10 + cx[0][1] * y11;
                    const float top1 = cx[1][0] * y00 + cx[1][1] * y01, bot1 = cx[1][0] * y10 + cx[1][1] * y11;
                    wd[0][blk] = cy0 * top0 + cy1 * bot0;
                    wd[1][blk] = cy0 * top1 + cy1 * bot1;
                    const float *q = db + (4 * t) * 64 + blk * 32;
                    float rc[3];
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        rc[j] = sr1 * q[(r1 * PW + j) * 64] + sr2 * q[(r2 * PW + j) * 64];
                    vv[0][blk] = vc[0][0] * rc[0] + vc[0][1] * rc[1] + vc[0][2] * rc[2];
                    vv[1][blk] = vc[1][0] * rc[0] + vc[1][1] * rc[1] + vc[1][2] * rc[2];
                }
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[e][i][j] = mfma32(wd[e][i], vv[e][j], acc[e][i][j]);
            }
            __syncthreads();          // everyone is done with this band
            if (has_next) {
#pragma unroll
                for (int k = 0; k < kMaxD; ++k) {
                    const int idx = tid + k * THREADS;
                    if (idx < nd_rows * PW * 16)
                        *reinterpret_cast<float4 *>(s_d + (2 * tr) * PW * 64 + idx * 4) = rd[k];
                }
#pragma unroll
                for (int k = 0; k < kDy; ++k)
                    *reinterpret_cast<float4 *>(s_dy + (2 * tr) * TW * 64 + (tid + k * THREADS) * 4) = rdy[k];
            }
        }
        __syncthreads();              // the refilled tile is complete before the next tile starts
    }
    // partial: part[block][pos = a*4 + b][co][ci]
    float *dst = part + (size_t)blockIdx.x * (16 * 4096);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int pos = a * 4 + 2 * pair + e;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    dst[pos * 4096 + (i * 32 + mfma32_row(r, lane)) * 64 + j * 32 + li] = acc[e][i][j][r];
    }
}

// sum the per-block partials (fp64) : q[16*4096]
__global__ __launch_bounds__(1024) void wgrad_wino_reduce_kernel(const float *__restrict__ part,
                                                                 int nparts, float *__restrict__ q)
{
    __shared__ double s_acc[16][64];
    const int tx = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + tx;
    double s = 0.0;
    for (int p = slice; p < nparts; p += 16) s += (double)part[(size_t)p * (16 * 4096) + idx];
    s_acc[slice][tx] = s;
    __syncthreads();
    if (slice == 0) {
        double t = 0.0;
        for (int j = 0; j < 16; ++j) t += s_acc[j][tx];
        q[idx] = (float)t;
    }
}

// dW[co][ci][r][t] = sum_{a,b} G[a][r] * Q[a*4+b][co][ci] * G[b][t]   (OIHW)
__global__ void wgrad_wino_final_kernel(const float *__restrict__ q, float *__restrict__ dw)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;     // over [co][ci]
    if (idx >= 4096) return;
    const float G[4][3] = {{1.f, 0.f, 0.f}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0.f, 0.f, 1.f}};
    float Q[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) Q[p] = q[p * 4096 + idx];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            float s = 0.f;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) s += G[a][r] * G[b][t] * Q[a * 4 + b];
            dw[idx * 9 + r * 3 + t] = s;
        }
}

}  // namespace

// act, dz NHWC [B,H,W,64]; dw OIHW [64,64,3,3]; ws >= (grid*16*4096 + 16*4096) floats, which
// cova_conv3x3_wgrad_workspace_floats covers
COVA_API int cova_conv3x3_wgrad_wino(const float *act, const float *dz, float *dw, float *ws, int B,
                                     int H, int W, void *stream)
{
    COVA_REQUIRE(act && dz && dw && ws && B > 0 && H > 0 && W > 0);
    const int tiles_x = cdiv(W, wgw::TW), tiles_y = cdiv(H, wgw::TH);
    const int ntiles = B * tiles_x * tiles_y;
    const int grid = cova_internal_persistent_grid(ntiles);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(conv3x3_wgrad_wino_kernel, dim3(grid), dim3(wgw::THREADS), 0, st, act, dz, ws, H,
                       W, tiles_x, tiles_y, ntiles);
    COVA_LAUNCH_CHECK();
    float *q = ws + (size_t)grid * (16 * 4096);
    hipLaunchKernelGGL(wgrad_wino_reduce_kernel, dim3(16 * 4096 / 64), dim3(1024), 0, st, ws, grid, q);
    COVA_LAUNCH_CHECK();
    hipLaunchKernelGGL(wgrad_wino_final_kernel, dim3(16), dim3(256), 0, st, q, dw);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}
