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
#include "../../cova-web-object-detection_amd/csrc/common.h"

namespace {

// 256 zero bytes in global memory: refill loads of halo pixels outside the image (and of idle
// threads) read from here, so the value needs no select afterwards (plain kernels)
__device__ __attribute__((aligned(256))) float g_wino_zero_page[64];

// ReLU mask + BatchNorm-backward sums in the epilogue.  The mask is act > 0, or, when the
// activation was never materialised (act == nullptr), fma(msc, z, msh) > 0 -- the same expression
// the affine-on-load prologue evaluates, so both sides take identical decisions.
struct BnBwdEpiW {
    const float *act, *z, *mean, *invstd, *msc, *msh;
    int epi_relu;          // BN == 3 (inference epilogue  out = f(msc*y + msh + addend)): f = ReLU
};

// Affine-on-load prologue: the conv input is  f(A[c]*in + B[c]*in2 + C[c])  (f = ReLU or identity),
// zero outside the image.  Covers (a) a BatchNorm+ReLU folded into the consuming conv
// (A = scale, B = 0, C = shift, relu) and (b) the BatchNorm-backward "apply" folded into the
// consumers of dz:  dz = scale*(dy - c1 - xhat*c2) = A*dy + B*z + C.
struct ProIn {
    const float *abc;      // [3][64] = A | B | C, or nullptr
    const float *in2;      // second tensor (same NHWC shape as the input) or nullptr (B ignored)
    int relu;
};

// Two geometries of the forward / data-gradient kernel:
//  GeoA: 8x32-pixel tiles, 512 threads, one block per CU.  Wave w = (tile row w&3, co half w>>2);
//        weights in LDS as [ci 4][co 64] rows of 16 positions padded to 20 floats.
//  GeoB: 8x16-pixel tiles, 256 threads, TWO blocks per CU (each wave alone on its SIMD within its
//        block, paired with a wave of the other block that is in a different phase: barrier waits,
//        LDS latencies and the epilogue's memory round trips of one block hide under the MFMAs of
//        the other).  Wave w = (tile-row pair w&1, co half w>>1); weight rows unpadded (the two
//        blocks must fit 2 x 80 KB) with the four 16-byte slots of a row XOR-swizzled by
//        (row >> 2) & 3, which makes the b128 operand reads of 16 consecutive rows conflict free.
struct GeoA {
    static constexpr int TH = 8, TW = 32, PH = TH + 2, PW = TW + 2;
    static constexpr int NPIX = PH * PW;                 // 340
    static constexpr int PIXS = 341;                     // odd channel-plane stride
    static constexpr int IN_FLOATS = 64 * PIXS;          // 21,824 floats = 87.3 KB
    static constexpr int UROW = 20;                      // one (ci, co) row = 16 positions + 4 pad floats
    static constexpr int UCH = 4 * 64 * UROW;            // 5,120 floats = 20 KB per chunk in LDS
    static constexpr int THREADS = 512, NWAVES = 8, BLOCKS_PER_CU = 1;
    static constexpr int NUL = 2;                        // float4 of a weight chunk per thread
    static constexpr int RED_FLOATS = NWAVES * 64;       // per-wave running statistics (32 channels x 2 kinds)
    static constexpr int LDS_FLOATS = IN_FLOATS + 2 * UCH + RED_FLOATS;   // 130.8 KB
    __device__ static int cbp(int wave) { return wave >> 2; }
    __device__ static int trow(int wave, int ti) { return wave & 3; }     // Winograd tile row / column of lane ti
    __device__ static int tcol(int ti) { return ti; }
    __device__ static constexpr int nrowgroups() { return 4; }             // waves sharing a co half
    __device__ static int wave_of(int half, int g) { return half * 4 + g; }
    // float4 index f (global chunk layout [k 4][co 64][pos 16]) -> LDS float offset inside a chunk
    __device__ static int u_lds_off(int f) { return ((f * 4) >> 4) * UROW + ((f * 4) & 15); }
    // LDS float offset of the operands of position group p4 in row `row`
    __device__ static int u_op_off(int row, int p4) { return row * UROW + p4 * 4; }
};
struct GeoB {
    static constexpr int TH = 8, TW = 16, PH = TH + 2, PW = TW + 2;
    static constexpr int NPIX = PH * PW;                 // 180
    static constexpr int PIXS = 181;
    static constexpr int IN_FLOATS = 64 * PIXS;          // 11,584 floats = 46.3 KB
    static constexpr int UROW = 16;
    static constexpr int UCH = 4 * 64 * UROW;            // 4,096 floats = 16 KB per chunk
    static constexpr int THREADS = 256, NWAVES = 4, BLOCKS_PER_CU = 2;
    static constexpr int NUL = 4;
    static constexpr int RED_FLOATS = NWAVES * 64;
    static constexpr int LDS_FLOATS = IN_FLOATS + 2 * UCH + RED_FLOATS;   // 79.9 KB
    __device__ static int cbp(int wave) { return wave >> 1; }
    __device__ static int trow(int wave, int ti) { return 2 * (wave & 1) + (ti >> 3); }
    __device__ static int tcol(int ti) { return ti & 7; }
    __device__ static constexpr int nrowgroups() { return 2; }
    __device__ static int wave_of(int half, int g) { return half * 2 + g; }
    __device__ static int u_lds_off(int f)
    {
        const int row = f >> 2, p4 = f & 3;
        return row * UROW + ((p4 ^ ((row >> 2) & 3)) << 2);
    }
    __device__ static int u_op_off(int row, int p4) { return row * UROW + ((p4 ^ ((row >> 2) & 3)) << 2); }
};
constexpr int UCH_G = 16 * 4 * 64;             // 4,096 floats per weight chunk in global memory
// L2 prefetch of the epilogue operands during the chunk loop was measured 4% slower and removed.

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c)
{
    // v_mfma_f32_16x16x4_f32: lane l supplies A[i = l&15][k = l>>4], B[k = l>>4][j = l&15];
    // D register r of lane l is D[row = (l>>4)*4 + r][col = l&15]
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Input transform of one 4x4 patch: V = B^T d B (32 adds), d read from LDS with immediate offsets.
template <int PW>
__device__ __forceinline__ void wino_input_transform(const float *__restrict__ a, float (&V)[16])
{
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

// BN: 0 = plain epilogue, 1 = ReLU mask from z (fma(msc, z, msh) > 0) + BatchNorm-backward sums,
//     2 = ReLU mask from the materialised activation + BatchNorm-backward sums,
//     3 = inference: out = f(msc*y + msh + addend), the BatchNorm (running statistics) + residual +
//         ReLU that follows the conv, in cova_bn_act_fwd's operation order (bit-identical to it)
template <class G, bool STATS, bool PRO, bool ADD, int BN>
__global__ __launch_bounds__(G::THREADS, G::BLOCKS_PER_CU) void conv3x3_c64_wino_kernel(
    const float *__restrict__ in, const float *__restrict__ ug, const float *__restrict__ addend,
    float *__restrict__ out, float *__restrict__ stat_part, int H, int W, int tiles_x, int tiles_y,
    int ntiles, const BnBwdEpiW bn, const ProIn pro, int abl_arg)
{
    const int abl = COVA_ABL(abl_arg);
    // abl (tools/conv_bench.py only, 0 in production): 1 no epilogue, 2 no refill stores,
    // 4 no refill loads, 8 no weight restaging, 16 no per-chunk barrier, 32 no input transform,
    // 64 no statistics reduction, 128 no epilogue operand loads, 256 no output stores,
    // 512 no L2 prefetch of the epilogue operands, 1024 no MFMA/VALU/LDS interleave hints
    constexpr int TH = G::TH, TW = G::TW, PW = G::PW, NPIX = G::NPIX, PIXS = G::PIXS;
    constexpr int IN_FLOATS = G::IN_FLOATS, UROW = G::UROW, UCH = G::UCH, THREADS = G::THREADS;
    constexpr int LDS_FLOATS = G::LDS_FLOATS, NUL = G::NUL;
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS + (PRO ? 192 : 0)];
    float *s_in = lds;
    float *s_u = lds + IN_FLOATS;
    float *s_red = lds + IN_FLOATS + 2 * UCH;
    float *s_pro = lds + LDS_FLOATS;               // A | B | C per channel
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // STATS: s_red[wave][0..31 sums | 32..63 second moments] are running totals over all tiles of this
    // block, updated by the same four lanes of each wave every tile (no barrier); the waves' rows are
    // added up once, after the tile loop (one partial row per block)
    if (STATS && tid < G::NWAVES * 64) s_red[tid] = 0.f;       // (ordered before its first use by the barriers below)
    const int cbp = G::cbp(wave);
    const int ti = lane & 15, kq = lane >> 4;
    const int trow = G::trow(wave, ti), tcol = G::tcol(ti);      // this lane's Winograd tile in the block tile
    const bool pro2 = PRO && pro.in2 != nullptr;
    if (PRO) {
        if (tid < 192) s_pro[tid] = pro.abc[tid];
        __syncthreads();
    }
    // staged value of an in-image element: f(A*v + B*w + C) on 4 consecutive channels c..c+3
    auto pro_apply = [&](float4 v, float4 w, int c) {
        const float4 A = *reinterpret_cast<const float4 *>(s_pro + c);
        const float4 Bc = *reinterpret_cast<const float4 *>(s_pro + 64 + c);
        const float4 Cc = *reinterpret_cast<const float4 *>(s_pro + 128 + c);
        float4 r;
        r.x = fmaf(A.x, v.x, pro2 ? fmaf(Bc.x, w.x, Cc.x) : Cc.x);
        r.y = fmaf(A.y, v.y, pro2 ? fmaf(Bc.y, w.y, Cc.y) : Cc.y);
        r.z = fmaf(A.z, v.z, pro2 ? fmaf(Bc.z, w.z, Cc.z) : Cc.z);
        r.w = fmaf(A.w, v.w, pro2 ? fmaf(Bc.w, w.w, Cc.w) : Cc.w);
        if (pro.relu) {
            r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f);
        }
        return r;
    };

    int tile = blockIdx.x;       // XCD-aware order (see conv.hip)
    if ((gridDim.x & 7) == 0) tile = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (tile >= ntiles) return;

    // transformed-weight chunk s: 1024 float4 in global memory, NUL per thread
    // (named scalars, not arrays: LLVM otherwise promotes the small arrays to LDS / scratch)
    const int uo0 = G::u_lds_off(tid), uo1 = G::u_lds_off(tid + THREADS);
    const int uo2 = NUL > 2 ? G::u_lds_off(tid + 2 * THREADS) : 0, uo3 = NUL > 2 ? G::u_lds_off(tid + 3 * THREADS) : 0;

    // ---- first tile: whole halo'd input tile, channel-major in LDS
    {
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
#pragma unroll 1
        for (int idx = tid; idx < NPIX * 16; idx += THREADS) {
            const int px = idx >> 4, c = (idx & 15) * 4;
            const int r = px / PW, cc = px - r * PW;
            const int gy = ty * TH + r - 1, gx = tx * TW + cc - 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
                const size_t o = ((size_t)b * H * W + (size_t)gy * W + gx) * 64 + c;
                v = *reinterpret_cast<const float4 *>(in + o);
                if (PRO) {
                    float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (pro2) w = *reinterpret_cast<const float4 *>(pro.in2 + o);
                    v = pro_apply(v, w, c);
                }
            }
            s_in[(c + 0) * PIXS + px] = v.x;
            s_in[(c + 1) * PIXS + px] = v.y;
            s_in[(c + 2) * PIXS + px] = v.z;
            s_in[(c + 3) * PIXS + px] = v.w;
        }
        const float4 *ug4 = reinterpret_cast<const float4 *>(ug);
        *reinterpret_cast<float4 *>(s_u + uo0) = ug4[tid];
        *reinterpret_cast<float4 *>(s_u + uo1) = ug4[tid + THREADS];
        if (NUL > 2) {
            *reinterpret_cast<float4 *>(s_u + uo2) = ug4[tid + 2 * THREADS];
            *reinterpret_cast<float4 *>(s_u + uo3) = ug4[tid + 3 * THREADS];
        }
    }
    __syncthreads();

    // Streaming refill: chunk s of a tile reads only channel planes 4s..4s+3, so once its barrier
    // has passed those planes are dead and are overwritten with the NEXT tile's data two chunks
    // later.  The tile buffer is therefore refilled during the MFMAs, 4 planes per chunk: no
    // tile-sized prefetch registers, no refill phase between tiles, and HBM sees a perfectly steady
    // stream.  Thread px (< 340) owns pixel px.  Two register rings (even / odd chunks) hold the
    // data in flight, so a load has two full chunks (~3 us) to land before it is needed.
    // Every global load in the loop is unconditional (clamped address, value discarded when not
    // needed): a load inside a branch would force s_waitcnt vmcnt(0) at the next counted wait.
    struct Ring {
        float4 v, w;        // planes 4s..4s+3 of the next tile's pixel (w: second input of PRO)
        int plane;          // first plane, or -1: nothing to write
        bool in;            // pixel lies inside the image
    };
    const int rpx = tid < NPIX ? tid : 0;
    const int rrow = rpx / PW, rcol = rpx - rrow * PW;
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    Ring ringA{zero4, zero4, -1, false}, ringB{zero4, zero4, -1, false};

    const float *a_lane = s_in + kq * PIXS + (2 * trow) * PW + 2 * tcol;  // this lane's patch origin
    // weight operands: rows (ci kq, co cbp*32 + ti) and +16; one LDS offset per position group
    // (the +16 rows of the second co block only add the immediate 16 * UROW: (row >> 2) & 3 is unchanged)
    const int urow = kq * 64 + cbp * 32 + ti;
    const int uq0 = G::u_op_off(urow, 0), uq1 = G::u_op_off(urow, 1), uq2 = G::u_op_off(urow, 2),
              uq3 = G::u_op_off(urow, 3);
    float V[16], V1[16];            // transformed input of the even / odd chunk (double buffer)
    wino_input_transform<PW>(a_lane, V);                                   // chunk 0 of the first tile
    for (; tile < ntiles; tile += gridDim.x) {
        const int tx = tile % tiles_x;
        const int ty = (tile / tiles_x) % tiles_y;
        const int b = tile / (tiles_x * tiles_y);
        const int y0 = ty * TH, x0 = tx * TW;
        const bool has_next = tile + gridDim.x < ntiles;
        const int next = has_next ? tile + gridDim.x : tile;       // always a valid tile
        const int ntx = next % tiles_x, nty = (next / tiles_x) % tiles_y;
        const int ngy = nty * TH + rrow - 1, ngx = ntx * TW + rcol - 1;
        const bool nload = has_next && tid < NPIX && ngy >= 0 && ngy < H && ngx >= 0 && ngx < W;
        const size_t noff = (size_t)(next / (tiles_x * tiles_y)) * H * W * 64 +
                            ((size_t)min(max(ngy, 0), H - 1) * W + min(max(ngx, 0), W - 1)) * 64;
        // plain kernels: out-of-image pixels load zeros from the zero page (no select later); the
        // prologue variants load a clamped address and select after the affine transform
        const float *nsrc = (PRO || nload) ? in + noff : g_wino_zero_page;
        const float *nsrc2 = pro2 ? pro.in2 + noff : nsrc;
        const size_t img = (size_t)b * H * W * 64;

        f32x4 acc[2][16];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int p = 0; p < 16; ++p) acc[c][p] = f32x4{0.f, 0.f, 0.f, 0.f};

        // np / np2: refill source of the chunk PAIR (advanced once per loop iteration); the chunk's own
        // 16 bytes are an immediate offset
        auto chunk = [&](const int s, const int e, const float *np, const float *np2, Ring &ring,
                         const float (&V)[16], float (&Vn)[16]) {
            // (1) refill: planes consumed two chunks ago <- data of the tile after theirs
            if (ring.plane >= 0 && tid < NPIX && !(abl & 2)) {
                float4 v = zero4;
                if (PRO) {
                    if (ring.in) v = pro_apply(ring.v, ring.w, ring.plane);
                } else {
                    v = ring.v;                       // already zero outside the image (zero page)
                }
                float *dst = s_in + (size_t)ring.plane * PIXS + rpx;
                dst[0] = v.x;
                dst[PIXS] = v.y;
                dst[2 * PIXS] = v.z;
                dst[3 * PIXS] = v.w;
            }
            // (2) next chunk of transformed weights (chunk 0 again for the next tile).  Issued FIRST:
            //     vmcnt retires in order, so the wait for these (end of this chunk) must not have
            //     the long-latency HBM loads below in front of it.
            const int ns = (s + 1) & 15;
            float4 un0 = zero4, un1 = zero4, un2 = zero4, un3 = zero4;
            if (!(abl & 8)) {
                // wave-uniform chunk base + this thread's fixed 32-bit byte offset (scalar-base addressing)
                const char *ugn = reinterpret_cast<const char *>(ug + (size_t)ns * UCH_G);
                const unsigned toff = (unsigned)tid * 16u;
                un0 = *reinterpret_cast<const float4 *>(ugn + toff);
                un1 = *reinterpret_cast<const float4 *>(ugn + toff + THREADS * 16u);
                if (NUL > 2) {
                    un2 = *reinterpret_cast<const float4 *>(ugn + toff + 2u * THREADS * 16u);
                    un3 = *reinterpret_cast<const float4 *>(ugn + toff + 3u * THREADS * 16u);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // (3) fetch this chunk's planes for the next tile; consumed two chunks from now
            if (!(abl & 4)) {
                ring.v = *reinterpret_cast<const float4 *>(np + 4 * e);
                if (pro2) ring.w = *reinterpret_cast<const float4 *>(np2 + 4 * e);
            }
            ring.plane = has_next ? 4 * s : -1;
            ring.in = nload;
            __builtin_amdgcn_sched_barrier(0);

            // (4) 32 MFMAs of this chunk
            // the 16 positions of a (ci, co) pair are contiguous: 4 ds_read_b128 per channel block
            const float *bp = s_u + (s & 1) * UCH;            // chunk s sits in ring slot s & 1
#pragma unroll
            for (int p4 = 0; p4 < 4; ++p4) {
                const int uq = p4 == 0 ? uq0 : p4 == 1 ? uq1 : p4 == 2 ? uq2 : uq3;
                const float4 u0 = *reinterpret_cast<const float4 *>(bp + uq);
                const float4 u1 = *reinterpret_cast<const float4 *>(bp + uq + 16 * UROW);
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
            //     planes were refilled at chunk 2 of this tile) into the other V buffer.  The asm
            //     uses pin it inside this chunk (LLVM otherwise sinks it below the barrier, next to
            //     its first use, where both waves of a SIMD run it with the matrix pipe idle) and the
            //     group barriers weave its 8 LDS reads and 32 adds between the MFMAs.
            if (!(abl & 32)) {
                wino_input_transform<PW>(a_lane + (4 * ns) * PIXS, Vn);
#pragma unroll
                for (int k = 0; k < 16; ++k) asm volatile("" ::"v"(Vn[k]));
            }
            if (!(abl & 1024)) {
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);        // DS read: first operands
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);    // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    // DS read
                    __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);    // VALU
                }
            }
            // (6) publish the next weight chunk
            if (!(abl & 8)) {
                float *ud = s_u + ((s & 1) ^ 1) * UCH;
                *reinterpret_cast<float4 *>(ud + uo0) = un0;
                *reinterpret_cast<float4 *>(ud + uo1) = un1;
                if (NUL > 2) {
                    *reinterpret_cast<float4 *>(ud + uo2) = un2;
                    *reinterpret_cast<float4 *>(ud + uo3) = un3;
                }
            }
            __builtin_amdgcn_sched_barrier(0);      // nothing of this chunk sinks below its barrier
            if (!(abl & 16)) __syncthreads();
        };
#pragma unroll 1
        for (int s2 = 0; s2 < 8; ++s2) {
            const float *np = nsrc + 8 * s2, *np2 = nsrc2 + 8 * s2;
            chunk(2 * s2, 0, np, np2, ringA, V, V1);
            chunk(2 * s2 + 1, 1, np, np2, ringB, V1, V);
        }

        // ---- output transform Y = A^T M A in registers, straight to HBM.
        // D layout (rows = co, cols = tiles): this lane holds, for c2 = 0..1 and q = 0..3, channel
        // cbp*32 + c2*16 + kq*4 + q of Winograd tile (trow, tcol) -- four consecutive channels per
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
        } else {
            // (a) output transform of both channel groups: the 128 accumulator registers die here
            float y[2][4][4];                      // [c2][pixel = yy*2 + xx][q]
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
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
                    y[c2][0][q] = s0[0] + s0[1] + s0[2];
                    y[c2][1][q] = s0[1] - s0[2] - s0[3];
                    y[c2][2][q] = s1[0] + s1[1] + s1[2];
                    y[c2][3][q] = s1[1] - s1[2] - s1[3];
                    ssum[c2][q] = 0.f;
                    ssq[c2][q] = 0.f;
                }
            __builtin_amdgcn_sched_barrier(0);
            // (b) every epilogue operand (2 channel groups x 4 pixels x up to 3 tensors, float4 each)
            //     is requested before the first one is used -- one memory round trip per tile --
            //     and nothing is conditional (out-of-image pixels read a clamped address and are
            //     discarded), so the waits are counted instead of s_waitcnt vmcnt(0).
            int co0 = cbp * 32 + kq * 4;
            asm volatile("" : "+v"(co0));      // keeps the per-channel constant loads out of the main loop
            // wave-uniform image base + 32-bit in-image element offsets (scalar-base addressing)
            const float *add_b = addend + img, *z_b = bn.z + img, *act_b = bn.act + img;
            float *out_b = out + img;
            bool ok[4];
            unsigned off[4], offc[4];
#pragma unroll
            for (int pq = 0; pq < 4; ++pq) {
                const int oy = y0 + 2 * trow + (pq >> 1), ox = x0 + 2 * tcol + (pq & 1);
                ok[pq] = oy < H && ox < W;
                off[pq] = (unsigned)((oy * W + ox) * 64 + co0);
                offc[pq] = (unsigned)((min(oy, H - 1) * W + min(ox, W - 1)) * 64 + co0);
            }
            struct Ops {
                float4 ad[4], z4[4], a4[4], mu4, is4, sc4, sh4;
            };
            auto load_ops = [&](const int c2, Ops &o) {
                o.mu4 = o.is4 = o.sc4 = o.sh4 = zero4;
                if (BN == 1 || BN == 2) {
                    o.mu4 = *reinterpret_cast<const float4 *>(bn.mean + co0 + c2 * 16);
                    o.is4 = *reinterpret_cast<const float4 *>(bn.invstd + co0 + c2 * 16);
                }
                if (BN == 1 || BN == 3) {
                    o.sc4 = *reinterpret_cast<const float4 *>(bn.msc + co0 + c2 * 16);
                    o.sh4 = *reinterpret_cast<const float4 *>(bn.msh + co0 + c2 * 16);
                }
#pragma unroll
                for (int pq = 0; pq < 4; ++pq) {
                    o.ad[pq] = o.a4[pq] = o.z4[pq] = zero4;
                    const size_t e = (size_t)(offc[pq] + c2 * 16);
                    if (ADD && !(abl & 128)) o.ad[pq] = *reinterpret_cast<const float4 *>(add_b + e);
                    if ((BN == 1 || BN == 2) && !(abl & 128)) o.z4[pq] = *reinterpret_cast<const float4 *>(z_b + e);
                    if (BN == 2 && !(abl & 128)) o.a4[pq] = *reinterpret_cast<const float4 *>(act_b + e);
                }
            };
            // (c) addend, ReLU mask, BatchNorm-backward sums (or plain statistics); result in place
            auto finish = [&](const int c2, const Ops &o) {
                const float mu[4] = {o.mu4.x, o.mu4.y, o.mu4.z, o.mu4.w};
                const float is[4] = {o.is4.x, o.is4.y, o.is4.z, o.is4.w};
                const float msc[4] = {o.sc4.x, o.sc4.y, o.sc4.z, o.sc4.w};
                const float msh[4] = {o.sh4.x, o.sh4.y, o.sh4.z, o.sh4.w};
#pragma unroll
                for (int pq = 0; pq < 4; ++pq) {
                    const float adv[4] = {o.ad[pq].x, o.ad[pq].y, o.ad[pq].z, o.ad[pq].w};
                    const float zv[4] = {o.z4[pq].x, o.z4[pq].y, o.z4[pq].z, o.z4[pq].w};
                    const float aa[4] = {o.a4[pq].x, o.a4[pq].y, o.a4[pq].z, o.a4[pq].w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float v = ADD ? y[c2][pq][q] + adv[q] : y[c2][pq][q];     // (y + 0.f is not folded)
                        if (BN == 3) {
                            v = fmaf(msc[q], y[c2][pq][q], msh[q]);
                            if (ADD) v += adv[q];
                            if (bn.epi_relu) v = v > 0.f ? v : 0.f;
                        } else if (BN != 0) {
                            const float av = BN == 2 ? aa[q] : fmaf(msc[q], zv[q], msh[q]);
                            if (!(av > 0.f) || !ok[pq]) v = 0.f;
                            ssum[c2][q] += v;
                            ssq[c2][q] += v * ((zv[q] - mu[q]) * is[q]);
                        } else if (STATS) {
                            if (!ok[pq]) v = 0.f;
                            ssum[c2][q] += v;
                            ssq[c2][q] += v * v;
                        }
                        y[c2][pq][q] = v;
                    }
                }
            };
            constexpr int NT = (ADD ? 1 : 0) + (BN == 1 || BN == 2 ? 1 : 0) + (BN == 2 ? 1 : 0);
            if (NT <= 1) {              // everything in flight at once: one round trip
                Ops o0, o1;
                load_ops(0, o0);
                load_ops(1, o1);
                __builtin_amdgcn_sched_barrier(0);
                finish(0, o0);
                finish(1, o1);
            } else {                    // one channel group at a time (registers): two round trips
                Ops o;
                load_ops(0, o);
                __builtin_amdgcn_sched_barrier(0);
                finish(0, o);
                __builtin_amdgcn_sched_barrier(0);
                load_ops(1, o);
                __builtin_amdgcn_sched_barrier(0);
                finish(1, o);
            }
            // (d) the eight stores last, so that no load ever waits behind a (conditional) store
            __builtin_amdgcn_sched_barrier(0);
            if (!(abl & 256)) {
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                    for (int pq = 0; pq < 4; ++pq)
                        if (ok[pq])
                            *reinterpret_cast<float4 *>(out_b + (size_t)(off[pq] + c2 * 16)) =
                                make_float4(y[c2][pq][0], y[c2][pq][1], y[c2][pq][2], y[c2][pq][3]);
            }
        }
        if (STATS && !(abl & 64)) {
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    ssum[c2][q] = row16_sum(ssum[c2][q]);        // over the 16 tiles of this lane row
                    ssq[c2][q] = row16_sum(ssq[c2][q]);
                }
            if (ti == 0) {         // 32 channels of this wave: two float4 read-modify-writes per kind
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                    float4 *ps = reinterpret_cast<float4 *>(s_red + wave * 64 + c2 * 16 + kq * 4);
                    float4 *pq = reinterpret_cast<float4 *>(s_red + wave * 64 + 32 + c2 * 16 + kq * 4);
                    float4 a = *ps, b = *pq;
                    a.x += ssum[c2][0]; a.y += ssum[c2][1]; a.z += ssum[c2][2]; a.w += ssum[c2][3];
                    b.x += ssq[c2][0]; b.y += ssq[c2][1]; b.z += ssq[c2][2]; b.w += ssq[c2][3];
                    *ps = a;
                    *pq = b;
                }
            }
        }
        // (no barrier here: the next tile's first LDS writes that could clash -- the refill of planes
        //  read two chunks ago, the weight ring -- are ordered by the chunk barriers, as inside a tile)
    }
    if (STATS) {
        __syncthreads();
        if (tid < 128) {
            const int which = tid >> 6, ch = tid & 63;        // 0 = sum, 1 = second moment
            const int half = ch >> 5, idx = ch & 31;
            float tsum = 0.f;
#pragma unroll
            for (int g = 0; g < G::nrowgroups(); ++g) tsum += s_red[G::wave_of(half, g) * 64 + which * 32 + idx];
            stat_part[(size_t)blockIdx.x * 128 + tid] = tsum;
        }
    }
    // (one row [sum 64 | second moment 64] per block: cova_conv3x3_wino_num_partials rows, no fold pass)
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

// (this library's own copies of the product library's launch helpers: persistent grid = one block per CU and blocks_per_cu,
// capped for the tests; ablation mask of -DCOVA_ABLATE builds)
static int g_f2_grid_cap = 0, g_f2_ablate = 0;
static int cova_internal_persistent_grid2(int ntiles, int blocks_per_cu)
{
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
    if (g_f2_grid_cap > 0 && g > g_f2_grid_cap) g = g_f2_grid_cap;
    return g;
}
static int cova_internal_persistent_grid(int ntiles) { return cova_internal_persistent_grid2(ntiles, 1); }
static int cova_internal_ablate() { return g_f2_ablate; }

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
struct WinoArgs {
    const float *in, *u, *addend;
    float *out, *stat_part;
    int H, W, tiles_x, tiles_y, ntiles;
    BnBwdEpiW bn;
    ProIn pro;
    int abl;
    dim3 grid;
    hipStream_t st;
};

template <class G, bool STATS, bool PRO, bool ADD, int BN>
static void launch_wino_variant(const WinoArgs &a)
{
    hipLaunchKernelGGL((conv3x3_c64_wino_kernel<G, STATS, PRO, ADD, BN>), a.grid, dim3(G::THREADS), 0, a.st,
                       a.in, a.u, a.addend, a.out, a.stat_part, a.H, a.W, a.tiles_x, a.tiles_y, a.ntiles,
                       a.bn, a.pro, a.abl);
}

template <class G, bool STATS, int BN>
static void launch_wino_epi(const WinoArgs &a)
{
    const bool p = a.pro.abc != nullptr, ad = a.addend != nullptr;
    if (p && ad) launch_wino_variant<G, STATS, true, true, BN>(a);
    else if (p) launch_wino_variant<G, STATS, true, false, BN>(a);
    else if (ad) launch_wino_variant<G, STATS, false, true, BN>(a);
    else launch_wino_variant<G, STATS, false, false, BN>(a);
}

// cova_set_option(6, 1 | 2): GeoA (8x32, 1 block/CU; default) | GeoB (8x16, 2 blocks/CU).  Measured:
// B is 3-6 % faster on the epilogue-heavy variants and equal on the plain ones -- the kernel is bound
// by instruction issue (MFMA + VALU + LDS share the SIMD's issue slots), not by latencies a second
// block could hide -- and 0.5-1 % slower over the whole step (twice the statistics partials, more halo).
static int g_wino_geometry = 1;
COVA_API int cova_wino_f2x2_set_option(int key, int v)
{
    if (key == 2) { g_f2_grid_cap = v; return COVA_OK; }
    if (key == 5) { g_f2_ablate = v; return COVA_OK; }
    if (key != 6 || (v != 1 && v != 2)) return COVA_ERR_BAD_ARG;
    g_wino_geometry = v;
    return COVA_OK;
}

COVA_API int cova_conv3x3_wgrad_workspace_floats(int B, int H, int W)
{
    const int ntiles = B * cdiv(W, 32) * cdiv(H, 4);
    return cova_internal_persistent_grid(ntiles) * 2 * 9 * 4096 + 16 * 4096;   // + Q buffer of the Winograd form
}

template <class G>
static int launch_wino_geo(const float *in, const float *u, const float *addend, const BnBwdEpiW bn,
                           const ProIn pro, float *out, float *stat_part, int B, int H, int W, void *stream)
{
    WinoArgs a;
    a.in = in; a.u = u; a.addend = addend; a.out = out; a.stat_part = stat_part;
    a.H = H; a.W = W;
    a.tiles_x = cdiv(W, G::TW); a.tiles_y = cdiv(H, G::TH);
    a.ntiles = B * a.tiles_x * a.tiles_y;
    a.bn = bn; a.pro = pro;
    a.abl = cova_internal_ablate();
    a.grid = dim3(cova_internal_persistent_grid2(a.ntiles, G::BLOCKS_PER_CU));
    a.st = (hipStream_t)stream;
    const int mode = bn.z == nullptr ? (bn.msc != nullptr ? 3 : 0)
                                     : (bn.act == nullptr ? 1 : 2);       // BN epilogues 1, 2 need stat_part
    if (mode == 3) launch_wino_epi<G, false, 3>(a);
    else if (mode == 1) launch_wino_epi<G, true, 1>(a);
    else if (mode == 2) launch_wino_epi<G, true, 2>(a);
    else if (stat_part) launch_wino_epi<G, true, 0>(a);
    else launch_wino_epi<G, false, 0>(a);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

static int launch_wino(const float *in, const float *u, const float *addend, const BnBwdEpiW bn,
                       const ProIn pro, float *out, float *stat_part, int B, int H, int W, void *stream)
{
    // 32-bit in-image offsets in the epilogue
    COVA_REQUIRE((long long)H * W * 64 < (1ll << 31));
    if (g_wino_geometry == 1) return launch_wino_geo<GeoA>(in, u, addend, bn, pro, out, stat_part, B, H, W, stream);
    return launch_wino_geo<GeoB>(in, u, addend, bn, pro, out, stat_part, B, H, W, stream);
}

// rows of the statistics partials written by cova_conv3x3_wino(_pro): one per (persistent) block of
// the active geometry
COVA_API int cova_conv3x3_wino_num_partials(int B, int H, int W)
{
    if (g_wino_geometry == 1)
        return cova_internal_persistent_grid2(B * cdiv(W, GeoA::TW) * cdiv(H, GeoA::TH), GeoA::BLOCKS_PER_CU);
    return cova_internal_persistent_grid2(B * cdiv(W, GeoB::TW) * cdiv(H, GeoB::TH), GeoB::BLOCKS_PER_CU);
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
    return launch_wino(in, u, addend, BnBwdEpiW{act, z, mean, invstd, nullptr, nullptr, 0},
                       ProIn{nullptr, nullptr, 0}, out, stat_part, B, H, W, stream);
}

// Inference form: out = f(scale[c]*conv(in) + shift[c] + addend), f = ReLU if relu -- the
// BatchNorm (running statistics), residual add and ReLU that follow the conv in a BasicBlock
// (torchvision resnet.py BasicBlock.forward; models.py:49-51), in the epilogue instead of a
// cova_bn_act_fwd pass; same operation order, bit-identical result.
COVA_API int cova_conv3x3_wino_bnact(const float *in, const float *u, const float *addend,
                                     const float *scale, const float *shift, int relu, float *out,
                                     int B, int H, int W, void *stream)
{
    COVA_REQUIRE(in && u && scale && shift && out && B > 0 && H > 0 && W > 0);
    return launch_wino(in, u, addend, BnBwdEpiW{nullptr, nullptr, nullptr, nullptr, scale, shift, relu},
                       ProIn{nullptr, nullptr, 0}, out, nullptr, B, H, W, stream);
}

// cova_conv3x3_wino whose input is  f(A[c]*in + B[c]*in2 + C[c])  applied on load (f = ReLU if
// pro_relu, identity otherwise; zero padding stays zero).  pro_abc = [3][64] floats (A | B | C);
// in2 may be NULL (then B is ignored).  Folds a BatchNorm+ReLU, or the BatchNorm-backward apply
// dz = A*dy + B*z + C, into the convolution that consumes it.  In the epilogue the ReLU mask may
// come from z itself (act == NULL): fma(mask_scale, z, mask_shift) > 0.
COVA_API int cova_conv3x3_wino_pro(const float *in, const float *in2, const float *pro_abc,
                                   int pro_relu, const float *u, const float *addend,
                                   const float *act, const float *mask_scale,
                                   const float *mask_shift, const float *z, const float *mean,
                                   const float *invstd, float *out, float *stat_part, int B, int H,
                                   int W, void *stream)
{
    COVA_REQUIRE(in && u && out && B > 0 && H > 0 && W > 0);
    COVA_REQUIRE((z == nullptr) || ((act || (mask_scale && mask_shift)) && mean && invstd && stat_part));
    return launch_wino(in, u, addend,
                       BnBwdEpiW{act, z, mean, invstd, z ? mask_scale : nullptr, z ? mask_shift : nullptr, 0},
                       ProIn{pro_abc, pro_abc ? in2 : nullptr, pro_relu}, out, stat_part, B, H, W, stream);
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
constexpr bool WG_PIPE = true;
constexpr int TH = 8, TW = 32, PH = 10, PW = 34;
constexpr int D_FLOATS = PH * PW * 64;       // 21,760 floats
constexpr int DY_FLOATS = TH * TW * 64;      // 16,384 floats
constexpr int THREADS = 512;
}  // namespace wgw

// PROA: the activation operand is f(A*act + C) on load (BatchNorm+ReLU never materialised);
// PROD: the gradient operand is A*dz + B*dz2 + C on load (BatchNorm-backward apply folded in).
template <bool PROA, bool PROD>
__global__ __launch_bounds__(wgw::THREADS) void conv3x3_wgrad_wino_kernel(
    const float *__restrict__ act, const float *__restrict__ dz, float *__restrict__ part, int H, int W,
    int tiles_x, int tiles_y, int ntiles, const ProIn proa, const ProIn prod)
{
    using namespace wgw;
    __shared__ __attribute__((aligned(16))) float lds[D_FLOATS + DY_FLOATS + 384];
    float *s_d = lds;
    float *s_dy = lds + D_FLOATS;
    float *s_pa = lds + D_FLOATS + DY_FLOATS;      // A | B | C of the activation prologue
    float *s_pd = s_pa + 192;                      // A | B | C of the gradient prologue
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (PROA || PROD) {
        if (PROA && tid < 192) s_pa[tid] = proa.abc[tid];
        if (PROD && tid >= 192 && tid < 384) s_pd[tid - 192] = prod.abc[tid - 192];
        __syncthreads();
    }
    auto affine_regs = [&](const float4 A, const float4 Bc, const float4 Cc, float4 v, float4 w, int relu,
                           bool useb) {
        float4 r;      // same expression as the forward prologue (identical ReLU decisions)
        r.x = fmaf(A.x, v.x, useb ? fmaf(Bc.x, w.x, Cc.x) : Cc.x);
        r.y = fmaf(A.y, v.y, useb ? fmaf(Bc.y, w.y, Cc.y) : Cc.y);
        r.z = fmaf(A.z, v.z, useb ? fmaf(Bc.z, w.z, Cc.z) : Cc.z);
        r.w = fmaf(A.w, v.w, useb ? fmaf(Bc.w, w.w, Cc.w) : Cc.w);
        if (relu) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
        return r;
    };
    auto affine = [&](const float *tab, float4 v, float4 w, int c, int relu, bool useb) {
        const float4 A = *reinterpret_cast<const float4 *>(tab + c);
        const float4 Bc = *reinterpret_cast<const float4 *>(tab + 64 + c);
        const float4 Cc = *reinterpret_cast<const float4 *>(tab + 128 + c);
        float4 r;      // same expression as the forward prologue (identical ReLU decisions)
        r.x = fmaf(A.x, v.x, useb ? fmaf(Bc.x, w.x, Cc.x) : Cc.x);
        r.y = fmaf(A.y, v.y, useb ? fmaf(Bc.y, w.y, Cc.y) : Cc.y);
        r.z = fmaf(A.z, v.z, useb ? fmaf(Bc.z, w.z, Cc.z) : Cc.z);
        r.w = fmaf(A.w, v.w, useb ? fmaf(Bc.w, w.w, Cc.w) : Cc.w);
        if (relu) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
        return r;
    };
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int li = lane & 31, kh2 = lane >> 5;
    // per-wave (uniform) transform coefficients
    const int a = wave >> 1, pair = wave & 1;
    // Operand formation with as few vector-ALU instructions as possible (on gfx950 the f32 MFMA and
    // the VALU share the issue slot: measured, their times add up).  Everything is evaluated on
    // f32x2 = (channel block 0, channel block 1) pairs (v_pk_fma_f32 / v_pk_add_f32) and the +-1 / 0
    // transform coefficients are wave-uniform scalars; row / column choices live in the LDS
    // addresses.  Three sign flips are left to wgrad_wino_final_kernel (wino_wgrad_sign):
    //   A dY A^T row a:  R_j = y[i0][j] + sa * y[1][j]      i0 = (a == 3), sa = 0, +1, -1, 0   (a = 3 flipped)
    //   columns b0, b1:  wd0 = R_0 + al * R_1,  wd1 = be * R_0 + R_1                           (b = 3 flipped)
    //   B^T d B  row a:  u = q[r1] + sr * q[r2]             (a = 2 flipped: d1 - d2 instead of d2 - d1)
    //   columns b0, b1:  vv0 = u[X] - u[Z],  vv1 = u[P] + sc * u[Q]
    const int i0 = a == 3 ? 1 : 0;
    const float sa = a == 1 ? 1.f : (a == 2 ? -1.f : 0.f);
    const float al = pair == 0 ? 0.f : -1.f, be = pair == 0 ? 1.f : 0.f;
    const int r1 = a == 0 ? 0 : 1, r2 = a == 3 ? 3 : 2;
    const float sr = a == 1 ? 1.f : -1.f;
    const int cX = pair == 0 ? 0 : 2, cZ = pair == 0 ? 2 : 1, cP = 1, cQ = pair == 0 ? 2 : 3;
    const float sc = pair == 0 ? 1.f : -1.f;
    // the coefficients as VGPR pairs, so that the compiler keeps the packed forms
    f32x2 sa2 = {sa, sa}, al2 = {al, al}, be2 = {be, be}, sr2 = {sr, sr}, sc2 = {sc, sc};
    asm volatile("" : "+v"(sa2), "+v"(al2), "+v"(be2), "+v"(sr2), "+v"(sc2));

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

    // fetch helpers: float4 slot `idx` of input rows [row0, ...) / dy rows of tile (ty, tx) of batch
    // image b_off (element offset); `ok` = inside the image, v2 = second tensor of the prologue
    auto load_d = [&](size_t b_off, int ty, int tx, int row0, int idx, float4 &v, bool &ok) {
        const int px = idx >> 4, c4 = idx & 15;
        const int r = row0 + px / PW, c = px % PW;
        const int gy = ty * TH + r - 1, gx = tx * TW + c - 1;
        v = zero4;
        ok = gy >= 0 && gy < H && gx >= 0 && gx < W;
        if (ok) v = *reinterpret_cast<const float4 *>(act + b_off + ((size_t)gy * W + gx) * 64 + c4 * 4);
    };
    auto load_dy = [&](size_t b_off, int ty, int tx, int row0, int idx, float4 &v, float4 &v2, bool &ok) {
        const int px = idx >> 4, c4 = idx & 15;
        const int r = row0 + px / TW, c = px % TW;
        const int gy = ty * TH + r, gx = tx * TW + c;
        v = zero4;
        v2 = zero4;
        ok = gy < H && gx < W;
        if (ok) {
            const size_t o = b_off + ((size_t)gy * W + gx) * 64 + c4 * 4;
            v = *reinterpret_cast<const float4 *>(dz + o);
            if (PROD && prod.in2 != nullptr) v2 = *reinterpret_cast<const float4 *>(prod.in2 + o);
        }
    };
    auto fin_d = [&](float4 v, bool ok, int idx) {      // value stored to LDS for an input slot
        return (PROA && ok) ? affine(s_pa, v, zero4, (idx & 15) * 4, proa.relu, false) : v;
    };
    auto fin_dy = [&](float4 v, float4 v2, bool ok, int idx) {
        return (PROD && ok) ? affine(s_pd, v, v2, (idx & 15) * 4, prod.relu, prod.in2 != nullptr) : v;
    };

    // LDS pixel layout: channel c = blk*32 + l sits at float 2*l + blk, so that a lane's two channel
    // blocks are ONE ds_read_b64 with a 16-bit immediate offset (no per-read address arithmetic).
    // A staged float4 (channels 4k..4k+3 of one block) becomes two ds_write2_b32.
    auto stage4 = [&](float *tile_base, int idx, float4 v) {
        const int px = idx >> 4, c = (idx & 15) * 4;
        float *dst = tile_base + px * 64 + ((c & 31) << 1) + (c >> 5);
        dst[0] = v.x;
        dst[2] = v.y;
        dst[4] = v.z;
        dst[6] = v.w;
    };

    if (tile < ntiles) {      // first tile: everything
        const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, b = tile / (tiles_x * tiles_y);
        const size_t b_off = (size_t)b * H * W * 64;
#pragma unroll 1
        for (int idx = tid; idx < PH * PW * 16; idx += THREADS) {
            float4 v;
            bool ok;
            load_d(b_off, ty, tx, 0, idx, v, ok);
            stage4(s_d, idx, fin_d(v, ok, idx));
        }
#pragma unroll 1
        for (int idx = tid; idx < TH * TW * 16; idx += THREADS) {
            float4 v, v2;
            bool ok;
            load_dy(b_off, ty, tx, 0, idx, v, v2, ok);
            stage4(s_dy, idx, fin_dy(v, v2, ok, idx));
        }
    }
    __syncthreads();

    for (; tile < ntiles; tile += gridDim.x) {
        const int next = tile + gridDim.x;
        const bool has_next = next < ntiles;
        const int ntx = next % tiles_x, nty = (next / tiles_x) % tiles_y;
        const size_t nb_off = (size_t)(next / (tiles_x * tiles_y)) * H * W * 64;
#pragma unroll
        for (int tr = 0; tr < 4; ++tr) {
            // band tr of the NEXT tile: input rows 2tr, 2tr+1 (+ rows 8, 9 with the last band) and dY
            // rows 2tr, 2tr+1; fetched now, written after this tile row's barrier
            constexpr int kMaxD = 5, kDy = 2;
            const int nd_rows = tr == 3 ? 4 : 2;
            float4 rd[kMaxD], rdy[kDy], rdy2[kDy];
            bool okd[kMaxD], okdy[kDy];
            if (has_next) {
#pragma unroll
                for (int k = 0; k < kMaxD; ++k) {
                    const int idx = tid + k * THREADS;
                    if (idx < nd_rows * PW * 16) load_d(nb_off, nty, ntx, 2 * tr, idx, rd[k], okd[k]);
                }
#pragma unroll
                for (int k = 0; k < kDy; ++k)
                    load_dy(nb_off, nty, ntx, 2 * tr, tid + k * THREADS, rdy[k], rdy2[k], okdy[k]);
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- 8 k-pairs: tiles (tr, 2t + kh2)
            const float *dyb = s_dy + ((2 * tr) * TW + 2 * kh2) * 64 + 2 * li;
            const float *db = s_d + ((2 * tr) * PW + 2 * kh2) * 64 + 2 * li;
            // Software pipeline over the 8 k-pairs of the band: the 20 LDS reads of pair t+1 are
            // issued before the 8 MFMAs of pair t and its ~40 FMAs are woven between them, so the
            // matrix pipe is not left idle while a wave forms its operands.
            struct Raw {
                f32x2 y[4], q[8];               // dY: (i0,0) (i0,1) (1,0) (1,1); input: rows r1, r2 x cols X Z P Q
            };
            auto pair2 = [&](const float *p) { return *reinterpret_cast<const f32x2 *>(p); };   // both channel blocks
            auto read_ops = [&](const int t, Raw &r) {
                const float *p = dyb + (4 * t) * 64;                          // tile col 2t+kh2 -> px 4t+2kh2
                r.y[0] = pair2(p + i0 * TW * 64);
                r.y[1] = pair2(p + i0 * TW * 64 + 64);
                r.y[2] = pair2(p + TW * 64);
                r.y[3] = pair2(p + TW * 64 + 64);
                const float *q = db + (4 * t) * 64;
                r.q[0] = pair2(q + (r1 * PW + cX) * 64);
                r.q[1] = pair2(q + (r1 * PW + cZ) * 64);
                r.q[2] = pair2(q + (r1 * PW + cP) * 64);
                r.q[3] = pair2(q + (r1 * PW + cQ) * 64);
                r.q[4] = pair2(q + (r2 * PW + cX) * 64);
                r.q[5] = pair2(q + (r2 * PW + cZ) * 64);
                r.q[6] = pair2(q + (r2 * PW + cP) * 64);
                r.q[7] = pair2(q + (r2 * PW + cQ) * 64);
            };
            auto form_ops = [&](const Raw &r, float (&wd)[2][2], float (&vv)[2][2]) {   // [column e][channel block]
                // `keep` = empty asm on the 64-bit pair: stops LLVM from splitting the packed op into
                // two scalar ones when its halves are consumed separately (by the MFMAs)
                auto keep = [](f32x2 x) { asm("" : "+v"(x)); return x; };
                const f32x2 R0 = keep(sa2 * r.y[2] + r.y[0]), R1 = keep(sa2 * r.y[3] + r.y[1]);
                const f32x2 w0 = keep(al2 * R1 + R0), w1 = keep(be2 * R0 + R1);
                const f32x2 u10 = keep(r.q[0] - r.q[1]), u20 = keep(r.q[4] - r.q[5]);
                const f32x2 u11 = keep(sc2 * r.q[3] + r.q[2]), u21 = keep(sc2 * r.q[7] + r.q[6]);
                const f32x2 v0 = keep(sr2 * u20 + u10), v1 = keep(sr2 * u21 + u11);
                wd[0][0] = w0.x; wd[0][1] = w0.y; wd[1][0] = w1.x; wd[1][1] = w1.y;
                vv[0][0] = v0.x; vv[0][1] = v0.y; vv[1][0] = v1.x; vv[1][1] = v1.y;
            };
            float wd[2][2][2], vv[2][2][2];     // double buffered operands
            {
                Raw r;
                read_ops(0, r);
                form_ops(r, wd[0], vv[0]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int cur = t & 1;
                Raw r;
                if (t < 7) read_ops(t + 1, r);
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[e][i][j] = mfma32(wd[cur][e][i], vv[cur][e][j], acc[e][i][j]);
                if (t < 7) {
                    form_ops(r, wd[cur ^ 1], vv[cur ^ 1]);
#pragma unroll
                    for (int e = 0; e < 2; ++e)
#pragma unroll
                        for (int i = 0; i < 2; ++i) asm volatile("" ::"v"(wd[cur ^ 1][e][i]), "v"(vv[cur ^ 1][e][i]));
                    if (WG_PIPE) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);      // LDS reads of pair t+1
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
#pragma unroll
                        for (int m = 0; m < 6; ++m) {
                            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);   // packed FMAs of pair t+1
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();          // everyone is done with this band
            if (has_next) {
                // every slot of this thread is the same channel quad (THREADS % 16 == 0): the affine
                // tables are read once per band, all values are formed, then stored
                if (PROA || PROD) {
                    const int c = (tid & 15) * 4;
                    if (PROA) {
                        const float4 A = *reinterpret_cast<const float4 *>(s_pa + c);
                        const float4 Cc = *reinterpret_cast<const float4 *>(s_pa + 128 + c);
#pragma unroll
                        for (int k = 0; k < kMaxD; ++k)
                            if (tid + k * THREADS < nd_rows * PW * 16 && okd[k])
                                rd[k] = affine_regs(A, zero4, Cc, rd[k], zero4, proa.relu, false);
                    }
                    if (PROD) {
                        const bool useb = prod.in2 != nullptr;
                        const float4 A = *reinterpret_cast<const float4 *>(s_pd + c);
                        const float4 Bc = *reinterpret_cast<const float4 *>(s_pd + 64 + c);
                        const float4 Cc = *reinterpret_cast<const float4 *>(s_pd + 128 + c);
#pragma unroll
                        for (int k = 0; k < kDy; ++k)
                            if (okdy[k]) rdy[k] = affine_regs(A, Bc, Cc, rdy[k], rdy2[k], prod.relu, useb);
                    }
                }
#pragma unroll
                for (int k = 0; k < kMaxD; ++k) {
                    const int idx = tid + k * THREADS;
                    if (idx < nd_rows * PW * 16) stage4(s_d + (2 * tr) * PW * 64, idx, rd[k]);
                }
#pragma unroll
                for (int k = 0; k < kDy; ++k) stage4(s_dy + (2 * tr) * TW * 64, tid + k * THREADS, rdy[k]);
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
// Fold of the per-block partials and the final transform in ONE launch (they were two):
//   Q[pos][co][ci] = sum over blocks of part[block][pos][co][ci]                 (fp64, fixed order)
//   dW[co][ci][r][t] = sum_{a,b} G[a][r] * Q[a*4+b][co][ci] * G[b][t]            (OIHW)
// block = 16 (co, ci) pairs x 16 slices of partial rows; 256 blocks.
// blockIdx.y = which of up to four convolutions (one launch finishes all weight gradients of the step's 3x3 stack)
struct FinishJobs { const float *part[4]; float *dw[4]; };
__global__ __launch_bounds__(256) void wgrad_wino_finish_kernel(const FinishJobs jobs, int nparts)
{
    __shared__ double s_acc[16][16][16];         // [position][slice][pair]
    const float *__restrict__ part = jobs.part[blockIdx.y];
    float *__restrict__ dw = jobs.dw[blockIdx.y];
    const int tx = threadIdx.x & 15, slice = threadIdx.x >> 4;
    const int idx = blockIdx.x * 16 + tx;        // (co, ci) pair
    double acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) acc[p] = 0.0;
    for (int r = slice; r < nparts; r += 16) {
        const float *row = part + (size_t)r * (16 * 4096) + idx;
#pragma unroll
        for (int p = 0; p < 16; ++p) acc[p] += (double)row[p * 4096];
    }
#pragma unroll
    for (int p = 0; p < 16; ++p) s_acc[p][slice][tx] = acc[p];
    __syncthreads();
    if (slice == 0) {
        const float G[4][3] = {{1.f, 0.f, 0.f}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0.f, 0.f, 1.f}};
        float Q[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            double t = 0.0;
            for (int j = 0; j < 16; ++j) t += s_acc[p][j][tx];
            // sign flips left open by the operand formation of conv3x3_wgrad_wino_kernel
            const int a = p >> 2, b = p & 3;
            const float sgn = ((a == 3) != (b == 3)) != (a == 2) ? -1.f : 1.f;
            Q[p] = sgn * (float)t;
        }
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                float sm = 0.f;
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) sm += G[a][r] * G[b][t] * Q[a * 4 + b];
                dw[idx * 9 + r * 3 + t] = sm;
            }
    }
}

}  // namespace

static int launch_wgrad_wino(const float *act, const float *dz, float *dw, float *ws, int B, int H, int W,
                             const ProIn proa, const ProIn prod, void *stream, bool finish = true)
{
    const int tiles_x = cdiv(W, wgw::TW), tiles_y = cdiv(H, wgw::TH);
    const int ntiles = B * tiles_x * tiles_y;
    const int grid = cova_internal_persistent_grid(ntiles);
    hipStream_t st = (hipStream_t)stream;
    const dim3 g(grid), blk(wgw::THREADS);
    if (proa.abc && prod.abc)
        hipLaunchKernelGGL((conv3x3_wgrad_wino_kernel<true, true>), g, blk, 0, st, act, dz, ws, H, W,
                           tiles_x, tiles_y, ntiles, proa, prod);
    else if (proa.abc)
        hipLaunchKernelGGL((conv3x3_wgrad_wino_kernel<true, false>), g, blk, 0, st, act, dz, ws, H, W,
                           tiles_x, tiles_y, ntiles, proa, prod);
    else if (prod.abc)
        hipLaunchKernelGGL((conv3x3_wgrad_wino_kernel<false, true>), g, blk, 0, st, act, dz, ws, H, W,
                           tiles_x, tiles_y, ntiles, proa, prod);
    else
        hipLaunchKernelGGL((conv3x3_wgrad_wino_kernel<false, false>), g, blk, 0, st, act, dz, ws, H, W,
                           tiles_x, tiles_y, ntiles, proa, prod);
    COVA_LAUNCH_CHECK();
    if (!finish) return COVA_OK;
    FinishJobs jobs{};
    jobs.part[0] = ws;
    jobs.dw[0] = dw;
    hipLaunchKernelGGL(wgrad_wino_finish_kernel, dim3(4096 / 16), dim3(256), 0, st, jobs, grid);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

// act, dz NHWC [B,H,W,64]; dw OIHW [64,64,3,3]; ws >= (grid*16*4096 + 16*4096) floats, which
// cova_conv3x3_wgrad_workspace_floats covers
COVA_API int cova_conv3x3_wgrad_wino(const float *act, const float *dz, float *dw, float *ws, int B,
                                     int H, int W, void *stream)
{
    COVA_REQUIRE(act && dz && dw && ws && B > 0 && H > 0 && W > 0);
    return launch_wgrad_wino(act, dz, dw, ws, B, H, W, ProIn{nullptr, nullptr, 0},
                             ProIn{nullptr, nullptr, 0}, stream);
}

// Same with operands transformed on load: activation = f(A*act + C) (act_abc [3,64], B row ignored;
// NULL = plain), gradient = A*dz + B*dz2 + C (dz_abc [3,64], dz2 nullable; NULL = plain).
COVA_API int cova_conv3x3_wgrad_wino_pro(const float *act, const float *act_abc, int act_relu,
                                         const float *dz, const float *dz2, const float *dz_abc,
                                         float *dw, float *ws, int B, int H, int W, void *stream)
{
    COVA_REQUIRE(act && dz && dw && ws && B > 0 && H > 0 && W > 0);
    return launch_wgrad_wino(act, dz, dw, ws, B, H, W, ProIn{act_abc, nullptr, act_relu},
                             ProIn{dz_abc, dz2, 0}, stream);
}

// The same in two steps, so that ONE launch can finish the weight gradients of several convolutions: the per-block
// partial sums only (each convolution needs its own workspace) ...
COVA_API int cova_conv3x3_wgrad_wino_partial(const float *act, const float *act_abc, int act_relu,
                                             const float *dz, const float *dz2, const float *dz_abc,
                                             float *ws, int B, int H, int W, void *stream)
{
    COVA_REQUIRE(act && dz && ws && B > 0 && H > 0 && W > 0);
    return launch_wgrad_wino(act, dz, nullptr, ws, B, H, W, ProIn{act_abc, nullptr, act_relu},
                             ProIn{dz_abc, dz2, 0}, stream, false);
}

// ... and fold + final transform of up to four of them (pairs 1..3 nullable) into their OIHW gradients
COVA_API int cova_conv3x3_wgrad_wino_finish(const float *ws0, float *dw0, const float *ws1, float *dw1,
                                            const float *ws2, float *dw2, const float *ws3, float *dw3, int B, int H,
                                            int W, void *stream)
{
    COVA_REQUIRE(ws0 && dw0 && B > 0 && H > 0 && W > 0);
    COVA_REQUIRE((ws1 == nullptr) == (dw1 == nullptr) && (ws2 == nullptr) == (dw2 == nullptr) &&
                 (ws3 == nullptr) == (dw3 == nullptr));
    const float *ws[4] = {ws0, ws1, ws2, ws3};
    float *dw[4] = {dw0, dw1, dw2, dw3};
    FinishJobs jobs{};
    int n = 0;
    for (int i = 0; i < 4; ++i)
        if (ws[i]) { jobs.part[n] = ws[i]; jobs.dw[n] = dw[i]; ++n; }
    const int grid = cova_internal_persistent_grid(B * cdiv(W, wgw::TW) * cdiv(H, wgw::TH));
    hipLaunchKernelGGL(wgrad_wino_finish_kernel, dim3(4096 / 16, n), dim3(256), 0, (hipStream_t)stream, jobs, grid);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}
