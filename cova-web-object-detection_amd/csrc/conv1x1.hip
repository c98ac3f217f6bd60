// 1x1 convolutions of the ResNet-50-stem extension (BASELINE.json configs[2] / configs[4]: the
// representation network is torchvision resnet50 `children()[:-5]`, i.e. conv1, bn1, relu, maxpool
// and layer1 = 3 Bottleneck blocks: 1x1 Cin->64, 3x3 64->64, 1x1 64->256, + a 1x1 64->256
// downsample branch in block 0).  The reference itself only wires resnet18 (models.py:49); this
// file is the build-defined extension of that call site, with the same NHWC layout and the same
// prologue / epilogue conventions as the Winograd 3x3 kernels (conv_wino.hip):
//
//   out[r, co] = sum_ci f(A[ci]*in[r,ci] + B[ci]*in2[r,ci] + C[ci]) * w[co, ci]      r = pixel row
//
// * prologue  : BatchNorm(+ReLU) of the producer, or the BatchNorm-backward apply
//               dz = A*dy + B*z + C, evaluated on load (the normalised map / dz is never written);
// * epilogue  : per-channel statistics partials (sum y, sum y^2) for the train-mode BatchNorm that
//               follows, or -- data gradient -- (+ residual-branch gradient) * ReLU mask and the
//               BatchNorm-backward sums (sum dy, sum dy*xhat) of the layer in front.
//
// Implicit GEMM M = pixels, N = Cout, K = Cin on v_mfma_f32_32x32x2_f32 (exact f32).  A wave owns
// 32 pixel rows: lane (p = lane&31, h = lane>>5) supplies row p's channels of K-half h, so its
// operand is 128 contiguous bytes per 32-channel chunk -- loaded straight from HBM into registers
// (8 float4).  For the 64-channel operand that is done row per lane; the 256-channel operand (round 3) is fetched
// coalesced, 8 adjacent lanes per 128-byte line, and turned into the row-per-lane MFMA layout by a wave-private LDS
// transposer (see CO in the kernel: 4.2 vs 6.2 TB/s of load bandwidth between the two patterns).  K is permuted (step s of half h = channel h*Cin/2+s)
// identically for both operands, which leaves every dot product an exact f32 FMA chain.  The
// weight [Cout][Cin+4] stays resident in LDS for the whole persistent launch; a lane's four
// consecutive k are one ds_read_b128 (row stride Cin+4 floats: the 16 lanes of a b128 group hit
// 16 distinct 16-byte slots).  Output: lane owns channel n*32+p of 16 rows per accumulator, so
// channel statistics are plain per-lane sums kept in registers across all tiles of the launch.
//
// Roofline bookkeeping: 2*Cin*Cout FLOP and 4*(Cin+Cout) compulsory bytes per pixel --
// 64->256 / 256->64: 32,768 FLOP vs 1,280 B (25.6 FLOP/B: at the f32-MFMA / HBM ridge),
// 64->64: 8,192 FLOP vs 512 B (HBM bound).
#include "bf3.h"

int cova_internal_persistent_grid2(int ntiles, int blocks_per_cu);

namespace {

__device__ __forceinline__ f32x16 c11_mfma_bf(u32x4 a, u32x4 b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

struct C11Args {
    const float *in, *in2, *pro;          // pro [3][Cin] = A | B | C (nullable)
    const float *w;                       // [Cout][Cin], or [Cin][Cout] when w_trans
    const float *addend;                  // nullable [R][Cout]
    const float *act;                     // mask source (act > 0), or null: fma(msc, z, msh) > 0
    const float *msc, *msh;
    const float *z, *mean, *invstd;       // xhat source of the BatchNorm-backward sums
    const float *z2, *mean2, *invstd2;    // optional second xhat source (block 0's downsample BN)
    float *out, *part, *part2;
    long long R;
    int pro_relu, w_trans;
    // EXT (linear form of a conv+BatchNorm backward, conv1x1_lin.hip): 64 more K channels
    // relu?(A3*in3 + C3) with weight w3 [64][Cout], and a per-output-channel constant cvec
    const float *in3, *pro3, *w3, *cvec;
    int pro3_relu;
    // SIDE: the prologue's result f(A*in + B*in2 + C) is also written here ([R][Cin]): the consumer of a
    // Bottleneck output materialises it (relu(bn3(z3) + x)) while it reads its operands anyway
    float *side;
    // ... and its ReLU decisions as one bit per element ([R][Cin/32] words, bit = channel & 31; nullable): the data
    // gradient that needs this map only as a mask source reads 1/32 of the bytes
    unsigned *side_bits;
    const unsigned *act_bits;             // mask source of the EPI 3 epilogue in that form (instead of `act`)
};

template <int CIN, int COUT, bool EXT = false, bool BF_OK = true>
struct C11Geo {
    static constexpr int XK = EXT ? 64 : 0;            // extra K channels (second operand tensor)
    static constexpr int WSTR = CIN + XK + 4;
    static constexpr int KH = CIN / 2;                 // channels per half-wave
    static constexpr int CHUNKS = KH / 32;             // 32-channel register chunks per half
    static constexpr int NT = COUT / 32;               // 32-column accumulators per row tile
    static constexpr int NTP = NT > 4 ? 4 : NT;        // accumulators per pass (register budget)
    static constexpr int PASSES = NT / NTP;
    // BF: the product runs on the bf16 matrix pipe, every f32 operand as three round-to-nearest bf16 pieces and the six
    // products of order <= 2 accumulated in f32 (bf3.h; DESIGN.md sections 11.8 / 12.8) -- 2.67x less matrix time than
    // v_mfma_f32_32x32x2_f32, on a pipe the vector work does not share: the launches become HBM bound.
    // The weights sit in LDS as [piece 3][co][slots of 8 channels as packed bf16] with one slot of padding per row (an odd
    // number of 16-byte slots: the 16 lanes of a ds_read_b128 group hit 16 distinct slots).
    static constexpr bool BF = BF_OK;
    static constexpr int WROW = (CIN + XK) / 2 + 4;    // dwords per (piece, co) row of the bf16 image
    static constexpr int W_FLOATS = BF ? 3 * COUT * WROW : COUT * WSTR;
    static constexpr int RED_FLOATS = 8 * 3 * COUT;    // aliased onto the weights after the tile loop (<= 8 waves)
    static constexpr int LDS_FLOATS = W_FLOATS + 3 * CIN + 3 * XK;
};

// PRO: 0 plain input | 1 f(A*in + C) | 2 f(A*in + B*in2 + C).  EPI: 0 store | 1 store + (sum y, sum y^2)
// | 2 (acc + addend) * mask, (sum dy, sum dy*xhat[, sum dy*xhat2]) | 3 (acc + addend) * mask, no sums.
// NW = waves per block (4: two blocks per CU; 8: one block per CU when the weights need > 80 KB of LDS).
template <int CIN, int COUT, int PRO, int EPI, bool HAS_ADD, bool MASK_ACT, bool HAS_X2, bool EXT = false,
          int NW = 4, bool SIDE = false>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void conv1x1_kernel(const C11Args a)
{
    // (the 64 -> 256 data-gradient variants WITH BatchNorm sums stay on the f32 MFMA: their epilogue keeps up to four operands
    // per output in flight and the bf16 path's operand pieces would spill 35-160 registers; the step does not launch them
    // when the linear form of conv1x1_lin.hip is on)
    using G = C11Geo<CIN, COUT, EXT, !(EPI == 2 && COUT == 256)>;
    constexpr int NTHR = NW * 64;
    // CO: the 256-channel operand is fetched COALESCED -- 8 adjacent lanes on one 128-byte line of a row -- and brought
    // into the MFMA layout (lane = row) through a wave-private 8 KB LDS transposer.  Row-per-lane loads (8 x 16 bytes
    // walking the lane's own line, 64 lines per wave instruction) reach 4.2 TB/s on this part, the coalesced pattern
    // 6.2 TB/s (tools/lane_pattern_probe2.py); the element-wise prologue and the side output run in the coalesced layout
    // (per-lane channel constants, 128-byte stores).  Needs the 8-wave / one-block-per-CU geometry for the LDS room.
    constexpr bool CO = NW == 8;
    // (BF: the transposer takes the tile's 32 rows in two rounds of 16 -- 4 KB per wave -- which is what lets the 110 KB
    // piece image of a 64 -> 256 weight and eight transposers share the CU's 160 KB)
    constexpr bool BF = G::BF;
    constexpr int TRF4 = BF ? 256 : 512;                    // float4 slots of a wave's transposer
    __shared__ __attribute__((aligned(16))) float lds[G::LDS_FLOATS + (CO ? NW * TRF4 * 4 : 0)];
    float4 *s_tr = reinterpret_cast<float4 *>(lds + G::LDS_FLOATS) + (threadIdx.x >> 6) * TRF4;     // this wave's transposer
    const int ca = (threadIdx.x & 63) >> 4, chh = ((threadIdx.x & 63) >> 3) & 1, cq = threadIdx.x & 7;   // CO lane -> (row & 3, K-half, 16-byte piece)
    float *Ws = lds, *s_pro = lds + G::W_FLOATS, *s_pro3 = lds + G::W_FLOATS + 3 * CIN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = lane & 31, h = lane >> 5;

    // ---- stage the weight [co][WSTR] (+ the prologue table) once per block
    if constexpr (BF) {                                    // (co, slot g = channels 8g .. 8g+7) -> three packed-bf16 u32x4
        u32x4 *Wp = reinterpret_cast<u32x4 *>(lds);
        constexpr int SL = (CIN + G::XK) / 8;
        for (int i = tid; i < COUT * SL; i += NTHR) {
            const int co = i / SL, g = i - co * SL;
            float wv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int ci = 8 * g + k;
                if (EXT && ci >= CIN) wv[k] = a.w3[(size_t)(ci - CIN) * COUT + co];               // w3 [k 64][COUT]
                else wv[k] = a.w_trans ? a.w[(size_t)ci * COUT + co] : a.w[(size_t)co * CIN + ci];
            }
            u32x4 q0, q1, q2;
            bf3_split8(wv, q0, q1, q2);
            Wp[(0 * COUT + co) * (G::WROW / 4) + g] = q0;
            Wp[(1 * COUT + co) * (G::WROW / 4) + g] = q1;
            Wp[(2 * COUT + co) * (G::WROW / 4) + g] = q2;
        }
    } else if (!a.w_trans) {
        for (int i = tid; i < COUT * (CIN / 4); i += NTHR) {
            const int co = i / (CIN / 4), c4 = i - co * (CIN / 4);
            *reinterpret_cast<float4 *>(Ws + co * G::WSTR + 4 * c4) =
                *reinterpret_cast<const float4 *>(a.w + (size_t)co * CIN + 4 * c4);
        }
    } else {
        for (int i = tid; i < CIN * COUT; i += NTHR) {
            const int ci = i / COUT, co = i - ci * COUT;
            Ws[co * G::WSTR + ci] = a.w[i];
        }
    }
    if (EXT) {
        for (int i = tid; i < (BF ? 0 : 64 * COUT); i += NTHR) {          // w3 [k 64][COUT] (BF: staged above)
            const int k = i / COUT, co = i - k * COUT;
            Ws[co * G::WSTR + CIN + k] = a.w3[i];
        }
        for (int i = tid; i < 3 * 64; i += NTHR) s_pro3[i] = a.pro3[i];
    }
    if (PRO != 0)
        for (int i = tid; i < 3 * CIN; i += NTHR) s_pro[i] = a.pro[i];
    __syncthreads();

    const long long ntiles = (a.R + 31) / 32;
    const long long stride = (long long)gridDim.x * NW;
    float su[G::NT], sq[G::NT], sq2[G::NT];
#pragma unroll
    for (int n = 0; n < G::NT; ++n) su[n] = sq[n] = sq2[n] = 0.f;

    // operand chunk of (tile, chunk c): 8 float4 = channels h*KH + 32c .. +31 of row tile*32 + p
    auto issue = [&](long long tile, int c, float4 (&v)[8], float4 (&v2)[8]) {
        if constexpr (CO) {                                     // v[j]: row 4j + ca, K-half chh, piece cq of the chunk
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                long long row = tile * 32 + 4 * j + ca;
                if (row >= a.R) row = a.R - 1;
                if (EXT && c == G::CHUNKS) {
                    v[j] = *reinterpret_cast<const float4 *>(a.in3 + (size_t)row * 64 + chh * 32 + 4 * cq);
                } else {
                    const size_t o = (size_t)row * CIN + chh * G::KH + c * 32 + 4 * cq;
                    v[j] = *reinterpret_cast<const float4 *>(a.in + o);
                    if (PRO == 2) v2[j] = *reinterpret_cast<const float4 *>(a.in2 + o);
                }
            }
            return;
        }
        long long row = tile * 32 + p;
        if (row >= a.R) row = a.R - 1;                       // clamped: loads stay unconditional
        if (EXT && c == G::CHUNKS) {                         // the extra 64 channels: half h takes 32h .. 32h+31
            const size_t o3 = (size_t)row * 64 + h * 32;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4 *>(a.in3 + o3 + 4 * j);
            return;
        }
        const size_t o = (size_t)row * CIN + h * G::KH + c * 32;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const float4 *>(a.in + o + 4 * j);
        if (PRO == 2) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v2[j] = *reinterpret_cast<const float4 *>(a.in2 + o + 4 * j);
        }
    };
    auto prologue = [&](long long tile, int c, float4 (&v)[8], const float4 (&v2)[8], float (&x)[32]) {
        if constexpr (CO) {
            const bool ext = EXT && c == G::CHUNKS;
            int ch = ext ? chh * 32 + 4 * cq : chh * G::KH + c * 32 + 4 * cq;      // this lane's channel quad, all 8 rows
            asm volatile("" : "+v"(ch));
            float Aa[4] = {1.f, 1.f, 1.f, 1.f}, Ba[4] = {0.f, 0.f, 0.f, 0.f}, Ca[4] = {0.f, 0.f, 0.f, 0.f};
            bool relu = false;
            if (ext) {
                const float4 A = *reinterpret_cast<const float4 *>(s_pro3 + ch), C = *reinterpret_cast<const float4 *>(s_pro3 + 128 + ch);
                Aa[0] = A.x; Aa[1] = A.y; Aa[2] = A.z; Aa[3] = A.w; Ca[0] = C.x; Ca[1] = C.y; Ca[2] = C.z; Ca[3] = C.w;
                relu = a.pro3_relu != 0;
            } else if (PRO != 0) {
                const float4 A = *reinterpret_cast<const float4 *>(s_pro + ch), C = *reinterpret_cast<const float4 *>(s_pro + 2 * CIN + ch);
                Aa[0] = A.x; Aa[1] = A.y; Aa[2] = A.z; Aa[3] = A.w; Ca[0] = C.x; Ca[1] = C.y; Ca[2] = C.z; Ca[3] = C.w;
                if (PRO == 2) {
                    const float4 B = *reinterpret_cast<const float4 *>(s_pro + CIN + ch);
                    Ba[0] = B.x; Ba[1] = B.y; Ba[2] = B.z; Ba[3] = B.w;
                }
                relu = a.pro_relu != 0;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float e[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
                if (ext || PRO != 0) {
                    const float e2[4] = {v2[j].x, v2[j].y, v2[j].z, v2[j].w};
#pragma unroll
                    for (int k = 0; k < 4; ++k)         // (the same expressions as the row-per-lane path)
                        e[k] = (PRO == 2 && !ext) ? fmaf(Aa[k], e[k], fmaf(Ba[k], e2[k], Ca[k])) : fmaf(Aa[k], e[k], Ca[k]);
                    if (relu) {
#pragma unroll
                        for (int k = 0; k < 4; ++k) e[k] = e[k] > 0.f ? e[k] : 0.f;
                    }
                }
                const float4 e4 = make_float4(e[0], e[1], e[2], e[3]);
                const long long rj = tile * 32 + 4 * j + ca;
                if (SIDE && !ext && rj < a.R) *reinterpret_cast<float4 *>(a.side + (size_t)rj * CIN + ch) = e4;
                // transposer slot of (row r, half, piece q): (2 r + half) * 8 + (q ^ (r & 7)) -- the eight lanes of a line
                // write one 128-byte row, the eight readers of consecutive rows hit eight different 16-byte bank groups
                const int rl = 4 * j + ca;
                if constexpr (BF) {                     // two rounds of 16 rows through a 4 KB transposer (same wave: LDS operations stay in order)
                    const int rr = rl & 15;
                    s_tr[(2 * rr + chh) * 8 + (cq ^ (rr & 7))] = e4;
                    if (j == 3 || j == 7) {
                        if ((p >> 4) == (j >> 2)) {
                            const int pr = p & 15;
#pragma unroll
                            for (int q = 0; q < 8; ++q) {
                                const float4 t = s_tr[(2 * pr + h) * 8 + (q ^ (pr & 7))];
                                x[4 * q] = t.x; x[4 * q + 1] = t.y; x[4 * q + 2] = t.z; x[4 * q + 3] = t.w;
                            }
                        }
                    }
                } else {
                    s_tr[(2 * rl + chh) * 8 + (cq ^ (rl & 7))] = e4;
                }
            }
            if constexpr (!BF) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float4 t = s_tr[(2 * p + h) * 8 + (q ^ (p & 7))];
                    x[4 * q] = t.x; x[4 * q + 1] = t.y; x[4 * q + 2] = t.z; x[4 * q + 3] = t.w;
                }
            }
            return;
        }
        const long long srow = tile * 32 + p;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float e[4] = {v[j].x, v[j].y, v[j].z, v[j].w};
            if (EXT && c == G::CHUNKS) {
                int ch = h * 32 + 4 * j;
                asm volatile("" : "+v"(ch));
                const float4 A = *reinterpret_cast<const float4 *>(s_pro3 + ch);
                const float4 C = *reinterpret_cast<const float4 *>(s_pro3 + 128 + ch);
                const float Aa[4] = {A.x, A.y, A.z, A.w}, Ca[4] = {C.x, C.y, C.z, C.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) e[k] = fmaf(Aa[k], e[k], Ca[k]);
                if (a.pro3_relu) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) e[k] = e[k] > 0.f ? e[k] : 0.f;
                }
            } else if (PRO != 0) {
                int ch = h * G::KH + c * 32 + 4 * j;
                asm volatile("" : "+v"(ch));           // keep the table reads here: hoisted, they cost 96 registers
                const float4 A = *reinterpret_cast<const float4 *>(s_pro + ch);
                const float4 C = *reinterpret_cast<const float4 *>(s_pro + 2 * CIN + ch);
                const float Aa[4] = {A.x, A.y, A.z, A.w}, Ca[4] = {C.x, C.y, C.z, C.w};
                if (PRO == 2) {
                    const float4 B = *reinterpret_cast<const float4 *>(s_pro + CIN + ch);
                    const float Ba[4] = {B.x, B.y, B.z, B.w};
                    const float e2[4] = {v2[j].x, v2[j].y, v2[j].z, v2[j].w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) e[k] = fmaf(Aa[k], e[k], fmaf(Ba[k], e2[k], Ca[k]));
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) e[k] = fmaf(Aa[k], e[k], Ca[k]);
                }
                if (a.pro_relu) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) e[k] = e[k] > 0.f ? e[k] : 0.f;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) x[4 * j + k] = e[k];
            if (SIDE && !(EXT && c == G::CHUNKS) && srow < a.R)
                *reinterpret_cast<float4 *>(a.side + (size_t)srow * CIN + h * G::KH + c * 32 + 4 * j) =
                    make_float4(e[0], e[1], e[2], e[3]);
        }
    };
    // SIDE: the ReLU decisions of a chunk's 32 values as one word (bit = channel & 31).  x = relu(.) >= +0 here (the
    // materialising launch always applies the ReLU), so x > 0 <=> its bit pattern != 0: min(bits, 1), no compare
    auto pack_bits = [&](const float (&x)[32]) {
        unsigned word = 0u;
#pragma unroll
        for (int i = 0; i < 32; ++i) word |= min(__float_as_uint(x[i]), 1u) << i;
        return word;
    };

    // Cross-tile operand prefetch, except in the 256-channel data-gradient variants: those move ~5x more
    // bytes in their epilogue (addend / mask / xhat operands) than through the MFMA operand, are bound by
    // loads in flight, and need the 64 registers for a second resident wave per SIMD instead.
    constexpr bool PREFETCH = !((EPI == 2 || EPI == 3) && COUT == 256);
    // TR: transposed product for the variants without per-channel sums.  With the weights as the MFMA's A
    // operand the accumulator is D[channel][pixel]: a lane holds pixel p and, per register quad, FOUR
    // CONSECUTIVE CHANNELS (32n + 8j + 4h ..+3), so the epilogue moves float4s (a quarter of the memory
    // instructions of the channel-per-lane layout, whose 4-byte accesses bound the 256-channel epilogues).
    constexpr bool TR = EPI == 3 && COUT == 256 && !EXT;
    // accumulators per pass.  BF + TR (the block-input gradient 64 -> 256 with addend and mask): two, and the pass's epilogue
    // operands are requested BEFORE its MFMAs -- that launch moves 5x more bytes through its epilogue than through the MFMA
    // operand and was bound by loads in flight (two waves per SIMD), not by the matrix pipe
    constexpr bool EARLY = BF && TR;
    constexpr int NTP = EARLY ? 2 : G::NTP, PASSES = G::NT / NTP;
    constexpr int TC = G::CHUNKS + (EXT ? 1 : 0);          // register chunks per tile incl. the extra operand
    long long tile = (long long)blockIdx.x * NW + wave;
    float4 nv[8], nv2[8];
    if (PREFETCH && tile < ntiles) issue(tile, 0, nv, nv2);
    for (; tile < ntiles; tile += stride) {
        const long long r0 = tile * 32;
        if (!PREFETCH) issue(tile, 0, nv, nv2);
        // Cin = 64: the tile's operand is 32 registers and is reused by both 128-channel passes of a
        // 256-channel output (so only 4 accumulators are live at a time).  Cin = 256: four chunks stream
        // through a single 64-channel pass.
        float x[32];
        if (TC == 1) {
            prologue(tile, 0, nv, nv2, x);
            if (PREFETCH && tile + stride < ntiles) issue(tile + stride, 0, nv, nv2);      // next tile in flight
        }
        // BF: the tile's operand as three packed-bf16 pieces (A operand of v_mfma_f32_32x32x16_bf16: K-step s = the lane's
        // channels 8s .. 8s+7).  The bf16 MFMA drops product bits below ~2^-26 of the largest exponent of a step toward
        // -infinity (tools/probe/mfma_round_probe.hip): every second row tile is multiplied with NEGATED activations and its
        // accumulators are negated back, so that the bias has no common sign over the map (as conv_wino4_split.h).
        u32x4 xp[3][4];
        const bool neg = BF && (tile & 1);
        auto split_x = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                float t8[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) t8[k] = neg ? -x[8 * s4 + k] : x[8 * s4 + k];
                bf3_split8(t8, xp[0][s4], xp[1][s4], xp[2][s4]);
            }
        };
        if constexpr (BF && TC == 1) split_x();
        // TR epilogue in two halves: the requests (addend, mask source) and the arithmetic + stores
        long long trow = r0 + p;
        const bool tr_ok = trow < a.R;
        if (!tr_ok) trow = a.R - 1;
        const size_t rbase = (size_t)trow * COUT + 4 * h;
        const bool bits = a.act_bits != nullptr;                  // (wave-uniform)
        auto tr_request = [&](int ps, int nn, float4 (&ad)[2][4], float4 (&mk)[2][4], unsigned (&mb)[2]) __attribute__((always_inline)) {
            if (EPI != 3) return;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (bits) mb[u] = a.act_bits[(size_t)trow * (COUT / 32) + ps * NTP + nn + u] >> (4 * h);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const size_t o = rbase + (ps * NTP + nn + u) * 32 + 8 * j;
                    if (HAS_ADD) ad[u][j] = *reinterpret_cast<const float4 *>(a.addend + o);
                    if (!bits) mk[u][j] = *reinterpret_cast<const float4 *>(a.act + o);
                }
            }
        };
        auto tr_finish = [&](int ps, int nn, const f32x16 (&acc)[NTP], const float4 (&ad)[2][4], const float4 (&mk)[2][4],
                             const unsigned (&mb)[2]) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float4 v = make_float4(acc[nn + u][4 * j], acc[nn + u][4 * j + 1], acc[nn + u][4 * j + 2], acc[nn + u][4 * j + 3]);
                    if (EPI == 3) {
                        if (HAS_ADD) { v.x += ad[u][j].x; v.y += ad[u][j].y; v.z += ad[u][j].z; v.w += ad[u][j].w; }
                        if (bits) {                    // bit 8j + 4h + k of the word = channel 32n + 8j + 4h + k
                            const unsigned q = mb[u] >> (8 * j);
                            if (!(q & 1u)) v.x = 0.f;
                            if (!(q & 2u)) v.y = 0.f;
                            if (!(q & 4u)) v.z = 0.f;
                            if (!(q & 8u)) v.w = 0.f;
                        } else {
                            if (!(mk[u][j].x > 0.f)) v.x = 0.f;
                            if (!(mk[u][j].y > 0.f)) v.y = 0.f;
                            if (!(mk[u][j].z > 0.f)) v.z = 0.f;
                            if (!(mk[u][j].w > 0.f)) v.w = 0.f;
                        }
                    }
                    if (tr_ok) *reinterpret_cast<float4 *>(a.out + rbase + (ps * NTP + nn + u) * 32 + 8 * j) = v;
                }
        };
#pragma unroll
        for (int ps = 0; ps < PASSES; ++ps) {
            f32x16 acc[NTP];
            float4 e_ad[2][4], e_mk[2][4];
            unsigned e_mb[2] = {0u, 0u};
            if constexpr (EARLY) {
                tr_request(ps, 0, e_ad, e_mk, e_mb);
                __builtin_amdgcn_sched_barrier(0);         // the requests stay ahead of the MFMAs (not sunk to their uses)
            }
#pragma unroll
            for (int n = 0; n < NTP; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
#pragma unroll
            for (int c = 0; c < TC; ++c) {
                if (TC > 1) {
                    prologue(tile, c, nv, nv2, x);
                    if (SIDE && CIN == 256 && a.side_bits != nullptr && tile * 32 + p < a.R)     // (a lane owns one word per chunk)
                        a.side_bits[(size_t)(tile * 32 + p) * (CIN / 32) + h * 4 + c] = pack_bits(x);
                    // next chunk (or the next tile's first chunk) in flight during this chunk's MFMAs
                    if (c + 1 < TC) issue(tile, c + 1, nv, nv2);
                    else if (tile + stride < ntiles) issue(tile + stride, 0, nv, nv2);
                    if constexpr (BF) split_x();
                    __builtin_amdgcn_sched_barrier(0);     // one chunk of loads in flight, not all four
                }
                // weight columns of this chunk: the lane's K-half of the main operand, or of the extra one
                const int wcol = (EXT && c == G::CHUNKS) ? CIN + h * 32 : h * G::KH + c * 32;
                if constexpr (BF) {
                    const u32x4 *Wp = reinterpret_cast<const u32x4 *>(lds);
                    // one accumulator at a time (a dependent chain of bf16 MFMAs issues back to back: tools/probe/mfma32_probe.hip):
                    // three weight pieces = 12 registers in flight instead of 48
#pragma unroll
                    for (int n = 0; n < NTP; ++n) {
#pragma unroll
                        for (int s4 = 0; s4 < 4; ++s4) {
                            u32x4 b[3];
#pragma unroll
                            for (int pc = 0; pc < 3; ++pc)
                                b[pc] = Wp[(pc * COUT + (ps * NTP + n) * 32 + p) * (G::WROW / 4) + wcol / 8 + s4];
                            // the six products of order <= 2, smallest first: x2 w0, x0 w2, x1 w1, x1 w0, x0 w1, x0 w0
                            auto mf = [&](int xi, int wi) __attribute__((always_inline)) {
                                acc[n] = TR ? c11_mfma_bf(b[wi], xp[xi][s4], acc[n]) : c11_mfma_bf(xp[xi][s4], b[wi], acc[n]);
                            };
                            mf(2, 0); mf(0, 2); mf(1, 1); mf(1, 0); mf(0, 1); mf(0, 0);
                        }
                    }
                } else
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    float4 b[NTP];
#pragma unroll
                    for (int n = 0; n < NTP; ++n)
                        b[n] = *reinterpret_cast<const float4 *>(
                            Ws + ((ps * NTP + n) * 32 + p) * G::WSTR + wcol + 4 * q);
#pragma unroll
                    for (int n = 0; n < NTP; ++n)
                        acc[n] = TR ? mfma32(b[n].x, x[4 * q + 0], acc[n]) : mfma32(x[4 * q + 0], b[n].x, acc[n]);
#pragma unroll
                    for (int n = 0; n < NTP; ++n)
                        acc[n] = TR ? mfma32(b[n].y, x[4 * q + 1], acc[n]) : mfma32(x[4 * q + 1], b[n].y, acc[n]);
#pragma unroll
                    for (int n = 0; n < NTP; ++n)
                        acc[n] = TR ? mfma32(b[n].z, x[4 * q + 2], acc[n]) : mfma32(x[4 * q + 2], b[n].z, acc[n]);
#pragma unroll
                    for (int n = 0; n < NTP; ++n)
                        acc[n] = TR ? mfma32(b[n].w, x[4 * q + 3], acc[n]) : mfma32(x[4 * q + 3], b[n].w, acc[n]);
                }
            }
            if (neg) {
#pragma unroll
                for (int n = 0; n < NTP; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[n][r] = -acc[n][r];
            }
            if (TR) {
                // ---- float4 epilogue: register quad j of accumulator nn = channels 32n + 8j + 4h ..+3 of row r0 + p
#pragma unroll
                for (int nn = 0; nn < NTP; nn += 2) {       // two accumulators = 8 float4 per tensor in flight
                    float4 ad[2][4], mk[2][4];
                    unsigned mb[2] = {0u, 0u};
                    if (!EARLY) tr_request(ps, nn, ad, mk, mb);
                    tr_finish(ps, nn, acc, EARLY ? e_ad : ad, EARLY ? e_mk : mk, EARLY ? e_mb : mb);
                }
                continue;                                      // next pass
            }
            // ---- epilogue of this pass: lane owns channel n*32 + p of rows mfma32_row(r, lane); 8 rows at
            // a time so that the operands in flight (up to 4 per row) stay within the register budget
#pragma unroll
            for (int nn = 0; nn < NTP; ++nn) {
                const int n = ps * NTP + nn;
                const int cn = n * 32 + p;
                float mu = 0.f, is = 0.f, mu2 = 0.f, is2 = 0.f, msc = 0.f, msh = 0.f, cv = 0.f;
                if (EPI == 2) {
                    mu = a.mean[cn]; is = a.invstd[cn];
                    if (!MASK_ACT) { msc = a.msc[cn]; msh = a.msh[cn]; }
                    if (HAS_X2) { mu2 = a.mean2[cn]; is2 = a.invstd2[cn]; }
                }
                if (EXT) cv = a.cvec[cn];
                constexpr int RB = (HAS_X2 && COUT == 256) ? 4 : 8;    // rows per operand batch
#pragma unroll
                for (int rh = 0; rh < 16; rh += RB) {
                    float ad[RB], mk[RB], zz[RB], z2v[RB];
                    if (EPI == 2 || EPI == 3) {            // request every operand first, store last
#pragma unroll
                        for (int r = 0; r < RB; ++r) {
                            long long row = r0 + mfma32_row(rh + r, lane);
                            if (row >= a.R) row = a.R - 1;
                            const size_t o = (size_t)row * COUT + cn;
                            if (HAS_ADD) ad[r] = a.addend[o];
                            if (MASK_ACT) mk[r] = a.act[o];
                            if (EPI == 2) zz[r] = a.z[o];
                            if (HAS_X2) z2v[r] = a.z2[o];
                        }
                    }
#pragma unroll
                    for (int r = 0; r < RB; ++r) {
                        const long long row = r0 + mfma32_row(rh + r, lane);
                        if (row < a.R) {
                            float v = acc[nn][rh + r];
                            if (EXT) v += cv;
                            if (EPI == 2) {
                                if (HAS_ADD) v += ad[r];
                                const float m = MASK_ACT ? mk[r] : fmaf(msc, zz[r], msh);
                                if (!(m > 0.f)) v = 0.f;
                                su[n] += v;
                                sq[n] += v * ((zz[r] - mu) * is);
                                if (HAS_X2) sq2[n] += v * ((z2v[r] - mu2) * is2);
                            } else if (EPI == 3) {
                                if (HAS_ADD) v += ad[r];
                                if (!(mk[r] > 0.f)) v = 0.f;
                            } else if (EPI == 1) {
                                su[n] += v;
                                sq[n] += v * v;
                            }
                            a.out[(size_t)row * COUT + cn] = v;
                        }
                    }
                }
            }
        }
    }
    if (EPI == 1 || EPI == 2) {
        __syncthreads();                                   // every wave is done with the weights
        float *s_red = lds;                                // [NW][3][COUT]
#pragma unroll
        for (int n = 0; n < G::NT; ++n) {
            const float s0 = su[n] + __shfl_xor(su[n], 32, 64);
            const float s1 = sq[n] + __shfl_xor(sq[n], 32, 64);
            const float s2 = HAS_X2 ? sq2[n] + __shfl_xor(sq2[n], 32, 64) : 0.f;
            if (h == 0) {
                s_red[(wave * 3 + 0) * COUT + n * 32 + p] = s0;
                s_red[(wave * 3 + 1) * COUT + n * 32 + p] = s1;
                s_red[(wave * 3 + 2) * COUT + n * 32 + p] = s2;
            }
        }
        __syncthreads();
        for (int i = tid; i < 2 * COUT; i += NTHR) {
            const int which = i / COUT, c = i - which * COUT;
            float t = 0.f;
            for (int w = 0; w < NW; ++w) t += s_red[(w * 3 + which) * COUT + c];
            a.part[(size_t)blockIdx.x * 2 * COUT + i] = t;
        }
        if (HAS_X2)
            for (int i = tid; i < 2 * COUT; i += NTHR) {
                const int which = i / COUT, c = i - which * COUT;
                float t = 0.f;
                for (int w = 0; w < NW; ++w) t += s_red[(w * 3 + (which ? 2 : 0)) * COUT + c];
                a.part2[(size_t)blockIdx.x * 2 * COUT + i] = t;
            }
    }
}

// ------------------------------------------------------------------------------------ weight gradient
// dw[co][ci] = sum_r fz(dz[r,co]) * fa(act[r,ci]):  GEMM M = co, N = ci, K = pixels.  One MFMA k-step is a
// pixel pair (half-wave h takes pixel 2s+h); lane p supplies channels {2p, 2p+1} of its 64-channel block as
// one float2, so accumulator (i, j) holds co = 64u + 2*row + i, ci = 64v + 2*col + j.  A wave owns one
// 64x64 unit of dw: with 4 units (256x64 / 64x256) the four waves share the pixel range, with one unit
// (64x64) they split it and are summed through LDS.  Every block writes one fp32 partial [CO][CI]; the
// reduce kernel folds them in fp64 (fixed order: deterministic).
struct W11Args {
    const float *dz, *dz2, *dz_abc;       // dz_abc [3][CO] nullable
    const float *act, *act_abc;           // act_abc [3][CI] nullable (A | unused | C)
    float *ws;
    long long R;
    int act_relu;
};

template <int CO, int CI, bool PRO_DZ, bool PRO_ACT>
__global__ __launch_bounds__(256) void conv1x1_wgrad_kernel(const W11Args a)
{
    constexpr int UNITS = (CO / 64) * (CI / 64);
    static_assert(UNITS == 1 || UNITS == 4, "64x64, 256x64 or 64x256");
    __shared__ float s_sum[UNITS == 1 ? 4 * 4096 : 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = lane & 31, h = lane >> 5;
    const int u = UNITS == 4 ? wave : 0;
    const int cob = (CO == 256 ? u : 0) * 64, cib = (CI == 256 ? u : 0) * 64;

    float zA[2] = {1.f, 1.f}, zB[2] = {0.f, 0.f}, zC[2] = {0.f, 0.f}, aA[2] = {1.f, 1.f}, aC[2] = {0.f, 0.f};
    if (PRO_DZ) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            zA[k] = a.dz_abc[cob + 2 * p + k];
            zB[k] = a.dz_abc[CO + cob + 2 * p + k];
            zC[k] = a.dz_abc[2 * CO + cob + 2 * p + k];
        }
    }
    if (PRO_ACT) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            aA[k] = a.act_abc[cib + 2 * p + k];
            aC[k] = a.act_abc[2 * CI + cib + 2 * p + k];
        }
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // pixel pairs of this block: a contiguous range; with one unit the waves interleave inside it
    const long long npairs = (a.R + 1) / 2;
    const long long per = (npairs + gridDim.x - 1) / gridDim.x;
    const long long s_lo = (long long)blockIdx.x * per;
    long long s_hi = s_lo + per;
    if (s_hi > npairs) s_hi = npairs;
    constexpr int U = 8;                                    // pixel pairs per batch; two batches in flight
    const int wstep = UNITS == 1 ? 4 : 1;
    const int woff = UNITS == 1 ? __builtin_amdgcn_readfirstlane(wave) : 0;
    float2 g[2][U], g2[2][U], x[2][U];
    // full batches: wave-uniform base + one 32-bit lane offset, so the 3*U loads of a batch share three
    // address registers (scalar base, immediate / scalar strides) instead of 3*U 64-bit lane addresses
    const int lz = h * CO + cob + 2 * p, lx = h * CI + cib + 2 * p;
    auto issue = [&](long long s0, float2 (&gg)[U], float2 (&gg2)[U], float2 (&xx)[U]) {
        const float *bz = a.dz + (size_t)(2 * s0) * CO, *bx = a.act + (size_t)(2 * s0) * CI;
        const float *bz2 = PRO_DZ ? a.dz2 + (size_t)(2 * s0) * CO : nullptr;
#pragma unroll
        for (int k = 0; k < U; ++k) {
            gg[k] = *reinterpret_cast<const float2 *>(bz + lz + k * 2 * CO);
            if (PRO_DZ) gg2[k] = *reinterpret_cast<const float2 *>(bz2 + lz + k * 2 * CO);
            xx[k] = *reinterpret_cast<const float2 *>(bx + lx + k * 2 * CI);
        }
    };
    auto mac = [&](float d0, float d1, float g0, float g1, float x0, float x1, bool valid) {
        if (PRO_DZ) {
            d0 = fmaf(zA[0], d0, fmaf(zB[0], g0, zC[0]));
            d1 = fmaf(zA[1], d1, fmaf(zB[1], g1, zC[1]));
        }
        if (PRO_ACT) {
            x0 = fmaf(aA[0], x0, aC[0]);
            x1 = fmaf(aA[1], x1, aC[1]);
            if (a.act_relu) { x0 = x0 > 0.f ? x0 : 0.f; x1 = x1 > 0.f ? x1 : 0.f; }
        }
        if (!valid) d0 = d1 = 0.f;
        acc[0][0] = mfma32(d0, x0, acc[0][0]);
        acc[0][1] = mfma32(d0, x1, acc[0][1]);
        acc[1][0] = mfma32(d1, x0, acc[1][0]);
        acc[1][1] = mfma32(d1, x1, acc[1][1]);
    };
    auto consume = [&](const float2 (&gg)[U], const float2 (&gg2)[U], const float2 (&xx)[U]) {
#pragma unroll
        for (int k = 0; k < U; ++k)
            mac(gg[k].x, gg[k].y, PRO_DZ ? gg2[k].x : 0.f, PRO_DZ ? gg2[k].y : 0.f, xx[k].x, xx[k].y, true);
    };
    const long long step = (long long)wstep * U;
    // batches [s0, s0+U) that lie fully inside the block's range AND inside the map (2*(s0+U) <= R)
    long long full_hi = s_hi;
    if (2 * full_hi > a.R) full_hi = a.R / 2;
    long long s0 = s_lo + (long long)woff * U;
    if (s0 + U <= full_hi) issue(s0, g[0], g2[0], x[0]);
    while (s0 + U <= full_hi) {                             // the next batch's loads fly under this one's MFMAs
        const bool more = s0 + step + U <= full_hi;
        if (more) issue(s0 + step, g[1], g2[1], x[1]);
        consume(g[0], g2[0], x[0]);
        s0 += step;
        if (!more) break;
        const bool more2 = s0 + step + U <= full_hi;
        if (more2) issue(s0 + step, g[0], g2[0], x[0]);
        consume(g[1], g2[1], x[1]);
        s0 += step;
        if (!more2) break;
    }
    // ragged tail of the range (at most one partial batch per wave): clamped, predicated loads
    for (; s0 < s_hi; s0 += step) {
        for (int k = 0; k < U; ++k) {
            const long long sp = s0 + k;
            long long row = 2 * sp + h;
            const bool valid = sp < s_hi && row < a.R;
            if (row >= a.R) row = a.R - 1;
            const float2 d = *reinterpret_cast<const float2 *>(a.dz + (size_t)row * CO + cob + 2 * p);
            float2 d2 = make_float2(0.f, 0.f);
            if (PRO_DZ) d2 = *reinterpret_cast<const float2 *>(a.dz2 + (size_t)row * CO + cob + 2 * p);
            const float2 xv = *reinterpret_cast<const float2 *>(a.act + (size_t)row * CI + cib + 2 * p);
            mac(d.x, d.y, d2.x, d2.y, xv.x, xv.y, valid);
        }
    }
    float *dst = a.ws + (size_t)blockIdx.x * CO * CI;
    if (UNITS == 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    s_sum[wave * 4096 + (2 * mfma32_row(r, lane) + i) * 64 + 2 * p + j] = acc[i][j][r];
        __syncthreads();
        for (int i = tid; i < 4096; i += 256)
            dst[i] = (s_sum[i] + s_sum[4096 + i]) + (s_sum[2 * 4096 + i] + s_sum[3 * 4096 + i]);
    } else {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    dst[(size_t)(cob + 2 * mfma32_row(r, lane) + i) * CI + cib + 2 * p + j] = acc[i][j][r];
    }
}

// block = 64 entries x 16 slices of the partial list; fp64, fixed order
__global__ __launch_bounds__(1024) void conv1x1_wgrad_reduce_kernel(const float *__restrict__ ws, int nparts,
                                                                    int n, float *__restrict__ dw)
{
    __shared__ double s_acc[16][64];
    const int tx = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + tx;
    double s = 0.0;
    if (i < n)
        for (int q0 = slice; q0 < nparts; q0 += 16 * 8) {        // eight partial rows in flight, added in the same order
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ws[(size_t)(q0 + 16 * u < nparts ? q0 + 16 * u : q0) * n + i];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (q0 + 16 * u < nparts) s += (double)v[u];
        }
    s_acc[slice][tx] = s;
    __syncthreads();
    if (slice == 0 && i < n) {
        double t = 0.0;
        for (int j = 0; j < 16; ++j) t += s_acc[j][tx];
        dw[i] = (float)t;
    }
}

template <int CIN, int COUT>
int launch_c11(const C11Args &a, int epi, bool has_add, bool mask_act, bool has_x2, int pro, int grid,
               hipStream_t st)
{
    // eight waves, one block per CU (room for the coalesced-operand transposer): grid = c11_grid(R), which is also the
    // number of statistics rows (cova_conv1x1_num_partials)
#define C11_LAUNCH(PRO, EPI, ADD, MA, X2)                                                            \
    do {                                                                                             \
        hipLaunchKernelGGL((conv1x1_kernel<CIN, COUT, PRO, EPI, ADD, MA, X2, false, 8>), dim3(grid), dim3(512), \
                           0, st, a);                                                                \
        return COVA_OK;                                                                              \
    } while (0)
    if (a.side) {           // materialising consumer: 256 -> 64 forward with the two-tensor prologue
        if constexpr (CIN == 256) {
            if (pro != 2 || epi > 1) return COVA_ERR_BAD_ARG;
            if (epi == 1)
                hipLaunchKernelGGL((conv1x1_kernel<CIN, COUT, 2, 1, false, false, false, false, 8, true>), dim3(grid),
                                   dim3(512), 0, st, a);
            else
                hipLaunchKernelGGL((conv1x1_kernel<CIN, COUT, 2, 0, false, false, false, false, 8, true>), dim3(grid),
                                   dim3(512), 0, st, a);
            return COVA_OK;
        }
        return COVA_ERR_BAD_ARG;
    }
    if (epi == 0) {
        if (pro == 0) C11_LAUNCH(0, 0, false, false, false);
        if (pro == 1) C11_LAUNCH(1, 0, false, false, false);
        C11_LAUNCH(2, 0, false, false, false);
    }
    if (epi == 1) {
        if (pro == 0) C11_LAUNCH(0, 1, false, false, false);
        if (pro == 1) C11_LAUNCH(1, 1, false, false, false);
        C11_LAUNCH(2, 1, false, false, false);
    }
    // data-gradient epilogue: always with the two-input prologue (dz = A*dy + B*z + C)
    if (pro != 2) return COVA_ERR_BAD_ARG;
    if (epi == 3) {          // (acc + addend) * mask(act), no sums: the 64->256 block-input gradient
        if constexpr (COUT == 256) {
            if (!mask_act || has_x2) return COVA_ERR_BAD_ARG;
            if (has_add) C11_LAUNCH(2, 3, true, true, false);
            C11_LAUNCH(2, 3, false, true, false);
        }
        return COVA_ERR_BAD_ARG;
    }
    if (has_add) {
        if (mask_act) {
            if (has_x2) C11_LAUNCH(2, 2, true, true, true);
            C11_LAUNCH(2, 2, true, true, false);
        }
        if (has_x2) return COVA_ERR_BAD_ARG;
        C11_LAUNCH(2, 2, true, false, false);
    }
    if (has_x2) return COVA_ERR_BAD_ARG;
    if (mask_act) C11_LAUNCH(2, 2, false, true, false);
    C11_LAUNCH(2, 2, false, false, false);
#undef C11_LAUNCH
}

inline int c11_grid(long long R)            // 32-row tiles, eight waves per block, one block per CU
{
    const long long ntiles = (R + 31) / 32, nb = (ntiles + 7) / 8;
    return cova_internal_persistent_grid2(nb > (1 << 30) ? (1 << 30) : (int)nb, 1);
}

}  // namespace

// ====================================================================================
// C ABI
// ====================================================================================
COVA_API int cova_conv1x1_num_partials(long long R, int Cin, int Cout)
{
    (void)Cin; (void)Cout;
    return c11_grid(R);
}

// nn.Conv2d(Cin, Cout, 1, bias=False) on NHWC rows (torchvision Bottleneck conv1 / conv3 / downsample[0]
// behind models.py:49-51 for the resnet50 extension), forward or -- with w_trans -- data gradient.
// (Cin, Cout) in {(64,64), (64,256), (256,64)}.  Arguments as cova_conv3x3_wino_pro; stat_part rows =
// cova_conv1x1_num_partials, [2][Cout] each: (sum y, sum y^2) when z == NULL, else (sum dy, sum dy*xhat);
// stat_part2 (with z2/mean2/invstd2): (sum dy, sum dy*xhat2) of a second BatchNorm fed by the same dy.
COVA_API int cova_conv1x1(const float *in, const float *in2, const float *pro_abc, int pro_relu,
                          const float *w, int w_trans, const float *addend, const float *act,
                          const unsigned *act_bits, const float *mask_scale, const float *mask_shift, const float *z,
                          const float *mean, const float *invstd, const float *z2, const float *mean2,
                          const float *invstd2, float *out, float *stat_part, float *stat_part2,
                          long long R, int Cin, int Cout, void *stream)
{
    COVA_REQUIRE(in && w && out && R > 0);
    COVA_REQUIRE(!in2 || pro_abc);
    COVA_REQUIRE(!act_bits || !z);           // the bit form only feeds the sum-free data gradient
    const int pro = !pro_abc ? 0 : (in2 ? 2 : 1);
    int epi = 0;
    if (z) {
        COVA_REQUIRE(mean && invstd && stat_part && (act || (mask_scale && mask_shift)));
        COVA_REQUIRE(!z2 || (mean2 && invstd2 && stat_part2));
        epi = 2;
    } else if (act || act_bits) {        // masked data gradient whose BatchNorm sums are taken elsewhere (conv1x1_lin.hip)
        COVA_REQUIRE(!z2 && !stat_part && Cout == 256 && !(act && act_bits));
        epi = 3;
    } else {
        COVA_REQUIRE(!addend && !z2);
        epi = stat_part ? 1 : 0;
    }
    const C11Args a{in, in2, pro_abc, w, addend, act, mask_scale, mask_shift, z, mean, invstd, z2, mean2,
                    invstd2, out, stat_part, stat_part2, R, pro_relu, w_trans,
                    nullptr, nullptr, nullptr, nullptr, 0, nullptr, nullptr, act_bits};
    const bool masked = act != nullptr || act_bits != nullptr;
    const int grid = c11_grid(R);
    hipStream_t st = (hipStream_t)stream;
    int rc = COVA_ERR_BAD_ARG;
    if (Cin == 64 && Cout == 64) rc = launch_c11<64, 64>(a, epi, addend != nullptr, masked, z2 != nullptr, pro, grid, st);
    else if (Cin == 64 && Cout == 256) rc = launch_c11<64, 256>(a, epi, addend != nullptr, masked, z2 != nullptr, pro, grid, st);
    else if (Cin == 256 && Cout == 64) rc = launch_c11<256, 64>(a, epi, addend != nullptr, masked, z2 != nullptr, pro, grid, st);
    if (rc != COVA_OK) return rc;
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

// cova_conv1x1 (Cin = 256, Cout = 64, forward) whose input is the previous Bottleneck's output
// relu(A*in + B*in2 + C) = relu(bn3(z3) + identity-or-downsample branch), formed on load AND written to `side`
// [R,256]: the block output is materialised by its first consumer instead of by an element-wise pass
// (cova_bn_act_fwd / cova_bn_act2_fwd: one read of the 256-channel map less).
// side_bits (nullable) [R][8] words: the map's ReLU decisions, bit (c & 31) of word c >> 5 = (side[r][c] > 0) -- what
// cova_conv1x1's `act_bits` reads.
COVA_API int cova_conv1x1_materialize(const float *in, const float *in2, const float *pro_abc, const float *w,
                                      float *side, unsigned *side_bits, float *out, float *stat_part, long long R,
                                      void *stream)
{
    COVA_REQUIRE(in && in2 && pro_abc && w && side && out && R > 0);
    const C11Args a{in, in2, pro_abc, w, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr,
                    nullptr, nullptr, out, stat_part, nullptr, R, 1, 0, nullptr, nullptr, nullptr, nullptr, 0, side,
                    side_bits, nullptr};
    const int rc = launch_c11<256, 64>(a, stat_part ? 1 : 0, false, false, false, 2, c11_grid(R), (hipStream_t)stream);
    if (rc != COVA_OK) return rc;
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

// Data gradient of (1x1 conv 64->256, train-mode BatchNorm) in its LINEAR form (conv1x1_lin.hip): with
// z = a*W^T the BatchNorm-backward apply dz = A*v + B*z + C never has to be formed,
//   dz*W = (A.v)*W + a*(W^T diag(B) W) + C^T W = (A.v)*W + a*M + cvec,
// so the 256-channel z is not read: K = 256 channels of v (scaled by `avec` = A | 0 | 0 on load) plus 64
// channels of a = relu?(act_abc[0]*act + act_abc[2]).  Epilogue as cova_conv1x1's data gradient:
// (+ addend) * [fma(mask_scale, z, mask_shift) > 0], (sum dy, sum dy*xhat(z)) partials for the 64-channel
// BatchNorm in front.  w [256][64] (the conv's OIHW weight), m [64][64], cvec [64] from cova_conv1x1_lin_finish.
COVA_API int cova_conv1x1_lin_dgrad_num_partials(long long R)
{
    const long long ntiles = (R + 31) / 32, nb = (ntiles + 7) / 8;
    return cova_internal_persistent_grid2(nb > (1 << 30) ? (1 << 30) : (int)nb, 1);
}

COVA_API int cova_conv1x1_lin_dgrad(const float *v, const float *avec, const float *w, const float *act,
                                    const float *act_abc, int act_relu, const float *m, const float *cvec,
                                    const float *addend, const float *mask_scale, const float *mask_shift,
                                    const float *z, const float *mean, const float *invstd, float *out,
                                    float *stat_part, long long R, void *stream)
{
    COVA_REQUIRE(v && avec && w && act && act_abc && m && cvec && out && R > 0);
    COVA_REQUIRE(mask_scale && mask_shift && z && mean && invstd && stat_part);
    const C11Args a{v, nullptr, avec, w, addend, nullptr, mask_scale, mask_shift, z, mean, invstd, nullptr,
                    nullptr, nullptr, out, stat_part, nullptr, R, 0, 1, act, act_abc, m, cvec, act_relu, nullptr, nullptr,
                    nullptr};
    const int grid = cova_conv1x1_lin_dgrad_num_partials(R);
    hipStream_t st = (hipStream_t)stream;
    if (addend)
        hipLaunchKernelGGL((conv1x1_kernel<256, 64, 1, 2, true, false, false, true, 8>), dim3(grid), dim3(512), 0, st, a);
    else
        hipLaunchKernelGGL((conv1x1_kernel<256, 64, 1, 2, false, false, false, true, 8>), dim3(grid), dim3(512), 0, st, a);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

COVA_API int cova_conv1x1_wgrad_workspace_floats(long long R, int Co, int Ci)
{
    (void)R;
    return cova_internal_persistent_grid2(1 << 30, 2) * Co * Ci;
}

// Weight gradient of the same convolution: dw [Co][Ci] (= OIHW [Co,Ci,1,1]) = sum_r fz(dz)[r,co] * fa(act)[r,ci]
// with fz = dz_abc[0]*dz + dz_abc[1]*dz2 + dz_abc[2] (dz_abc nullable: plain dz) and
// fa = relu?(act_abc[0]*act + act_abc[2]) (act_abc nullable).  (Co, Ci) in {(64,64), (256,64), (64,256)}.
COVA_API int cova_conv1x1_wgrad(const float *dz, const float *dz2, const float *dz_abc, const float *act,
                                const float *act_abc, int act_relu, float *dw, float *ws, long long R,
                                int Co, int Ci, void *stream)
{
    COVA_REQUIRE(dz && act && dw && ws && R > 0);
    COVA_REQUIRE(!dz_abc || dz2);
    const W11Args a{dz, dz2, dz_abc, act, act_abc, ws, R, act_relu};
    const long long npairs = (R + 1) / 2, want = (npairs + 63) / 64;       // >= 64 pairs per block
    int grid = cova_internal_persistent_grid2(want > (1 << 30) ? (1 << 30) : (int)want, 2);
    hipStream_t st = (hipStream_t)stream;
    const bool pz = dz_abc != nullptr, pa = act_abc != nullptr;
#define W11_LAUNCH(CO, CI)                                                                                  \
    do {                                                                                                    \
        if (pz && pa) hipLaunchKernelGGL((conv1x1_wgrad_kernel<CO, CI, true, true>), dim3(grid), dim3(256), 0, st, a);   \
        else if (pz) hipLaunchKernelGGL((conv1x1_wgrad_kernel<CO, CI, true, false>), dim3(grid), dim3(256), 0, st, a);   \
        else if (pa) hipLaunchKernelGGL((conv1x1_wgrad_kernel<CO, CI, false, true>), dim3(grid), dim3(256), 0, st, a);   \
        else hipLaunchKernelGGL((conv1x1_wgrad_kernel<CO, CI, false, false>), dim3(grid), dim3(256), 0, st, a);          \
    } while (0)
    if (Co == 64 && Ci == 64) W11_LAUNCH(64, 64);
    else if (Co == 256 && Ci == 64) W11_LAUNCH(256, 64);
    else if (Co == 64 && Ci == 256) W11_LAUNCH(64, 256);
    else return COVA_ERR_BAD_ARG;
#undef W11_LAUNCH
    COVA_LAUNCH_CHECK();
    hipLaunchKernelGGL(conv1x1_wgrad_reduce_kernel, dim3(cdiv(Co * Ci, 64)), dim3(1024), 0, st, ws, grid,
                       Co * Ci, dw);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}
