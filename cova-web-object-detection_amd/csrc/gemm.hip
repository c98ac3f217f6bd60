// fp32 GEMM on v_mfma_f32_32x32x2_f32 for the dense layers of the hot path:
//   Wh = h [W_i;W_j]^T (models.py:188,193), decoder Linear(T,T) (models.py:85) and their
//   backward products (dX = dY W, dW = dY^T X).
// Exact f32 arithmetic (the MFMA is an fmaf chain), so results differ from the reference's
// CPU GEMM only by summation order.  Row-major operands with leading dimensions; any M, N, K.
//   C[M,N] (+)= op(A)[M,K] * op(B)[K,N] (+ bias[N])
//   transA = 0: A is [M,K] (lda >= K);  transA = 1: A is stored [K,M] (lda >= M)
//   transB = 0: B is [K,N] (ldb >= N);  transB = 1: B is stored [N,K] (ldb >= K)
// Block tile 64x64, BK = 64, 4 waves (2x2), each wave one 32x32 accumulator.  LDS tiles are
// k-major ([k][m]) so that MFMA operand reads are conflict-free ds_read_b32 across lanes; the
// next k-tile is fetched into registers (4 x float4 per operand per thread) while the current one
// feeds 32 MFMAs per wave (these GEMMs are a few GFLOP each: the loop is latency bound, so fewer, longer
// k-tiles -- half the barriers and round trips of BK = 32 -- matter more than occupancy).  ~3 GFLOP per call on this path: latency bound, not roofline relevant.
//
// Round 5, measured and NOT the default: sgemm_bf_kernel runs the product on the bf16 matrix pipe -- every f32 operand as three
// round-to-nearest bf16 pieces, the six products of order <= 2 accumulated in f32 (bf3.h; DESIGN.md sections 11.8 / 12.6): same
// error class as the f32 chain, 2.67x less matrix time -- and is SLOWER on this path's GEMMs (configs[1]: step 9.65 against
// 9.52 ms alternated in one process; configs[2]: 0.85 / 1.00 / 1.11 ms against 0.77 / 0.70 / 0.67 ms for the three layouts):
// these launches are bound by the k-tile round trip (fetch, barrier, stash, barrier), not by the matrix pipe, and splitting
// the tile on its way into LDS (88 vector operations per thread and tile, cross-lane pairing for the row-contiguous
// layouts, 4-way conflicting dword stores) lengthens exactly that.  Kept behind cova_set_option(11, 0) for A/B.  Same tiling (64 x 64 x 64, four waves 2 x 2, two k-groups per block); every element is split ONCE, by the thread
// that fetched it, on its way into LDS: tiles are [piece 3][row 64][32 packed k-pairs + 4 pad] dwords, so an operand of
// v_mfma_f32_32x32x16_bf16 (8 consecutive k of a row) is one ds_read_b128 per piece (rows of 144 B: conflict-free).
#include "bf3.h"

namespace {

constexpr int BM = 64, BN = 64, BK = 64, LDT = 68;
constexpr int EPT = BK * 64 / 256;      // elements of an operand tile per thread (16)

// Fetch this thread's EPT elements of a (rows x BK) operand tile.
//  k-contiguous storage (X[row][k]):   thread -> row = tid & 63, k = (tid >> 6) * EPT .. +EPT-1
//  row-contiguous storage (X[k][row]): thread -> k = tid >> 2,   row = (tid & 3) * EPT .. +EPT-1
template <bool KCONTIG>
__device__ __forceinline__ void fetch_tile(const float *__restrict__ X, int ld, int row0, int nrows,
                                           int k0, int K, int tid, bool vec_ok, float (&v)[EPT])
{
    if (KCONTIG) {
        const int r = row0 + (tid & 63), k = k0 + (tid >> 6) * EPT;
        const float *p = X + (size_t)r * ld + k;
        if (vec_ok && r < nrows && k + EPT - 1 < K) {
#pragma unroll
            for (int q = 0; q < EPT / 4; ++q) {
                const float4 a = *reinterpret_cast<const float4 *>(p + 4 * q);
                v[4 * q] = a.x; v[4 * q + 1] = a.y; v[4 * q + 2] = a.z; v[4 * q + 3] = a.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < EPT; ++j) v[j] = (r < nrows && k + j < K) ? p[j] : 0.f;
        }
    } else {
        const int k = k0 + (tid >> 2), r = row0 + (tid & 3) * EPT;
        const float *p = X + (size_t)k * ld + r;
        if (vec_ok && k < K && r + EPT - 1 < nrows) {
#pragma unroll
            for (int q = 0; q < EPT / 4; ++q) {
                const float4 a = *reinterpret_cast<const float4 *>(p + 4 * q);
                v[4 * q] = a.x; v[4 * q + 1] = a.y; v[4 * q + 2] = a.z; v[4 * q + 3] = a.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < EPT; ++j) v[j] = (k < K && r + j < nrows) ? p[j] : 0.f;
        }
    }
}

template <bool KCONTIG>
__device__ __forceinline__ void stash_tile(float (*S)[LDT], int tid, const float (&v)[EPT])
{
    if (KCONTIG) {
        const int r = tid & 63, k = (tid >> 6) * EPT;
#pragma unroll
        for (int j = 0; j < EPT; ++j) S[k + j][r] = v[j];
    } else {
        const int k = tid >> 2, r = (tid & 3) * EPT;
#pragma unroll
        for (int q = 0; q < EPT / 4; ++q)
            *reinterpret_cast<float4 *>(&S[k][r + 4 * q]) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
    }
}

// KS = 2: the block is TWO groups of four waves, each with its own LDS tiles, that take alternate k-tiles of the same
// 64x64 output tile and add their accumulators up at the end (through LDS, fixed order).  These GEMMs leave ~1.4 waves
// per SIMD and every k-tile is a dependent chain (barrier, LDS round trip, 32 MFMAs on one accumulator): two chains per
// output tile halve the serial length and give every SIMD a second wave to switch to.
// PF2: the operand tiles are fetched TWO k-tiles ahead (two register sets, the loop body twice per trip): a tile's loads
// have two MFMA phases to land instead of one (cova_set_option(19, 1))
// MODE 2 (cova_set_option(19, 2)): TWO LDS buffers per k-group -- the next tile is stashed into the other buffer in the middle of
// the current tile's MFMAs and ONE barrier per k-tile is left (the single-buffer loop: barrier, stash, barrier, MFMAs)
template <bool TA, bool TB, int KS, int MODE = 0>
__global__ __launch_bounds__(256 * KS) void sgemm_kernel(const float *__restrict__ A, int lda,
                                                         const float *__restrict__ Bm, int ldb,
                                                         float *__restrict__ C, int ldc,
                                                         const float *__restrict__ bias, int M, int N,
                                                         int K, int accumulate, int vecA, int vecB,
                                                         const uint8_t *__restrict__ emask, float einv)
{
    constexpr bool PF2 = MODE == 1, DB = MODE == 2;
    constexpr int NBUF = DB ? 2 : 1;
    __shared__ __attribute__((aligned(16))) float As_[KS * NBUF][BK][LDT];
    __shared__ __attribute__((aligned(16))) float Bs_[KS * NBUF][BK][LDT];
    const int grp = KS == 2 ? (int)(threadIdx.x >> 8) : 0;       // k-group of this wave
    float (*As)[LDT] = As_[grp * NBUF];
    float (*Bs)[LDT] = Bs_[grp * NBUF];
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int li = lane & 31, kh2 = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    // A tile: rows = m; stored k-contiguous unless transposed.  B tile: rows = n; stored
    // k-contiguous when transB (B is [N,K]).
    // (both groups run the same number of iterations -- a past-the-end k-tile fetches zeros -- so the block barriers match)
    constexpr int KSTEP = BK * KS;
    float va[EPT], vb[EPT];
    fetch_tile<!TA>(A, lda, m0, M, grp * BK, K, tid, vecA != 0, va);
    fetch_tile<TB>(Bm, ldb, n0, N, grp * BK, K, tid, vecB != 0, vb);
    if (DB) {
        // (every condition below is the same for both k-groups: k0 - grp * BK does not depend on the group)
        stash_tile<!TA>(As_[grp * NBUF], tid, va);
        stash_tile<TB>(Bs_[grp * NBUF], tid, vb);
        __syncthreads();
        if (KSTEP < K) {
            fetch_tile<!TA>(A, lda, m0, M, grp * BK + KSTEP, K, tid, vecA != 0, va);
            fetch_tile<TB>(Bm, ldb, n0, N, grp * BK + KSTEP, K, tid, vecB != 0, vb);
        }
        int cur = 0;
        for (int k0 = grp * BK; k0 - grp * BK < K; k0 += KSTEP, cur ^= 1) {
            float (*Ac)[LDT] = As_[grp * NBUF + cur];
            float (*Bc)[LDT] = Bs_[grp * NBUF + cur];
            const bool more = k0 + KSTEP - grp * BK < K;
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) acc = mfma32(Ac[2 * kk + kh2][wm * 32 + li], Bc[2 * kk + kh2][wn * 32 + li], acc);
            if (more) {                         // the other buffer was last read before the barrier that ended the previous tile
                stash_tile<!TA>(As_[grp * NBUF + (cur ^ 1)], tid, va);
                stash_tile<TB>(Bs_[grp * NBUF + (cur ^ 1)], tid, vb);
                if (k0 + 2 * KSTEP - grp * BK < K) {
                    fetch_tile<!TA>(A, lda, m0, M, k0 + 2 * KSTEP, K, tid, vecA != 0, va);
                    fetch_tile<TB>(Bm, ldb, n0, N, k0 + 2 * KSTEP, K, tid, vecB != 0, vb);
                }
            }
#pragma unroll
            for (int kk = BK / 4; kk < BK / 2; ++kk) acc = mfma32(Ac[2 * kk + kh2][wm * 32 + li], Bc[2 * kk + kh2][wn * 32 + li], acc);
            __syncthreads();                    // this tile fully consumed, the next one fully stashed
        }
    } else if (PF2) {
        float va1[EPT], vb1[EPT];
        fetch_tile<!TA>(A, lda, m0, M, grp * BK + KSTEP, K, tid, vecA != 0, va1);      // (past K: zeros, no loads)
        fetch_tile<TB>(Bm, ldb, n0, N, grp * BK + KSTEP, K, tid, vecB != 0, vb1);
        auto mma = [&]() {
#pragma unroll
            for (int kk = 0; kk < BK / 2; ++kk) {
                const float a = As[2 * kk + kh2][wm * 32 + li];
                const float b = Bs[2 * kk + kh2][wn * 32 + li];
                acc = mfma32(a, b, acc);
            }
        };
        // (every condition below is the same for both k-groups: k0 - grp * BK does not depend on the group)
        for (int k0 = grp * BK; k0 - grp * BK < K; k0 += 2 * KSTEP) {
            __syncthreads();
            stash_tile<!TA>(As, tid, va);
            stash_tile<TB>(Bs, tid, vb);
            __syncthreads();
            if (k0 + 2 * KSTEP - grp * BK < K) {
                fetch_tile<!TA>(A, lda, m0, M, k0 + 2 * KSTEP, K, tid, vecA != 0, va);
                fetch_tile<TB>(Bm, ldb, n0, N, k0 + 2 * KSTEP, K, tid, vecB != 0, vb);
            }
            mma();
            if (k0 + KSTEP - grp * BK < K) {
                __syncthreads();
                stash_tile<!TA>(As, tid, va1);
                stash_tile<TB>(Bs, tid, vb1);
                __syncthreads();
                if (k0 + 3 * KSTEP - grp * BK < K) {
                    fetch_tile<!TA>(A, lda, m0, M, k0 + 3 * KSTEP, K, tid, vecA != 0, va1);
                    fetch_tile<TB>(Bm, ldb, n0, N, k0 + 3 * KSTEP, K, tid, vecB != 0, vb1);
                }
                mma();
            }
        }
    } else
    for (int k0 = grp * BK; k0 - grp * BK < K; k0 += KSTEP) {
        __syncthreads();                    // previous tile fully consumed
        stash_tile<!TA>(As, tid, va);
        stash_tile<TB>(Bs, tid, vb);
        __syncthreads();
        if (k0 + KSTEP - grp * BK < K) {    // this group's next tile in flight while this one computes
            fetch_tile<!TA>(A, lda, m0, M, k0 + KSTEP, K, tid, vecA != 0, va);
            fetch_tile<TB>(Bm, ldb, n0, N, k0 + KSTEP, K, tid, vecB != 0, vb);
        }
        // (measured: requesting all 64 operands of the k-tile from LDS before the first MFMA -- instead of hipcc's read, read,
        // wait, two MFMAs -- is SLOWER, 43 against 38 us and 65 against 55 us per launch: the second wave of the SIMD
        // already covers the LDS latencies, and 64 more registers cost a resident block)
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const float a = As[2 * kk + kh2][wm * 32 + li];
            const float b = Bs[2 * kk + kh2][wn * 32 + li];
            acc = mfma32(a, b, acc);
        }
    }
    if (KS == 2) {                          // group 1 hands its accumulator over (its A tile area is free now)
        __syncthreads();
        float *red = &As_[0][0][0] + wave * (16 * 64);         // [wave][r][lane]: 4 x 4 KB <= one A tile
        if (grp == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[r * 64 + lane] = acc[r];
        }
        __syncthreads();
        if (grp == 1) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += red[r * 64 + lane];
    }
    const int gn = n0 + wn * 32 + li;
    if (gn < N) {
        const float bv = bias ? bias[gn] : 0.f;
        // what the epilogue reads (the old values under `accumulate`, the keep mask of the Dropout epilogue) is requested for
        // all 16 rows before the first is used: one round trip instead of sixteen
        float old[16];
        uint8_t keep[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gm = min(m0 + wm * 32 + mfma32_row(r, lane), M - 1);       // (clamped: a valid address; not stored)
            old[r] = accumulate ? C[(size_t)gm * ldc + gn] : 0.f;
            keep[r] = emask != nullptr ? emask[(size_t)gm * N + gn] : (uint8_t)1;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gm = m0 + wm * 32 + mfma32_row(r, lane);
            if (gm < M) {
                float v = acc[r] + bv;
                if (accumulate) v += old[r];
                if (emask != nullptr) v = keep[r] ? v * einv : 0.f;             // cova_dropout_bwd on the result
                C[(size_t)gm * ldc + gn] = v;
            }
        }
    }
}

constexpr int PROW = 36;                           // dwords per (piece, row) of a bf16 tile: 32 k-pairs + 4 pad (144 B)
constexpr int PTILE = 3 * 64 * PROW;               // one operand tile: 27,648 B

__device__ __forceinline__ f32x16 gemm_mfma_bf(u32x4 a, u32x4 b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// The fetched elements of a tile (fetch_tile's ownership) -> packed bf16 pieces in LDS.  `flip`: sign bits of both halves
// (0x80008000) for the operand of a block that multiplies with the negated matrix (see the kernel), else 0.
//  k-contiguous: the thread holds k = k0 .. k0+15 of row r: eight k-pairs, two 16-byte stores per piece.
//  row-contiguous: the thread holds rows r0 .. r0+15 of ONE k; its neighbour four lanes up holds k + 1 of the same rows.
//  The even-k thread pairs rows r0 .. r0+7 (its values below, the neighbour's above), the odd-k thread rows r0+8 .. r0+15.
template <bool KCONTIG>
__device__ __forceinline__ void stash_tile_bf(uint32_t *S, int tid, const float (&v)[EPT], uint32_t flip)
{
    if (KCONTIG) {
        const int r = tid & 63, kp = (tid >> 6) * (EPT / 2);
        u32x4 q[3][2];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            uint32_t a, b, c;
            bf3_split_pair(v[2 * i], v[2 * i + 1], a, b, c);
            q[0][i >> 2][i & 3] = a ^ flip; q[1][i >> 2][i & 3] = b ^ flip; q[2][i >> 2][i & 3] = c ^ flip;
        }
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
            u32x4 *d = reinterpret_cast<u32x4 *>(S + (pc * 64 + r) * PROW + kp);
            d[0] = q[pc][0];
            d[1] = q[pc][1];
        }
    } else {
        const int k = tid >> 2, r0 = (tid & 3) * EPT, odd = k & 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // even k: pair (own v[j], neighbour's v[j]); odd k: pair (neighbour's v[8 + j], own v[8 + j])
            const float mine = odd ? v[8 + j] : v[j];
            const float give = odd ? v[j] : v[8 + j];                  // what the neighbour pairs with its own
            const float got = __shfl_xor(give, 4, 64);
            uint32_t a, b, c;
            bf3_split_pair(odd ? got : mine, odd ? mine : got, a, b, c);
            uint32_t *d = S + (size_t)(r0 + 8 * odd + j) * PROW + (k >> 1);
            d[0] = a ^ flip;
            d[64 * PROW] = b ^ flip;
            d[2 * 64 * PROW] = c ^ flip;
        }
    }
}

// Same tiling and k-group scheme as sgemm_kernel.  The bf16 MFMA drops low product bits toward -infinity
// (tools/probe/mfma_round_probe.hip): blocks of odd (x + y) multiply with the NEGATED A tile and negate their sums back, so
// that the bias has no common sign over the output.
template <bool TA, bool TB, int KS>
__global__ __launch_bounds__(256 * KS) void sgemm_bf_kernel(const float *__restrict__ A, int lda,
                                                            const float *__restrict__ Bm, int ldb,
                                                            float *__restrict__ C, int ldc,
                                                            const float *__restrict__ bias, int M, int N,
                                                            int K, int accumulate, int vecA, int vecB)
{
    __shared__ __attribute__((aligned(16))) uint32_t As_[KS][PTILE];
    __shared__ __attribute__((aligned(16))) uint32_t Bs_[KS][PTILE];
    const int grp = KS == 2 ? (int)(threadIdx.x >> 8) : 0;       // k-group of this wave
    uint32_t *As = As_[grp], *Bs = Bs_[grp];
    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int li = lane & 31, kh2 = lane >> 5;
    const bool neg = ((blockIdx.x + blockIdx.y) & 1) != 0;
    const uint32_t flip = neg ? 0x80008000u : 0u;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    constexpr int KSTEP = BK * KS;
    float va[EPT], vb[EPT];
    fetch_tile<!TA>(A, lda, m0, M, grp * BK, K, tid, vecA != 0, va);
    fetch_tile<TB>(Bm, ldb, n0, N, grp * BK, K, tid, vecB != 0, vb);
    const u32x4 *pa = reinterpret_cast<const u32x4 *>(As + (wm * 32 + li) * PROW) + kh2;      // + 2 s: K-step s
    const u32x4 *pb = reinterpret_cast<const u32x4 *>(Bs + (wn * 32 + li) * PROW) + kh2;
    constexpr int PSTR = 64 * PROW / 4;                          // u32x4 per piece plane
    for (int k0 = grp * BK; k0 - grp * BK < K; k0 += KSTEP) {
        __syncthreads();                    // previous tile fully consumed
        stash_tile_bf<!TA>(As, tid, va, flip);
        stash_tile_bf<TB>(Bs, tid, vb, 0u);
        __syncthreads();
        if (k0 + KSTEP - grp * BK < K) {    // this group's next tile in flight while this one computes
            fetch_tile<!TA>(A, lda, m0, M, k0 + KSTEP, K, tid, vecA != 0, va);
            fetch_tile<TB>(Bm, ldb, n0, N, k0 + KSTEP, K, tid, vecB != 0, vb);
        }
#pragma unroll
        for (int s4 = 0; s4 < BK / 16; ++s4) {
            const u32x4 a0 = pa[2 * s4], a1 = pa[PSTR + 2 * s4], a2 = pa[2 * PSTR + 2 * s4];
            const u32x4 b0 = pb[2 * s4], b1 = pb[PSTR + 2 * s4], b2 = pb[2 * PSTR + 2 * s4];
            // the six products of order <= 2, smallest first
            acc = gemm_mfma_bf(a2, b0, acc);
            acc = gemm_mfma_bf(a0, b2, acc);
            acc = gemm_mfma_bf(a1, b1, acc);
            acc = gemm_mfma_bf(a1, b0, acc);
            acc = gemm_mfma_bf(a0, b1, acc);
            acc = gemm_mfma_bf(a0, b0, acc);
        }
    }
    if (neg) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = -acc[r];
    }
    if (KS == 2) {                          // group 1 hands its accumulator over (its A tile area is free now)
        __syncthreads();
        float *red = reinterpret_cast<float *>(&As_[0][0]) + wave * (16 * 64);         // [wave][r][lane]: 4 x 4 KB <= one A tile
        if (grp == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[r * 64 + lane] = acc[r];
        }
        __syncthreads();
        if (grp == 1) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += red[r * 64 + lane];
    }
    const int gn = n0 + wn * 32 + li;
    if (gn < N) {
        const float bv = bias ? bias[gn] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gm = m0 + wm * 32 + mfma32_row(r, lane);
            if (gm < M) {
                float v = acc[r] + bv;
                float *dst = C + (size_t)gm * ldc + gn;
                if (accumulate) v += *dst;
                *dst = v;
            }
        }
    }
}


// ---- register-direct form (cova_set_option(15, 1)) ----
// No LDS tiles, no barriers in the k loop: every wave streams the operands of its own 32 x 32 accumulator straight from
// L1 / L2 into registers, one 64-deep k-tile ahead of the MFMAs that use it (two register sets, ping-pong).  The K index of
// an MFMA slot is free as long as both operands agree: slot kk of lane half kh is k = 8 (kk >> 2) + 4 kh + (kk & 3) of the
// tile, so that a k-contiguous operand (X[row][k]) is eight float4 loads per lane and tile at constant offsets, and a
// row-contiguous one (X[k][row]) 32 coalesced dword loads (lanes 0..31 one row of 128 B, lanes 32..63 the row four below)
// through per-slot byte offsets kept in registers (uniform base + 32-bit lane offset: no address arithmetic in the loop).
// KS k-groups of four waves take alternate k-tiles of the same 64 x 64 output tile and meet through LDS at the end (fixed
// order).  Needs K % 4 == 0 and 16-byte aligned k-contiguous operands; everything else takes sgemm_kernel.
template <bool KC>
struct DirectOperand {
    const char *base;          // k-contiguous: this lane's row at k = 4 kh (lane pointer); row-contiguous: the matrix (uniform)
    uint32_t lane_off;         // row-contiguous: bytes of (row 4 kh, this lane's column)
    long long ld4;             // row-contiguous: bytes per k row (uniform)
};

__device__ __forceinline__ int direct_kmap(int kk) { return 8 * (kk >> 2) + (kk & 3); }      // (+ 4 kh)

template <bool KC>
__device__ __forceinline__ void direct_init(DirectOperand<KC> &o, const float *X, int ld, int row0, int nrows, int li, int kh)
{
    int r = row0 + li;
    if (r > nrows - 1) r = nrows - 1;                 // rows past the matrix: a valid address, the result is not stored
    if (KC) {
        o.base = reinterpret_cast<const char *>(X + (size_t)r * ld + 4 * kh);
        o.lane_off = 0u;
        o.ld4 = 0;
    } else {
        o.base = reinterpret_cast<const char *>(X);
        o.lane_off = (uint32_t)(((size_t)4 * kh * ld + r) * 4);
        o.ld4 = (long long)ld * 4;
    }
}

// tile t (k0 = 64 t) -> v[32]; slots at or past K are zero
template <bool KC>
__device__ __forceinline__ void direct_load(const DirectOperand<KC> &o, int t, int ntiles, int K, int kh, float (&v)[32])
{
    const int k0 = t * 64;                             // (t < ntiles: the caller's wave-uniform guard)
    const bool full = k0 + 64 <= K;                    // (wave-uniform) every tile but the last: no per-slot predicate
    if (KC) {
        const char *p = o.base + (long long)t * 256;
        if (full) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 q = *reinterpret_cast<const float4 *>(p + 32 * j);
                v[4 * j] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w;
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool ok = k0 + 8 * j + 4 * kh < K;   // (K % 4 == 0: a float4 is inside or outside)
            const float4 q = *reinterpret_cast<const float4 *>(ok ? p + 32 * j : o.base);
            v[4 * j] = ok ? q.x : 0.f; v[4 * j + 1] = ok ? q.y : 0.f; v[4 * j + 2] = ok ? q.z : 0.f; v[4 * j + 3] = ok ? q.w : 0.f;
        }
    } else {
        const char *p = o.base + (long long)k0 * o.ld4;           // uniform: the per-slot row offsets below stay scalar
        if (full) {
#pragma unroll
            for (int kk = 0; kk < 32; ++kk)
                v[kk] = *reinterpret_cast<const float *>(p + (long long)direct_kmap(kk) * o.ld4 + o.lane_off);
            return;
        }
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) {
            const bool ok = k0 + direct_kmap(kk) + 4 * kh < K;
            const char *pk = ok ? p + (long long)direct_kmap(kk) * o.ld4 : o.base;
            const float q = *reinterpret_cast<const float *>(pk + o.lane_off);
            v[kk] = ok ? q : 0.f;
        }
    }
}

template <bool TA, bool TB, int KS>
__global__ __launch_bounds__(256 * KS) void sgemm_direct_kernel(const float *__restrict__ A, int lda,
                                                                const float *__restrict__ Bm, int ldb,
                                                                float *__restrict__ C, int ldc,
                                                                const float *__restrict__ bias, int M, int N,
                                                                int K, int accumulate,
                                                                const uint8_t *__restrict__ emask, float einv)
{
    __shared__ float s_red[KS > 1 ? (KS - 1) * 4 * 16 * 64 : 1];
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3;
    const int grp = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM + wm * 32, n0 = blockIdx.x * BN + wn * 32;
    const int li = lane & 31, kh = lane >> 5;
    const int ntiles = (K + 63) / 64;
    DirectOperand<!TA> oa;
    DirectOperand<TB> ob;
    direct_init<!TA>(oa, A, lda, m0, M, li, kh);
    direct_init<TB>(ob, Bm, ldb, n0, N, li, kh);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float a0[32], b0[32], a1[32], b1[32];
    if (grp < ntiles) {
        direct_load<!TA>(oa, grp, ntiles, K, kh, a0);
        direct_load<TB>(ob, grp, ntiles, K, kh, b0);
    }
    for (int t = grp; t < ntiles; t += 2 * KS) {        // (every guard below is wave-uniform)
        const bool second = t + KS < ntiles;
        if (second) {
            direct_load<!TA>(oa, t + KS, ntiles, K, kh, a1);
            direct_load<TB>(ob, t + KS, ntiles, K, kh, b1);
        }
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) acc = mfma32(a0[kk], b0[kk], acc);
        if (t + 2 * KS < ntiles) {
            direct_load<!TA>(oa, t + 2 * KS, ntiles, K, kh, a0);
            direct_load<TB>(ob, t + 2 * KS, ntiles, K, kh, b0);
        }
        if (second) {
#pragma unroll
            for (int kk = 0; kk < 32; ++kk) acc = mfma32(a1[kk], b1[kk], acc);
        }
    }
    if (KS > 1) {                            // groups 1.. hand their accumulators to group 0 (fixed order)
        if (grp > 0) {
            float *red = s_red + ((grp - 1) * 4 + wave) * (16 * 64);
#pragma unroll
            for (int r = 0; r < 16; ++r) red[r * 64 + lane] = acc[r];
        }
        __syncthreads();
        if (grp > 0) return;
#pragma unroll
        for (int g = 1; g < KS; ++g) {
            const float *red = s_red + ((g - 1) * 4 + wave) * (16 * 64);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += red[r * 64 + lane];
        }
    }
    const int gn = n0 + li;
    if (gn < N) {
        const float bv = bias ? bias[gn] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gm = m0 + mfma32_row(r, lane);
            if (gm < M) {
                float v = acc[r] + bv;
                float *dst = C + (size_t)gm * ldc + gn;
                if (accumulate) v += *dst;
                if (emask != nullptr) v = emask[(size_t)gm * N + gn] ? v * einv : 0.f;
                *dst = v;
            }
        }
    }
}

int g_sgemm_direct = 0;
int g_sgemm_pf2 = 0;

int g_sgemm_f32 = 1;          // default: the f32-MFMA kernel (the bf16-split one measured slower, see the header)

inline int vec_ok(const float *p, int ld) { return (((uintptr_t)p & 15) == 0 && (ld & 3) == 0) ? 1 : 0; }

}  // namespace

int cova_internal_set_sgemm_f32(int v) { g_sgemm_f32 = v != 0; return COVA_OK; }
int cova_internal_set_sgemm_direct(int v) { g_sgemm_direct = v; return COVA_OK; }
int cova_internal_set_sgemm_pf2(int v) { g_sgemm_pf2 = v; return COVA_OK; }

static int sgemm_launch(int transA, int transB, int M, int N, int K, const float *A, int lda,
                        const float *B, int ldb, float *C, int ldc, const float *bias,
                        int accumulate, const uint8_t *emask, float einv, void *stream)
{
    COVA_REQUIRE(A && B && C && M >= 0 && N >= 0 && K >= 0);
    if (M == 0 || N == 0) return COVA_OK;
    const dim3 grid(cdiv(N, BN), cdiv(M, BM));
    hipStream_t st = (hipStream_t)stream;
    const int va = vec_ok(A, lda), vb = vec_ok(B, ldb);
    // two k-groups per block when there are at least four k-tiles and the grid alone does not fill the chip twice over
    const bool split = K >= 4 * BK && (long long)grid.x * grid.y < 2 * 4 * 256;
    // register-direct form: k-contiguous operands as float4 (K % 4 == 0, aligned rows); 32-bit lane offsets of the others
    const bool direct_ok = g_sgemm_direct != 0 && K % 4 == 0 && K >= 64 && (transA || va) && (!transB || vb) &&
                           (long long)(transA ? K : 1) * lda * 4 + (long long)M * 4 < (1ll << 31) &&
                           (long long)(transB ? 1 : K) * ldb * 4 + (long long)N * 4 < (1ll << 31);
    if (direct_ok) {
        const int ks = g_sgemm_direct == 2 ? 2 : g_sgemm_direct == 3 ? 1 : (split ? 2 : 1);       // (2 / 3: force two / one k-group)
#define SGEMM_DIRECT(TA_, TB_)                                                                                              \
        do {                                                                                                                \
            if (ks >= 2) hipLaunchKernelGGL((sgemm_direct_kernel<TA_, TB_, 2>), grid, dim3(512), 0, st, A, lda, B, ldb, C, ldc, bias, M, N, K, accumulate, emask, einv); \
            else hipLaunchKernelGGL((sgemm_direct_kernel<TA_, TB_, 1>), grid, dim3(256), 0, st, A, lda, B, ldb, C, ldc, bias, M, N, K, accumulate, emask, einv); \
        } while (0)
        if (!transA && !transB) SGEMM_DIRECT(false, false);
        else if (!transA && transB) SGEMM_DIRECT(false, true);
        else if (transA && !transB) SGEMM_DIRECT(true, false);
        else SGEMM_DIRECT(true, true);
#undef SGEMM_DIRECT
        COVA_LAUNCH_CHECK();
        return COVA_OK;
    }
#define SGEMM_LAUNCH(TA_, TB_)                                                                                              \
    do {                                                                                                                    \
        if (!g_sgemm_f32 && emask == nullptr) {                                                                             \
            if (split) hipLaunchKernelGGL((sgemm_bf_kernel<TA_, TB_, 2>), grid, dim3(512), 0, st, A, lda, B, ldb, C, ldc, bias, M, N, K, accumulate, va, vb); \
            else hipLaunchKernelGGL((sgemm_bf_kernel<TA_, TB_, 1>), grid, dim3(256), 0, st, A, lda, B, ldb, C, ldc, bias, M, N, K, accumulate, va, vb);      \
        } else if (split && g_sgemm_pf2 == 2) hipLaunchKernelGGL((sgemm_kernel<TA_, TB_, 2, 2>), grid, dim3(512), 0, st, A, lda, B, ldb, C, ldc, bias, M, N, K, accumulate, va, vb, emask, einv); \
        else if (split && g_sgemm_pf2 == 1) hipLaunchKernelGGL((sgemm_kernel<TA_, TB_, 2, 1>), grid, dim3(512), 0, st, A, lda, B, ldb, C, ldc, bias, M, N, K, accumulate, va, vb, emask, einv); \
        else if (split) hipLaunchKernelGGL((sgemm_kernel<TA_, TB_, 2>), grid, dim3(512), 0, st, A, lda, B, ldb, C, ldc, bias, M, N, K, accumulate, va, vb, emask, einv); \
        else hipLaunchKernelGGL((sgemm_kernel<TA_, TB_, 1>), grid, dim3(256), 0, st, A, lda, B, ldb, C, ldc, bias, M, N, K, accumulate, va, vb, emask, einv); \
    } while (0)
    if (!transA && !transB) SGEMM_LAUNCH(false, false);
    else if (!transA && transB) SGEMM_LAUNCH(false, true);
    else if (transA && !transB) SGEMM_LAUNCH(true, false);
    else SGEMM_LAUNCH(true, true);
#undef SGEMM_LAUNCH
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

COVA_API int cova_sgemm(int transA, int transB, int M, int N, int K, const float *A, int lda,
                        const float *B, int ldb, float *C, int ldc, const float *bias,
                        int accumulate, void *stream)
{
    return sgemm_launch(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, accumulate, nullptr, 1.f, stream);
}

// cova_sgemm followed by cova_dropout_bwd on its result, in the GEMM's epilogue: C = keep ? (op(A) op(B)) / (1 - p) : 0 with
// keep [M, N] uint8 (contiguous) -- the decoder's first Dropout backward (models.py:84) behind the data gradient of its
// Linear (models.py:85).  Same arithmetic per element as the two launches (bit-identical); always the f32-MFMA kernel.
COVA_API int cova_sgemm_dropout_bwd(int transA, int transB, int M, int N, int K, const float *A, int lda,
                                    const float *B, int ldb, float *C, int ldc, const uint8_t *keep, float p,
                                    void *stream)
{
    COVA_REQUIRE(keep && p >= 0.f && p < 1.f);
    return sgemm_launch(transA, transB, M, N, K, A, lda, B, ldb, C, ldc, nullptr, 0, keep, 1.f / (1.f - p), stream);
}
