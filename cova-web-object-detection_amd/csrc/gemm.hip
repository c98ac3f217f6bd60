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
// Measured and not kept (rounds 5: DESIGN.md 12.6 / 12.9, profiles/r05_gemm_bench.txt; the code left the library in round 6): the
// product on the bf16 matrix pipe with three-piece operands (slower: these launches are bound by the k-tile round trip -- fetch,
// barrier, stash, barrier -- and splitting a tile on its way into LDS lengthens exactly that), a register-direct form without LDS
// tiles, operand tiles two k-tiles ahead, two LDS buffers per k-group.
#include "common.h"

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
template <bool TA, bool TB, int KS>
__global__ __launch_bounds__(256 * KS) void sgemm_kernel(const float *__restrict__ A, int lda,
                                                         const float *__restrict__ Bm, int ldb,
                                                         float *__restrict__ C, int ldc,
                                                         const float *__restrict__ bias, int M, int N,
                                                         int K, int accumulate, int vecA, int vecB,
                                                         const uint8_t *__restrict__ emask, float einv)
{
    __shared__ __attribute__((aligned(16))) float As_[KS][BK][LDT];
    __shared__ __attribute__((aligned(16))) float Bs_[KS][BK][LDT];
    const int grp = KS == 2 ? (int)(threadIdx.x >> 8) : 0;       // k-group of this wave
    float (*As)[LDT] = As_[grp];
    float (*Bs)[LDT] = Bs_[grp];
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

inline int vec_ok(const float *p, int ld) { return (((uintptr_t)p & 15) == 0 && (ld & 3) == 0) ? 1 : 0; }

}  // namespace

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
#define SGEMM_LAUNCH(TA_, TB_)                                                                                              \
    do {                                                                                                                    \
        if (split) hipLaunchKernelGGL((sgemm_kernel<TA_, TB_, 2>), grid, dim3(512), 0, st, A, lda, B, ldb, C, ldc, bias, M, N, K, accumulate, va, vb, emask, einv); \
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
