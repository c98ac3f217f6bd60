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
#include <type_traits>

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


// ---------------------------------------------------------------------------------------------------------------------------
// The same product with the operand tiles brought in by global -> LDS copies (global_load_lds_dwordx4: no staging registers,
// no stash instructions, one LDS-only block barrier per k-tile) -- the form cova_sgemm takes whenever the operands allow 16-byte
// pieces (round 6; the six products of the head: 260 -> 197 us with 64 x 64 tiles, profiles/r06_gemm_bench.txt).
//   * block tile (32 WM) x 64, waves of 32 x 32 (v_mfma_f32_32x32x2_f32), k-tiles of 32 in a ring of DS stages per k-group; the
//     copies of the k-tile DS - 1 ahead are issued right behind the barrier that releases the slot they land in;
//   * KS = 2: two k-groups of waves per block take alternate k-tiles (each with its own ring) and add their accumulators at the
//     end through LDS, group 0 + group 1: for grids that leave CUs without a block of their own otherwise;
//   * a k-contiguous operand (X[row][k]) lands as 16-byte pieces in the order the MFMA's lanes read them with ds_read_b128:
//     copy piece c = (row block rb = c >> 1, k half kb = c & 1), lane = s * 16 + r -> row 16 rb + r, k = 16 kb + 4 s: a copy
//     instruction covers 16 rows x 64 bytes (16 cache lines), a read group of 16 lanes covers the 16 different 16-byte columns of
//     the LDS (conflict-free); the lane with k-half kh uses k = 16 kh + kk for MFMA kk -- the order of k inside a tile is free as
//     long as both operands agree;
//   * a row-contiguous operand (X[k][row]) lands as [k][rows of the tile] and is read with ds_read_b32 (lanes = consecutive rows);
//   * two accumulators per wave (even / odd kk, added at the end in a fixed order): a dependent chain of 32x32x2 MFMAs on one
//     accumulator issues every ~128 cycles, two chains keep the pipe busy from one wave;
//   * rows past M / N are clamped (their outputs are never stored); a k past K is clamped and both of its operands are zeroed
//     in registers (last k-tile only).
typedef __attribute__((address_space(3))) void gd_lds_void;

__device__ __forceinline__ void gd_copy16(const char *base, unsigned off_bytes, unsigned lds_base_bytes)
{
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_base_bytes), "v"(off_bytes), "s"(base) : "memory", "m0");
}

template <int N_>
__device__ __forceinline__ void gd_wait_vm()
{
    static_assert(N_ >= 0 && N_ < 64, "vmcnt");
    if constexpr (N_ == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N_ == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N_ == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N_ == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N_ == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else static_assert(N_ == 0, "add the count");
}

template <bool TA, bool TB, int WM, int KS, int DS>
__global__ __launch_bounds__(128 * WM * KS) void sgemm_dma_kernel(const float *__restrict__ A, int lda, const float *__restrict__ Bm,
                                                                  int ldb, float *__restrict__ C, int ldc,
                                                                  const float *__restrict__ bias, int M, int N, int K, int accumulate,
                                                                  const uint8_t *__restrict__ emask, float einv)
{
    constexpr int KT = 32;                               // k-tile
    constexpr int RA = 32 * WM, RB = 64;                 // rows of the A / B tile of a stage
    constexpr int NW = 2 * WM;                           // waves of a k-group
    constexpr int NA = (RA / 8) / NW, NB = (RB / 8) / NW;       // this wave's 1 KB copies per stage (a tile of R rows = R / 8 pieces)
    constexpr int A_FLOATS = RA * KT, STAGE_FLOATS = (RA + RB) * KT;
    constexpr bool AK = !TA, BKC = TB;                   // operand stored k-contiguous?
    __shared__ __attribute__((aligned(1024))) float s_t[KS][DS][STAGE_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = KS == 2 ? wave_all / NW : 0, wave = wave_all % NW;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * RA, n0 = blockIdx.x * RB;
    const int li = lane & 31, kh = lane >> 5;
    const unsigned lds0 = (unsigned)(size_t)(gd_lds_void *)&s_t[grp][0][0];

    // ---- copy pieces.  k-contiguous tile: piece c = (rb = c >> 1, kb = c & 1), lane = s * 16 + r -> row 16 rb + r, k = 16 kb + 4 s.
    // Row-contiguous tile of R rows: R / 4 lanes per k-row, 256 / R k-rows per piece: k = c * (256 / R) + lane / (R / 4), rows 4 (lane % (R / 4)) ...
    int a_row[NA], b_row[NB];                            // (clamped to the operand; fixed over the k-tiles)
#pragma unroll
    for (int i = 0; i < NA; ++i)
        a_row[i] = AK ? min(m0 + 16 * ((wave + i * NW) >> 1) + (lane & 15), M - 1) : min(m0 + 4 * (lane % (RA / 4)), M - 4);
#pragma unroll
    for (int i = 0; i < NB; ++i)
        b_row[i] = BKC ? min(n0 + 16 * ((wave + i * NW) >> 1) + (lane & 15), N - 1) : min(n0 + 4 * (lane % (RB / 4)), N - 4);
    auto issue = [&](int t, int slot) __attribute__((always_inline)) {
        const int k0 = t * KT;
        const unsigned dst = lds0 + (unsigned)slot * (unsigned)(STAGE_FLOATS * 4);
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int c = wave + i * NW;
            unsigned o;
            if (AK) o = (unsigned)(a_row[i] * lda + min(k0 + 16 * (c & 1) + 4 * (lane >> 4), K - 4)) * 4u;
            else o = (unsigned)(min(k0 + c * (256 / RA) + lane / (RA / 4), K - 1) * lda + a_row[i]) * 4u;
            gd_copy16(reinterpret_cast<const char *>(A), o, dst + (unsigned)c * 1024u);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int c = wave + i * NW;
            unsigned o;
            if (BKC) o = (unsigned)(b_row[i] * ldb + min(k0 + 16 * (c & 1) + 4 * (lane >> 4), K - 4)) * 4u;
            else o = (unsigned)(min(k0 + c * (256 / RB) + lane / (RB / 4), K - 1) * ldb + b_row[i]) * 4u;
            gd_copy16(reinterpret_cast<const char *>(Bm), o, dst + (unsigned)(A_FLOATS * 4) + (unsigned)c * 1024u);
        }
    };

    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }

    // operand addresses of this lane inside a stage (floats)
    const int a_rd = AK ? ((wm * 2 + (li >> 4)) * 2 + kh) * 256 + (li & 15) * 4 : (kh * 16) * RA + wm * 32 + li;
    const int b_rd = A_FLOATS + (BKC ? ((wn * 2 + (li >> 4)) * 2 + kh) * 256 + (li & 15) * 4 : (kh * 16) * RB + wn * 32 + li);

    const int ntiles = (K + KT - 1) / KT;
    const int nit = (ntiles + KS - 1) / KS;              // k-tiles grp, grp + KS, ... : the same count for every group (the barriers match)
#pragma unroll
    for (int it = 0; it < DS - 1; ++it) issue(grp + KS * it, it);
#pragma unroll 1
    for (int it = 0; it < nit; ++it) {
        const int t = grp + KS * it;
        gd_wait_vm<(NA + NB) * (DS - 2)>();             // this wave's copies of k-tile t have landed (the DS - 2 younger groups may be in flight) ...
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // ... and everybody's; the slot of the k-tile before has been read by all
                                                       // (LDS-only barrier: __syncthreads() would wait for the copies in flight)
        issue(grp + KS * (it + DS - 1), (it + DS - 1) % DS);                 // (past the end: clamped addresses, never read)
        const float *st = &s_t[grp][it % DS][0];
        const float *ta = st + a_rd, *tb = st + b_rd;
        float av[16], bv[16];
        if (AK) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4 *>(ta + 64 * q);
                av[4 * q] = v.x; av[4 * q + 1] = v.y; av[4 * q + 2] = v.z; av[4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) av[kk] = ta[RA * kk];
        }
        if (BKC) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4 *>(tb + 64 * q);
                bv[4 * q] = v.x; bv[4 * q + 1] = v.y; bv[4 * q + 2] = v.z; bv[4 * q + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) bv[kk] = tb[RB * kk];
        }
        const int kv = K - t * KT - 16 * kh;            // this lane's k = 16 kh + kk is inside K for kk < kv
        if (kv < 16) {                                 // (last k-tile of a K that is not a multiple of 32; a k-tile past the end)
#pragma unroll
            for (int kk = 0; kk < 16; ++kk)
                if (kk >= kv) { av[kk] = 0.f; bv[kk] = 0.f; }
        }
#pragma unroll
        for (int kk = 0; kk < 16; kk += 2) {
            acc0 = mfma32(av[kk], bv[kk], acc0);
            acc1 = mfma32(av[kk + 1], bv[kk + 1], acc1);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (the copies past the end have landed before the LDS is reused / released)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] += acc1[r];
    if (KS == 2) {                                     // group 1 hands its sum over through its (idle) ring: [wave][r][lane]
        __syncthreads();
        float *red = &s_t[KS - 1][0][0] + wave * (16 * 64);
        if (grp == 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[r * 64 + lane] = acc0[r];
        }
        __syncthreads();
        if (grp == 1) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] += red[r * 64 + lane];
    }
    const int gn = n0 + wn * 32 + li;
    if (gn >= N) return;
    const float bv = bias ? bias[gn] : 0.f;
    // three epilogues, chosen block-uniformly: plain, += C, Dropout backward -- each with its 16 operands requested by unconditional
    // loads in front of the first use (a load behind a branch costs a vmcnt(0) at the join: sixteen round trips instead of one)
    auto epilogue = [&](auto mode_t) __attribute__((always_inline)) {
        constexpr int MODE = decltype(mode_t)::value;
        float old[16];
        uint8_t keep[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gm = min(m0 + wm * 32 + mfma32_row(r, lane), M - 1);       // (clamped: a valid address; not stored)
            if (MODE == 1) old[r] = C[(size_t)gm * ldc + gn];
            if (MODE == 2) keep[r] = emask[(size_t)gm * N + gn];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int gm = m0 + wm * 32 + mfma32_row(r, lane);
            if (gm < M) {
                float v = acc0[r] + bv;
                if (MODE == 1) v += old[r];
                if (MODE == 2) v = keep[r] ? v * einv : 0.f;                      // cova_dropout_bwd on the result
                C[(size_t)gm * ldc + gn] = v;
            }
        }
    };
    if (emask != nullptr) epilogue(std::integral_constant<int, 2>{});
    else if (accumulate) epilogue(std::integral_constant<int, 1>{});
    else epilogue(std::integral_constant<int, 0>{});
}

int g_sgemm_dma = 1;         // cova_set_option(22, .): 1 = the LDS-DMA kernel wherever the operands allow it (default), 0 = never

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
    // 16-byte pieces of both operands: aligned bases and leading dimensions, K a multiple of 4, a row-contiguous operand's row
    // count a multiple of 4 (a piece never straddles the edge); 32-bit byte offsets inside an operand
    const long long a_bytes = 4ll * (transA ? (long long)K * lda : (long long)M * lda), b_bytes = 4ll * (transB ? (long long)N * ldb : (long long)K * ldb);
    const bool dma = g_sgemm_dma && va && vb && (K & 3) == 0 && K >= 4 && (!transA || ((M & 3) == 0 && M >= 4)) &&
                     (transB || ((N & 3) == 0 && N >= 4)) && a_bytes < (1ll << 31) && b_bytes < (1ll << 31);
    if (dma) {
        // tile shape by the grid it makes (measured on the six products of the head, profiles/r06_gemm_bench.txt): 32 x 64 tiles with two
        // k-groups per block when even those leave at most one block per CU (the GAT weight gradient, 768 x 608 x 1440: 33 -> 22 us);
        // 64 x 64 tiles otherwise, with two k-groups when they leave at most one block per CU and K is long (eight waves on the CU
        // instead of four); 32 x 64 tiles without the k-groups lost wherever they were tried (two waves per block: 20 -> 30, 33 -> 50 us)
        const long long blocks64 = (long long)cdiv(M, 64) * cdiv(N, 64), blocks32 = (long long)cdiv(M, 32) * cdiv(N, 64);
        const int shape = blocks32 <= 256 && K >= 8 * 32 ? 2 : (blocks64 <= 256 && K >= 16 * 32 ? 1 : 0);
        const dim3 g32(cdiv(N, 64), cdiv(M, 32));
#define SGEMM_DMA(TA_, TB_)                                                                                                      \
    do {                                                                                                                        \
        if (shape == 0) hipLaunchKernelGGL((sgemm_dma_kernel<TA_, TB_, 2, 1, 4>), grid, dim3(256), 0, st, A, lda, B, ldb, C, ldc, bias, M, N, K, accumulate, emask, einv); \
        else if (shape == 1) hipLaunchKernelGGL((sgemm_dma_kernel<TA_, TB_, 2, 2, 3>), grid, dim3(512), 0, st, A, lda, B, ldb, C, ldc, bias, M, N, K, accumulate, emask, einv); \
        else hipLaunchKernelGGL((sgemm_dma_kernel<TA_, TB_, 1, 2, 4>), g32, dim3(256), 0, st, A, lda, B, ldb, C, ldc, bias, M, N, K, accumulate, emask, einv); \
    } while (0)
        if (!transA && !transB) SGEMM_DMA(false, false);
        else if (!transA && transB) SGEMM_DMA(false, true);
        else if (transA && !transB) SGEMM_DMA(true, false);
        else SGEMM_DMA(true, true);
#undef SGEMM_DMA
        COVA_LAUNCH_CHECK();
        return COVA_OK;
    }
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

int cova_internal_get_sgemm_dma() { return g_sgemm_dma; }
int cova_internal_set_sgemm_dma(int v) { g_sgemm_dma = v != 0; return COVA_OK; }

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
