// fp32 GEMM on v_mfma_f32_32x32x2_f32 for the dense layers of the hot path:
//   Wh = h [W_i;W_j]^T (models.py:188,193), decoder Linear(T,T) (models.py:85) and their
//   backward products (dX = dY W, dW = dY^T X).
// Exact f32 arithmetic (the MFMA is an fmaf chain), so results differ from the reference's
// CPU GEMM only by summation order.  Row-major operands with leading dimensions; any M, N, K.
//   C[M,N] (+)= op(A)[M,K] * op(B)[K,N] (+ bias[N])
//   transA = 0: A is [M,K] (lda >= K);  transA = 1: A is stored [K,M] (lda >= M)
//   transB = 0: B is [K,N] (ldb >= N);  transB = 1: B is stored [N,K] (ldb >= K)
// Block tile 64x64, BK = 16, 4 waves (2x2), each wave one 32x32 accumulator.  LDS tiles are
// k-major ([k][m]) so that MFMA operand reads are conflict-free ds_read_b32 across lanes.
#include "common.h"

namespace {

constexpr int BM = 64, BN = 64, BK = 16, LDT = 65;   // +1 pad: transposing stores spread banks

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void sgemm_kernel(const float *__restrict__ A, int lda,
                                                    const float *__restrict__ Bm, int ldb,
                                                    float *__restrict__ C, int ldc,
                                                    const float *__restrict__ bias, int M, int N,
                                                    int K, int accumulate)
{
    __shared__ float As[BK][LDT], Bs[BK][LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int li = lane & 31, kh2 = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    for (int k0 = 0; k0 < K; k0 += BK) {
        // ---- stage A tile (BM x BK) as As[k][m]
        if (!TA) {
            const int r = tid >> 2, kq = (tid & 3) * 4;     // row r, 4 consecutive k
            const int gm = m0 + r;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gk = k0 + kq + j;
                As[kq + j][r] = (gm < M && gk < K) ? A[(size_t)gm * lda + gk] : 0.f;
            }
        } else {
            const int kk = tid >> 4, mq = (tid & 15) * 4;   // k row kk, 4 consecutive m
            const int gk = k0 + kk;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gm = m0 + mq + j;
                As[kk][mq + j] = (gm < M && gk < K) ? A[(size_t)gk * lda + gm] : 0.f;
            }
        }
        // ---- stage B tile (BK x BN) as Bs[k][n]
        if (TB) {
            const int r = tid >> 2, kq = (tid & 3) * 4;     // n row r of stored [N,K]
            const int gn = n0 + r;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gk = k0 + kq + j;
                Bs[kq + j][r] = (gn < N && gk < K) ? Bm[(size_t)gn * ldb + gk] : 0.f;
            }
        } else {
            const int kk = tid >> 4, nq = (tid & 15) * 4;
            const int gk = k0 + kk;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int gn = n0 + nq + j;
                Bs[kk][nq + j] = (gn < N && gk < K) ? Bm[(size_t)gk * ldb + gn] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const float a = As[2 * kk + kh2][wm * 32 + li];
            const float b = Bs[2 * kk + kh2][wn * 32 + li];
            acc = mfma32(a, b, acc);
        }
        __syncthreads();
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

}  // namespace

COVA_API int cova_sgemm(int transA, int transB, int M, int N, int K, const float *A, int lda,
                        const float *B, int ldb, float *C, int ldc, const float *bias,
                        int accumulate, void *stream)
{
    COVA_REQUIRE(A && B && C && M >= 0 && N >= 0 && K >= 0);
    if (M == 0 || N == 0) return COVA_OK;
    const dim3 grid(cdiv(N, BN), cdiv(M, BM)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (!transA && !transB)
        hipLaunchKernelGGL((sgemm_kernel<false, false>), grid, block, 0, st, A, lda, B, ldb, C, ldc, bias, M, N, K, accumulate);
    else if (!transA && transB)
        hipLaunchKernelGGL((sgemm_kernel<false, true>), grid, block, 0, st, A, lda, B, ldb, C, ldc, bias, M, N, K, accumulate);
    else if (transA && !transB)
        hipLaunchKernelGGL((sgemm_kernel<true, false>), grid, block, 0, st, A, lda, B, ldb, C, ldc, bias, M, N, K, accumulate);
    else
        hipLaunchKernelGGL((sgemm_kernel<true, true>), grid, block, 0, st, A, lda, B, ldb, C, ldc, bias, M, N, K, accumulate);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}
