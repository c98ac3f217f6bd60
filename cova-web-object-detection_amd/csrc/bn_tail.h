// BatchNorm finalize as the TAIL of the kernel that produced the statistics partials: the last block to finish (ticket
// counter) folds the partial rows and writes what cova_bn_finalize_fwd / cova_bn_finalize_bwd_abc would have written
// in a separate launch (same fp64 arithmetic, same summation order: bit-identical results).  Used by the 3x3 / 7x7
// convolution kernels with 64 channels; SyncBN and frozen BatchNorm keep the separate kernels (a collective / a host
// decision sits between the partials and the finalize there).
#pragma once
#include "common.h"

#include "../../include/cova_hip.h"
typedef cova_bn_tail BnTail;      // (a plain C struct of device pointers; passed to the kernels by value)

// per-channel results from the fp64 totals -- shared with bn.hip's stand-alone kernels
// (values of the channel's parameters passed in: the tail requests them before its partial sums, see bn_tail_run)
__device__ __forceinline__ void bn_fwd_channel_v(double sum, double sumsq, double count, int c, float gamma_c, float beta_c,
                                                 float rm_c, float rv_c, float *running_mean, float *running_var,
                                                 float momentum, float eps, float *scale, float *shift, float *mean,
                                                 float *invstd)
{
    const double mu = sum / count;
    double var = sumsq / count - mu * mu;
    if (var < 0.0) var = 0.0;
    const float is = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma_c * is;
    mean[c] = (float)mu;
    invstd[c] = is;
    scale[c] = sc;
    shift[c] = beta_c - (float)mu * sc;
    if (running_mean != nullptr) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_mean[c] = (1.f - momentum) * rm_c + momentum * (float)mu;
        running_var[c] = (1.f - momentum) * rv_c + momentum * (float)unbiased;
    }
}

__device__ __forceinline__ void bn_fwd_channel(double sum, double sumsq, double count, int c, const float *gamma,
                                               const float *beta, float *running_mean, float *running_var,
                                               float momentum, float eps, float *scale, float *shift, float *mean,
                                               float *invstd)
{
    bn_fwd_channel_v(sum, sumsq, count, c, gamma[c], beta[c], running_mean != nullptr ? running_mean[c] : 0.f,
                     running_mean != nullptr ? running_var[c] : 0.f, running_mean, running_var, momentum, eps, scale, shift,
                     mean, invstd);
}

__device__ __forceinline__ void bn_bwd_abc_channel_v(double s1, double s2, double count, int c, int C, float *dgamma,
                                                     float *dbeta, float mean_c, float invstd_c, float scale_c, float *abc)
{
    if (dbeta) dbeta[c] = (float)s1;
    if (dgamma) dgamma[c] = (float)s2;
    const double c1 = s1 / count, c2 = s2 / count;
    const double sc = scale_c, is = invstd_c, mu = mean_c;
    abc[c] = (float)sc;
    abc[C + c] = (float)(-sc * is * c2);
    abc[2 * C + c] = (float)(sc * (mu * is * c2 - c1));
}

__device__ __forceinline__ void bn_bwd_abc_channel(double s1, double s2, double count, int c, int C, float *dgamma,
                                                   float *dbeta, const float *mean, const float *invstd,
                                                   const float *scale, float *abc)
{
    bn_bwd_abc_channel_v(s1, s2, count, c, C, dgamma, dbeta, mean[c], invstd[c], scale[c], abc);
}

// The publish protocol below (relaxed agent-scope stores, s_waitcnt(0), relaxed ticket) is ordered only where stores
// and non-returning atomics are counted in vmcnt and sc1 stores write through: gfx9 / CDNA.  An architecture with a
// separate store counter (gfx10+) would race silently -- refuse to build for anything else.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__)
#error "bn_tail.h: the fence-free partial-row hand-off is written for gfx90a / gfx942 / gfx950 (stores in vmcnt); use release / acquire at agent scope on the ticket for other targets"
#endif

// A block's partial row is written with device-scope (agent) relaxed atomic stores and read back by the last block with
// device-scope atomic loads: per-access coherence across the eight XCDs' L2s.  A __threadfence() instead would write
// back and invalidate the whole L2 of the XCD at the end of every block -- measured: +0.6 ms per train step, because
// the next kernel then finds neither the output map nor the weights in L2.
__device__ __forceinline__ void bn_tail_store(float *p, float v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Call at the very end of the kernel, by ALL threads of every block, after this block's partial row(s) have been
// written with bn_tail_store (`partial` = [nparts][2][64]).  `s_dbl` = at least 2049 doubles of shared memory that is
// free by now (2 x 1024 partial sums + the block's ticket flag).  blockDim.x must be a multiple of 64.
__device__ __forceinline__ void bn_tail_run(const BnTail &t, float *partial, int nparts, double *s_dbl)
{
    if (t.mode == 0) return;
    int &s_last = *reinterpret_cast<int *>(s_dbl + 2048);
    __builtin_amdgcn_s_waitcnt(0);        // this thread's row stores have been acknowledged at device scope ...
    __syncthreads();                      // ... all of the block's have ...
    if (threadIdx.x == 0)                 // ... before its ticket is taken
        s_last = __hip_atomic_fetch_add(t.counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
    __syncthreads();
    if (!s_last) return;
    double *s_a = s_dbl, *s_b = s_dbl + 1024;
    const int c = threadIdx.x & 63, nsl = blockDim.x >> 6;
    // what the finalize of channel c reads goes out in front of the partial sums (it was two more dependent round trips behind them)
    float q0 = 0.f, q1 = 0.f, q2 = 0.f, q3 = 0.f;
    if (threadIdx.x < 64) {
        if (t.mode == 1) {
            q0 = t.gamma[c];
            q1 = t.beta[c];
            if (t.running_mean != nullptr) { q2 = t.running_mean[c]; q3 = t.running_var[c]; }
        } else {
            q0 = t.mean[c];
            q1 = t.invstd[c];
            q2 = t.scale[c];
        }
    }
    for (int slice = threadIdx.x >> 6; slice < 16; slice += nsl) {      // bn.hip's combine_partials: 16 slices of rows
        double a = 0.0, b = 0.0;
        // eight rows of the slice (16 device-scope loads) are requested before the first is added: this loop runs in ONE block
        // after every other block has finished -- a serial tail of the launch -- and one row per round trip was 16 dependent
        // trips to the far side of the L2s per slice (256 partial rows, two slices per wave); same adds in the same order
        for (int p0 = slice; p0 < nparts; p0 += 16 * 8) {
            float va[8], vb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int p = p0 + 16 * u < nparts ? p0 + 16 * u : p0;
                va[u] = __hip_atomic_load(partial + ((size_t)p * 2 + 0) * 64 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                vb[u] = __hip_atomic_load(partial + ((size_t)p * 2 + 1) * 64 + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (p0 + 16 * u < nparts) {
                    a += (double)va[u];
                    b += (double)vb[u];
                }
        }
        s_a[slice * 64 + c] = a;
        s_b[slice * 64 + c] = b;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        double ta = 0.0, tb = 0.0;
        for (int j = 0; j < 16; ++j) {
            ta += s_a[j * 64 + c];
            tb += s_b[j * 64 + c];
        }
        if (t.mode == 1) {
            if (threadIdx.x == 0 && t.num_batches_tracked != nullptr) *t.num_batches_tracked += 1;
            bn_fwd_channel_v(ta, tb, t.count, c, q0, q1, q2, q3, t.running_mean, t.running_var, t.momentum, t.eps,
                             t.scale, t.shift, t.mean, t.invstd);
        } else {
            bn_bwd_abc_channel_v(ta, tb, t.count, c, 64, t.dgamma, t.dbeta, q0, q1, q2, t.abc);
        }
    }
    if (threadIdx.x == 0) __hip_atomic_store(t.counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
}
