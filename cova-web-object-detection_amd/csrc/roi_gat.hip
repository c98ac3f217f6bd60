// RoIPool over the NHWC feature map, the positional (bbox) encoder input features, and the
// graph-attention layer's sparse part (scores, masked softmax, neighbour gather) fwd + bwd.
//
// Reference: torchvision.ops.RoIPool as used at models.py:58,125 (algorithm restated in
// oracle/roipool_ref.c), CoVA._get_bbox_features (models.py:129-148) and
// GraphAttentionLayer.forward (models.py:171-212).
//
// GAT formulation: W_j is linear and bias-free, so W_j(h_pad[ctx]) == (W_j h)_pad[ctx]; the
// dense projections Wh = h [W_i;W_j]^T are one GEMM per node (gemm.hip) and this file does the
// O(N*K) part: e_ik = LeakyReLU(s_i + t_ctx(i,k)), s = a_i.Wh_i + b, t = a_j.Wh_j (t = 0 for the
// -1 pad row), mask -> -9e15, softmax over the K slots, h'_i = sum_k alpha_ik Wh_j[ctx(i,k)].
// One wavefront per node: the K neighbour slots live in lanes (ceil(K / 64) passes, K <= 1024) for the softmax (shuffle
// reductions), the D hidden channels live in lanes for the gather (256-byte coalesced rows).
#include "bn_tail.h"

int cova_internal_persistent_grid2(int ntiles, int blocks_per_cu);

namespace {

// ------------------------------------------------------------------------------------ RoIPool
// one wave per (roi, bin); C multiple of 64 handled by a loop over 64-channel blocks
// LAZY: the feature map is not read but formed on the fly as relu(fma(scale, z, shift) + x) -- the last
// BasicBlock's bn2 + residual + ReLU (same expression as cova_bn_act_fwd), never written to HBM.
struct LazyFeat {
    const float *x, *scale, *shift;       // feat points at z
};

// U: float4 loads per lane and map in flight per trip (4 U pixels of the bin; a bin of a large box is hundreds of pixels, and
// a trip is a dependent memory round trip); XCD: consecutive blocks go to the eight XCDs in turn -- block b takes work block
// (b % 8) * (grid / 8) + b / 8, so that the boxes of a page (adjacent in the box list) meet in ONE XCD's L2.
// 64-channel maps launch <8, true> (0.189 -> 0.174 ms inside the step against <4, false>; <8, false>: 0.178, <4, true>: 0.186),
// wider maps <4, false> (bandwidth bound, eight loads per lane cost occupancy: configs[2] 1.51 against 1.59 ms).
template <bool LAZY, int U = 4, bool XCD = false>
__global__ __launch_bounds__(256) void roipool_fwd_kernel(
    const float *__restrict__ feat, const float *__restrict__ rois, int n_rois, int B, int C, int H,
    int W, int PH, int PW, float spatial_scale, float *__restrict__ out, int ld_out,
    int32_t *__restrict__ argmax, float *__restrict__ zmax, const LazyFeat lz)
{
    int wb = blockIdx.x;
    if (XCD) {
        const int per = gridDim.x / 8;
        if (wb < per * 8) wb = (wb % 8) * per + wb / 8;
    }
    const int task = wb * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (task >= n_rois * PH * PW) return;
    const int n = task / (PH * PW), bin = task - n * (PH * PW);
    const int ph = bin / PW, pw = bin - ph * PW;
    const float *roi = rois + 5 * n;
    const int b = (int)roi[0];
    const int rs_w = (int)roundf(roi[1] * spatial_scale);
    const int rs_h = (int)roundf(roi[2] * spatial_scale);
    const int re_w = (int)roundf(roi[3] * spatial_scale);
    const int re_h = (int)roundf(roi[4] * spatial_scale);
    const int roi_w = max(re_w - rs_w + 1, 1);
    const int roi_h = max(re_h - rs_h + 1, 1);
    const float bin_h = (float)roi_h / (float)PH;
    const float bin_w = (float)roi_w / (float)PW;
    int hstart = (int)floorf((float)ph * bin_h);
    int hend = (int)ceilf((float)(ph + 1) * bin_h);
    int wstart = (int)floorf((float)pw * bin_w);
    int wend = (int)ceilf((float)(pw + 1) * bin_w);
    hstart = min(max(hstart + rs_h, 0), H);
    hend = min(max(hend + rs_h, 0), H);
    wstart = min(max(wstart + rs_w, 0), W);
    wend = min(max(wend + rs_w, 0), W);
    // a page index outside the batch (torchvision asserts / faults) reads nothing: bin = 0, argmax = -1
    const bool bad_page = b < 0 || b >= B;
    if (bad_page) hend = hstart;
    const bool empty = (hend <= hstart) || (wend <= wstart);
    const float *fb = feat + (size_t)(bad_page ? 0 : b) * H * W * C;
    const float *xb = LAZY ? lz.x + (size_t)(bad_page ? 0 : b) * H * W * C : nullptr;
    // lane = (pixel phase ps = lane >> 4, channels 4*(lane & 15)..+3): one float4 per lane covers 4 pixels x 256 B
    // per load instruction; 4 loads (16 pixels) per map are in flight before the first compare.  Every lane scans
    // its pixels (linear window index == ps mod 4) in ascending order with a strict >, then the four phases are
    // merged preferring the larger value and, on ties, the smaller index: exactly the first maximum in row-major
    // scan order that the reference's sequential loop finds.
    const int ps = lane >> 4, cq = 4 * (lane & 15);
    const int nw = wend - wstart, npx = empty ? 0 : (hend - hstart) * nw;
    for (int cb = 0; cb < C; cb += 64) {
        const int c4 = cb + cq;
        float best[4], bz[4];
        int bi[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { best[e] = empty ? 0.f : -FLT_MAX; bz[e] = 0.f; bi[e] = -1; }
        float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), sh = sc;
        if (LAZY) {
            sc = *reinterpret_cast<const float4 *>(lz.scale + c4);
            sh = *reinterpret_cast<const float4 *>(lz.shift + c4);
        }
        for (int i0 = 0; i0 < npx; i0 += 4 * U) {
            float4 zv[U], xv[U];
            int pos[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int i = i0 + 4 * u + ps;
                const int ii = i < npx ? i : npx - 1;            // clamped: loads stay unconditional
                const int hh = ii / nw;
                pos[u] = (hstart + hh) * W + wstart + (ii - hh * nw);
                zv[u] = *reinterpret_cast<const float4 *>(fb + (size_t)pos[u] * C + c4);
                if (LAZY) xv[u] = *reinterpret_cast<const float4 *>(xb + (size_t)pos[u] * C + c4);
                if (i >= npx) pos[u] = -1;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (pos[u] < 0) continue;
                const float z4[4] = {zv[u].x, zv[u].y, zv[u].z, zv[u].w};
                float v4[4] = {z4[0], z4[1], z4[2], z4[3]};
                if (LAZY) {                                       // bn2 + residual + ReLU on the fly
                    const float x4[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
                    const float s4[4] = {sc.x, sc.y, sc.z, sc.w}, h4[4] = {sh.x, sh.y, sh.z, sh.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float y = fmaf(s4[e], z4[e], h4[e]) + x4[e];
                        v4[e] = y > 0.f ? y : 0.f;
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (v4[e] > best[e]) { best[e] = v4[e]; bz[e] = z4[e]; bi[e] = pos[u]; }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int o = 16; o <= 32; o <<= 1) {
                const float ov = __shfl_xor(best[e], o, 64), oz = __shfl_xor(bz[e], o, 64);
                const int oi = __shfl_xor(bi[e], o, 64);
                // (a lane that saw no pixel still holds the initial value with index -1: it never wins a tie
                // against a real pixel, and loses to any larger value)
                const bool take = ov > best[e] || (ov == best[e] && oi >= 0 && (bi[e] < 0 || oi < bi[e]));
                if (take) { best[e] = ov; bz[e] = oz; bi[e] = oi; }
            }
        }
        if (ps == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // reference layout [N, C, PH, PW] flattened per roi: c*(PH*PW) + bin  (models.py:125-127)
                const size_t o = (size_t)(c4 + e) * (PH * PW) + bin;
                out[(size_t)n * ld_out + o] = best[e];
                argmax[(size_t)n * (C * PH * PW) + o] = bi[e];
                // the pre-activation z at the arg-max: with it the backward takes the producer's BatchNorm sums
                // per pooled entry and never has to read the maps again
                if (LAZY && zmax != nullptr) zmax[(size_t)n * (C * PH * PW) + o] = bz[e];
            }
        }
    }
}

// RoIPool backward WITHOUT atomics (deterministic): the reference's scatter `grad_in[b,c,argmax] += grad_out`
// collides all the time on web pages (DOM parents contain their children), and float atomics make two runs of
// the same step differ in the last bits.  Here every (page, feature row, 64-channel block, x segment) has ONE
// owner wave: it finds the boxes whose bins touch its row (lane-parallel geometry test + ballot, ascending box
// order), adds their contributions into an LDS row accumulator in program order (lanes = channels, so lanes never
// collide), and writes the finished row -- which also replaces the zero-fill of the gradient map.
// BN: the row is masked by the ReLU of the map's producer and the BatchNorm-backward sums are taken on the way
// out (only where the gradient is non-zero: ~1 % of the map).
constexpr int ROI_XW = 40;                        // pixels per owner wave (10 KB of LDS: 16 waves per CU)

struct RoiGeo {
    int b, rs_h, rs_w;
    float bin_h, bin_w;
};

__device__ __forceinline__ RoiGeo roi_geo(const float *__restrict__ roi, float spatial_scale, int PH, int PW)
{
    RoiGeo g;
    g.b = (int)roi[0];
    g.rs_w = (int)roundf(roi[1] * spatial_scale);
    g.rs_h = (int)roundf(roi[2] * spatial_scale);
    const int re_w = (int)roundf(roi[3] * spatial_scale);
    const int re_h = (int)roundf(roi[4] * spatial_scale);
    g.bin_h = (float)max(re_h - g.rs_h + 1, 1) / (float)PH;
    g.bin_w = (float)max(re_w - g.rs_w + 1, 1) / (float)PW;
    return g;
}
__device__ __forceinline__ int bin_lo(int i, float bin, int rs, int lim) { return min(max((int)floorf((float)i * bin) + rs, 0), lim); }
__device__ __forceinline__ int bin_hi(int i, float bin, int rs, int lim) { return min(max((int)ceilf((float)(i + 1) * bin) + rs, 0), lim); }

// [first, last] box index of every page (boxes of a page are contiguous in the collate layout, datasets.py:
// 170-178; if they are not, the range merely contains foreign boxes, which the page test rejects)
__global__ void roipool_page_range_kernel(const float *__restrict__ rois, int n_rois, int B, int *__restrict__ range)
{
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_rois) return;
    const int b = (int)rois[5 * n];
    if (b < 0 || b >= B) return;
    // only the first / last box of a run of equal page indices can extend the range: 2 atomics per page for the
    // sorted collate layout instead of n_rois contended ones (integer min / max: order-independent)
    if (n == 0 || (int)rois[5 * (n - 1)] != b) atomicMin(range + 2 * b, n);
    if (n == n_rois - 1 || (int)rois[5 * (n + 1)] != b) atomicMax(range + 2 * b + 1, n);
}

__global__ void roipool_page_range_init_kernel(int B, int *__restrict__ range)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) { range[2 * b] = 0x7fffffff; range[2 * b + 1] = -1; }
}

// both steps by ONE block (init, block barrier, scan): the form the launches of the step use
__device__ __forceinline__ void page_ranges_by_block(const float *__restrict__ rois, int n_rois, int B,
                                                     int *__restrict__ range)
{
    for (int b = threadIdx.x; b < B; b += blockDim.x) { range[2 * b] = 0x7fffffff; range[2 * b + 1] = -1; }
    __syncthreads();
    for (int n = threadIdx.x; n < n_rois; n += blockDim.x) {
        const int b = (int)rois[5 * n];
        if (b < 0 || b >= B) continue;
        if (n == 0 || (int)rois[5 * (n - 1)] != b) atomicMin(range + 2 * b, n);
        if (n == n_rois - 1 || (int)rois[5 * (n + 1)] != b) atomicMax(range + 2 * b + 1, n);
    }
}

__global__ __launch_bounds__(1024) void roipool_page_range_block_kernel(const float *__restrict__ rois, int n_rois, int B,
                                                                       int *__restrict__ range)
{
    page_ranges_by_block(rois, n_rois, B, range);
}

// Owner = one wave per (page, feature row, 40-pixel segment), 64 channels (blockIdx.y = channel block); the four
// waves of a block are independent (own LDS slice, own task stream, no block barrier).
//   accumulate: lane = channel; boxes touching the row are found 64 at a time (geometry test + ballot), visited
//               in ascending order, their arg-max hits added into the LDS row -- program order, so deterministic;
//   write out : lane = (pixel t*4 + lane/16, channels 4*(lane%16)..+3): one float4 per lane, 1 KB per store.
// pooled != NULL: the contribution of an entry is masked by pooled > 0 -- the pooled value IS the map's value at
// the arg-max, so this is the ReLU mask of the map's producer without reading the map.
// gT / amT: the (masked) contributions and arg-max positions in [box][bin][channel] order (roipool_bwd_prep_kernel)
// so that a wave's loads are 256 contiguous bytes.
// NBX: boxes of a row segment visited per round trip (4 measured slower in round 5: 0.146 against 0.123 ms for entry pass + rows
// -- 72 loads of which most bins miss the row)
template <bool P33, int NBX = 2>                  // P33: the reference's 3x3 bins (models.py:58) as compile-time constants
__global__ __launch_bounds__(256) void roipool_bwd_rows_kernel(
    const float *__restrict__ gT, const int32_t *__restrict__ amT, const float *__restrict__ rois,
    const int *__restrict__ page_range, int n_rois, int B, int C, int H, int W, int PH_, int PW_,
    float spatial_scale, float *__restrict__ gfeat)
{
    __shared__ __attribute__((aligned(16))) float lds[4 * ROI_XW * 64];
    const int PH = P33 ? 3 : PH_, PW = P33 ? 3 : PW_;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *acc = lds + wave * ROI_XW * 64;
    const int cb = blockIdx.y * 64;
    const int c = cb + lane;                          // accumulate phase
    const int ps = lane >> 4, c4 = cb + 4 * (lane & 15);   // write-out phase
    const int nx = (W + ROI_XW - 1) / ROI_XW;
    const long long ntask = (long long)B * H * nx;
    for (long long task = (long long)blockIdx.x * 4 + wave; task < ntask; task += (long long)gridDim.x * 4) {
        const int xs = (int)(task % nx);
        const int y = (int)((task / nx) % H);
        const int b = (int)(task / ((long long)nx * H));
        const int x0 = xs * ROI_XW, x1 = min(x0 + ROI_XW, W);
#pragma unroll
        for (int i = 0; i < ROI_XW * 16 / 64; ++i)
            reinterpret_cast<float4 *>(acc)[i * 64 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int n_lo = page_range[2 * b], n_hi = page_range[2 * b + 1];
        for (int n0 = n_lo; n0 <= n_hi; n0 += 64) {
            const int n = n0 + lane;
            bool hit = false;
            if (n <= n_hi) {
                const RoiGeo g = roi_geo(rois + 5 * n, spatial_scale, PH, PW);
                hit = g.b == b && y >= bin_lo(0, g.bin_h, g.rs_h, H) && y < bin_hi(PH - 1, g.bin_h, g.rs_h, H) &&
                      bin_lo(0, g.bin_w, g.rs_w, W) < x1 && bin_hi(PW - 1, g.bin_w, g.rs_w, W) > x0;
            }
            unsigned long long m = __ballot(hit);
            if (P33) {
                // boxes touching this segment, ascending (fixed order); NBX at a time so that their geometry
                // and arg-max / gradient operands are one round trip (NBX = 4, 72 loads in flight: measured slower)
                while (m) {
                    int nb[NBX];
                    RoiGeo g[NBX];
                    int mi[NBX][9];
                    float gg[NBX][9];
#pragma unroll
                    for (int u = 0; u < NBX; ++u) {
                        nb[u] = m ? n0 + __ffsll((long long)m) - 1 : -1;
                        m &= m - 1;                          // (0 & anything stays 0)
                        const int nn = nb[u] >= 0 ? nb[u] : n_lo;
                        g[u] = roi_geo(rois + 5 * nn, spatial_scale, PH, PW);
                        const int32_t *am = amT + (size_t)nn * 9 * C + c;
                        const float *gv = gT + (size_t)nn * 9 * C + c;
#pragma unroll
                        for (int q = 0; q < 9; ++q) {
                            mi[u][q] = am[(size_t)q * C];
                            gg[u][q] = gv[(size_t)q * C];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < NBX; ++u) {
#pragma unroll
                        for (int ph = 0; ph < 3; ++ph) {
                            const bool rowhit = nb[u] >= 0 && y >= bin_lo(ph, g[u].bin_h, g[u].rs_h, H) &&
                                                y < bin_hi(ph, g[u].bin_h, g[u].rs_h, H);
#pragma unroll
                            for (int pw = 0; pw < 3; ++pw) {
                                const int q = ph * 3 + pw;
                                const int x = mi[u][q] - y * W;
                                if (rowhit && mi[u][q] >= 0 && x >= x0 && x < x1) acc[(x - x0) * 64 + lane] += gg[u][q];
                            }
                        }
                    }
                }
            } else {
                while (m) {
                    const int nb = n0 + __ffsll((long long)m) - 1;
                    m &= m - 1;
                    const RoiGeo g = roi_geo(rois + 5 * nb, spatial_scale, PH, PW);
                    const int32_t *am = amT + (size_t)nb * PH * PW * C + c;
                    const float *gv = gT + (size_t)nb * PH * PW * C + c;
                    for (int q = 0; q < PH * PW; ++q) {
                        const int ph = q / PW;
                        if (y < bin_lo(ph, g.bin_h, g.rs_h, H) || y >= bin_hi(ph, g.bin_h, g.rs_h, H)) continue;
                        const int mi = am[(size_t)q * C];
                        const int x = mi - y * W;
                        if (mi >= 0 && x >= x0 && x < x1) acc[(x - x0) * 64 + lane] += gv[(size_t)q * C];
                    }
                }
            }
        }
        const size_t row = ((size_t)b * H + y) * W;
#pragma unroll
        for (int t = 0; t < ROI_XW / 4; ++t) {
            const int x = x0 + t * 4 + ps;
            if (x < x1)
                *reinterpret_cast<float4 *>(gfeat + (row + x) * C + c4) =
                    *reinterpret_cast<const float4 *>(acc + (t * 4 + ps) * 64 + 4 * (lane & 15));
        }
    }
}

// Entry pass of the backward: g' = gout (* (pooled > 0): the ReLU mask of the map's producer -- the pooled value
// IS the map's value at the arg-max) and the arg-max positions, transposed from the reference's [box][c*bins+bin]
// order into [box][bin][channel] (the rows kernel then reads 256 contiguous bytes per wave), and -- STATS -- the
// producer's BatchNorm-backward sums per pooled entry instead of per map element: both are linear in the routed
// contributions, and zmax holds the pre-activation at each arg-max:
//   partial[blk] = (sum g', sum g' * (zmax - mean) * invstd) per channel; boxes -> waves by a fixed rule.
// Block (0,0) also builds the per-page box ranges the rows kernel needs (page_range != NULL), and -- TAIL (C = 64) -- the
// last block to finish turns the partial rows into dgamma / dbeta / the dz coefficients (bn_tail.h): no launch of its own
// for either.
// NB = 9: the reference's 3 x 3 bins as a compile-time constant -- the 36 operands of a box are requested before the first is
// used (one round trip per box instead of nine; same sums in the same order); NB = 0: any bin count.
template <bool STATS, bool TAIL, int NB = 0>
__global__ __launch_bounds__(256) void roipool_bwd_prep_kernel(
    const float *__restrict__ gout, int ld_g, const float *__restrict__ pooled, int ld_p,
    const float *__restrict__ zmax, const int32_t *__restrict__ argmax, int n_rois, int C, int bins,
    const float *__restrict__ mean, const float *__restrict__ invstd, float *__restrict__ gT,
    int32_t *__restrict__ amT, float *__restrict__ partial, const float *__restrict__ rois, int B,
    int *__restrict__ page_range, const BnTail tail)
{
    __shared__ float s_red[4][2][64];
    __shared__ double s_tail[TAIL ? 2049 : 1];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (page_range != nullptr && blockIdx.x == 0 && blockIdx.y == 0) page_ranges_by_block(rois, n_rois, B, page_range);
    const int c = blockIdx.y * 64 + lane;
    const float mu = STATS ? mean[c] : 0.f, is = STATS ? invstd[c] : 0.f;
    float su = 0.f, sq = 0.f;
    for (int n = blockIdx.x * 4 + wave; n < n_rois; n += gridDim.x * 4) {
        const size_t e = (size_t)c * bins;
        if (NB > 0) {
            int mi[NB > 0 ? NB : 1];
            float g[NB > 0 ? NB : 1], pl[NB > 0 ? NB : 1], zm[NB > 0 ? NB : 1];
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                mi[q] = argmax[(size_t)n * C * NB + e + q];
                g[q] = gout[(size_t)n * ld_g + e + q];
                if (STATS) {
                    pl[q] = pooled[(size_t)n * ld_p + e + q];
                    zm[q] = zmax[(size_t)n * C * NB + e + q];
                }
            }
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                if (STATS) {
                    if (!(mi[q] >= 0 && pl[q] > 0.f)) g[q] = 0.f;
                    su += g[q];
                    sq += g[q] * ((zm[q] - mu) * is);
                }
                gT[((size_t)n * NB + q) * C + c] = g[q];
                amT[((size_t)n * NB + q) * C + c] = mi[q];
            }
            continue;
        }
        for (int q = 0; q < bins; ++q) {
            const int mi = argmax[(size_t)n * C * bins + e + q];
            float g = gout[(size_t)n * ld_g + e + q];
            if (STATS) {
                if (!(mi >= 0 && pooled[(size_t)n * ld_p + e + q] > 0.f)) g = 0.f;
                su += g;
                sq += g * ((zmax[(size_t)n * C * bins + e + q] - mu) * is);
            }
            gT[((size_t)n * bins + q) * C + c] = g;
            amT[((size_t)n * bins + q) * C + c] = mi;
        }
    }
    if (STATS) {
        s_red[wave][0][lane] = su;
        s_red[wave][1][lane] = sq;
        __syncthreads();
        if (threadIdx.x < 128) {
            const int which = threadIdx.x >> 6;
            const float v = (s_red[0][which][lane] + s_red[1][which][lane]) + (s_red[2][which][lane] + s_red[3][which][lane]);
            float *dst = partial + ((size_t)blockIdx.x * 2 + which) * C + blockIdx.y * 64 + lane;
            if (TAIL) bn_tail_store(dst, v);
            else *dst = v;
        }
        if (TAIL) bn_tail_run(tail, partial, (int)gridDim.x, s_tail);
    }
}

// ------------------------------------------------------------------------------------ RoIAlign (extension)
// north_star names RoIAlign; the reference itself calls torchvision.ops.RoIPool (models.py:58), which stays the
// parity operator.  This is the additional variant: torchvision's RoIAlign (bilinear samples, sampling_ratio^2 per
// bin or ceil(roi / bin)^2 when sampling_ratio <= 0, `aligned` half-pixel shift), coordinates in float like its
// C++ / CUDA code.  Self-oracle: oracle/cova_oracle.py::roi_align.
struct AlignGeo {
    int b, gh, gw;
    float sw, sh, bh, bw, inv_count;
};

__device__ __forceinline__ AlignGeo align_geo(const float *__restrict__ roi, float scale, int PH, int PW,
                                              int sampling_ratio, int aligned)
{
    AlignGeo g;
    const float off = aligned ? 0.5f : 0.f;
    g.b = (int)roi[0];
    g.sw = roi[1] * scale - off;
    g.sh = roi[2] * scale - off;
    float rw = (roi[3] * scale - off) - g.sw, rh = (roi[4] * scale - off) - g.sh;
    if (!aligned) { rw = fmaxf(rw, 1.f); rh = fmaxf(rh, 1.f); }
    g.bh = rh / (float)PH;
    g.bw = rw / (float)PW;
    g.gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)PH);
    g.gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)PW);
    g.inv_count = 1.f / (float)max(g.gh * g.gw, 1);
    return g;
}

struct Bilin {
    int yl, yh, xl, xh;
    float hy, ly, hx, lx;
    bool ok;
};

__device__ __forceinline__ Bilin bilin_setup(float y, float x, int H, int W)
{
    Bilin q;
    q.ok = !(y < -1.f || y > (float)H || x < -1.f || x > (float)W);
    y = fmaxf(y, 0.f);
    x = fmaxf(x, 0.f);
    q.yl = (int)y;
    q.xl = (int)x;
    if (q.yl >= H - 1) { q.yl = q.yh = H - 1; y = (float)q.yl; } else { q.yh = q.yl + 1; }
    if (q.xl >= W - 1) { q.xl = q.xh = W - 1; x = (float)q.xl; } else { q.xh = q.xl + 1; }
    q.ly = y - (float)q.yl;
    q.lx = x - (float)q.xl;
    q.hy = 1.f - q.ly;
    q.hx = 1.f - q.lx;
    return q;
}

__device__ __forceinline__ float align_coord(float start, int bin, float bsz, int i, int grid)
{
    return (start + (float)bin * bsz) + (((float)i + 0.5f) * bsz) / (float)grid;
}

// one wave per (roi, bin); lanes = channels (256-byte coalesced rows of the NHWC map)
__global__ __launch_bounds__(256) void roialign_fwd_kernel(
    const float *__restrict__ feat, const float *__restrict__ rois, int n_rois, int B, int C, int H, int W,
    int PH, int PW, float spatial_scale, int sampling_ratio, int aligned, float *__restrict__ out, int ld_out)
{
    const int task = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (task >= n_rois * PH * PW) return;
    const int n = task / (PH * PW), bin = task - n * (PH * PW);
    const int ph = bin / PW, pw = bin - ph * PW;
    const AlignGeo g = align_geo(rois + 5 * n, spatial_scale, PH, PW, sampling_ratio, aligned);
    const bool bad_page = g.b < 0 || g.b >= B;
    const float *fb = feat + (size_t)(bad_page ? 0 : g.b) * H * W * C;
    for (int c = lane; c < C; c += 64) {
        float acc = 0.f;
        if (!bad_page)
            for (int iy = 0; iy < g.gh; ++iy) {
                const float y = align_coord(g.sh, ph, g.bh, iy, g.gh);
                for (int ix = 0; ix < g.gw; ++ix) {
                    const float x = align_coord(g.sw, pw, g.bw, ix, g.gw);
                    const Bilin q = bilin_setup(y, x, H, W);
                    if (!q.ok) continue;
                    const float v1 = fb[((size_t)q.yl * W + q.xl) * C + c], v2 = fb[((size_t)q.yl * W + q.xh) * C + c];
                    const float v3 = fb[((size_t)q.yh * W + q.xl) * C + c], v4 = fb[((size_t)q.yh * W + q.xh) * C + c];
                    acc = acc + (q.hy * q.hx) * v1 + (q.hy * q.lx) * v2 + (q.ly * q.hx) * v3 + (q.ly * q.lx) * v4;
                }
            }
        out[(size_t)n * ld_out + c * (PH * PW) + bin] = acc * g.inv_count;
    }
}

// Backward, deterministic like RoIPool's: one owner wave per (page, feature row, 40-pixel segment), lanes = channels;
// boxes whose sample rows can touch the row are found 64 at a time (ballot) and visited in ascending order, every
// sample's two row neighbours are tested against the owner row and its column pair added into the LDS row.
__global__ __launch_bounds__(256) void roialign_bwd_rows_kernel(
    const float *__restrict__ gout, int ld_g, const float *__restrict__ rois, const int *__restrict__ page_range,
    int n_rois, int B, int C, int H, int W, int PH, int PW, float spatial_scale, int sampling_ratio, int aligned,
    float *__restrict__ gfeat)
{
    __shared__ __attribute__((aligned(16))) float lds[4 * ROI_XW * 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *acc = lds + wave * ROI_XW * 64;
    const int cb = blockIdx.y * 64, c = cb + lane;
    const int ps = lane >> 4, c4 = cb + 4 * (lane & 15);
    const int nx = (W + ROI_XW - 1) / ROI_XW;
    const long long ntask = (long long)B * H * nx;
    for (long long task = (long long)blockIdx.x * 4 + wave; task < ntask; task += (long long)gridDim.x * 4) {
        const int xs = (int)(task % nx);
        const int y = (int)((task / nx) % H);
        const int b = (int)(task / ((long long)nx * H));
        const int x0 = xs * ROI_XW, x1 = min(x0 + ROI_XW, W);
#pragma unroll
        for (int i = 0; i < ROI_XW * 16 / 64; ++i)
            reinterpret_cast<float4 *>(acc)[i * 64 + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int n_lo = page_range[2 * b], n_hi = page_range[2 * b + 1];
        for (int n0 = n_lo; n0 <= n_hi; n0 += 64) {
            const int n = n0 + lane;
            bool hit = false;
            if (n <= n_hi) {
                const AlignGeo g = align_geo(rois + 5 * n, spatial_scale, PH, PW, sampling_ratio, aligned);
                // rows / columns any sample of the box can touch (conservative: +-1 pixel around the roi)
                const float ylo = g.sh, yhi = g.sh + (float)PH * g.bh, xlo = g.sw, xhi = g.sw + (float)PW * g.bw;
                hit = g.b == b && (float)y >= ylo - 2.f && (float)y <= yhi + 2.f && (float)x1 >= xlo - 2.f &&
                      (float)x0 <= xhi + 2.f;
            }
            unsigned long long m = __ballot(hit);
            while (m) {
                const int nb = n0 + __ffsll((long long)m) - 1;
                m &= m - 1;
                const AlignGeo g = align_geo(rois + 5 * nb, spatial_scale, PH, PW, sampling_ratio, aligned);
                const float *gv = gout + (size_t)nb * ld_g + (size_t)c * (PH * PW);
                for (int ph = 0; ph < PH; ++ph)
                    for (int iy = 0; iy < g.gh; ++iy) {
                        const float yy = align_coord(g.sh, ph, g.bh, iy, g.gh);
                        if (yy < -1.f || yy > (float)H) continue;
                        const Bilin qy = bilin_setup(yy, 0.f, H, W);
                        const float wy = (qy.yl == y ? qy.hy : 0.f) + (qy.yh == y ? qy.ly : 0.f);
                        if (qy.yl != y && qy.yh != y) continue;
                        for (int pw = 0; pw < PW; ++pw) {
                            const float gval = gv[ph * PW + pw] * g.inv_count;
                            for (int ix = 0; ix < g.gw; ++ix) {
                                const float xx = align_coord(g.sw, pw, g.bw, ix, g.gw);
                                const Bilin q = bilin_setup(yy, xx, H, W);
                                if (!q.ok) continue;
                                if (q.xl >= x0 && q.xl < x1) acc[(q.xl - x0) * 64 + lane] += (wy * q.hx) * gval;
                                if (q.xh >= x0 && q.xh < x1) acc[(q.xh - x0) * 64 + lane] += (wy * q.lx) * gval;
                            }
                        }
                    }
            }
        }
        const size_t row = ((size_t)b * H + y) * W;
#pragma unroll
        for (int t = 0; t < ROI_XW / 4; ++t) {
            const int x = x0 + t * 4 + ps;
            if (x < x1)
                *reinterpret_cast<float4 *>(gfeat + (row + x) * C + c4) =
                    *reinterpret_cast<const float4 *>(acc + (t * 4 + ps) * 64 + 4 * (lane & 15));
        }
    }
}

// ------------------------------------------------------------------------------------ bbox
// raw = [x1, y1, w, h, w/h]; z = raw W^T + b  (models.py:134-144 up to the Linear)
__global__ void bbox_linear_fwd_kernel(const float *__restrict__ bboxes,
                                       const float *__restrict__ Wt /*[Hd,5]*/,
                                       const float *__restrict__ bias, float *__restrict__ raw /*[N,5]*/,
                                       float *__restrict__ z /*[N,Hd]*/, int N, int Hd)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * Hd) return;
    const int n = i / Hd, j = i - n * Hd;
    const float *bb = bboxes + 5 * n;
    const float x1 = bb[1], y1 = bb[2];
    const float w = bb[3] - x1, h = bb[4] - y1;
    const float f[5] = {x1, y1, w, h, w / h};
    if (j == 0) {
#pragma unroll
        for (int k = 0; k < 5; ++k) raw[5 * n + k] = f[k];
    }
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 5; ++k) acc += f[k] * Wt[j * 5 + k];
    z[i] = acc + bias[j];
}

// dW[j][k] = sum_n dz[n][j]*raw[n][k], db[j] = sum_n dz[n][j]; one block per j
__global__ __launch_bounds__(256) void bbox_linear_bwd_kernel(const float *__restrict__ dz,
                                                              const float *__restrict__ raw,
                                                              float *__restrict__ dW,
                                                              float *__restrict__ db, int N, int Hd)
{
    __shared__ float s[6][256];
    const int j = blockIdx.x;
    float a[6] = {0, 0, 0, 0, 0, 0};
    for (int n = threadIdx.x; n < N; n += 256) {
        const float g = dz[(size_t)n * Hd + j];
#pragma unroll
        for (int k = 0; k < 5; ++k) a[k] += g * raw[5 * n + k];
        a[5] += g;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) s[k][threadIdx.x] = a[k];
    __syncthreads();
    if (threadIdx.x < 6) {
        float t = 0.f;
        for (int i = 0; i < 256; ++i) t += s[threadIdx.x][i];
        if (threadIdx.x < 5) dW[j * 5 + threadIdx.x] = t;
        else db[j] = t;
    }
}

// ------------------------------------------------------------------------------------ GAT
// s[n] = a[0:D].Wh[n][0:D] + b ; t[n] = a[D:2D].Wh[n][D:2D]       (one wave per node)
__global__ __launch_bounds__(256) void gat_scores_kernel(const float *__restrict__ Wh, int ldw,
                                                         const float *__restrict__ att_w,
                                                         const float *__restrict__ att_b,
                                                         float *__restrict__ s, float *__restrict__ t,
                                                         int N, int D)
{
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    const float *row = Wh + (size_t)n * ldw;
    float a = 0.f, b = 0.f;
    for (int d = lane; d < D; d += 64) {
        a += att_w[d] * row[d];
        b += att_w[D + d] * row[D + d];
    }
    a = wave_sum(a);
    b = wave_sum(b);
    if (lane == 0) { s[n] = a + att_b[0]; t[n] = b; }
}

// KP = number of 64-slot passes a wave makes over the K neighbour slots (K <= 64 KP; KP = 1 is the reference's usual
// range, -cs <= 32); slot k lives in lane k & 63 of pass k >> 6.  models.py:171-177 accepts any n_context.
template <int KP>
__global__ __launch_bounds__(256) void gat_fwd_kernel(
    const float *__restrict__ Wh, int ldw, const float *__restrict__ s, const float *__restrict__ t,
    const int64_t *__restrict__ ctx, int N, int K, int D, float slope, float *__restrict__ attn,
    float *__restrict__ hprime, int ldh)
{
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    int jj[KP];
    float e[KP];
    float m = -INFINITY;
#pragma unroll
    for (int p = 0; p < KP; ++p) {
        const int k = 64 * p + lane;
        long long j = -1;
        e[p] = -INFINITY;
        if (k < K) {
            j = ctx[(size_t)n * K + k];
            if (j >= N) j = -1;                                 // out-of-range id: memory-safe, acts as a pad
            const float u = s[n] + (j >= 0 ? t[j] : 0.f);
            const float lr = u > 0.f ? u : slope * u;
            e[p] = j >= 0 ? lr : -9e15f;                       // models.py:202-203
        }
        jj[p] = (int)j;
        m = fmaxf(m, e[p]);
    }
    m = wave_max(m);
    float pr[KP], psum = 0.f;
#pragma unroll
    for (int p = 0; p < KP; ++p) {
        pr[p] = 64 * p + lane < K ? expf(e[p] - m) : 0.f;
        psum += pr[p];
    }
    const float denom = wave_sum(psum);
    float aw[KP];
#pragma unroll
    for (int p = 0; p < KP; ++p) {
        const float alpha = pr[p] / denom;
        if (64 * p + lane < K) attn[(size_t)n * K + 64 * p + lane] = alpha;
        aw[p] = jj[p] >= 0 ? alpha : 0.f;                      // pads: weight 0 on a valid (clamped) row
    }
    for (int d0 = 0; d0 < D; d0 += 64) {
        const int d = d0 + lane, dd = d < D ? d : 0;
        float acc = 0.f;
#pragma unroll
        for (int p = 0; p < KP; ++p) {
            const int Kp = min(64, K - 64 * p);                 // slots of this pass
            for (int k0 = 0; k0 < Kp; k0 += 8) {                // 8 neighbour rows in flight, added in slot order
                float v[8], a[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int kk = min(k0 + u, Kp - 1);
                    const int jk = __shfl(jj[p], kk, 64);
                    a[u] = k0 + u < Kp ? __shfl(aw[p], kk, 64) : 0.f;
                    v[u] = Wh[(size_t)(jk >= 0 ? jk : 0) * ldw + D + dd];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = a[u] != 0.f ? fmaf(a[u], v[u], acc) : acc;
            }
        }
        if (d < D) hprime[(size_t)n * ldh + d] = acc;
    }
}

// K <= 64 (the reference's range), D <= 64 ND: the same wave-per-node arithmetic as gat_fwd_kernel<1> with the loops of the
// aggregation exchanged -- neighbour batches outside, the ND 64-channel chunks of a row inside -- so that one batch has
// 8 ND loads in flight instead of 8 and a node makes ceil(K / 8) dependent round trips instead of ceil(K / 8) * D / 64
// (18 -> 3 at configs[1]: 18.7 us of latency).  Every output still adds its neighbours in slot order: identical bits.
template <int ND>
__global__ __launch_bounds__(256) void gat_fwd_wide_kernel(
    const float *__restrict__ Wh, int ldw, const float *__restrict__ s, const float *__restrict__ t,
    const int64_t *__restrict__ ctx, int N, int K, int D, float slope, float *__restrict__ attn,
    float *__restrict__ hprime, int ldh)
{
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    long long j = -1;
    float e = -INFINITY;
    if (lane < K) {
        j = ctx[(size_t)n * K + lane];
        if (j >= N) j = -1;                                 // out-of-range id: memory-safe, acts as a pad
        const float u = s[n] + (j >= 0 ? t[j] : 0.f);
        const float lr = u > 0.f ? u : slope * u;
        e = j >= 0 ? lr : -9e15f;                           // models.py:202-203
    }
    const int jj = (int)j;
    const float m = wave_max(e);
    const float pr = lane < K ? expf(e - m) : 0.f;
    float psum = 0.f;
    psum += pr;
    const float denom = wave_sum(psum);
    const float alpha = pr / denom;
    if (lane < K) attn[(size_t)n * K + lane] = alpha;
    const float aw = jj >= 0 ? alpha : 0.f;                 // pads: weight 0 on a valid (clamped) row
    float acc[ND];
    int dd[ND];
#pragma unroll
    for (int c = 0; c < ND; ++c) {
        acc[c] = 0.f;
        dd[c] = 64 * c + lane < D ? 64 * c + lane : 0;
    }
    for (int k0 = 0; k0 < K; k0 += 8) {                     // 8 neighbour rows x ND chunks in flight, added in slot order
        float v[ND][8], a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int kk = min(k0 + u, K - 1);
            const int jk = __shfl(jj, kk, 64);
            a[u] = k0 + u < K ? __shfl(aw, kk, 64) : 0.f;
            const float *row = Wh + (size_t)(jk >= 0 ? jk : 0) * ldw + D;
#pragma unroll
            for (int c = 0; c < ND; ++c) v[c][u] = row[dd[c]];
        }
#pragma unroll
        for (int c = 0; c < ND; ++c)
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[c] = a[u] != 0.f ? fmaf(a[u], v[c][u], acc[c]) : acc[c];
    }
#pragma unroll
    for (int c = 0; c < ND; ++c)
        if (64 * c + lane < D) hprime[(size_t)n * ldh + 64 * c + lane] = acc[c];
}

// backward of the sparse part.  g = dL/dh' [N, D] (ld = ldg).
//   dalpha_k = g . Wh_j[ctx_k];  de = alpha*(dalpha - sum alpha*dalpha) (0 on masked slots);
//   du = de * LeakyReLU'(u);  ds_i = sum_k du_k;  dt[ctx_k] += du_k;
//   dWh_i[i] = ds_i * a_i;  dWh_j[ctx_k] += alpha_k*g_i (+ dt_j*a_j added by gat_bwd_finish)
// dWh [N, 2D] must be zeroed in its second half (and dt zeroed) before the launch.
template <int KP>
__global__ __launch_bounds__(256) void gat_bwd_kernel(
    const float *__restrict__ g, int ldg, const float *__restrict__ Wh, int ldw,
    const float *__restrict__ s, const float *__restrict__ t, const float *__restrict__ attn,
    const int64_t *__restrict__ ctx, const float *__restrict__ att_w, int N, int K, int D,
    float slope, float *__restrict__ dWh, int lddw, float *__restrict__ ds, float *__restrict__ dt)
{
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    int jj[KP];
    float alpha[KP], dalpha[KP];
#pragma unroll
    for (int p = 0; p < KP; ++p) {
        const int k = 64 * p + lane;
        long long j = -1;
        alpha[p] = 0.f;
        dalpha[p] = 0.f;
        if (k < K) {
            j = ctx[(size_t)n * K + k];
            if (j >= N) j = -1;
            alpha[p] = attn[(size_t)n * K + k];
        }
        jj[p] = (int)j;
    }
    // dalpha for every slot: lanes over channels, one wave reduction per slot
#pragma unroll
    for (int p = 0; p < KP; ++p)
        for (int k = 0; k < min(64, K - 64 * p); ++k) {
            const int jk = __shfl(jj[p], k, 64);
            float part = 0.f;
            if (jk >= 0)
                for (int d = lane; d < D; d += 64)
                    part += g[(size_t)n * ldg + d] * Wh[(size_t)jk * ldw + D + d];
            part = wave_sum(part);
            if (lane == k) dalpha[p] = part;
        }
    float ad = 0.f;
#pragma unroll
    for (int p = 0; p < KP; ++p) ad += alpha[p] * dalpha[p];
    const float dot = wave_sum(ad);
    float dus = 0.f;
#pragma unroll
    for (int p = 0; p < KP; ++p) {
        float du = 0.f;
        if (64 * p + lane < K && jj[p] >= 0) {
            const float de = alpha[p] * (dalpha[p] - dot);
            const float u = s[n] + t[jj[p]];
            du = de * (u > 0.f ? 1.f : slope);
            atomicAdd(dt + jj[p], du);
        }
        dus += du;
    }
    const float dsn = wave_sum(dus);
    if (lane == 0) ds[n] = dsn;
    for (int d = lane; d < D; d += 64) dWh[(size_t)n * lddw + d] = dsn * att_w[d];
#pragma unroll
    for (int p = 0; p < KP; ++p)
        for (int k = 0; k < min(64, K - 64 * p); ++k) {
            const int jk = __shfl(jj[p], k, 64);
            const float ak = __shfl(alpha[p], k, 64);
            if (jk < 0) continue;
            for (int d = lane; d < D; d += 64)
                atomicAdd(dWh + (size_t)jk * lddw + D + d, ak * g[(size_t)n * ldg + d]);
        }
}

// ---- transposed neighbour index (CSR over destination nodes): lets the backward GATHER what the
// reference's autograd scatters (index_select backward), with a fixed summation order -> no float atomics,
// bit-identical reruns, for ARBITRARY context_indices (models.py:171-177 accepts any ids), not only the
// +-context_size windows the dataset builds (datasets.py:121-128).
//   deg[j] = #{(i,k): ctx[i,k] = j};  row_ptr = exclusive scan;  edges of row j = the flat slots e = i*K + k,
//   ascending (the integer atomics below only decide a provisional slot; every row is then rank-sorted).
__global__ void csr_count_kernel(const int64_t *__restrict__ ctx, int E, int N, int *__restrict__ deg)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const long long j = ctx[e];
    if (j >= 0 && j < N) atomicAdd(deg + j, 1);
}

__global__ __launch_bounds__(1024) void csr_scan_kernel(const int *__restrict__ deg, int N, int *__restrict__ row_ptr)
{
    __shared__ int s_part[1024];
    const int tid = threadIdx.x;
    const int per = (N + 1023) / 1024, lo = min(tid * per, N), hi = min(lo + per, N);
    int t = 0;
    for (int i = lo; i < hi; ++i) t += deg[i];
    s_part[tid] = t;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int i = 0; i < 1024; ++i) { const int v = s_part[i]; s_part[i] = run; run += v; }
        row_ptr[N] = run;
    }
    __syncthreads();
    int run = s_part[tid];
    for (int i = lo; i < hi; ++i) { row_ptr[i] = run; run += deg[i]; }
}

__global__ void csr_fill_kernel(const int64_t *__restrict__ ctx, int E, int N, const int *__restrict__ row_ptr,
                                int *__restrict__ cursor, int *__restrict__ tmp)
{
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= E) return;
    const long long j = ctx[e];
    if (j >= 0 && j < N) tmp[row_ptr[j] + atomicAdd(cursor + j, 1)] = e;
}

// one wave per row: rank every entry among the row's entries (entries are distinct) and store it at its rank
__global__ __launch_bounds__(256) void csr_sort_rows_kernel(const int *__restrict__ row_ptr, const int *__restrict__ tmp,
                                                            int N, int *__restrict__ edges, int *__restrict__ clean)
{
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= N) return;
    if (clean != nullptr && lane < 2) clean[lane * N + j] = 0;      // deg[j], cursor[j]: zero again for the next call
    const int lo = row_ptr[j], deg = row_ptr[j + 1] - lo;
    for (int c0 = 0; c0 < deg; c0 += 64) {
        const int mine = c0 + lane < deg ? tmp[lo + c0 + lane] : 0x7fffffff;
        int rank = 0;
        for (int d0 = 0; d0 < deg; d0 += 64) {
            const int other = d0 + lane < deg ? tmp[lo + d0 + lane] : 0x7fffffff;
            const int cnt = min(64, deg - d0);
            for (int t = 0; t < cnt; ++t) rank += __shfl(other, t, 64) < mine ? 1 : 0;
        }
        if (c0 + lane < deg) edges[lo + rank] = mine;
    }
}

// Three launches instead of seven for a workspace that is REUSED from call to call (cova_gat_transpose_reuse): its
// counters are zero on entry and left zero (the row sort clears them), so no memsets; and the exclusive scan is done by
// the counting kernel's last block (ticket counter; agent-scope atomic loads of the degrees, as bn_tail.h).
__global__ __launch_bounds__(256) void csr_count_scan_kernel(const int64_t *__restrict__ ctx, int E, int N,
                                                             int *__restrict__ deg, int *__restrict__ ticket,
                                                             int *__restrict__ row_ptr)
{
    __shared__ int s_part[256];
    __shared__ int s_last;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < E) {
        const long long j = ctx[e];
        if (j >= 0 && j < N) atomicAdd(deg + j, 1);
    }
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (threadIdx.x == 0)
        s_last = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1;
    __syncthreads();
    if (!s_last) return;
    const int tid = threadIdx.x;
    const int per = (N + 255) / 256, lo = min(tid * per, N), hi = min(lo + per, N);
    int t = 0;
    for (int i = lo; i < hi; ++i) t += __hip_atomic_load(deg + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_part[tid] = t;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int i = 0; i < 256; ++i) { const int v = s_part[i]; s_part[i] = run; run += v; }
        row_ptr[N] = run;
        __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    int run = s_part[tid];
    for (int i = lo; i < hi; ++i) {
        row_ptr[i] = run;
        run += __hip_atomic_load(deg + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// backward, source side (one wave per node i): everything that stays with node i -- du [N,K] (0 on pads),
// ds, dWh_i = ds * a_i.  The contributions to OTHER nodes (dt[ctx], dWh_j[ctx]) are gathered by
// gat_bwd_dst_kernel from du / attn / g through the transposed index.
template <int KP>
__global__ __launch_bounds__(256) void gat_bwd_src_kernel(
    const float *__restrict__ g, int ldg, const float *__restrict__ Wh, int ldw,
    const float *__restrict__ s, const float *__restrict__ t, const float *__restrict__ attn,
    const int64_t *__restrict__ ctx, const float *__restrict__ att_w, int N, int K, int D,
    float slope, float *__restrict__ dWh, int lddw, float *__restrict__ ds, float *__restrict__ du_out)
{
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    int jj[KP];
    float alpha[KP], dalpha[KP];
#pragma unroll
    for (int p = 0; p < KP; ++p) {
        const int k = 64 * p + lane;
        long long j = -1;
        alpha[p] = 0.f;
        dalpha[p] = 0.f;
        if (k < K) {
            j = ctx[(size_t)n * K + k];
            if (j >= N) j = -1;
            alpha[p] = attn[(size_t)n * K + k];
        }
        jj[p] = (int)j;
    }
#pragma unroll
    for (int p = 0; p < KP; ++p) {
        const int Kp = min(64, K - 64 * p);
        for (int k0 = 0; k0 < Kp; k0 += 4) {                   // four neighbour rows in flight per channel slice
            int jk[4];
            float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < 4; ++u) jk[u] = __shfl(jj[p], min(k0 + u, Kp - 1), 64);
            for (int d = lane; d < D; d += 64) {
                const float gv = g[(size_t)n * ldg + d];
                float w[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) w[u] = Wh[(size_t)(jk[u] >= 0 ? jk[u] : 0) * ldw + D + d];
#pragma unroll
                for (int u = 0; u < 4; ++u) part[u] = fmaf(gv, w[u], part[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float tot = wave_sum(part[u]);
                if (lane == k0 + u && jk[u] >= 0) dalpha[p] = tot;
            }
        }
    }
    float ad = 0.f;
#pragma unroll
    for (int p = 0; p < KP; ++p) ad += alpha[p] * dalpha[p];
    const float dot = wave_sum(ad);
    float dus = 0.f;
#pragma unroll
    for (int p = 0; p < KP; ++p) {
        float du = 0.f;
        if (64 * p + lane < K && jj[p] >= 0) {
            const float de = alpha[p] * (dalpha[p] - dot);
            const float u = s[n] + t[jj[p]];
            du = de * (u > 0.f ? 1.f : slope);
        }
        if (64 * p + lane < K) du_out[(size_t)n * K + 64 * p + lane] = du;
        dus += du;
    }
    const float dsn = wave_sum(dus);
    if (lane == 0) ds[n] = dsn;
    for (int d = lane; d < D; d += 64) dWh[(size_t)n * lddw + d] = dsn * att_w[d];
}

// backward, destination side (one wave per node j): dt[j] = sum_e du[e];
// dWh_j[j] = sum_e alpha[e] * g[i_e] + dt[j] * a_j, e over row j of the transposed index, ascending
__global__ __launch_bounds__(256) void gat_bwd_dst_kernel(
    const float *__restrict__ g, int ldg, const float *__restrict__ attn, const float *__restrict__ du,
    const int *__restrict__ row_ptr, const int *__restrict__ edges, const float *__restrict__ att_w, int N,
    int K, int D, float *__restrict__ dWh, int lddw, float *__restrict__ dt)
{
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= N) return;
    const int lo = row_ptr[j], deg = row_ptr[j + 1] - lo;
    // dt[j]: lanes over the row's edges, butterfly sums -- a fixed association, independent of timing
    float dtj = 0.f;
    for (int c0 = 0; c0 < deg; c0 += 64) dtj += wave_sum(c0 + lane < deg ? du[edges[lo + c0 + lane]] : 0.f);
    if (lane == 0) dt[j] = dtj;
    for (int d0 = 0; d0 < D; d0 += 64) {
        const int d = d0 + lane;
        const int dd = d < D ? d : 0;
        float accv = 0.f;
        for (int c0 = 0; c0 < deg; c0 += 64) {
            const int e = c0 + lane < deg ? edges[lo + c0 + lane] : 0;
            const float al = c0 + lane < deg ? attn[e] : 0.f;     // 0 weight on the padding lanes
            const int cnt = min(64, deg - c0);
            for (int q0 = 0; q0 < cnt; q0 += 8) {                  // 8 neighbour rows in flight, added in edge order
                float gv[8], aq[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int qq = min(q0 + u, 63);
                    const int i = __shfl(e, qq, 64) / K;
                    aq[u] = q0 + u < cnt ? __shfl(al, qq, 64) : 0.f;
                    const float gl = g[(size_t)i * ldg + dd];      // (unconditional load; padding slots read row 0 ...)
                    gv[u] = q0 + u < cnt ? gl : 0.f;               // ... and must not turn a non-finite g[0] into NaN: 0 * inf
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) accv = fmaf(aq[u], gv[u], accv);
            }
        }
        if (d < D) dWh[(size_t)j * lddw + D + d] = accv + dtj * att_w[D + d];
    }
}

// gat_bwd_src_kernel<1> / gat_bwd_dst_kernel with every 64-channel chunk of a row in flight per neighbour batch (see
// gat_fwd_wide_kernel): K <= 64, D <= 64 ND; identical bits (each sum keeps its order: channels ascending inside a dot
// product, edges ascending inside a gathered row).
template <int ND>
__global__ __launch_bounds__(256) void gat_bwd_src_wide_kernel(
    const float *__restrict__ g, int ldg, const float *__restrict__ Wh, int ldw,
    const float *__restrict__ s, const float *__restrict__ t, const float *__restrict__ attn,
    const int64_t *__restrict__ ctx, const float *__restrict__ att_w, int N, int K, int D,
    float slope, float *__restrict__ dWh, int lddw, float *__restrict__ ds, float *__restrict__ du_out)
{
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    long long j = -1;
    float alpha = 0.f, dalpha = 0.f;
    if (lane < K) {
        j = ctx[(size_t)n * K + lane];
        if (j >= N) j = -1;
        alpha = attn[(size_t)n * K + lane];
    }
    const int jj = (int)j;
    float gv[ND];
    int dd[ND];
#pragma unroll
    for (int c = 0; c < ND; ++c) {
        const bool in = 64 * c + lane < D;
        dd[c] = in ? 64 * c + lane : 0;
        gv[c] = in ? g[(size_t)n * ldg + dd[c]] : 0.f;       // (channels past D: weight 0 on a valid address)
    }
    for (int k0 = 0; k0 < K; k0 += 4) {                      // four neighbour rows x ND chunks in flight
        int jk[4];
        float w[ND][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            jk[u] = __shfl(jj, min(k0 + u, K - 1), 64);
            const float *row = Wh + (size_t)(jk[u] >= 0 ? jk[u] : 0) * ldw + D;
#pragma unroll
            for (int c = 0; c < ND; ++c) w[c][u] = row[dd[c]];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float part = 0.f;
#pragma unroll
            for (int c = 0; c < ND; ++c)
                part = 64 * c + lane < D ? fmaf(gv[c], w[c][u], part) : part;      // (the channels gat_bwd_src_kernel's loop visits)
            const float tot = wave_sum(part);
            if (lane == k0 + u && jk[u] >= 0) dalpha = tot;
        }
    }
    // (the sum below is gat_bwd_src_kernel's statement for statement: written as wave_sum(alpha * dalpha), hipcc's default
    // -ffp-contract=fast fuses the product into the first add of the reduction -- an unrounded product, and for K > 32,
    // where the partner lane holds a value, other bits than the chunk kernel's)
    float ad = 0.f;
    ad += alpha * dalpha;
    const float dot = wave_sum(ad);
    float du = 0.f;
    if (lane < K && jj >= 0) {
        const float de = alpha * (dalpha - dot);
        const float u = s[n] + t[jj];
        du = de * (u > 0.f ? 1.f : slope);
    }
    if (lane < K) du_out[(size_t)n * K + lane] = du;
    float dus = 0.f;
    dus += du;
    const float dsn = wave_sum(dus);
    if (lane == 0) ds[n] = dsn;
    for (int d = lane; d < D; d += 64) dWh[(size_t)n * lddw + d] = dsn * att_w[d];
}

template <int ND>
__global__ __launch_bounds__(256) void gat_bwd_dst_wide_kernel(
    const float *__restrict__ g, int ldg, const float *__restrict__ attn, const float *__restrict__ du,
    const int *__restrict__ row_ptr, const int *__restrict__ edges, const float *__restrict__ att_w, int N,
    int K, int D, float *__restrict__ dWh, int lddw, float *__restrict__ dt)
{
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= N) return;
    const int lo = row_ptr[j], deg = row_ptr[j + 1] - lo;
    float dtj = 0.f;
    for (int c0 = 0; c0 < deg; c0 += 64) dtj += wave_sum(c0 + lane < deg ? du[edges[lo + c0 + lane]] : 0.f);
    if (lane == 0) dt[j] = dtj;
    float accv[ND];
    int dd[ND];
#pragma unroll
    for (int c = 0; c < ND; ++c) {
        accv[c] = 0.f;
        dd[c] = 64 * c + lane < D ? 64 * c + lane : 0;
    }
    for (int c0 = 0; c0 < deg; c0 += 64) {
        const int e = c0 + lane < deg ? edges[lo + c0 + lane] : 0;
        const float al = c0 + lane < deg ? attn[e] : 0.f;         // 0 weight on the padding lanes
        const int cnt = min(64, deg - c0);
        for (int q0 = 0; q0 < cnt; q0 += 8) {                      // 8 neighbour rows x ND chunks in flight, added in edge order
            float gq[ND][8], aq[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int qq = min(q0 + u, 63);
                const int i = __shfl(e, qq, 64) / K;
                const bool in = q0 + u < cnt;
                aq[u] = in ? __shfl(al, qq, 64) : 0.f;
                const float *row = g + (size_t)i * ldg;
#pragma unroll
                for (int c = 0; c < ND; ++c) {
                    const float gl = row[dd[c]];                   // (unconditional load; padding slots read row 0 ...)
                    gq[c][u] = in ? gl : 0.f;                      // ... and must not turn a non-finite g[0] into NaN: 0 * inf
                }
            }
#pragma unroll
            for (int c = 0; c < ND; ++c)
#pragma unroll
                for (int u = 0; u < 8; ++u) accv[c] = fmaf(aq[u], gq[c][u], accv[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < ND; ++c)
        if (64 * c + lane < D) dWh[(size_t)j * lddw + D + 64 * c + lane] = accv[c] + dtj * att_w[D + 64 * c + lane];
}

// dWh_j[n] += dt[n]*a_j  (elementwise), and the attention-vector gradients
//   d att_w[d] = sum_n ds[n]*Wh[n][d] (d < D), sum_n dt[n]*Wh[n][d] (d >= D); d att_b = sum ds
__global__ void gat_bwd_addt_kernel(float *__restrict__ dWh, int lddw, const float *__restrict__ dt,
                                    const float *__restrict__ att_w, int N, int D)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * D) return;
    const int n = i / D, d = i - n * D;
    dWh[(size_t)n * lddw + D + d] += dt[n] * att_w[D + d];
}

// block = 64 columns; V = 4: a thread owns 4 adjacent columns (float4 per row) and the block walks 64 row slices
// (V = 1: 16 slices, any D); the extra block past the 2D columns sums ds (the bias gradient)
template <int V>
__global__ __launch_bounds__(1024) void gat_bwd_att_kernel(const float *__restrict__ Wh, int ldw,
                                                           const float *__restrict__ ds,
                                                           const float *__restrict__ dt,
                                                           float *__restrict__ d_att_w,
                                                           float *__restrict__ d_att_b, int N, int D)
{
    constexpr int TX = 64 / V, NS = 1024 / TX;
    __shared__ float s_acc[NS][64];
    const int tx = threadIdx.x % TX, slice = threadIdx.x / TX;
    const int col = blockIdx.x * 64 + tx * V;     // 0 .. 2D (+ the bias column)
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (col < 2 * D) {
        const float *w = col < D ? ds : dt;       // (D % V == 0: a thread's columns lie on one side)
        for (int n = slice; n < N; n += NS) {
            const float wn = w[n];
            if (V == 4) {
                const float4 v = *reinterpret_cast<const float4 *>(Wh + (size_t)n * ldw + col);
                acc[0] += wn * v.x; acc[1] += wn * v.y; acc[2] += wn * v.z; acc[3] += wn * v.w;
            } else {
                acc[0] += wn * Wh[(size_t)n * ldw + col];
            }
        }
    } else if (col == 2 * D) {
        for (int n = slice; n < N; n += NS) acc[0] += ds[n];
    }
#pragma unroll
    for (int j = 0; j < V; ++j) s_acc[slice][tx * V + j] = acc[j];
    __syncthreads();
    const int oc = blockIdx.x * 64 + threadIdx.x;
    if (threadIdx.x < 64 && oc <= 2 * D) {
        float t = 0.f;
        for (int j = 0; j < NS; ++j) t += s_acc[j][threadIdx.x];
        if (oc == 2 * D) d_att_b[0] = t;
        else d_att_w[oc] = t;
    }
}

// 64-channel chunks per row the wide GAT kernels are instantiated for (0: D too large, or switched off by
// cova_set_option(16, 0): the chunk-at-a-time kernels)
int g_gat_wide = 1;
inline int gat_wide_nd(int D)
{
    const int nd = (D + 63) / 64;
    if (!g_gat_wide || nd > 8) return 0;
    return nd <= 4 ? nd : (nd <= 6 ? 6 : 8);
}

}  // namespace

int cova_internal_set_gat_wide(int v) { g_gat_wide = v != 0; return COVA_OK; }
int cova_internal_get_gat_wide() { return (int)g_gat_wide; }

// ====================================================================================
// C ABI
// ====================================================================================
// feat NHWC [B,H,W,C]; rois [N,5]; out row n at out + n*ld_out, C*PH*PW entries in the
// reference's channel-major order; argmax [N, C*PH*PW] int32 (h*W + w, or -1)
COVA_API int cova_roipool_fwd(const float *feat, const float *rois, int n_rois, int B, int C, int H,
                              int W, int PH, int PW, float spatial_scale, float *out, int ld_out,
                              int32_t *argmax, void *stream)
{
    COVA_REQUIRE(feat && rois && out && argmax && n_rois >= 0 && B > 0 && C > 0 && C % 64 == 0 && PH > 0 && PW > 0);
    if (n_rois == 0) return COVA_OK;
#define COVA_ROIPOOL_FWD(LAZY_, U_, XCD_, F_, ZM_, LZ_)                                                                     \
    hipLaunchKernelGGL((roipool_fwd_kernel<LAZY_, U_, XCD_>), dim3(cdiv(n_rois * PH * PW, 4)), dim3(256), 0,              \
                       (hipStream_t)stream, F_, rois, n_rois, B, C, H, W, PH, PW, spatial_scale, out, ld_out, argmax, ZM_, LZ_)
    {
        const LazyFeat none{nullptr, nullptr, nullptr};
        // (wider maps: the channel-block loop multiplies the trips, the launch is bandwidth bound and loses occupancy with eight
        // loads per lane -- configs[2]: 1.51 ms with variant 0, 1.59 with 3; the default applies to 64-channel maps only)
        if (C > 64) COVA_ROIPOOL_FWD(false, 4, false, feat, nullptr, none);
        else COVA_ROIPOOL_FWD(false, 8, true, feat, nullptr, none);
    }
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

// RoIPool over feat = relu(scale * z + shift + x) formed on the fly (z, x NHWC [B,H,W,C]; scale, shift
// [C]): the last BasicBlock's bn2 + residual + ReLU without the pass that would materialise it.
COVA_API int cova_roipool_fwd_bn(const float *z, const float *x, const float *scale,
                                 const float *shift, const float *rois, int n_rois, int B, int C,
                                 int H, int W, int PH, int PW, float spatial_scale, float *out,
                                 int ld_out, int32_t *argmax, float *zmax, void *stream)
{
    COVA_REQUIRE(z && x && scale && shift && rois && out && argmax && n_rois >= 0 && B > 0 && C > 0 && C % 64 == 0 && PH > 0 && PW > 0);
    if (n_rois == 0) return COVA_OK;
    {
        const LazyFeat lzf{x, scale, shift};
        // (wider maps: the channel-block loop multiplies the trips, the launch is bandwidth bound and loses occupancy with eight
        // loads per lane -- configs[2]: 1.51 ms with variant 0, 1.59 with 3; the default applies to 64-channel maps only)
        if (C > 64) COVA_ROIPOOL_FWD(true, 4, false, z, zmax, lzf);
        else COVA_ROIPOOL_FWD(true, 8, true, z, zmax, lzf);
    }
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

static int roipool_bwd_grid(int B, int H, int W)
{
    const long long ntask = ((long long)B * H * ((W + ROI_XW - 1) / ROI_XW) + 3) / 4;       // 4 owner waves per block
    return cova_internal_persistent_grid2(ntask > (1 << 30) ? (1 << 30) : (int)ntask, 4);
}

// gfeat NHWC [B,H,W,C] = sum over (box, bin) of gout routed to the arg-max positions; fully written here
// (no zero-fill, no atomics: every row of the map has one owner wave).  C % 64 == 0.
static int roipool_page_ranges(const float *rois, int n_rois, int B, int *range, hipStream_t st)
{
    if (n_rois <= (1 << 18)) {
        hipLaunchKernelGGL(roipool_page_range_block_kernel, dim3(1), dim3(1024), 0, st, rois, n_rois, B, range);
        COVA_LAUNCH_CHECK();
        return COVA_OK;
    }
    hipLaunchKernelGGL(roipool_page_range_init_kernel, dim3(cdiv(B, 256)), dim3(256), 0, st, B, range);
    COVA_LAUNCH_CHECK();
    hipLaunchKernelGGL(roipool_page_range_kernel, dim3(cdiv(n_rois, 256)), dim3(256), 0, st, rois, n_rois, B, range);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

COVA_API int cova_roipool_bwd_bn_num_partials(int n_rois)
{
    const int g = cdiv(n_rois > 0 ? n_rois : 1, 4);
    return g < 256 ? g : 256;          // (64 blocks left the entry pass latency bound: 40 us for 1 440 boxes)
}

// 4-byte words of workspace of cova_roipool_bwd / cova_roipool_bwd_bn: transposed contributions + positions, page ranges
COVA_API int cova_roipool_bwd_workspace_words(int n_rois, int B, int C, int PH, int PW)
{
    return 2 * n_rois * C * PH * PW + 2 * B + 16;
}

static int launch_roipool_bwd(const float *gout, int ld_g, const float *pooled, int ld_p, const float *zmax,
                              const float *rois, const int32_t *argmax, int n_rois, int B, int C, int H, int W,
                              int PH, int PW, float spatial_scale, const float *mean, const float *invstd,
                              float *gfeat, float *partial, void *ws, hipStream_t st, const cova_bn_tail *tail = nullptr)
{
    const size_t ne = (size_t)n_rois * C * PH * PW;
    float *gT = (float *)ws;
    int32_t *amT = (int32_t *)ws + ne;
    int *page_range = (int *)ws + 2 * ne;
    const dim3 pgrid(cova_roipool_bwd_bn_num_partials(n_rois), C / 64);
    BnTail t{};
    if (tail != nullptr && tail->mode != 0) {
        if (!(partial && C == 64 && tail->mode == 2)) return COVA_ERR_BAD_ARG;
        t = *tail;
    }
    if (PH * PW == 9 && t.mode != 0)
        hipLaunchKernelGGL((roipool_bwd_prep_kernel<true, true, 9>), pgrid, dim3(256), 0, st, gout, ld_g, pooled, ld_p, zmax,
                           argmax, n_rois, C, 9, mean, invstd, gT, amT, partial, rois, B, page_range, t);
    else if (PH * PW == 9 && partial)
        hipLaunchKernelGGL((roipool_bwd_prep_kernel<true, false, 9>), pgrid, dim3(256), 0, st, gout, ld_g, pooled, ld_p, zmax,
                           argmax, n_rois, C, 9, mean, invstd, gT, amT, partial, rois, B, page_range, t);
    else if (PH * PW == 9)
        hipLaunchKernelGGL((roipool_bwd_prep_kernel<false, false, 9>), pgrid, dim3(256), 0, st, gout, ld_g, pooled, ld_p, zmax,
                           argmax, n_rois, C, 9, mean, invstd, gT, amT, partial, rois, B, page_range, t);
    else if (t.mode != 0)
        hipLaunchKernelGGL((roipool_bwd_prep_kernel<true, true>), pgrid, dim3(256), 0, st, gout, ld_g, pooled, ld_p, zmax,
                           argmax, n_rois, C, PH * PW, mean, invstd, gT, amT, partial, rois, B, page_range, t);
    else if (partial)
        hipLaunchKernelGGL((roipool_bwd_prep_kernel<true, false>), pgrid, dim3(256), 0, st, gout, ld_g, pooled, ld_p, zmax,
                           argmax, n_rois, C, PH * PW, mean, invstd, gT, amT, partial, rois, B, page_range, t);
    else
        hipLaunchKernelGGL((roipool_bwd_prep_kernel<false, false>), pgrid, dim3(256), 0, st, gout, ld_g, pooled, ld_p, zmax,
                           argmax, n_rois, C, PH * PW, mean, invstd, gT, amT, partial, rois, B, page_range, t);
    COVA_LAUNCH_CHECK();
    hipLaunchKernelGGL((PH == 3 && PW == 3 ? roipool_bwd_rows_kernel<true, 2>
                                           : roipool_bwd_rows_kernel<false>),
                       dim3(roipool_bwd_grid(B, H, W), C / 64), dim3(256), 0, st, gT, amT, rois, page_range, n_rois,
                       B, C, H, W, PH, PW, spatial_scale, gfeat);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

COVA_API int cova_roipool_bwd(const float *gout, int ld_g, const float *rois, const int32_t *argmax,
                              int n_rois, int B, int C, int H, int W, int PH, int PW,
                              float spatial_scale, float *gfeat, void *ws, void *stream)
{
    COVA_REQUIRE(gout && rois && argmax && gfeat && ws && B > 0 && n_rois >= 0 && C > 0 && C % 64 == 0);
    return launch_roipool_bwd(gout, ld_g, nullptr, 0, nullptr, rois, argmax, n_rois, B, C, H, W, PH, PW,
                              spatial_scale, nullptr, nullptr, gfeat, nullptr, ws, (hipStream_t)stream);
}

// cova_roipool_bwd for a map that is the output of relu(bn(z) + residual) (cova_roipool_fwd_bn): the routed
// gradient is masked by that ReLU -- pooled > 0, the pooled value being the map's value at the arg-max -- and the
// BatchNorm-backward partial sums [num_partials][2][C] of (g', g' * (zmax - mean) * invstd) are taken per pooled
// entry (zmax from cova_roipool_fwd_bn).  No map is read.  gfeat then holds the ReLU-masked gradient.
COVA_API int cova_roipool_bwd_bn(const float *gout, int ld_g, const float *pooled, int ld_p,
                                 const float *zmax, const float *rois, const int32_t *argmax, int n_rois,
                                 int B, int C, int H, int W, int PH, int PW, float spatial_scale,
                                 const float *mean, const float *invstd, float *gfeat, float *partial,
                                 void *ws, void *stream)
{
    COVA_REQUIRE(gout && pooled && zmax && rois && argmax && mean && invstd && gfeat && partial && ws);
    COVA_REQUIRE(B > 0 && n_rois >= 0 && C > 0 && C % 64 == 0);
    return launch_roipool_bwd(gout, ld_g, pooled, ld_p, zmax, rois, argmax, n_rois, B, C, H, W, PH, PW,
                              spatial_scale, mean, invstd, gfeat, partial, ws, (hipStream_t)stream);
}

// ... with the BatchNorm-backward finalize of those sums (cova_bn_finalize_bwd_abc) as the tail of the entry pass
// (tail: host pointer, mode 2, C = 64; NULL or mode 0 = cova_roipool_bwd_bn)
COVA_API int cova_roipool_bwd_bn_tail(const float *gout, int ld_g, const float *pooled, int ld_p,
                                      const float *zmax, const float *rois, const int32_t *argmax, int n_rois,
                                      int B, int C, int H, int W, int PH, int PW, float spatial_scale,
                                      const float *mean, const float *invstd, float *gfeat, float *partial,
                                      void *ws, const cova_bn_tail *tail, void *stream)
{
    COVA_REQUIRE(gout && pooled && zmax && rois && argmax && mean && invstd && gfeat && partial && ws);
    COVA_REQUIRE(B > 0 && n_rois >= 0 && C > 0 && C % 64 == 0);
    return launch_roipool_bwd(gout, ld_g, pooled, ld_p, zmax, rois, argmax, n_rois, B, C, H, W, PH, PW,
                              spatial_scale, mean, invstd, gfeat, partial, ws, (hipStream_t)stream, tail);
}

// RoIAlign (extension; see roialign_fwd_kernel): same tensor conventions as cova_roipool_fwd
COVA_API int cova_roialign_fwd(const float *feat, const float *rois, int n_rois, int B, int C, int H, int W,
                               int PH, int PW, float spatial_scale, int sampling_ratio, int aligned, float *out,
                               int ld_out, void *stream)
{
    COVA_REQUIRE(feat && rois && out && n_rois >= 0 && B > 0 && C > 0 && PH > 0 && PW > 0);
    if (n_rois == 0) return COVA_OK;
    hipLaunchKernelGGL(roialign_fwd_kernel, dim3(cdiv(n_rois * PH * PW, 4)), dim3(256), 0, (hipStream_t)stream,
                       feat, rois, n_rois, B, C, H, W, PH, PW, spatial_scale, sampling_ratio, aligned, out, ld_out);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

// gfeat NHWC [B,H,W,C], fully written, no atomics (one owner wave per feature row); ws >= 2*B + 16 ints
COVA_API int cova_roialign_bwd(const float *gout, int ld_g, const float *rois, int n_rois, int B, int C, int H,
                               int W, int PH, int PW, float spatial_scale, int sampling_ratio, int aligned,
                               float *gfeat, void *ws, void *stream)
{
    COVA_REQUIRE(gout && rois && gfeat && ws && B > 0 && n_rois >= 0 && C > 0 && C % 64 == 0);
    hipStream_t st = (hipStream_t)stream;
    int *page_range = (int *)ws;
    const int rc = roipool_page_ranges(rois, n_rois, B, page_range, st);
    if (rc != COVA_OK) return rc;
    hipLaunchKernelGGL(roialign_bwd_rows_kernel, dim3(roipool_bwd_grid(B, H, W), C / 64), dim3(256), 0, st, gout,
                       ld_g, rois, page_range, n_rois, B, C, H, W, PH, PW, spatial_scale, sampling_ratio, aligned,
                       gfeat);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

COVA_API int cova_bbox_linear_fwd(const float *bboxes, const float *W, const float *bias, float *raw,
                                  float *z, int N, int Hd, void *stream)
{
    COVA_REQUIRE(bboxes && W && bias && raw && z && N >= 0 && Hd > 0);
    if (N == 0) return COVA_OK;
    hipLaunchKernelGGL(bbox_linear_fwd_kernel, dim3(cdiv(N * Hd, 256)), dim3(256), 0,
                       (hipStream_t)stream, bboxes, W, bias, raw, z, N, Hd);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

COVA_API int cova_bbox_linear_bwd(const float *dz, const float *raw, float *dW, float *db, int N,
                                  int Hd, void *stream)
{
    COVA_REQUIRE(dz && raw && dW && db && Hd > 0);
    hipLaunchKernelGGL(bbox_linear_bwd_kernel, dim3(Hd), dim3(256), 0, (hipStream_t)stream, dz, raw,
                       dW, db, N, Hd);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

// Wh [N, 2D] (ld = ldw): first D columns W_i h, last D columns W_j h.
// Outputs: s, t [N]; attn [N, K]; hprime rows at hprime + n*ldh (D entries).
COVA_API int cova_gat_fwd(const float *Wh, int ldw, const float *att_w, const float *att_b,
                          const int64_t *ctx, int N, int K, int D, float slope, float *s, float *t,
                          float *attn, float *hprime, int ldh, void *stream)
{
    COVA_REQUIRE(Wh && att_w && att_b && ctx && s && t && attn && hprime && K > 0 && K <= COVA_GAT_MAX_K && D > 0);
    if (N == 0) return COVA_OK;
    hipLaunchKernelGGL(gat_scores_kernel, dim3(cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, Wh,
                       ldw, att_w, att_b, s, t, N, D);
    COVA_LAUNCH_CHECK();
    const dim3 grid(cdiv(N, 4)), blk(256);
    hipStream_t st = (hipStream_t)stream;
    if (K <= 64 && gat_wide_nd(D) > 0) {        // every 64-channel chunk of a neighbour row in flight (identical bits)
#define COVA_GAT_FWD_WIDE(ND) hipLaunchKernelGGL(gat_fwd_wide_kernel<ND>, grid, blk, 0, st, Wh, ldw, s, t, ctx, N, K, D, slope, attn, hprime, ldh)
        switch (gat_wide_nd(D)) {
        case 1: COVA_GAT_FWD_WIDE(1); break;
        case 2: COVA_GAT_FWD_WIDE(2); break;
        case 3: COVA_GAT_FWD_WIDE(3); break;
        case 4: COVA_GAT_FWD_WIDE(4); break;
        case 6: COVA_GAT_FWD_WIDE(6); break;
        default: COVA_GAT_FWD_WIDE(8); break;
        }
#undef COVA_GAT_FWD_WIDE
        COVA_LAUNCH_CHECK();
        return COVA_OK;
    }
#define COVA_GAT_FWD(KP) hipLaunchKernelGGL(gat_fwd_kernel<KP>, grid, blk, 0, st, Wh, ldw, s, t, ctx, N, K, D, slope, attn, hprime, ldh)
    switch ((K + 63) / 64) {        // one wave per node, K slots in ceil(K / 64) passes over the lanes
    case 1: COVA_GAT_FWD(1); break;
    case 2: COVA_GAT_FWD(2); break;
    case 3: COVA_GAT_FWD(3); break;
    case 4: COVA_GAT_FWD(4); break;
    case 5: COVA_GAT_FWD(5); break;
    case 6: COVA_GAT_FWD(6); break;
    case 7: COVA_GAT_FWD(7); break;
    case 8: COVA_GAT_FWD(8); break;
    case 9: case 10: case 11: case 12: COVA_GAT_FWD(12); break;
    default: COVA_GAT_FWD(16); break;
    }
#undef COVA_GAT_FWD
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

// Transposed neighbour index of ctx [N,K]: csr[0..N] = row_ptr, csr[N+1 .. N+1+N*K) = the flat slots i*K+k that
// name node j, ascending; the rest of the workspace is scratch.  Build it once per batch: every head and
// layer of the GAT stack (and every backward call) shares it.
COVA_API int cova_gat_transpose_ints(int N, int K) { return (N + 1) + 2 * N * K + 2 * N + 16; }

COVA_API int cova_gat_transpose(const int64_t *ctx, int N, int K, int *csr, void *stream)
{
    COVA_REQUIRE(ctx && csr && N > 0 && K > 0);
    hipStream_t st = (hipStream_t)stream;
    const int E = N * K;
    int *row_ptr = csr, *edges = csr + N + 1, *tmp = edges + E, *deg = tmp + E, *cursor = deg + N;
    hipError_t e = hipMemsetAsync(deg, 0, sizeof(int) * 2 * (size_t)N, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(csr_count_kernel, dim3(cdiv(E, 256)), dim3(256), 0, st, ctx, E, N, deg);
    COVA_LAUNCH_CHECK();
    hipLaunchKernelGGL(csr_scan_kernel, dim3(1), dim3(1024), 0, st, deg, N, row_ptr);
    COVA_LAUNCH_CHECK();
    hipLaunchKernelGGL(csr_fill_kernel, dim3(cdiv(E, 256)), dim3(256), 0, st, ctx, E, N, row_ptr, cursor, tmp);
    COVA_LAUNCH_CHECK();
    hipLaunchKernelGGL(csr_sort_rows_kernel, dim3(cdiv(N, 4)), dim3(256), 0, st, row_ptr, tmp, N, edges, (int *)nullptr);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

// The same for a workspace that is kept from call to call: the ints csr[(N+1) + 2NK ..] (2N + 16 of them) must be ZERO on
// entry (zero the buffer once, when it is allocated) and are zero again when the call's kernels have run -- three
// launches, no memsets.  One workspace per stream.
COVA_API int cova_gat_transpose_reuse(const int64_t *ctx, int N, int K, int *csr, void *stream)
{
    COVA_REQUIRE(ctx && csr && N > 0 && K > 0);
    hipStream_t st = (hipStream_t)stream;
    const int E = N * K;
    int *row_ptr = csr, *edges = csr + N + 1, *tmp = edges + E, *deg = tmp + E, *cursor = deg + N, *ticket = cursor + N;
    hipLaunchKernelGGL(csr_count_scan_kernel, dim3(cdiv(E, 256)), dim3(256), 0, st, ctx, E, N, deg, ticket, row_ptr);
    COVA_LAUNCH_CHECK();
    hipLaunchKernelGGL(csr_fill_kernel, dim3(cdiv(E, 256)), dim3(256), 0, st, ctx, E, N, row_ptr, cursor, tmp);
    COVA_LAUNCH_CHECK();
    hipLaunchKernelGGL(csr_sort_rows_kernel, dim3(cdiv(N, 4)), dim3(256), 0, st, row_ptr, tmp, N, edges, deg);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}

// dWh [N, 2D] (ld = lddw) is fully written; ds, dt [N] scratch; d_att_w [2D], d_att_b [1].
// csr (from cova_gat_transpose) + du [N,K] scratch: gather form, no float atomics, bit-identical reruns.
// csr == NULL: the scatter form with float atomics (kept for A/B measurements).
COVA_API int cova_gat_bwd(const float *g, int ldg, const float *Wh, int ldw, const float *s,
                          const float *t, const float *attn, const int64_t *ctx, const float *att_w,
                          int N, int K, int D, float slope, float *dWh, int lddw, float *ds, float *dt,
                          float *d_att_w, float *d_att_b, const int *csr, float *du, void *stream)
{
    COVA_REQUIRE(g && Wh && s && t && attn && ctx && att_w && dWh && ds && dt && d_att_w && d_att_b);
    COVA_REQUIRE(K > 0 && K <= COVA_GAT_MAX_K && D > 0 && N > 0);
    COVA_REQUIRE(!csr || du);
    hipStream_t st = (hipStream_t)stream;
    const int kp = (K + 63) / 64;
    if (csr != nullptr && K <= 64 && gat_wide_nd(D) > 0) {
        const dim3 grid(cdiv(N, 4)), blk(256);
#define COVA_GAT_BWD_WIDE(ND)                                                                                               \
        do {                                                                                                                \
            hipLaunchKernelGGL(gat_bwd_src_wide_kernel<ND>, grid, blk, 0, st, g, ldg, Wh, ldw, s, t, attn, ctx, att_w, N, K, D, \
                               slope, dWh, lddw, ds, du);                                                                   \
            hipLaunchKernelGGL(gat_bwd_dst_wide_kernel<ND>, grid, blk, 0, st, g, ldg, attn, du, csr, csr + N + 1, att_w, N, K, \
                               D, dWh, lddw, dt);                                                                           \
        } while (0)
        switch (gat_wide_nd(D)) {
        case 1: COVA_GAT_BWD_WIDE(1); break;
        case 2: COVA_GAT_BWD_WIDE(2); break;
        case 3: COVA_GAT_BWD_WIDE(3); break;
        case 4: COVA_GAT_BWD_WIDE(4); break;
        case 6: COVA_GAT_BWD_WIDE(6); break;
        default: COVA_GAT_BWD_WIDE(8); break;
        }
#undef COVA_GAT_BWD_WIDE
        COVA_LAUNCH_CHECK();
    } else if (csr != nullptr) {
#define COVA_GAT_SRC(KP) hipLaunchKernelGGL(gat_bwd_src_kernel<KP>, dim3(cdiv(N, 4)), dim3(256), 0, st, g, ldg, Wh, ldw, s, t, \
                                            attn, ctx, att_w, N, K, D, slope, dWh, lddw, ds, du)
        switch (kp) {
        case 1: COVA_GAT_SRC(1); break;
        case 2: COVA_GAT_SRC(2); break;
        case 3: COVA_GAT_SRC(3); break;
        case 4: COVA_GAT_SRC(4); break;
        case 5: COVA_GAT_SRC(5); break;
        case 6: COVA_GAT_SRC(6); break;
        case 7: COVA_GAT_SRC(7); break;
        case 8: COVA_GAT_SRC(8); break;
        case 9: case 10: case 11: case 12: COVA_GAT_SRC(12); break;
        default: COVA_GAT_SRC(16); break;
        }
#undef COVA_GAT_SRC
        COVA_LAUNCH_CHECK();
        hipLaunchKernelGGL(gat_bwd_dst_kernel, dim3(cdiv(N, 4)), dim3(256), 0, st, g, ldg, attn, du, csr,
                           csr + N + 1, att_w, N, K, D, dWh, lddw, dt);
        COVA_LAUNCH_CHECK();
    } else {
        hipError_t e = hipMemsetAsync(dt, 0, sizeof(float) * (size_t)N, st);
        if (e != hipSuccess) return (int)e;
        e = hipMemset2DAsync(dWh + D, sizeof(float) * (size_t)lddw, 0, sizeof(float) * (size_t)D, (size_t)N, st);
        if (e != hipSuccess) return (int)e;
#define COVA_GAT_BWD(KP) hipLaunchKernelGGL(gat_bwd_kernel<KP>, dim3(cdiv(N, 4)), dim3(256), 0, st, g, ldg, Wh, ldw, s, t, attn, \
                                            ctx, att_w, N, K, D, slope, dWh, lddw, ds, dt)
        switch (kp) {
        case 1: COVA_GAT_BWD(1); break;
        case 2: COVA_GAT_BWD(2); break;
        case 3: COVA_GAT_BWD(3); break;
        case 4: COVA_GAT_BWD(4); break;
        case 5: COVA_GAT_BWD(5); break;
        case 6: COVA_GAT_BWD(6); break;
        case 7: COVA_GAT_BWD(7); break;
        case 8: COVA_GAT_BWD(8); break;
        case 9: case 10: case 11: case 12: COVA_GAT_BWD(12); break;
        default: COVA_GAT_BWD(16); break;
        }
#undef COVA_GAT_BWD
        COVA_LAUNCH_CHECK();
        hipLaunchKernelGGL(gat_bwd_addt_kernel, dim3(cdiv(N * D, 256)), dim3(256), 0, st, dWh, lddw, dt,
                           att_w, N, D);
        COVA_LAUNCH_CHECK();
    }
    const bool v4 = (D % 4 == 0) && (ldw % 4 == 0) && (((uintptr_t)Wh & 15) == 0);
    hipLaunchKernelGGL((v4 ? gat_bwd_att_kernel<4> : gat_bwd_att_kernel<1>), dim3(cdiv(2 * D + 1, 64)), dim3(1024), 0,
                       st, Wh, ldw, ds, dt, d_att_w, d_att_b, N, D);
    COVA_LAUNCH_CHECK();
    return COVA_OK;
}
