/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement of RoIPool forward/backward.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product path (cova-web-object-detection_amd/) never does.
 *
 * What it restates: the operator the reference instantiates at models.py:58
 * (`torchvision.ops.RoIPool(roi_output_size, spatial_scale)`) and calls at models.py:125.
 * torchvision==0.7.0 (requirements.txt:4) is a third-party dependency that is NOT vendored
 * under /root/reference and is not installed in this image, so this file restates the
 * published algorithm of that operator (Fast R-CNN RoI max pooling, Girshick 2015, as
 * implemented by torchvision 0.7.0 `ROIPool_cpu.cpp`):
 *
 *   roi = [batch_idx, x1, y1, x2, y2];  s = spatial_scale
 *   roi_start_w = round(x1*s)  roi_start_h = round(y1*s)      (C round(), half away from 0,
 *   roi_end_w   = round(x2*s)  roi_end_h   = round(y2*s)       product formed in float32)
 *   roi_w = max(roi_end_w - roi_start_w + 1, 1), roi_h likewise
 *   bin_h = (float)roi_h / PH,  bin_w = (float)roi_w / PW
 *   hstart = floor(ph*bin_h), hend = ceil((ph+1)*bin_h)       (float32 products)
 *   then + roi_start_h and clamp to [0, H]; same for w
 *   empty bin (hend<=hstart || wend<=wstart) -> 0, argmax -1
 *   else max over the window, strict '>' scan in row-major order from -FLT_MAX
 *   backward: grad_in[b, c, argmax] += grad_out (argmax -1 skipped)
 *
 * PARITY STATUS: *unpinned* -- the reference holds no test or golden vector for this
 * operator and its source is unavailable offline.  The restatement is cross-checked in
 * tests/test_oracle_cpu.py against torch.nn.functional.adaptive_max_pool2d on boxes that lie
 * fully inside the feature map.
 *
 * Layout here is the reference's: feat [B, C, H, W] (NCHW), out [N, C, PH, PW],
 * argmax [N, C, PH, PW] int32 = h*W + w.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

void oracle_roipool_fwd(const float *feat, const float *rois, int n_rois, int C, int H, int W,
                        int PH, int PW, float spatial_scale, float *out, int32_t *argmax)
{
    for (int n = 0; n < n_rois; ++n) {
        const float *roi = rois + 5 * n;
        int b = (int)roi[0];
        int rs_w = (int)roundf(roi[1] * spatial_scale);
        int rs_h = (int)roundf(roi[2] * spatial_scale);
        int re_w = (int)roundf(roi[3] * spatial_scale);
        int re_h = (int)roundf(roi[4] * spatial_scale);
        int roi_w = imax(re_w - rs_w + 1, 1);
        int roi_h = imax(re_h - rs_h + 1, 1);
        float bin_h = (float)roi_h / (float)PH;
        float bin_w = (float)roi_w / (float)PW;
        for (int ph = 0; ph < PH; ++ph) {
            int hstart = (int)floorf((float)ph * bin_h);
            int hend = (int)ceilf((float)(ph + 1) * bin_h);
            hstart = imin(imax(hstart + rs_h, 0), H);
            hend = imin(imax(hend + rs_h, 0), H);
            for (int pw = 0; pw < PW; ++pw) {
                int wstart = (int)floorf((float)pw * bin_w);
                int wend = (int)ceilf((float)(pw + 1) * bin_w);
                wstart = imin(imax(wstart + rs_w, 0), W);
                wend = imin(imax(wend + rs_w, 0), W);
                int empty = (hend <= hstart) || (wend <= wstart);
                for (int c = 0; c < C; ++c) {
                    const float *plane = feat + ((size_t)b * C + c) * H * W;
                    float maxval = empty ? 0.f : -FLT_MAX;
                    int maxidx = -1;
                    for (int h = hstart; h < hend; ++h)
                        for (int w = wstart; w < wend; ++w) {
                            float v = plane[h * W + w];
                            if (v > maxval) { maxval = v; maxidx = h * W + w; }
                        }
                    size_t o = (((size_t)n * C + c) * PH + ph) * PW + pw;
                    out[o] = maxval;
                    argmax[o] = maxidx;
                }
            }
        }
    }
}

void oracle_roipool_bwd(const float *grad_out, const float *rois, const int32_t *argmax,
                        int n_rois, int B, int C, int H, int W, int PH, int PW, float *grad_in)
{
    memset(grad_in, 0, sizeof(float) * (size_t)B * C * H * W);
    for (int n = 0; n < n_rois; ++n) {
        int b = (int)rois[5 * n];
        for (int c = 0; c < C; ++c) {
            float *plane = grad_in + ((size_t)b * C + c) * H * W;
            for (int p = 0; p < PH * PW; ++p) {
                size_t o = ((size_t)n * C + c) * PH * PW + p;
                if (argmax[o] >= 0) plane[argmax[o]] += grad_out[o];
            }
        }
    }
}

/*
 * GAT forward in the reference's own formulation (models.py:178-212): gather the K
 * neighbour feature rows (index -1 selects an appended zero row), apply W_j to every
 * gathered row, one 2*D-long dot product with the attention vector per (node, slot),
 * LeakyReLU(0.2), mask (-9e15 where index < 0), softmax over the K slots, weighted sum.
 * Plain loops, float32 accumulation in index order.  Used as an independent check of the
 * torch restatement in oracle/cova_oracle.py at small sizes.
 */
void oracle_gat_fwd(const float *h, const int64_t *ctx, const float *W_i, const float *W_j,
                    const float *att_w, float att_b, int N, int K, int F, int D, float alpha,
                    float *h_prime, float *attn)
{
    float whi[4096], whj[4096], e[1024];
    for (int i = 0; i < N; ++i) {
        for (int d = 0; d < D; ++d) {
            float s = 0.f;
            for (int f = 0; f < F; ++f) s += h[(size_t)i * F + f] * W_i[(size_t)d * F + f];
            whi[d] = s;
        }
        float emax = -INFINITY;
        for (int k = 0; k < K; ++k) {
            int64_t j = ctx[(size_t)i * K + k];
            float s = att_b, t = 0.f;
            for (int d = 0; d < D; ++d) s += att_w[d] * whi[d];
            for (int d = 0; d < D; ++d) {
                float w = 0.f;
                if (j >= 0)
                    for (int f = 0; f < F; ++f) w += h[(size_t)j * F + f] * W_j[(size_t)d * F + f];
                t += att_w[D + d] * w;
            }
            float v = s + t;
            v = v > 0.f ? v : alpha * v;
            if (j < 0) v = -9e15f;
            e[k] = v;
            if (v > emax) emax = v;
        }
        float denom = 0.f;
        for (int k = 0; k < K; ++k) { e[k] = expf(e[k] - emax); denom += e[k]; }
        for (int k = 0; k < K; ++k) attn[(size_t)i * K + k] = e[k] / denom;
        for (int d = 0; d < D; ++d) h_prime[(size_t)i * D + d] = 0.f;
        for (int k = 0; k < K; ++k) {
            int64_t j = ctx[(size_t)i * K + k];
            if (j < 0) continue;
            for (int d = 0; d < D; ++d) whj[d] = 0.f;
            for (int d = 0; d < D; ++d) {
                float w = 0.f;
                for (int f = 0; f < F; ++f) w += h[(size_t)j * F + f] * W_j[(size_t)d * F + f];
                whj[d] = w;
            }
            float a = attn[(size_t)i * K + k];
            for (int d = 0; d < D; ++d) h_prime[(size_t)i * D + d] += a * whj[d];
        }
    }
}
