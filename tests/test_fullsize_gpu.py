"""GPU tests at BASELINE.json's full sizes (1280x1280 pages, 320x320x64 feature maps).

The oracle is affordable here only on crops or for a single page, so parity at full size is
established through size-independent properties: crop equivalence against torch-CPU (translation
structure of convolution / pooling), adjoint identities tying the gradient kernels to the forward
kernel (<dW, V> = <dz, conv(x; V)>, <dgrad(dz), x> = <dz, conv(x)>), exactness of the fused
statistics, BatchNorm output moments, and one full single-page train step against the oracle.
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from cova_web_object_detection_amd import _lib, engine, synthetic, weights  # noqa: E402
from helpers import assert_routing_near_ties, compare_grads, routing_from_saved  # noqa: E402
from oracle import cova_oracle as O  # noqa: E402

call, query = _lib.call, _lib.query
DEV = "cuda:0"


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-9))


@pytest.mark.parametrize("fam", ["w4", "w2"])
def test_conv3x3_full_size_crops_adjoints_and_stats(fam):
    """The product's 3x3 kernels at the benchmark map size: F(4x4,3x3) (default) and F(2x2,3x3) forward / data gradient,
    the Winograd weight gradient."""
    B, H, W = 2, 320, 320
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(B, H, W, 64, device=DEV, generator=g)
    w = torch.randn(64, 64, 3, 3, device=DEV, generator=g) * 0.05
    v = torch.randn(64, 64, 3, 3, device=DEV, generator=g) * 0.05
    dz = torch.randn(B, H, W, 64, device=DEV, generator=g)

    def prep(wt):
        if fam == "w4":
            a, b_ = torch.empty(query("cova_conv3x3_wino4_u_floats"), device=DEV), torch.empty(query("cova_conv3x3_wino4_u_floats"), device=DEV)
            call("cova_conv3x3_wino4_prep", wt, a, b_)
        else:
            a, b_ = torch.empty(16, 16, 4, 64, device=DEV), torch.empty(16, 16, 4, 64, device=DEV)
            call("cova_conv3x3_prep_weights_wino", wt, a, b_)
        return a, b_

    def conv(inp, u, out, part):
        if fam == "w4":
            call("cova_conv3x3_wino4", inp, u, out, part, B, H, W)
        else:
            call("cova_conv3x3_wino", inp, u, None, None, None, None, None, out, part, B, H, W)

    wf, wd = prep(w)
    nt = query("cova_conv3x3_wino4_num_partials" if fam == "w4" else "cova_conv3x3_wino_num_partials", B, H, W)
    out, part = torch.empty_like(x), torch.empty(nt, 2, 64, device=DEV)
    conv(x, wf, out, part)
    # (1) crop equivalence against torch-CPU, including image corners / borders (zero padding)
    wc = w.cpu()
    for (b, y0, x0, h, ww) in [(0, 0, 0, 40, 48), (1, 280, 272, 40, 48), (0, 131, 0, 37, 64),
                               (1, 0, 250, 33, 70), (0, 150, 97, 64, 64)]:
        ys, ye, xs, xe = max(y0 - 1, 0), min(y0 + h + 1, H), max(x0 - 1, 0), min(x0 + ww + 1, W)
        crop = x[b, ys:ye, xs:xe].permute(2, 0, 1).unsqueeze(0).cpu()
        pad = (1 if x0 == 0 else 0, 1 if x0 + ww == W else 0, 1 if y0 == 0 else 0, 1 if y0 + h == H else 0)
        ref = F.conv2d(F.pad(crop, pad), wc)[0].permute(1, 2, 0)
        assert rel(out[b, y0:y0 + h, x0:x0 + ww], ref) < 1e-4
    # (2) fused statistics are the exact column sums of what was written
    flat = out.view(-1, 64).double()
    assert rel(part[:, 0].double().sum(0), flat.sum(0)) < 1e-5
    assert rel(part[:, 1].double().sum(0), (flat * flat).sum(0)) < 1e-5
    # (3) linearity in the input
    out2 = torch.empty_like(x)
    conv(x * 0.5 + 1.25 * dz, wf, out2, None)
    out3 = torch.empty_like(x)
    conv(dz, wf, out3, None)
    assert rel(out2, 0.5 * out + 1.25 * out3) < 1e-4
    # (4) adjoint identities: data gradient and weight gradient vs the forward kernel
    dx = torch.empty_like(x)
    conv(dz, wd, dx, None)
    lhs, rhs = (dx.double() * x.double()).sum(), (dz.double() * out.double()).sum()
    assert abs(float(lhs - rhs)) <= 2e-5 * float(out.double().norm() * dz.double().norm())
    ws = torch.empty(query("cova_conv3x3_wgrad_workspace_floats", B, H, W), device=DEV)
    dw = torch.empty(64, 64, 3, 3, device=DEV)
    call("cova_conv3x3_wgrad_wino", x, dz, dw, ws, B, H, W)
    vf, _ = prep(v)
    outv = torch.empty_like(x)
    conv(x, vf, outv, None)
    lhs, rhs = (dw.double() * v.double()).sum(), (dz.double() * outv.double()).sum()
    assert abs(float(lhs - rhs)) <= 2e-5 * float(outv.double().norm() * dz.double().norm())


def test_conv1_full_size_crops_and_adjoint():
    B, H, W = 1, 1280, 1280
    g = torch.Generator(device=DEV).manual_seed(2)
    img = torch.rand(B, 3, H, W, device=DEV, generator=g)
    w = torch.randn(64, 3, 7, 7, device=DEV, generator=g) * 0.1
    v = torch.randn(64, 3, 7, 7, device=DEV, generator=g) * 0.1
    wk, vk = torch.empty(154, 64, device=DEV), torch.empty(154, 64, device=DEV)
    call("cova_conv1_prep_weights", w, wk)
    call("cova_conv1_prep_weights", v, vk)
    nt = query("cova_conv1_num_partials", B, H, W)
    y, part = torch.empty(B, 640, 640, 64, device=DEV), torch.empty(nt, 2, 64, device=DEV)
    call("cova_conv1_fwd", img, wk, y, part, B, H, W)
    wc = w.cpu()
    for (y0, x0, h, ww) in [(0, 0, 24, 40), (616, 600, 24, 40), (300, 0, 17, 33), (0, 333, 20, 31)]:
        ys, ye = max(2 * y0 - 3, 0), min(2 * (y0 + h - 1) + 3 + 1, H)
        xs, xe = max(2 * x0 - 3, 0), min(2 * (x0 + ww - 1) + 3 + 1, W)
        pad = (xs - (2 * x0 - 3), (2 * (x0 + ww - 1) + 4) - xe, ys - (2 * y0 - 3), (2 * (y0 + h - 1) + 4) - ye)
        ref = F.conv2d(F.pad(img[:, :, ys:ye, xs:xe].cpu(), pad), wc, stride=2)[0].permute(1, 2, 0)
        assert rel(y[0, y0:y0 + h, x0:x0 + ww], ref) < 1e-4
    flat = y.view(-1, 64).double()
    assert rel(part[:, 0].double().sum(0), flat.sum(0)) < 1e-5
    dy = torch.randn(B, 640, 640, 64, device=DEV, generator=g)
    ws = torch.empty(query("cova_conv1_wgrad_workspace_floats", B, H, W), device=DEV)
    dw = torch.empty(64, 3, 7, 7, device=DEV)
    call("cova_conv1_wgrad", img, dy, dw, ws, B, H, W)
    yv = torch.empty_like(y)
    call("cova_conv1_fwd", img, vk, yv, None, B, H, W)
    lhs, rhs = (dw.double() * v.double()).sum(), (dy.double() * yv.double()).sum()
    assert abs(float(lhs - rhs)) <= 1e-5 * float(yv.double().norm() * dy.double().norm())


def test_bn_pool_full_size_properties():
    B, H1, W1 = 1, 640, 640
    g = torch.Generator(device=DEV).manual_seed(3)
    y = torch.randn(B, H1, W1, 64, device=DEV, generator=g) * 1.7 + 0.4
    gamma = (torch.rand(64, device=DEV, generator=g) + 0.5)
    beta = torch.randn(64, device=DEV, generator=g) * 0.2
    params = {"bn.weight": gamma, "bn.bias": beta}
    buffers = {"bn.running_mean": torch.zeros(64, device=DEV), "bn.running_var": torch.ones(64, device=DEV),
               "bn.num_batches_tracked": torch.zeros((), dtype=torch.long, device=DEV)}
    R = B * H1 * W1
    part, n = engine.colstats(y, 64, R, 64)
    st = engine.bn_params("bn.", params, buffers, 64, y, True, part, n, R)
    out = torch.empty_like(y)
    call("cova_bn_act_fwd", y, 64, st.scale, st.shift, None, 0, out, 64, R, 64, 0)
    o = out.view(-1, 64).double()
    assert rel(o.mean(0), beta.double()) < 1e-4           # BN output moments: mean = beta, std = gamma
    assert rel(o.std(0, unbiased=False), gamma.double()) < 1e-4
    H2 = W2 = 320
    p = torch.empty(B, H2, W2, 64, device=DEV)
    idx = torch.empty(B, H2, W2, 64, device=DEV, dtype=torch.uint8)
    call("cova_bn_relu_maxpool_fwd", y, st.scale, st.shift, p, idx, None, B, H1, W1)
    for (y0, x0, h, w) in [(0, 0, 20, 24), (300, 296, 20, 24), (100, 0, 9, 33)]:
        ys, ye, xs, xe = max(2 * y0 - 1, 0), min(2 * (y0 + h) , H1), max(2 * x0 - 1, 0), min(2 * (x0 + w), W1)
        crop = torch.relu(out[0, ys:ye, xs:xe]).permute(2, 0, 1).unsqueeze(0).cpu()
        pad = (1 if x0 == 0 else 0, 0, 1 if y0 == 0 else 0, 0)
        ref = F.max_pool2d(F.pad(crop, pad, value=float("-inf")), 3, 2, 0)[0].permute(1, 2, 0)
        assert torch.equal(p[0, y0:y0 + h, x0:x0 + w].cpu(), ref[:h, :w])
    # backward: the window-form reduction equals the pixel-form identity sum(dz) == 0 per channel
    dp = torch.randn(B, H2, W2, 64, device=DEV, generator=g)
    npart = query("cova_bn_relu_maxpool_bwd_num_partials", B, H1, W1)
    bpart = torch.empty(npart, 2, 64, device=DEV)
    call("cova_bn_relu_maxpool_bwd_reduce", dp, idx, y, st.scale, st.shift, st.mean, st.invstd, bpart, B, H1, W1)
    dg, db, coef = torch.empty(64, device=DEV), torch.empty(64, device=DEV), torch.empty(2, 64, device=DEV)
    call("cova_bn_finalize_bwd", bpart, npart, 64, float(R), dg, db, coef)
    dz = torch.empty_like(y)
    call("cova_bn_relu_maxpool_bwd_apply", dp, idx, y, st.scale, st.shift, st.mean, st.invstd, coef, dz, B, H1, W1)
    d = dz.view(-1, 64).double()
    scale = float(d.abs().sum(0).max())
    assert float(d.sum(0).abs().max()) <= 1e-5 * scale               # sum dz = 0 (BN backward identity)
    xh = (y.view(-1, 64).double() - st.mean.double()) * st.invstd.double()
    assert float((d * xh).sum(0).abs().max()) <= 1e-4 * scale         # sum dz*xhat = 0


def test_single_page_train_step_full_resolution_matches_oracle():
    cfg = dict(roi_output_size=(3, 3), n_classes=4, use_context=True, hidden_dim=384,
               bbox_hidden_dim=32, n_additional_feat=0, drop_prob=0.0)
    wcfg = {k: v for k, v in cfg.items() if k != "drop_prob"}
    sd = weights.seeded_state_dict(7, logit_gain=4.0, **wcfg)
    batch = synthetic.make_batch(1, img_h=1280, boxes_per_page=90, context_size=12, seed=7)
    params = {k: v.to(DEV) for k, v in sd.items() if k in O.param_keys(sd)}
    buffers = {k: v.to(DEV) for k, v in sd.items() if k not in params}
    args = [batch[k].to(DEV) for k in ("images", "bboxes", "additional_feats", "context_indices")]
    logits, sv = engine.model_fwd(cfg, params, buffers, *args, True)
    loss, dl, pred = engine.ce_sum(logits, batch["labels"].to(DEV))
    routing = routing_from_saved(sv)
    grads = engine.model_bwd(sv, dl, params)
    torch.set_num_threads(max(torch.get_num_threads(), 8))
    loss_ref, logits_ref, grads_ref, after, inter = O.loss_and_grads(
        sd, batch["images"], batch["bboxes"], batch["additional_feats"], batch["context_indices"],
        batch["labels"], cfg, None, routing)
    err = rel(logits, logits_ref)
    assert err < 5e-5, err                                                # ~5x the measured round-off (3.6e-6 / 5e-7 / 2.5e-5:
    assert abs(loss.item() - float(loss_ref)) <= 2e-5 * abs(float(loss_ref))     # profiles/r03_wino4_margin.txt, r02_grad_parity_1280*.txt)
    tol = 10 * max(err, 1e-6) * float(logits_ref.abs().max())
    top2 = torch.topk(logits_ref, 2, dim=1).values
    ok = (top2[:, 0] - top2[:, 1]) > tol
    assert torch.equal(pred.cpu()[ok], logits_ref.argmax(1)[ok])          # integer predictions, exact
    # all discrete decisions (max-pool / RoIPool argmax, 3e7 ReLU gates) forced to the HIP forward's:
    # the remaining difference is fp32 round-off
    compare_grads(grads, grads_ref, rtol=1e-4, outlier_frac=0.0)
    assert_routing_near_ties(routing, inter, batch["bboxes"], (3, 3), 0.25)
    for k in buffers:
        if not k.endswith("num_batches_tracked"):
            assert rel(buffers[k], after[k]) < 1e-4, k


@pytest.mark.parametrize("kw,H,W,n,cs", [(dict(backbone="resnet50", n_heads=2), 1280, 1280, 90, 12),
                                         (dict(backbone="resnet50", n_heads=2, n_gat_layers=2), 2048, 1280, 300, 24)])
def test_single_page_train_step_full_resolution_extension_matches_self_oracle(kw, H, W, n, cs):
    """configs[2] (ResNet-50 stem + 2-head GAT, 1280x1280, 90 boxes, K=24) and the configs[4] architecture /
    box count / K = 48 on a half-length page (2048x1280; the full 4096-row page is covered by
    tests/test_extension_gpu.py::test_full_size_long_page_train_step_properties), one page, whole train step against
    the self-oracle with every discrete decision forced: fp32 round-off only."""
    cfg = dict(roi_output_size=(3, 3), n_classes=4, use_context=True, hidden_dim=384,
               bbox_hidden_dim=32, n_additional_feat=0, drop_prob=0.0, **kw)
    wcfg = {k: v for k, v in cfg.items() if k != "drop_prob"}
    sd = weights.seeded_state_dict(9, logit_gain=4.0, **wcfg)
    batch = synthetic.make_batch(1, img_h=H, img_w=W, boxes_per_page=n, context_size=cs, seed=9)
    params = {k: v.to(DEV) for k, v in sd.items() if k in O.param_keys(sd)}
    buffers = {k: v.to(DEV) for k, v in sd.items() if k not in params}
    args = [batch[k].to(DEV) for k in ("images", "bboxes", "additional_feats", "context_indices")]
    logits, sv = engine.model_fwd(cfg, params, buffers, *args, True)
    loss, dl, pred = engine.ce_sum(logits, batch["labels"].to(DEV))
    routing = routing_from_saved(sv)
    grads = engine.model_bwd(sv, dl, params)
    torch.set_num_threads(max(torch.get_num_threads(), 8))
    ocfg = {k: v for k, v in cfg.items() if k not in kw}
    loss_ref, logits_ref, grads_ref, after, inter = O.loss_and_grads(
        sd, batch["images"], batch["bboxes"], batch["additional_feats"], batch["context_indices"],
        batch["labels"], ocfg, None, routing)
    err = rel(logits, logits_ref)
    assert err < 5e-5, err                                                # ~5x the measured round-off (3.6e-6 / 5e-7 / 2.5e-5:
    assert abs(loss.item() - float(loss_ref)) <= 2e-5 * abs(float(loss_ref))     # profiles/r03_wino4_margin.txt, r02_grad_parity_1280*.txt)
    tol = 10 * max(err, 1e-6) * float(logits_ref.abs().max())
    top2 = torch.topk(logits_ref, 2, dim=1).values
    ok = (top2[:, 0] - top2[:, 1]) > tol
    assert torch.equal(pred.cpu()[ok], logits_ref.argmax(1)[ok])          # integer predictions, exact
    assert set(grads) == set(grads_ref)
    compare_grads(grads, grads_ref, rtol=1e-4, outlier_frac=0.0)
    assert_routing_near_ties(routing, inter, batch["bboxes"], (3, 3), 0.25)
    for k in buffers:
        if not k.endswith("num_batches_tracked"):
            assert rel(buffers[k], after[k]) < 1e-4, k


def test_activations_beyond_2_31_elements_index_correctly():
    """96 pages of 1280x1280 (conv1 output 2.5 G elements, 10 GB): six copies of a 16-page batch share
    its BatchNorm statistics, so every copy's logits equal the 16-page logits and the gradient is six
    times the 16-page gradient (tools/large_batch_check.py; 176 pages = 4.6 G elements checked there)."""
    import re
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "large_batch_check.py")],
                         env=dict(os.environ, REPS="6"), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    m = re.search(r"pages\s+96 .*logits err ([0-9.e+-]+)\s+loss ratio ([0-9.]+)\s+grad err ([0-9.e+-]+)", out.stdout)
    assert m, out.stdout[-2000:]
    assert float(m.group(1)) < 1e-4 and abs(float(m.group(2)) - 1.0) < 1e-5 and float(m.group(3)) < 1e-3


def test_weight_gradient_of_a_batch_beyond_4_gb_per_tensor():
    """cova_conv3x3_wgrad4_partial on 165 maps of 320 x 320 x 64 -- 4.3 GB per operand: byte offsets from the tensor base no
    longer fit 32 bits (rounds 4-5 refused such batches; the kernel now addresses a page through its own 64-bit base).
    The weight gradient is a sum over pages: the 165-page launch equals the launches over pages [0, 100) and [100, 165) added
    (fp32 re-association of the fixed-order fold), and the side output of the LAST pages -- the ones beyond 4 GB -- is the operand
    A*dz + B*dz2 + C there."""
    B, H, W = 165, 320, 320
    assert B * H * W * 256 >= 1 << 32
    g = torch.Generator(device=DEV).manual_seed(7)
    mk = lambda: torch.randn((B, H, W, 64), device=DEV, generator=g)
    act, dz, dz2 = mk(), mk(), mk()
    abc_a = torch.randn((3, 64), device=DEV, generator=g)
    abc_d = torch.randn((3, 64), device=DEV, generator=g)

    def wgrad(lo, hi, want_out):
        n = hi - lo
        ws = torch.empty(query("cova_conv3x3_wgrad4_workspace_floats", n, H, W), device=DEV)
        out = torch.full((n, H, W, 64), float("nan"), device=DEV) if want_out else None
        dw = torch.empty(64, 64, 3, 3, device=DEV)
        call("cova_conv3x3_wgrad4_partial", act[lo:hi], abc_a, 1, dz[lo:hi], dz2[lo:hi], abc_d, out, ws, n, H, W)
        call("cova_conv3x3_wgrad4_finish", ws, dw, None, None, None, None, None, None, n, H, W)
        return dw, out

    dw_all, out_all = wgrad(0, B, True)
    dw_a, _ = wgrad(0, 100, False)
    dw_b, out_b = wgrad(100, B, True)
    assert rel(dw_all, dw_a + dw_b) < 2e-5
    assert torch.equal(out_all[100:], out_b)
    ref = torch.addcmul(torch.addcmul(abc_d[2].expand_as(dz[160:]), dz2[160:], abc_d[1]), dz[160:], abc_d[0])
    assert rel(out_all[160:], ref) < 1e-6
