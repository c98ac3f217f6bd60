"""CPU tests: the oracle (oracle/) against the golden vectors captured from the reference."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cova_oracle as O
from helpers import FULL_CASES, GOLDEN, check_grads, load_case
from cova_web_object_detection_amd import synthetic, weights


def test_state_dict_spec_matches_reference_count():
    spec = weights.state_dict_spec()
    assert len(spec) == 50
    n_params = sum(int(np.prod(s)) for k, s in spec
                   if not k.endswith(("running_mean", "running_var", "num_batches_tracked")))
    assert n_params == 1616485                               # README "1.6m"


def test_collate_layout_matches_reference():
    fx = np.load(GOLDEN + "/collate.npz")
    counts, cs = [int(c) for c in fx["counts"]], int(fx["context_size"])
    ctx = synthetic.collate_context([synthetic.context_window_indices(n, cs) for n in counts])
    assert np.array_equal(ctx, fx["context_indices"])
    assert fx["bboxes"].shape[1] == 5
    exp_idx = np.concatenate([np.full(n, i, np.float32) for i, n in enumerate(counts)])
    assert np.array_equal(fx["bboxes"][:, 0], exp_idx)
    # xywh -> xyxy happened in __getitem__ (datasets.py:115): x2 > x1
    assert (fx["bboxes"][:, 3] > fx["bboxes"][:, 1]).all()


def test_eval_decision_rule_matches_reference():
    fx = np.load(GOLDEN + "/evaluate.npz")
    logits, bboxes, labels = (torch.from_numpy(fx[k]) for k in ("logits", "bboxes", "labels"))
    for k in (1, 3):
        acc = np.asarray(O.eval_accuracy(logits, bboxes, labels, 4, k))
        assert np.array_equal(acc, fx["img_acc_k%d" % k][:, 1:])


@pytest.mark.parametrize("fixture", ["gat_layer", "gat_layer_k100"])
def test_gat_layer_matches_reference(fixture):
    fx = np.load(GOLDEN + "/%s.npz" % fixture)
    sd = {"gat." + k[2:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("w/")}
    h = torch.from_numpy(fx["h"]).requires_grad_(True)
    for k in sd:
        sd[k].requires_grad_(True)
    ctx = torch.from_numpy(fx["ctx"])
    hp, attn = O.gat(h, ctx, sd, return_attn_wts=True)
    assert torch.allclose(hp, torch.from_numpy(fx["h_prime"]), atol=1e-6, rtol=1e-5)
    assert torch.allclose(attn, torch.from_numpy(fx["attn"]), atol=1e-7, rtol=1e-5)
    # all -1 row: uniform weights, zero context
    assert np.allclose(fx["attn"][3], 1.0 / ctx.shape[1])
    assert np.abs(fx["h_prime"][3]).max() == 0
    (hp * torch.from_numpy(fx["g"])).sum().backward()
    assert torch.allclose(h.grad, torch.from_numpy(fx["grad_h"]), atol=1e-5, rtol=1e-4)
    # independent plain-C loop version
    hp_c, attn_c = O.gat_forward_c(h.detach(), ctx, sd["gat.W_i.weight"].detach(),
                                   sd["gat.W_j.weight"].detach(),
                                   sd["gat.attention_layer.weight"].detach(),
                                   sd["gat.attention_layer.bias"].item())
    assert torch.allclose(hp_c, hp.detach(), atol=2e-5, rtol=1e-4)
    assert torch.allclose(attn_c, attn.detach(), atol=1e-6, rtol=1e-4)


def test_roipool_matches_adaptive_max_pool_inside_map():
    rs = np.random.RandomState(3)
    feat = torch.from_numpy(rs.standard_normal((2, 5, 20, 24)).astype(np.float32))
    rois, exp = [], []
    for _ in range(40):
        b = rs.randint(0, 2)
        w0, h0 = rs.randint(0, 18), rs.randint(0, 14)
        w1, h1 = w0 + rs.randint(2, 6), h0 + rs.randint(2, 6)   # >= 3 px so every bin is non-empty
        w1, h1 = min(w1, 23), min(h1, 19)
        if w1 - w0 + 1 < 3 or h1 - h0 + 1 < 3:
            continue
        rois.append([b, w0 * 4.0, h0 * 4.0, w1 * 4.0, h1 * 4.0])   # scale 0.25 -> integer coords
        exp.append(F.adaptive_max_pool2d(feat[b:b + 1, :, h0:h1 + 1, w0:w1 + 1], (3, 3))[0])
    rois = torch.tensor(rois, dtype=torch.float32)
    out = O.roi_pool(feat, rois, (3, 3), 0.25)
    assert torch.equal(out, torch.stack(exp))


def test_roipool_edge_cases():
    feat = torch.arange(2 * 1 * 4 * 4, dtype=torch.float32).view(2, 1, 4, 4)
    rois = torch.tensor([[0, 0, 0, 3.9, 3.9],        # tiny box -> single pixel, repeated bins
                         [1, 8, 8, 100, 100],        # clipped by the border
                         [0, 40, 40, 50, 50]])       # fully outside -> empty bins -> 0
    out, arg = O.roi_pool_argmax(feat, rois, (3, 3), 0.25)
    assert (out[2] == 0).all() and (arg[2] == -1).all()
    assert out[1].max() == 31.0
    assert out[0, 0, 0, 0] == 0.0


@pytest.mark.parametrize("name", FULL_CASES)
def test_full_model_oracle_matches_reference(name):
    fx, cfg, sd, batch = load_case(name)
    args = (batch["images"], batch["bboxes"], batch["additional_feats"], batch["context_indices"])
    with torch.no_grad():
        logits, inter = O.forward(O.clone_state_dict(sd), *args, cfg, False, None, True)
    assert np.allclose(logits.numpy(), fx["eval/logits"], atol=1e-5, rtol=1e-5)
    assert np.array_equal(logits.argmax(1).numpy(), fx["eval/argmax"])
    assert np.allclose(inter["attn"].numpy(), fx["eval/attn"], atol=1e-6, rtol=1e-5)
    assert np.allclose(inter["visual"].numpy().reshape(-1)[::7], fx["eval/visual_sample"], atol=1e-6)
    top1 = np.stack([v.numpy()[0] for _, v in sorted(O.page_class_decisions(logits, batch["bboxes"]).items())])
    assert np.array_equal(top1, fx["eval/page_class_top1"])
    loss, tl, grads, after, _ = O.loss_and_grads(sd, *args, batch["labels"], cfg, None)
    assert abs(float(loss) - float(fx["train/loss"])) <= 1e-4 * abs(float(fx["train/loss"]))
    assert np.allclose(tl.numpy(), fx["train/logits"], atol=1e-4, rtol=1e-4)
    check_grads(fx, grads, rtol=1e-3)
    for k in [k for k in fx if k.startswith("buf/")]:
        assert np.allclose(after[k[4:]].numpy(), fx[k], atol=1e-6, rtol=1e-5), k


def test_collate_restatement_matches_reference_on_raw_inputs():
    """oracle.collate_reference (ToTensor + __getitem__ box part + custom_collate_fn) is pinned by
    the reference's own output on uint8 pages / csv rows (tests/golden/collate_raw.npz)."""
    fx = np.load(GOLDEN + "/collate_raw.npz")
    rows = np.split(fx["rows"], np.cumsum(fx["counts"])[:-1])
    got = O.collate_reference(fx["u8_pages"], rows, int(fx["context_size"]))
    for k in ("images", "bboxes", "labels", "context_indices"):
        assert np.array_equal(got[k].numpy(), fx[k]), k
    assert got["additional_feats"].shape == (fx["rows"].shape[0], 0)


def test_attention_rows_restatement_matches_reference_dump():
    """oracle.attention_rows on the oracle's own eval forward == the reference's dump
    (extract_attn_wts_and_visualize.py:104-135) within fp32 round-off of the attention weights;
    the geometric / integer columns are exact."""
    fx = np.load(GOLDEN + "/attn_export.npz")
    _, cfg, sd, batch = load_case(str(fx["source"]))
    _, inter = O.forward(sd, batch["images"], batch["bboxes"], batch["additional_feats"],
                         batch["context_indices"], cfg, training=False, return_intermediates=True)
    rows = O.attention_rows(batch["bboxes"], batch["context_indices"], batch["labels"], inter["attn"])
    ref = fx["rows"]
    assert rows.shape == ref.shape
    K = batch["context_indices"].shape[1]
    assert np.array_equal(rows[:, :5 + 4 * K].numpy(), ref[:, :5 + 4 * K])
    np.testing.assert_allclose(rows[:, 5 + 4 * K:].numpy(), ref[:, 5 + 4 * K:], atol=1e-6)


def _roipool_numpy(feat, rois, ph_n, pw_n, scale):
    """Independent second formulation of torchvision 0.7.0's RoIPool: explicit window
    loops in numpy with float32 arithmetic at every step the C code has it, written without looking at
    oracle/roipool_ref.c's structure (bins are materialised as numpy slices, the arg-max comes from
    np.argmax over the flattened window, which also returns the FIRST maximum in row-major order)."""
    f32 = np.float32
    B, C, H, W = feat.shape
    out = np.zeros((len(rois), C, ph_n, pw_n), f32)
    arg = np.full((len(rois), C, ph_n, pw_n), -1, np.int32)

    def rnd(v):                       # C round(): half away from zero, on the float32 product
        v = f32(v)
        return int(np.floor(v + f32(0.5))) if v >= 0 else -int(np.floor(-v + f32(0.5)))

    for n, (b, x1, y1, x2, y2) in enumerate(rois):
        b = int(b)
        sw, sh = rnd(f32(x1) * f32(scale)), rnd(f32(y1) * f32(scale))
        ew, eh = rnd(f32(x2) * f32(scale)), rnd(f32(y2) * f32(scale))
        rw, rh = max(ew - sw + 1, 1), max(eh - sh + 1, 1)
        bh, bw = f32(rh) / f32(ph_n), f32(rw) / f32(pw_n)
        for ph in range(ph_n):
            hs = min(max(int(np.floor(f32(ph) * bh)) + sh, 0), H)
            he = min(max(int(np.ceil(f32(ph + 1) * bh)) + sh, 0), H)
            for pw in range(pw_n):
                ws = min(max(int(np.floor(f32(pw) * bw)) + sw, 0), W)
                we = min(max(int(np.ceil(f32(pw + 1) * bw)) + sw, 0), W)
                if he <= hs or we <= ws:
                    continue
                win = feat[b, :, hs:he, ws:we].reshape(C, -1)
                k = win.argmax(axis=1)
                out[n, :, ph, pw] = win[np.arange(C), k]
                arg[n, :, ph, pw] = (hs + k // (we - ws)) * W + ws + k % (we - ws)
    return out, arg


def test_roipool_against_independent_numpy_formulation_edge_cases():
    """Border-crossing, negative, zero-area, inverted, sub-pixel and x.5-rounding boxes, plus tied maxima:
    oracle/roipool_ref.c and the numpy window-loop formulation must agree bit for bit (values and argmax)."""
    rs = np.random.RandomState(12)
    B, C, H, W = 2, 6, 13, 17
    feat = rs.standard_normal((B, C, H, W)).astype(np.float32)
    feat[0, :, 4:8, 4:8] = 1.5                       # plateau: ties -> first maximum wins
    feat[1, 2] = np.round(feat[1, 2])                # many exact ties
    s = 0.25
    rois = [[0, -30, -20, 10, 14], [1, -8, -8, -2, -2], [0, 60, 40, 200, 200], [1, 66, 50, 70, 54],
            [0, 10, 10, 10, 10], [1, 10, 10, 10.4, 10.2], [0, 30, 30, 10, 10], [1, 40, 8, 8, 40],
            [0, 2, 2, 6, 6], [1, 6, 6, 10, 10], [0, 10, 14, 30, 18], [1, 0, 0, 67.9, 51.9],
            [0, 0, 0, 4 * W + 40, 4 * H + 40], [1, 17.9, 13.9, 18.1, 14.1], [0, 1.99, 2.0, 2.01, 49.99],
            [0, 16, 16, 31, 31], [0, 15.9, 15.9, 28, 28]]
    for _ in range(200):
        x1, y1 = rs.uniform(-20, 4 * W + 10), rs.uniform(-20, 4 * H + 10)
        rois.append([rs.randint(0, B), x1, y1, x1 + rs.uniform(-5, 60), y1 + rs.uniform(-5, 60)])
    for _ in range(60):                              # coordinates on the .5 rounding boundary after scaling
        x1, y1 = 4 * rs.randint(-2, W) + 2, 4 * rs.randint(-2, H) + 2
        rois.append([rs.randint(0, B), x1, y1, x1 + 4 * rs.randint(0, 8) + 2, y1 + 4 * rs.randint(0, 8)])
    rois = np.asarray(rois, np.float32)
    for size in ((3, 3), (1, 1), (2, 5)):
        ref, ref_arg = _roipool_numpy(feat, rois, size[0], size[1], s)
        out, arg = O.roi_pool_argmax(torch.from_numpy(feat), torch.from_numpy(rois), size, s)
        assert np.array_equal(out.numpy(), ref)
        assert np.array_equal(arg.numpy(), ref_arg)
    assert (ref_arg == -1).any() and (ref_arg >= 0).any()


def test_extension_oracle_reduces_to_reference_formulation():
    """The multi-head / stacked GAT self-oracle with one head and one layer IS the reference's layer, and
    the extension state_dict layout only adds keys (defaults keep the reference's 50)."""
    from cova_web_object_detection_amd import weights
    fx = np.load(GOLDEN + "/gat_layer.npz")
    sd = {"gat." + k[2:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("w/")}
    h, ctx = torch.from_numpy(fx["h"]), torch.from_numpy(fx["ctx"])
    hp, attn = O.gat_stack(h, ctx, sd)
    assert torch.equal(hp, O.gat(h, ctx, sd)) and np.allclose(attn.numpy(), fx["attn"], atol=1e-7, rtol=1e-5)
    # two heads = two independent reference layers side by side; two layers = composition
    D = sd["gat.W_i.weight"].shape[0]
    sd2 = {}
    for hd in (0, 1):
        for k, v in sd.items():
            sd2[k.replace("gat.", "gat.layers.0.heads.%d." % hd)] = v * (1.0 + hd)
    rs = np.random.RandomState(0)
    for k, v in sd.items():
        shape = (D, 2 * D) if k.endswith(("W_i.weight", "W_j.weight")) else tuple(v.shape)
        sd2[k.replace("gat.", "gat.layers.1.heads.0.")] = torch.from_numpy(rs.standard_normal(shape).astype(np.float32)) * 0.1
    assert O.gat_prefixes(sd2) == [["gat.layers.0.heads.0.", "gat.layers.0.heads.1."], ["gat.layers.1.heads.0."]]
    assert O.gat_prefixes(sd2)[0] == weights.gat_prefixes(2, 2)[0]
    l0 = torch.cat([O.gat(h, ctx, sd2, prefix=p) for p in O.gat_prefixes(sd2)[0]], dim=1)
    assert torch.equal(l0[:, :D], hp)
    out, _ = O.gat_stack(h, ctx, sd2)
    assert torch.equal(out, O.gat(l0, ctx, sd2, prefix="gat.layers.1.heads.0."))
    assert len(weights.state_dict_spec()) == 50
    r50 = dict(weights.state_dict_spec(backbone="resnet50", n_heads=2))
    assert r50["convnet.4.0.downsample.0.weight"] == (256, 64, 1, 1) and r50["convnet.4.2.conv1.weight"] == (64, 256, 1, 1)
    assert r50["gat.layers.0.heads.1.W_j.weight"] == (192, 2336) and r50["decoder.1.weight"] == (2720, 2720)


def test_resnet50_oracle_wiring_is_torchvision_bottleneck():
    """The bottleneck path of oracle.convnet against an independently assembled nn.Module graph with
    torchvision's Bottleneck wiring (conv1x1-bn-relu, conv3x3-bn-relu, conv1x1-bn, (+downsample), add, relu)."""
    import torch.nn as nn
    from cova_web_object_detection_amd import weights
    sd = weights.seeded_state_dict(9, backbone="resnet50")

    class Bott(nn.Module):
        def __init__(self, cin, down):
            super().__init__()
            self.conv1, self.bn1 = nn.Conv2d(cin, 64, 1, bias=False), nn.BatchNorm2d(64)
            self.conv2, self.bn2 = nn.Conv2d(64, 64, 3, 1, 1, bias=False), nn.BatchNorm2d(64)
            self.conv3, self.bn3 = nn.Conv2d(64, 256, 1, bias=False), nn.BatchNorm2d(256)
            self.downsample = nn.Sequential(nn.Conv2d(cin, 256, 1, bias=False), nn.BatchNorm2d(256)) if down else None

        def forward(self, x):
            o = torch.relu(self.bn1(self.conv1(x)))
            o = torch.relu(self.bn2(self.conv2(o)))
            o = self.bn3(self.conv3(o))
            return torch.relu(o + (x if self.downsample is None else self.downsample(x)))

    net = nn.Sequential(nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(), nn.MaxPool2d(3, 2, 1),
                        nn.Sequential(Bott(64, True), Bott(256, False), Bott(256, False)))
    net.load_state_dict({k[len("convnet."):]: v for k, v in sd.items() if k.startswith("convnet.")}, strict=True)
    x = torch.rand(2, 3, 48, 80, generator=torch.Generator().manual_seed(1))
    for training in (False, True):
        net.train(training)
        got = O.convnet(x, O.clone_state_dict(sd), training)
        with torch.no_grad():
            ref = net(x)
        assert got.shape == (2, 256, 12, 20) and torch.allclose(got, ref, atol=1e-6, rtol=1e-5)


def test_roialign_self_oracle_properties():
    """The RoIAlign restatement (extension) on cases with known answers: a constant map pools to the constant, a box
    more than a pixel outside the map pools to 0, a 1x1-bin box centred on a pixel with one sample returns the
    bilinear value, and every inside sample distributes a total weight of 1 / count to the map (sum of the gradient)."""
    c = torch.full((1, 2, 9, 11), 3.25)
    box = torch.tensor([[0, 4.0, 6.0, 30.0, 28.0]])
    for sr, al in ((2, False), (0, False), (3, True)):
        assert torch.allclose(O.roi_align(c, box, (3, 3), 0.25, sr, al), torch.full((1, 2, 3, 3), 3.25), atol=1e-6)
    assert float(O.roi_align(c, torch.tensor([[0, 100.0, 100.0, 130.0, 120.0]]), (3, 3), 0.25).abs().max()) == 0.0
    assert float(O.roi_align(c, torch.tensor([[5, 4.0, 4.0, 20.0, 20.0]]), (3, 3), 0.25).abs().max()) == 0.0   # bad page
    rs = np.random.RandomState(1)
    f = torch.from_numpy(rs.standard_normal((1, 1, 6, 7)).astype(np.float32)).requires_grad_(True)
    # box [2.3, 1.6] .. +1 (feature units after scale 1): one 1x1 bin, one sample at its centre (2.8, 2.1)
    o = O.roi_align(f, torch.tensor([[0, 2.3, 1.6, 3.3, 2.6]]), (1, 1), 1.0, 1, False)
    x, y = 2.8, 2.1
    exp = (1 - (y - 2)) * (1 - (x - 2)) * f[0, 0, 2, 2] + (1 - (y - 2)) * (x - 2) * f[0, 0, 2, 3] + \
        (y - 2) * (1 - (x - 2)) * f[0, 0, 3, 2] + (y - 2) * (x - 2) * f[0, 0, 3, 3]
    assert abs(float(o) - float(exp)) < 1e-6
    o = O.roi_align(f, torch.tensor([[0, 0.5, 0.5, 5.0, 4.0]]), (2, 3), 1.0, 2, False)
    o.sum().backward()
    assert abs(float(f.grad.sum()) - 6.0) < 1e-5          # 6 bins, each a unit of weight, all samples inside


def _roialign_dense(feat, rois, PH, PW, scale, sampling_ratio, aligned):
    """A second, independent RoIAlign formulation (float64, no per-neighbour case analysis): bilinear interpolation is
    the tensor-product hat-function interpolant of the pixel grid, so a sample at (y, x) is  hat_y @ feat @ hat_x  with
    hat_y[i] = max(0, 1 - |clip(y, 0, H-1) - i|); a bin is the mean of its samples' DENSE weight maps; samples more
    than one pixel outside contribute nothing (Mask R-CNN, section 3 / torchvision ops.RoIAlign contract).  Returns
    (out [N,C,PH,PW], weight maps [N,PH,PW,H,W]) -- the transposed maps are the backward."""
    B, C, H, W = feat.shape
    N = rois.shape[0]
    wmap = np.zeros((N, PH, PW, H, W))
    off = 0.5 if aligned else 0.0
    for n in range(N):
        b, x1, y1, x2, y2 = [float(v) for v in rois[n]]
        sw, sh = x1 * scale - off, y1 * scale - off
        rw, rh = x2 * scale - off - sw, y2 * scale - off - sh
        if not aligned:
            rw, rh = max(rw, 1.0), max(rh, 1.0)
        gh = sampling_ratio if sampling_ratio > 0 else int(np.ceil(rh / PH))
        gw = sampling_ratio if sampling_ratio > 0 else int(np.ceil(rw / PW))
        if gh <= 0 or gw <= 0:
            continue
        ys = sh + (rh / PH) * (np.arange(PH)[:, None] + (np.arange(gh)[None, :] + 0.5) / gh)      # [PH, gh]
        xs = sw + (rw / PW) * (np.arange(PW)[:, None] + (np.arange(gw)[None, :] + 0.5) / gw)      # [PW, gw]
        hat_y = np.maximum(0.0, 1.0 - np.abs(np.clip(ys, 0, H - 1)[..., None] - np.arange(H)))   # [PH, gh, H]
        hat_x = np.maximum(0.0, 1.0 - np.abs(np.clip(xs, 0, W - 1)[..., None] - np.arange(W)))   # [PW, gw, W]
        hat_y *= ((ys >= -1.0) & (ys <= H))[..., None]
        hat_x *= ((xs >= -1.0) & (xs <= W))[..., None]
        wmap[n] = np.einsum("pih,qjw->pqhw", hat_y, hat_x) / (gh * gw)
    page = rois[:, 0].astype(int)
    out = np.einsum("npqhw,nchw->ncpq", wmap, feat.astype(np.float64)[page])
    return out, wmap


@pytest.mark.parametrize("sr,aligned", [(2, False), (0, False), (3, True), (0, True)])
def test_roialign_against_independent_dense_formulation(sr, aligned):
    """oracle.roi_align (the checker of the RoIAlign kernels; torchvision is not installed, so it is a restatement
    of the published algorithm) against the dense hat-function formulation above: values AND the backward, on boxes
    that cross every border, lie partly or wholly more than a pixel outside, are smaller than one feature pixel
    (clamped to 1 unless `aligned`), inverted, and on two pages."""
    rs = np.random.RandomState(21)
    B, C, H, W = 2, 3, 11, 14
    s = 0.25
    feat = rs.standard_normal((B, C, H, W)).astype(np.float32)
    rois = [[0, -9.3, -7.1, 20.7, 18.2], [1, 30.2, 21.9, 90.4, 77.7], [0, 50.1, 40.3, 54.2, 41.1],
            [1, 10.3, 10.2, 10.9, 10.6], [0, 3.3, 2.2, 51.7, 40.4], [1, -60.5, -50.5, -12.2, -11.1],
            [0, 44.4, 30.3, 70.7, 60.6], [1, 20.6, 30.1, 8.3, 12.2], [0, 0.7, 0.3, 55.1, 43.2]]
    for _ in range(40):
        x1, y1 = rs.uniform(-12, 4 * W + 6), rs.uniform(-12, 4 * H + 6)
        rois.append([rs.randint(0, B), x1, y1, x1 + rs.uniform(-3, 40), y1 + rs.uniform(-3, 40)])
    rois = np.asarray(rois, np.float32)
    PH, PW = 3, 2
    f = torch.from_numpy(feat).requires_grad_(True)
    out = O.roi_align(f, torch.from_numpy(rois), (PH, PW), s, sr, aligned)
    ref, wmap = _roialign_dense(feat, rois, PH, PW, s, sr, aligned)
    scale = np.abs(ref).max()
    assert np.abs(out.detach().numpy() - ref).max() <= 2e-5 * scale, np.abs(out.detach().numpy() - ref).max() / scale
    assert (np.abs(ref).reshape(len(rois), -1).max(1) == 0).any()           # a box wholly outside pools to exactly 0
    assert float(out[5].detach().abs().max()) == 0.0
    g = rs.standard_normal(out.shape).astype(np.float32)
    (out * torch.from_numpy(g)).sum().backward()
    gref = np.zeros((B, C, H, W))
    page = rois[:, 0].astype(int)
    contrib = np.einsum("npqhw,ncpq->nchw", wmap, g.astype(np.float64))
    for n in range(len(rois)):
        gref[page[n]] += contrib[n]
    assert np.abs(f.grad.numpy() - gref).max() <= 2e-5 * np.abs(gref).max()


def test_maxpool_commutes_with_the_monotone_bn_relu_apply():
    """DESIGN 12.7 (next lever): relu(fma(y, scale_c, shift_c)) is monotone in y with the direction of sign(scale_c), so the
    stem's MaxPool2d(3, 2, 1) of it (models.py:49-51, torchvision's conv1 -> bn1 -> relu -> maxpool) equals the same apply on the
    3x3 / stride-2 max (scale_c >= 0) or min (scale_c < 0) of y -- bit for bit, ties included."""
    g = torch.Generator().manual_seed(5)
    y = torch.randn(2, 64, 37, 50, generator=g) * 3
    y[0, :, 4:9, 4:9] = 0.25                                   # a flat patch: ties inside the windows
    gamma = torch.randn(64, generator=g)
    gamma[3] = 0.0
    invstd, mean, beta = torch.rand(64, generator=g) + 0.1, torch.randn(64, generator=g), torch.randn(64, generator=g)
    scale = (gamma * invstd).view(1, -1, 1, 1)
    shift = (beta - mean * gamma * invstd).view(1, -1, 1, 1)
    ref = F.max_pool2d(torch.relu(torch.addcmul(shift, y, scale)), 3, 2, 1)
    picked = torch.where(scale >= 0, F.max_pool2d(y, 3, 2, 1), -F.max_pool2d(-y, 3, 2, 1))
    assert torch.equal(ref, torch.relu(torch.addcmul(shift, picked, scale)))
