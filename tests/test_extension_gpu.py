"""GPU parity tests of the extensions BASELINE.json's configs name but the reference does not contain
(it wires resnet18 and one single-head GAT layer, models.py:49,78-79): the ResNet-50-stem representation
network (3 Bottlenecks: 1x1 / 3x3 / 1x1 convolutions, 256 channels), multi-head / stacked graph attention,
non-square pages (configs[4]: 1280 x 4096) and N = 300 boxes / K = 48 neighbours per page.

The oracle for the new pieces is the build's own torch-CPU restatement (oracle/cova_oracle.py: SELF-ORACLE,
parity unpinned by the reference); geometry cases on the reference's own model use the reference-pinned oracle.
Tolerances as in test_model_gpu.py (about 5x the measured round-off): forward 5e-5 of the tensor scale, loss 2e-5, gradients 1e-4 of each
tensor's scale under forced routing.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from cova_web_object_detection_amd import engine, synthetic, weights  # noqa: E402
from cova_web_object_detection_amd._lib import call, query  # noqa: E402
from cova_web_object_detection_amd.models import CoVA  # noqa: E402
from helpers import assert_gate_flips_near_zero, compare_grads, margins_ok, routing_from_saved  # noqa: E402
from oracle import cova_oracle as O  # noqa: E402

DEV = "cuda:0"


def close(got, ref, tol, what):
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    err = float((got - ref).abs().max() / max(float(ref.abs().max()), 1e-6))
    assert err < tol, "%s: %.3e" % (what, err)


def rnd(rs, *shape, scale=1.0):
    return torch.from_numpy((rs.standard_normal(shape) * scale).astype(np.float32))


# ------------------------------------------------------------------------------------ kernels
@pytest.mark.parametrize("cin,cout", [(64, 64), (64, 256), (256, 64)])
@pytest.mark.parametrize("R", [1, 37, 2048, 40000 + 13])
def test_conv1x1_forward_variants(cin, cout, R):
    rs = np.random.RandomState(cin + cout + R)
    x, x2 = rnd(rs, R, cin), rnd(rs, R, cin)
    w = rnd(rs, cout, cin, scale=0.1)
    abc = torch.stack([torch.from_numpy(rs.uniform(0.5, 1.5, cin).astype(np.float32)),
                       rnd(rs, cin, scale=0.5), rnd(rs, cin, scale=0.3)])
    xd, x2d, wd, abcd = x.to(DEV), x2.to(DEV), w.to(DEV), abc.to(DEV)
    n = query("cova_conv1x1_num_partials", R, cin, cout)
    for pro, relu, stats in ((0, 0, 1), (1, 1, 1), (1, 0, 0), (2, 0, 1), (0, 0, 0)):
        if pro == 0:
            a = x
        elif pro == 1:
            a = abc[0] * x + abc[2]
        else:
            a = abc[0] * x + abc[1] * x2 + abc[2]
        if relu:
            a = a.clamp_min(0)
        ref = a.double() @ w.double().t()
        out = torch.full((R, cout), 7.0, device=DEV)
        part = torch.full((n, 2, cout), 3.0, device=DEV) if stats else None
        engine.conv1x1(xd, x2d if pro == 2 else None, abcd if pro else None, relu, wd, 0, out, part, R, cin, cout)
        close(out, ref, 2e-5, "conv1x1 %d->%d pro %d" % (cin, cout, pro))
        if stats:
            close(part[:, 0].double().sum(0), ref.sum(0), 1e-4 * max(1.0, np.sqrt(R) / 10), "sum y")
            close(part[:, 1].double().sum(0), (ref * ref).sum(0), 1e-4, "sum y^2")
        # transposed weight argument (the data-gradient form) gives the same product
        out_t = torch.empty((R, cout), device=DEV)
        engine.conv1x1(xd, x2d if pro == 2 else None, abcd if pro else None, relu, w.t().contiguous().to(DEV), 1,
                       out_t, None, R, cin, cout)
        assert torch.equal(out_t, out)


@pytest.mark.parametrize("cin,cout,add,mask_act,x2", [(256, 64, False, False, False), (256, 64, True, False, False),
                                                      (64, 256, True, True, False), (64, 256, True, True, True),
                                                      (64, 64, False, True, False)])
def test_conv1x1_data_gradient_epilogue(cin, cout, add, mask_act, x2):
    rs = np.random.RandomState(11 + cin + 2 * cout + add + 3 * mask_act + 5 * x2)
    R = 5000 + 29
    dy, zin = rnd(rs, R, cin), rnd(rs, R, cin)
    w = rnd(rs, cin, cout, scale=0.1)                      # forward weight of the adjoint conv: [Cin_of_this, Cout]^T
    abc = torch.stack([rnd(rs, cin), rnd(rs, cin, scale=0.5), rnd(rs, cin, scale=0.3)])
    addend, act, z, zb = rnd(rs, R, cout), rnd(rs, R, cout), rnd(rs, R, cout), rnd(rs, R, cout)
    msc = torch.from_numpy(rs.uniform(-1, 1.5, cout).astype(np.float32))
    msh = rnd(rs, cout, scale=0.3)
    mean, invstd = rnd(rs, cout, scale=0.2), torch.from_numpy(rs.uniform(0.5, 1.5, cout).astype(np.float32))
    mean2, invstd2 = rnd(rs, cout, scale=0.2), torch.from_numpy(rs.uniform(0.5, 1.5, cout).astype(np.float32))
    dz = abc[0] * dy + abc[1] * zin + abc[2]
    ref = dz.double() @ w.double()                          # w given as [Cin, Cout] + w_trans
    if add:
        ref = ref + addend.double()
    m = (act > 0) if mask_act else (torch.addcmul(msh, msc, z) > 0)
    # decisions of the fma form the kernel evaluates (addcmul on CPU is not fused): exclude the ulp-close ones
    sure = torch.ones_like(m) if mask_act else ((msc * z + msh).abs() > 1e-5)
    ref = ref * m
    n = query("cova_conv1x1_num_partials", R, cin, cout)
    out = torch.empty((R, cout), device=DEV)
    part, part2 = torch.empty((n, 2, cout), device=DEV), (torch.empty((n, 2, cout), device=DEV) if x2 else None)
    d = lambda t: t.to(DEV)
    engine.conv1x1(d(dy), d(zin), d(abc), 0, d(w), 1, out, part, R, cin, cout, addend=d(addend) if add else None,
                   act=d(act) if mask_act else None, msc=None if mask_act else d(msc), msh=None if mask_act else d(msh),
                   z=d(z), mean=d(mean), invstd=d(invstd), z2=d(zb) if x2 else None, mean2=d(mean2) if x2 else None,
                   invstd2=d(invstd2) if x2 else None, part2=part2)
    got = out.cpu().double()
    assert float(((got - ref).abs() * sure).max()) < 2e-5 * float(ref.abs().max())
    xh = ((z - mean) * invstd).double()
    close(part[:, 0].double().sum(0), (got).sum(0), 1e-4, "sum dy")
    close(part[:, 1].double().sum(0), (got * xh).sum(0), 1e-4, "sum dy*xhat")
    if x2:
        xh2 = ((zb - mean2) * invstd2).double()
        close(part2[:, 0].double().sum(0), got.sum(0), 1e-4, "sum dy (2)")
        close(part2[:, 1].double().sum(0), (got * xh2).sum(0), 1e-4, "sum dy*xhat2")


@pytest.mark.parametrize("co,ci", [(64, 64), (256, 64), (64, 256)])
@pytest.mark.parametrize("R", [1, 2, 777, 100001])
def test_conv1x1_weight_gradient(co, ci, R):
    rs = np.random.RandomState(co + 3 * ci + R)
    dy, z, act = rnd(rs, R, co), rnd(rs, R, co), rnd(rs, R, ci)
    zabc = torch.stack([rnd(rs, co), rnd(rs, co, scale=0.5), rnd(rs, co, scale=0.3)])
    aabc = torch.stack([torch.from_numpy(rs.uniform(0.5, 1.5, ci).astype(np.float32)), rnd(rs, ci), rnd(rs, ci, scale=0.3)])
    ws = torch.empty(query("cova_conv1x1_wgrad_workspace_floats", R, co, ci), device=DEV)
    d = lambda t: t.to(DEV)
    for pz, pa, relu in ((True, True, 1), (True, False, 0), (False, True, 0), (False, False, 0)):
        dz = (zabc[0] * dy + zabc[1] * z + zabc[2]) if pz else dy
        a = (aabc[0] * act + aabc[2]) if pa else act
        if relu:
            a = a.clamp_min(0)
        ref = dz.double().t() @ a.double()
        dw = torch.empty((co, ci), device=DEV)
        call("cova_conv1x1_wgrad", d(dy), d(z) if pz else None, d(zabc) if pz else None, d(act),
             d(aabc) if pa else None, relu, dw, ws, R, co, ci)
        close(dw, ref, 1e-4, "wgrad %dx%d" % (co, ci))
        dw2 = torch.empty((co, ci), device=DEV)
        call("cova_conv1x1_wgrad", d(dy), d(z) if pz else None, d(zabc) if pz else None, d(act),
             d(aabc) if pa else None, relu, dw2, ws, R, co, ci)
        assert torch.equal(dw, dw2)                          # fixed reduction order: bit-identical reruns


@pytest.mark.parametrize("R", [1, 2, 777, 60001])
@pytest.mark.parametrize("pro", [True, False])
def test_linear_form_of_conv_bn_backward(R, pro):
    """(1x1 conv 64->256, train-mode BatchNorm) backward without reading the conv output (conv1x1_lin.hip):
    the four reductions, the BatchNorm sums, dW and the data gradient against torch autograd in fp64."""
    rs = np.random.RandomState(R + pro)
    d = lambda t: t.to(DEV)
    act, v = rnd(rs, R, 64), rnd(rs, R, 256) * (torch.from_numpy(rs.rand(R, 256)) > 0.4)
    aabc = torch.stack([torch.from_numpy(rs.uniform(0.5, 1.5, 64).astype(np.float32)), rnd(rs, 64), rnd(rs, 64, scale=0.3)])
    w = rnd(rs, 256, 64, scale=0.15)
    gamma = torch.from_numpy(rs.uniform(0.5, 1.5, 256).astype(np.float32))
    a = ((aabc[0] * act + aabc[2]).clamp_min(0) if pro else act).double()
    # ---- the four reductions
    nl = query("cova_conv1x1_lin_floats")
    lin, lin2 = torch.empty(nl, device=DEV), torch.empty(nl, device=DEV)
    ws = torch.empty(query("cova_conv1x1_vprod_workspace_floats", R), device=DEV)
    for out in (lin, lin2):
        call("cova_conv1x1_vprod", d(v), d(act), d(aabc) if pro else None, 1 if pro else 0, out, ws, R)
    assert torch.equal(lin, lin2)                            # fixed reduction order
    P, G, S, SU = lin[:16384].view(256, 64), lin[16384:20480].view(64, 64), lin[20480:20544], lin[20544:]
    close(P, v.double().t() @ a, 2e-5, "P")
    close(G, a.t() @ a, 2e-5, "G")
    close(S, a.sum(0), 2e-5, "S")
    close(SU, v.double().sum(0), 1e-4, "SU")
    if R < 2:
        return
    # ---- autograd reference of z = a W^T -> train-mode BatchNorm, upstream gradient v
    a_r, w_r, g_r = a.clone().requires_grad_(True), w.double().requires_grad_(True), gamma.double().requires_grad_(True)
    z = a_r @ w_r.t()
    mean, var = z.mean(0), z.var(0, unbiased=False)
    invstd = 1.0 / torch.sqrt(var + 1e-5)
    ((z - mean) * invstd * g_r).backward(v.double())
    mean32, invstd32 = mean.detach().float(), invstd.detach().float()
    part = torch.empty((1, 2, 256), device=DEV)
    call("cova_conv1x1_lin_bnsums", lin, d(w), d(mean32), d(invstd32), part)
    close(part[0, 1], g_r.grad, 1e-4, "dgamma = sum v*xhat")
    close(part[0, 0], v.double().sum(0), 1e-4, "dbeta")
    dg, db, abc = torch.empty(256, device=DEV), torch.empty(256, device=DEV), torch.empty((3, 256), device=DEV)
    call("cova_bn_finalize_bwd_abc", part, 1, 256, float(R), dg, db, d(mean32), d(invstd32),
         d((gamma * invstd32)), abc)
    dw, mm, cvec, avec = (torch.empty((256, 64), device=DEV), torch.empty((64, 64), device=DEV),
                          torch.empty(64, device=DEV), torch.empty((3, 256), device=DEV))
    call("cova_conv1x1_lin_finish", lin, abc, d(w), dw, mm, cvec, avec)
    close(dw, w_r.grad, 1e-4, "dW")
    assert torch.equal(avec[0], abc[0]) and not avec[1:].any()
    # ---- data gradient (+ addend), masked by the BatchNorm in front, with that layer's sums
    z2 = rnd(rs, R, 64)
    msc, msh = torch.from_numpy(rs.uniform(-1, 1.5, 64).astype(np.float32)), rnd(rs, 64, scale=0.3)
    mean2, invstd2 = rnd(rs, 64, scale=0.2), torch.from_numpy(rs.uniform(0.5, 1.5, 64).astype(np.float32))
    addend = rnd(rs, R, 64)
    ident = torch.stack([torch.ones(64), torch.zeros(64), torch.zeros(64)])
    n = query("cova_conv1x1_lin_dgrad_num_partials", R)
    for add in (False, True):
        out, sp = torch.empty((R, 64), device=DEV), torch.empty((n, 2, 64), device=DEV)
        call("cova_conv1x1_lin_dgrad", d(v), avec, d(w), d(act), d(aabc) if pro else d(ident), 1 if pro else 0, mm,
             cvec, d(addend) if add else None, d(msc), d(msh), d(z2), d(mean2), d(invstd2), out, sp, R)
        ref = a_r.grad + (addend.double() if add else 0.0)
        m = torch.addcmul(msh, msc, z2) > 0
        sure = (msc * z2 + msh).abs() > 1e-5
        ref = ref * m
        got = out.cpu().double()
        assert float(((got - ref).abs() * sure).max()) < 1e-4 * float(ref.abs().max()), (add, R)
        xh = ((z2 - mean2) * invstd2).double()
        close(sp[:, 0].double().sum(0), got.sum(0), 1e-4, "sum dy")
        close(sp[:, 1].double().sum(0), (got * xh).sum(0), 1e-4, "sum dy*xhat")


def test_conv1x1_masked_gradient_without_sums():
    """Block-input gradient once its BatchNorm sums come from cova_conv1x1_vprod: (acc + addend) * [act > 0]."""
    rs = np.random.RandomState(77)
    R = 3001
    dy, zin = rnd(rs, R, 64), rnd(rs, R, 64)
    w = rnd(rs, 64, 256, scale=0.1)
    abc = torch.stack([rnd(rs, 64), rnd(rs, 64, scale=0.5), rnd(rs, 64, scale=0.3)])
    addend, act = rnd(rs, R, 256), rnd(rs, R, 256)
    d = lambda t: t.to(DEV)
    ref = ((abc[0] * dy + abc[1] * zin + abc[2]).double() @ w.double() + addend.double()) * (act > 0)
    out = torch.empty((R, 256), device=DEV)
    engine.conv1x1(d(dy), d(zin), d(abc), 0, d(w), 1, out, None, R, 64, 256, addend=d(addend), act=d(act))
    close(out, ref, 2e-5, "masked data gradient")


@pytest.mark.parametrize("R", [1, 37, 5000 + 3])
def test_conv1x1_materializing_consumer(R):
    """256->64 forward on relu(A*in + B*in2 + C) that also writes that input (the previous Bottleneck's output)."""
    rs = np.random.RandomState(R)
    z, x = rnd(rs, R, 256), rnd(rs, R, 256)
    abc = torch.stack([torch.from_numpy(rs.uniform(0.5, 1.5, 256).astype(np.float32)),
                       torch.from_numpy(rs.uniform(0.5, 1.5, 256).astype(np.float32)), rnd(rs, 256, scale=0.3)])
    w = rnd(rs, 64, 256, scale=0.1)
    d = lambda t: t.to(DEV)
    a = (abc[0] * z + abc[1] * x + abc[2]).clamp_min(0).double()
    n = query("cova_conv1x1_num_partials", R, 256, 64)
    for stats in (True, False):
        side, out = torch.full((R, 256), 9.0, device=DEV), torch.empty((R, 64), device=DEV)
        part = torch.empty((n, 2, 64), device=DEV) if stats else None
        bits = torch.zeros((R, 8), dtype=torch.int32, device=DEV) if stats else None
        call("cova_conv1x1_materialize", d(z), d(x), d(abc), d(w), side, bits, out, part, R)
        close(side, a, 1e-6, "materialised input")
        if bits is not None:      # one bit per element: bit (c & 31) of word c >> 5 = (side[r, c] > 0)
            sh = torch.arange(32, device=DEV, dtype=torch.int32).view(1, 1, 32)
            got = ((bits.view(R, 8, 1) >> sh) & 1).view(R, 256).bool()
            assert torch.equal(got, side > 0)
            # ... and the data gradient that takes its mask from them equals the one reading the map
            rs2 = np.random.RandomState(R + 1)
            dy, zin, add = rnd(rs2, R, 64), rnd(rs2, R, 64), rnd(rs2, R, 256)
            abc2, w2 = rnd(rs2, 3, 64), rnd(rs2, 64, 256, scale=0.1)
            o_act, o_bits = torch.empty((R, 256), device=DEV), torch.empty((R, 256), device=DEV)
            engine.conv1x1(d(dy), d(zin), d(abc2), 0, d(w2), 1, o_act, None, R, 64, 256, addend=d(add), act=side)
            engine.conv1x1(d(dy), d(zin), d(abc2), 0, d(w2), 1, o_bits, None, R, 64, 256, addend=d(add), act_bits=bits)
            assert torch.equal(o_act, o_bits)
        # the product is taken on exactly the values that were written
        close(out, side.cpu().double() @ w.double().t(), 2e-5, "conv on the materialised input")
        if stats:
            close(part[:, 0].double().sum(0), out.cpu().double().sum(0), 1e-4, "sum y")
        ref = torch.empty((R, 64), device=DEV)
        engine.conv1x1(d(z), d(x), d(abc), 1, d(w), 0, ref, None, R, 256, 64)
        assert torch.equal(out, ref)


def test_bn_act2():
    rs = np.random.RandomState(5)
    R, C = 1000, 256
    z, z2 = rnd(rs, R, C), rnd(rs, R, C)
    s, h, s2, h2 = rnd(rs, C), rnd(rs, C), rnd(rs, C), rnd(rs, C)
    out = torch.empty((R, C), device=DEV)
    call("cova_bn_act2_fwd", z.to(DEV), s.to(DEV), h.to(DEV), z2.to(DEV), s2.to(DEV), h2.to(DEV), out, R, C, 1)
    close(out, ((s * z + h) + (s2 * z2 + h2)).clamp_min(0), 1e-6, "bn_act2")


# ------------------------------------------------------------------------------------ whole model
def run_case(cfg_kw, img_h, img_w, boxes, cs, seed, hidden=96, tight=1e-4, bbox_hidden=32):
    cfg = dict(roi_output_size=(3, 3), n_classes=4, use_context=True, hidden_dim=hidden, bbox_hidden_dim=bbox_hidden,
               n_additional_feat=0, drop_prob=0.0)
    sd = weights.seeded_state_dict(seed, logit_gain=4.0, **{k: v for k, v in cfg.items() if k != "drop_prob"},
                                   **cfg_kw)
    batch = synthetic.make_batch(len(boxes), img_h=img_h, img_w=img_w, boxes_per_page=boxes, context_size=cs,
                                 seed=seed)
    m = CoVA((3, 3), img_h, 4, True, hidden, bbox_hidden, 0, 0.0, None, **cfg_kw)
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV)
    args = [batch[k].to(DEV) for k in ("images", "bboxes", "additional_feats", "context_indices")]
    ocfg = dict(cfg)
    # ---- eval
    m.eval()
    with torch.no_grad():
        logits = m(*args)
    ref = O.forward(O.clone_state_dict(sd), batch["images"], batch["bboxes"], batch["additional_feats"],
                    batch["context_indices"], ocfg, False)
    err = float((logits.cpu() - ref).abs().max() / ref.abs().max())
    assert err < 5e-5, err                  # (gates at ~5x the measured errors, as tests/test_model_gpu.py)
    ok = margins_ok(ref, 10 * max(err, 1e-6) * float(ref.abs().max()))
    assert torch.equal(logits.argmax(1).cpu()[ok], ref.argmax(1)[ok])
    # ---- train: loss, gradients under forced routing, running statistics
    m.train()
    logits = m(*args)
    routing = routing_from_saved(logits.grad_fn.sv)
    loss = F.cross_entropy(logits, batch["labels"].to(DEV), reduction="sum")
    loss.backward()
    loss_ref, logits_ref, grads_ref, after, _ = O.loss_and_grads(
        sd, batch["images"], batch["bboxes"], batch["additional_feats"], batch["context_indices"],
        batch["labels"], ocfg, None, routing)
    assert float((logits.detach().cpu() - logits_ref).abs().max() / logits_ref.abs().max()) < 5e-5
    assert abs(loss.item() - float(loss_ref)) <= 2e-5 * abs(float(loss_ref))
    grads = {k: p.grad for k, p in m.named_parameters()}
    assert set(grads) == set(grads_ref)
    compare_grads(grads, grads_ref, rtol=tight, outlier_frac=0.0)
    tap = {}
    O.loss_and_grads(sd, batch["images"], batch["bboxes"], batch["additional_feats"], batch["context_indices"],
                     batch["labels"], ocfg, None, {"_tap": tap})
    assert_gate_flips_near_zero(routing, tap)
    for k, buf in m.named_buffers():
        if k.endswith("num_batches_tracked"):
            assert int(buf) == 1
        else:
            close(buf, after[k], 1e-4, k)
    return m, batch


def test_resnet50_two_head_gat_matches_self_oracle():                # BASELINE.json configs[2] architecture
    run_case(dict(backbone="resnet50", n_heads=2), 64, 64, [23, 31], 6, 31)


def test_resnet50_two_head_two_layer_nonsquare():                    # configs[4] architecture, 1:3.2 page
    run_case(dict(backbone="resnet50", n_heads=2, n_gat_layers=2), 160, 96, [40, 9], 24, 32)


def test_resnet50_odd_page_size_ragged_tiles():
    """100 x 132 pages: 25 x 33 feature map, 1650 pixel rows per batch -- partial tiles in every 1x1 / 3x3 kernel."""
    run_case(dict(backbone="resnet50", n_heads=2), 100, 132, [12, 7], 6, 35)


def test_multi_head_gat_on_reference_backbone():
    run_case(dict(n_heads=4, n_gat_layers=2), 64, 64, [17, 30], 12, 33)


def test_reference_model_on_config5_geometry():
    """configs[4]'s geometry on the reference's own architecture (reference-pinned oracle): a long,
    non-square page (H:W = 3.2), 300 boxes on it, K = 48 neighbours (-cs 24)."""
    run_case(dict(), 256, 80, [300], 24, 34, hidden=384)


def test_without_positional_encoder_bbox_hidden_dim_zero():
    """`-bbhd 0` (/root/reference utils.py:21, models.py:145-146: _get_bbox_features returns an empty [N, 0] tensor and
    the concatenation is visual || additional only) on the reference architecture, against the reference-pinned oracle:
    eval logits + decisions, train loss, all gradients, running statistics."""
    m, _ = run_case(dict(), 64, 64, [31, 18], 12, 51, hidden=384, bbox_hidden=0)
    assert not any(k.startswith("bbox_feat_encoder") for k in m.state_dict())


def test_context_size_32_fills_the_wave_wide_neighbour_table():
    """`-cs 32` => K = 64 neighbour slots, exactly one wavefront (64 lanes = 64 slots):
    pages of 70 / 9 boxes, so rows mix full windows, -1 padding and (second page) rows that are mostly padding."""
    run_case(dict(), 64, 64, [70, 9], 32, 52, hidden=384)


@pytest.mark.parametrize("cs,boxes", [(50, [130, 9]), (100, [230, 40]), (150, [330, 25]), (290, [600]), (512, [70, 520])])
def test_context_size_beyond_one_wavefront(cs, boxes):
    """`-cs 50` ... `-cs 512` => K = 100 ... 1024 neighbour slots (/root/reference models.py:171-177 and utils.py:19 take any
    value): two, four, five, twelve (K = 580) and sixteen 64-lane passes per node in the GAT kernels, against the
    reference-pinned oracle at model level (eval logits + decisions, train loss, every gradient)."""
    run_case(dict(), 64, 64, boxes, cs, 53, hidden=96)


def test_context_size_beyond_the_kernel_limit_is_refused():
    m = CoVA((3, 3), 64, 4, True, 32, 16, 0, 0.0, None).to(DEV)
    batch = synthetic.make_batch(1, img_h=64, boxes_per_page=[12], context_size=513, seed=5)
    with pytest.raises(ValueError, match="n_context > 1024"):
        m(*[batch[k].to(DEV) for k in ("images", "bboxes", "additional_feats", "context_indices")])


def test_eval_mode_backward_uses_frozen_batchnorm():
    """model.eval() + backward (frozen-BN fine-tuning, saliency): BatchNorm is a fixed affine map, so
    dz = scale*dy (torch's eval-mode batch_norm backward), not the train-mode formula."""
    for kw in (dict(), dict(backbone="resnet50", n_heads=2)):
        cfg = dict(roi_output_size=(3, 3), n_classes=4, use_context=True, hidden_dim=96, bbox_hidden_dim=32,
                   n_additional_feat=0, drop_prob=0.0)
        sd = weights.seeded_state_dict(41, logit_gain=4.0, **{k: v for k, v in cfg.items() if k != "drop_prob"}, **kw)
        batch = synthetic.make_batch(2, img_h=64, boxes_per_page=[19, 26], context_size=6, seed=41)
        m = CoVA((3, 3), 64, 4, True, 96, 32, 0, 0.0, None, **kw)
        m.load_state_dict(sd, strict=True)
        m = m.to(DEV).eval()
        args = [batch[k].to(DEV) for k in ("images", "bboxes", "additional_feats", "context_indices")]
        logits = m(*args)
        routing = routing_from_saved(logits.grad_fn.sv)
        F.cross_entropy(logits, batch["labels"].to(DEV), reduction="sum").backward()
        _, _, grads_ref, after, _ = O.loss_and_grads(sd, batch["images"], batch["bboxes"], batch["additional_feats"],
                                                     batch["context_indices"], batch["labels"], cfg, None, routing,
                                                     training=False)
        compare_grads({k: p.grad for k, p in m.named_parameters()}, grads_ref, rtol=2e-4, outlier_frac=0.0)
        for k, buf in m.named_buffers():                     # eval mode leaves the running statistics alone
            assert torch.equal(buf.cpu(), sd[k])


def test_full_size_long_page_train_step_properties():
    """One train step of the reference architecture on a full-size configs[4] page (4096 x 1280, 300 boxes,
    K = 48): finite, BatchNorm output moments, and the crop property of the conv stack against torch-CPU."""
    H, W = 4096, 1280
    cfg = dict(roi_output_size=(3, 3), n_classes=4, use_context=True, hidden_dim=384, bbox_hidden_dim=32,
               n_additional_feat=0, drop_prob=0.0)
    sd = weights.seeded_state_dict(7, **{k: v for k, v in cfg.items() if k != "drop_prob"})
    boxes = synthetic.make_boxes_only(1, H, W, 300, 24, 7)
    g = torch.Generator(device=DEV).manual_seed(7)
    images = torch.rand((1, 3, H, W), generator=g, device=DEV)
    params = {k: sd[k].to(DEV) for k in O.param_keys(sd)}
    buffers = {k: v.to(DEV) for k, v in sd.items() if k not in params}
    logits, sv = engine.model_fwd(cfg, params, buffers, images, boxes["bboxes"].to(DEV),
                                  boxes["additional_feats"].to(DEV), boxes["context_indices"].to(DEV), True)
    loss, dl, pred = engine.ce_sum(logits, boxes["labels"].to(DEV))
    grads = engine.model_bwd(sv, dl, params)
    assert logits.shape == (300, 4) and torch.isfinite(logits).all() and torch.isfinite(loss).all()
    for k, gk in grads.items():
        assert torch.isfinite(gk).all(), k
    # crop property: the top-left 64x64 corner of the feature map depends on the top-left ~300x300 pixels only
    # when BatchNorm uses fixed (eval) statistics
    fresh = {k: v.to(DEV) for k, v in sd.items() if k not in params}      # the train step advanced `buffers`
    feat, _ = engine.convstack_fwd(images, params, fresh, False, save=False)
    crop = images[:, :, :384, :384].cpu()
    ref = O.convnet(crop, O.clone_state_dict(sd), False)[:, :, :64, :64]
    close(feat[0, :64, :64].permute(2, 0, 1).unsqueeze(0), ref, 1e-4, "long-page crop")
    assert feat.shape == (1, 1024, 320, 64)


def _bn_moment_checks(conv, keys_bn):
    """BatchNorm identities of a train step (size independent): the normalised output of every conv-stack BatchNorm has
    per-channel mean 0 and variance 1 over the batch (checked through the saved pre-activations and the saved
    mean / invstd, in fp64)."""
    for z, st in keys_bn:
        zz = z.reshape(-1, st.C)
        mu = torch.zeros(st.C, dtype=torch.float64, device=zz.device)
        sq = torch.zeros(st.C, dtype=torch.float64, device=zz.device)
        for r0 in range(0, zz.shape[0], 1 << 22):           # fp64 sums in slabs (the maps are up to 3.4 GB)
            zd = zz[r0:r0 + (1 << 22)].double()
            mu += zd.sum(0)
            sq += (zd * zd).sum(0)
        mu /= zz.shape[0]
        var = (sq / zz.shape[0] - mu * mu).clamp_min(0)
        assert float((mu - st.mean.double()).abs().max() / (mu.abs().max() + 1e-6)) < 1e-4
        assert float((1.0 / torch.sqrt(var + 1e-5) - st.invstd.double()).abs().max() / st.invstd.double().abs().max()) < 1e-4


def test_config5_model_on_a_full_length_page():
    """BASELINE.json configs[4] with ITS OWN model (ResNet-50 stem, 2 heads x 2 layers, 300 boxes, K = 48) on one
    full 4096 x 1280 page: finite train step, BatchNorm moment identities at full size, bit-identical rerun, and the
    top 512 pixel rows against the self-oracle's conv stack with the step's batch statistics frozen in (eval-mode
    crop property: the first 96 feature rows depend on the first ~400 pixel rows only)."""
    H, W = 4096, 1280
    kw = dict(backbone="resnet50", n_heads=2, n_gat_layers=2)
    cfg = dict(roi_output_size=(3, 3), n_classes=4, use_context=True, hidden_dim=384, bbox_hidden_dim=32,
               n_additional_feat=0, drop_prob=0.0, **kw)
    sd = weights.seeded_state_dict(9, **{k: v for k, v in cfg.items() if k != "drop_prob"})
    boxes = synthetic.make_boxes_only(1, H, W, 300, 24, 9)
    g = torch.Generator(device=DEV).manual_seed(9)
    images = torch.rand((1, 3, H, W), generator=g, device=DEV)
    params = {k: sd[k].to(DEV) for k in O.param_keys(sd)}

    def step():
        buffers = {k: v.to(DEV) for k, v in sd.items() if k not in params}
        logits, sv = engine.model_fwd(cfg, params, buffers, images, boxes["bboxes"].to(DEV),
                                      boxes["additional_feats"].to(DEV), boxes["context_indices"].to(DEV), True)
        loss, dl, _ = engine.ce_sum(logits, boxes["labels"].to(DEV))
        grads = engine.model_bwd(sv, dl, params)
        return logits, loss, grads, sv, buffers

    logits, loss, grads, sv, buffers = step()
    assert logits.shape == (300, 4) and torch.isfinite(logits).all() and torch.isfinite(loss).all()
    assert set(grads) == set(params)
    for k, gk in grads.items():
        assert torch.isfinite(gk).all(), k
    conv = sv["conv"]
    pairs = [(conv["y1"], conv["bn1"])]
    for blk in conv["blocks"]:
        pairs += [(blk["z1"], blk["bn1"]), (blk["z2"], blk["bn2"]), (blk["z3"], blk["bn3"])]
    _bn_moment_checks(conv, pairs)
    logits2, loss2, grads2, _, _ = step()
    assert torch.equal(logits, logits2) and torch.equal(loss, loss2)
    for k in grads:
        assert torch.equal(grads[k], grads2[k]), k
    # eval-mode forward with the statistics this step left in the running buffers after ONE update from (0, 1):
    # running = 0.9 * init + 0.1 * batch; compare the top crop against the self-oracle given the same buffers
    feat, _ = engine.convstack_fwd(images, params, buffers, False, save=False)
    assert feat.shape == (1, 1024, 320, 256)
    sd_eval = O.clone_state_dict(sd)
    for k, v in buffers.items():
        sd_eval[k] = v.cpu()
    ref = O.convnet(images[:, :, :512, :].cpu(), sd_eval, False)[:, :, :96, :]
    close(feat[0, :96].permute(2, 0, 1).unsqueeze(0), ref, 2e-4, "configs[4] full-length page, top crop vs self-oracle")


def test_config3_batch_of_32_pages_reproducible_with_batchnorm_identities():
    """BASELINE.json configs[2] at its full batch (32 pages of 1280 x 1280, ResNet-50 stem + 2-head GAT): the
    256-channel maps have 0.84 G elements each (32 * 320 * 320 * 256 > 2^29 floats: 64-bit image bases), two train steps
    from the same state are bit-identical, and every conv-stack BatchNorm's saved statistics are the exact moments of
    its saved pre-activation."""
    from cova_web_object_detection_amd.trainer import HotPathTrainer
    kw = dict(backbone="resnet50", n_heads=2, n_gat_layers=1)
    cfg = dict(roi_output_size=(3, 3), n_classes=4, use_context=True, hidden_dim=384, bbox_hidden_dim=32,
               n_additional_feat=0, drop_prob=0.2, **kw)
    sd = weights.seeded_state_dict(123, **{k: v for k, v in cfg.items() if k != "drop_prob"})
    pages = 32
    b = synthetic.make_boxes_only(pages, 1280, 1280, 90, 12, 123)
    batch = {k: v.to(DEV) for k, v in b.items() if torch.is_tensor(v)}
    g = torch.Generator(device=DEV).manual_seed(123)
    batch["images"] = torch.rand((pages, 3, 1280, 1280), generator=g, device=DEV)
    outs = []
    for _ in range(2):
        tr = HotPathTrainer(cfg, sd, DEV)
        loss, pred = tr.train_step(batch)
        outs.append((loss.clone(), pred.clone(), tr.pbucket.flat.clone(), tr.gbucket.flat.clone()))
        del tr
        torch.cuda.empty_cache()
    for a, c in zip(*outs):
        assert torch.equal(a, c)
    assert torch.isfinite(outs[0][0]).all() and torch.isfinite(outs[0][3]).all()
    # BatchNorm identities on one forward of the same batch (drop_prob does not touch the conv stack)
    params = {k: sd[k].to(DEV) for k in O.param_keys(sd)}
    buffers = {k: v.to(DEV) for k, v in sd.items() if k not in params}
    feat, conv = engine.convstack_fwd(batch["images"], params, buffers, True, save=True, lazy_out=True)
    pairs = [(conv["y1"], conv["bn1"])]
    for blk in conv["blocks"]:
        pairs += [(blk["z1"], blk["bn1"]), (blk["z2"], blk["bn2"]), (blk["z3"], blk["bn3"])]
    _bn_moment_checks(conv, pairs)


@pytest.mark.parametrize("kw", [dict(), dict(backbone="resnet50")])
def test_piecewise_visual_features_backward(kw):
    """The members extract_attn_wts_and_visualize.py:117-124 calls piecewise are differentiable on their own:
    d(sum(visual * g)) through `_get_visual_features` (materialised feature map, stand-alone mask + BatchNorm sums)
    against autograd of the oracle's conv stack + RoIPool, routing forced to the HIP forward's decisions."""
    cfg = dict(roi_output_size=(3, 3), n_classes=4, use_context=True, hidden_dim=48, bbox_hidden_dim=16,
               n_additional_feat=0, drop_prob=0.0)
    sd = weights.seeded_state_dict(51, **{k: v for k, v in cfg.items() if k != "drop_prob"}, **kw)
    batch = synthetic.make_batch(2, img_h=64, img_w=96, boxes_per_page=[14, 20], context_size=6, seed=51)
    m = CoVA((3, 3), 64, 4, True, 48, 16, 0, 0.0, None, **kw)
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).train()
    vis = m._get_visual_features(batch["images"].to(DEV), batch["bboxes"].to(DEV))
    rs = np.random.RandomState(2)
    g = torch.from_numpy(rs.standard_normal(tuple(vis.shape)).astype(np.float32))
    conv_sv, roi_sv = vis.grad_fn.sv, vis.grad_fn.rsv
    (vis * g.to(DEV)).sum().backward()
    # oracle with the same discrete decisions
    fake = dict(conv=conv_sv, roi=roi_sv, n_vis=vis.shape[1], Hd=0, dec=dict(y=torch.zeros(1, 1, device=DEV)))
    routing = routing_from_saved(fake)
    routing.pop("gate_dec")
    work = O.clone_state_dict(sd)
    leaves = {k: work[k].requires_grad_(True) for k in O.param_keys(work) if k.startswith("convnet.")}
    feat = O.convnet(batch["images"], work, True, routing)
    ref = O.roi_pool(feat, batch["bboxes"], (3, 3), 0.25, routing["roi_argmax"]).reshape(vis.shape)
    close(vis, ref, 1e-4, "visual features")
    (ref * g).sum().backward()
    grads = {k: p.grad for k, p in m.named_parameters() if k.startswith("convnet.")}
    compare_grads(grads, {k: v.grad for k, v in leaves.items()}, rtol=2e-4, outlier_frac=0.0)


# ------------------------------------------------------------------------------------ RoIAlign variant
@pytest.mark.parametrize("sr,aligned,C", [(2, False, 64), (0, False, 64), (2, True, 256), (3, False, 64)])
def test_roialign_kernels_match_self_oracle(sr, aligned, C):
    rs = np.random.RandomState(sr * 7 + aligned + C)
    B, H, W = 2, 14, 90
    feat = torch.from_numpy(rs.standard_normal((B, C, H, W)).astype(np.float32)).requires_grad_(True)
    n = 40
    x1, y1 = rs.uniform(-20, 340, n), rs.uniform(-12, 50, n)
    rois = torch.from_numpy(np.stack([rs.randint(0, B, n), x1, y1, x1 + rs.uniform(0.5, 150, n),
                                      y1 + rs.uniform(0.5, 40, n)], 1).astype(np.float32))
    rois[0, 1:] = torch.tensor([500., 500., 600., 600.])                 # fully outside: zeros
    ref = O.roi_align(feat, rois, (3, 3), 0.25, sr, aligned)
    g = torch.from_numpy(rs.standard_normal((n, C * 9)).astype(np.float32))
    (ref.reshape(n, -1) * g).sum().backward()
    fd = feat.detach().permute(0, 2, 3, 1).contiguous().to(DEV)
    out = torch.full((n, C * 9 + 8), -3.0, device=DEV)
    sv = engine.roialign_fwd(fd, rois.to(DEV), (3, 3), 0.25, sr, aligned, out, C * 9 + 8)
    close(out[:, :C * 9], ref.detach().reshape(n, -1), 1e-5, "roialign fwd")
    assert (out[:, C * 9:] == -3.0).all() and (out[0, :C * 9] == 0).all()
    res = []
    for rep in range(2):
        res.append(engine.roialign_bwd(sv, g.to(DEV), C * 9))
    close(res[0].permute(0, 3, 1, 2), feat.grad, 1e-5, "roialign bwd")
    assert torch.equal(res[0], res[1])                                    # no atomics: bit-identical reruns


@pytest.mark.parametrize("kw", [dict(roi_op="align"), dict(roi_op="align", backbone="resnet50", n_heads=2, sampling_ratio=0)])
def test_model_with_roialign_matches_self_oracle(kw):
    cfg = dict(roi_output_size=(3, 3), n_classes=4, use_context=True, hidden_dim=64, bbox_hidden_dim=16,
               n_additional_feat=0, drop_prob=0.0)
    wkw = {k: v for k, v in kw.items() if k in ("backbone", "n_heads", "n_gat_layers")}
    sd = weights.seeded_state_dict(61, logit_gain=4.0, **{k: v for k, v in cfg.items() if k != "drop_prob"}, **wkw)
    batch = synthetic.make_batch(2, img_h=96, img_w=128, boxes_per_page=[15, 22], context_size=6, seed=61)
    m = CoVA((3, 3), 96, 4, True, 64, 16, 0, 0.0, None, **kw)
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).train()
    args = [batch[k].to(DEV) for k in ("images", "bboxes", "additional_feats", "context_indices")]
    logits = m(*args)
    sv = logits.grad_fn.sv
    fake = dict(sv, roi=dict(argmax=torch.zeros(1, dtype=torch.int32)))
    routing = routing_from_saved(fake)
    routing.pop("roi_argmax")
    F.cross_entropy(logits, batch["labels"].to(DEV), reduction="sum").backward()
    ocfg = dict(cfg, roi_op="align", sampling_ratio=kw.get("sampling_ratio", 2))
    loss_ref, logits_ref, grads_ref, after, _ = O.loss_and_grads(
        sd, batch["images"], batch["bboxes"], batch["additional_feats"], batch["context_indices"], batch["labels"],
        ocfg, None, routing)
    close(logits, logits_ref, 2e-4, "logits (RoIAlign)")
    compare_grads({k: p.grad for k, p in m.named_parameters()}, grads_ref, rtol=2e-4, outlier_frac=0.0)
