"""GPU parity tests, model level: the drop-in CoVA / GraphAttentionLayer modules against the golden
vectors captured from the reference (tests/golden/*.npz) and against the CPU oracle.

Tolerances (fp32 path; the gates sit at about 5x the errors measured on MI355X, profiles/r03_wino4_margin.txt and
profiles/r04_parity_margin.txt -- SURVEY.md section 8c's starting point was 1e-4 / 1e-3):
  forward logits (eval and train) 5e-5 of the logit scale (measured 3e-6 .. 9e-6), CE-sum loss 2e-5 (measured <= 1.4e-6);
  gradients, every discrete decision forced to the HIP forward's: 1e-4 of each tensor's scale (measured 1.1e-5 .. 4e-5;
  scale floored at 1 % of the largest gradient: several parameters have analytically zero gradient);
  gradients against the reference's own fixtures with NOTHING forced: 2e-4 when every discrete decision of the HIP forward
  is the unforced oracle's; otherwise bounded by the number and the place of the decisions that fell the other way (each
  within round-off of its threshold, asserted): 2e-4 + 5e-3 per head gate + 1e-2 per conv-stack gate on the tensors
  upstream of a flip (the head keeps 2e-4 under conv-stack flips, the last Linear always), and the fraction of fixture
  entries beyond 2e-4 at most 0.25 per head gate + 0.015 per conv-stack gate (measured footprints, see the test);
  integer outputs exact on rows whose top-2 logit margin exceeds 10x the observed fp error.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from cova_web_object_detection_amd import engine, synthetic, weights  # noqa: E402
from cova_web_object_detection_amd.models import CoVA, GraphAttentionLayer  # noqa: E402
from helpers import (FULL_CASES, GOLDEN, assert_gate_flips_near_zero, assert_routing_near_ties,  # noqa: E402
                     check_grads, compare_grads, load_case, margins_ok, routing_from_saved, unforced_fraction_above)

LOGIT_TOL, LOSS_TOL, GRAD_TOL = 5e-5, 2e-5, 1e-4
from oracle import cova_oracle as O  # noqa: E402

DEV = "cuda:0"


def build(cfg, img_h, sd):
    m = CoVA(cfg["roi_output_size"], img_h, cfg["n_classes"], cfg["use_context"], cfg["hidden_dim"],
             cfg["bbox_hidden_dim"], cfg["n_additional_feat"], cfg["drop_prob"], None)
    missing = m.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return m.to(DEV)


def dev_batch(batch):
    return [batch[k].to(DEV) for k in ("images", "bboxes", "additional_feats", "context_indices")]


def relerr(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    return np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-6)


@pytest.mark.parametrize("fixture", ["gat_layer", "gat_layer_k100"])     # K = 6; K = 100: two 64-lane passes per node
def test_gat_layer_matches_reference_fixture(fixture):
    fx = np.load(GOLDEN + "/%s.npz" % fixture)
    N, Fd = fx["h"].shape
    D = fx["w/W_i.weight"].shape[0]
    layer = GraphAttentionLayer(Fd, D)
    layer.load_state_dict({k[2:]: torch.from_numpy(fx[k]) for k in fx.files if k.startswith("w/")})
    layer = layer.to(DEV)
    h = torch.from_numpy(fx["h"]).to(DEV).requires_grad_(True)
    ctx = torch.from_numpy(fx["ctx"]).to(DEV)
    hp, attn = layer(h, ctx, return_attn_wts=True)
    assert relerr(hp.detach().cpu(), fx["h_prime"]) < 1e-5
    assert relerr(attn.cpu(), fx["attn"]) < 1e-5
    assert np.allclose(attn[3].cpu().numpy(), 1.0 / ctx.shape[1])      # all -1 row: uniform
    assert hp[3].abs().max().item() == 0.0                              # ... and zero context
    (hp * torch.from_numpy(fx["g"]).to(DEV)).sum().backward()
    assert relerr(h.grad.cpu(), fx["grad_h"]) < 1e-4
    for k, p in layer.named_parameters():
        assert relerr(p.grad.cpu(), fx["grad/" + k]) < 1e-4, k
    assert layer(h, ctx).shape == (N, D)                                # default return


@pytest.mark.parametrize("name", FULL_CASES)
def test_full_model_matches_reference_fixture(name):
    fx, cfg, sd, batch = load_case(name)
    img_h = int(fx["meta/img_h"])
    args = dev_batch(batch)
    # ---- eval mode: logits, attention, decisions
    m = build(cfg, img_h, sd)
    m.eval()
    with torch.no_grad():
        logits = m(*args)
        visual = m._get_visual_features(args[0], args[1])
        bbf = m._get_bbox_features(args[1])
        own = torch.cat((visual, bbf, m.bn_additional_feat(args[2])), dim=1)
        hp, attn = m.gat(own, args[3], return_attn_wts=True)
    err = relerr(logits.cpu(), fx["eval/logits"])
    assert err < LOGIT_TOL, err
    assert relerr(visual.cpu().numpy().reshape(-1)[::7], fx["eval/visual_sample"]) < LOGIT_TOL
    assert relerr(bbf.cpu(), fx["eval/bbox_feats"]) < LOGIT_TOL
    assert relerr(attn.cpu(), fx["eval/attn"]) < LOGIT_TOL
    assert relerr(hp.cpu().numpy().reshape(-1)[::5], fx["eval/context_sample"]) < LOGIT_TOL
    ref_logits = torch.from_numpy(fx["eval/logits"])
    tol = 10 * max(err, 1e-6) * float(ref_logits.abs().max())
    ok = margins_ok(ref_logits, tol)
    assert ok.float().mean() > 0.9                       # fixtures were built with decisive margins
    assert torch.equal(logits.argmax(1).cpu()[ok], torch.from_numpy(fx["eval/argmax"])[ok])
    # per page / per class top-1 box (train.py:144-153), exact when the deciding margin is clear
    dec = O.page_class_decisions(logits.cpu(), batch["bboxes"], 1)
    for p, row in enumerate(fx["eval/page_class_top1"]):
        o = ref_logits[batch["bboxes"][:, 0] == p]
        for c in range(o.shape[1]):
            top2 = torch.topk(o[:, c], 2).values
            if (top2[0] - top2[1]) > tol:
                assert int(dec[p][0, c]) == int(row[c])
    # ---- train mode: batch statistics, CE-sum, backward, running stats
    m = build(cfg, img_h, sd)
    m.train()
    logits = m(*args)
    routing = routing_from_saved(logits.grad_fn.sv)      # the HIP forward's max-pool / RoIPool routing
    loss = torch.nn.CrossEntropyLoss(reduction="sum")(logits, batch["labels"].to(DEV))
    loss.backward()
    assert relerr(logits.detach().cpu(), fx["train/logits"]) < LOGIT_TOL
    assert abs(loss.item() - float(fx["train/loss"])) <= LOSS_TOL * abs(float(fx["train/loss"]))
    grads = {k: p.grad for k, p in m.named_parameters()}
    # (1) the tight check: everything against the oracle forced to the HIP forward's discrete decisions (max-pool /
    #     RoIPool argmax, ReLU gates, LeakyReLU slopes of the attention scores); the oracle itself is pinned to the
    #     reference's gradients by tests/test_oracle_cpu.py.  What remains is fp32 round-off: 1e-4 of each tensor's scale
    b = batch
    _, _, grads_ref, after, inter = O.loss_and_grads(sd, b["images"], b["bboxes"], b["additional_feats"],
                                                     b["context_indices"], b["labels"], cfg, None, routing)
    compare_grads(grads, grads_ref, rtol=GRAD_TOL, outlier_frac=0.0)
    assert_routing_near_ties(routing, inter, b["bboxes"], (3, 3), m.roi_pool.spatial_scale)
    # ... and the decisions that were forced are the oracle's own, up to pre-activations within round-off of
    # zero / exact ties: counted and bounded against the UNFORCED oracle
    tap = {}
    O.loss_and_grads(sd, b["images"], b["bboxes"], b["additional_feats"], b["context_indices"], b["labels"],
                     cfg, None, {"_tap": tap})
    flips = assert_gate_flips_near_zero(routing, tap)
    # (2) the reference's OWN gradients, nothing forced.  When every discrete decision of the HIP forward is the unforced
    #     oracle's, this one-hop comparison is as tight as (1).  A decision that sits within round-off of its threshold and
    #     falls the other way is not an error of either side, but it moves whatever lies upstream of it: measured, ONE
    #     decoder ReLU gate at a pre-activation of -4.7e-6 (scale 7.8) moved 22 % of all entries by more than 2e-4 (worst
    #     4e-3, the decoder row of that unit); one gate in the conv stack ~450 entries of a 36,864-entry weight by ~1e-3.
    #     So: no flips -> tight bound; flips (counted and placed by assert_gate_flips_near_zero above) -> loose bounds.
    nflip = sum(flips.values())
    head = {k: v for k, v in fx.items() if not (k.startswith(("grad", "gradnorm", "gradsample"))
                                                  and "/convnet." in k)}
    conv_only = {k: v for k, v in fx.items() if k.startswith(("grad", "gradnorm", "gradsample"))
                 and "/convnet." in k}
    frac, total, worst, per = unforced_fraction_above(fx, grads, rtol=2e-4)
    print("%s: %d decisions differ from the unforced oracle's %s; %d of %d unforced fixture gradient entries beyond 2e-4 "
          "(%.4f %%), worst %s %.2e" % (name, nflip, {k: v for k, v in flips.items() if v}, round(frac * total), total,
                                        100 * frac, worst, per[worst][2]))
    # Bound by the COUNT and the LOCATION of the flips.  Measured footprints on these fixtures (64 x 64 .. 128 x 128 images: a
    # single pixel is 1e-4 .. 2e-4 of a map): one decoder / positional-encoder / attention gate that falls the other way moves
    # up to 22 % of all entries by more than 2e-4, worst 4e-3 of a tensor's scale; one gate in the conv stack moves up to
    # 0.9 % of the entries, worst 9e-3 (cova_h64_addfeat: two flips, 1.7 %, 8.98e-3 on convnet.4.0.bn1.bias).
    #   * a conv-stack flip changes nothing downstream of the stack: the head's gradients keep the tight bound;
    #   * the last Linear (behind the decoder's gate: its gradient does not pass through any gate) always keeps it;
    #   * per tensor upstream of a flip: 2e-4 + 5e-3 per head flip + 1e-2 per conv-stack flip (never above the 5e-2 of the
    #     earlier blanket bound); fraction of entries beyond 2e-4: 0.25 per head flip + 0.015 per conv-stack flip.
    n_head = sum(v for k, v in flips.items() if k in ("gate_dec", "gate_bbox") or k.endswith("leaky"))
    n_conv = nflip - n_head
    last = {k: v for k, v in fx.items() if k.startswith(("grad", "gradnorm", "gradsample")) and "/decoder.5." in k}
    assert last, "the fixture holds the last Linear's gradients"
    check_grads(last, grads, rtol=2e-4)
    if nflip == 0:
        check_grads(head, grads, rtol=2e-4)
        check_grads(conv_only, grads, rtol=2e-4)
        assert frac == 0.0, (frac, worst, per[worst])
    else:
        loose = min(5e-2, 2e-4 + 5e-3 * n_head + 1e-2 * n_conv)
        check_grads(head, grads, rtol=2e-4 if n_head == 0 else loose)
        check_grads(conv_only, grads, rtol=loose)
        assert frac <= min(1.0, 0.25 * n_head + 0.015 * n_conv), (frac, flips, worst, per[worst])
    for k, buf in m.named_buffers():
        if "buf/" + k in fx:
            assert relerr(buf.cpu(), fx["buf/" + k]) < 1e-4, k
        if k.endswith("num_batches_tracked"):
            assert int(buf) == 1


def test_fused_loss_and_engine_step_match_oracle():
    """engine.model_fwd + ce_sum + model_bwd (the bench path) against the CPU oracle, with
    injected dropout keep-masks (train-mode parity needs a shared mask: SURVEY 8a row D)."""
    cfg = dict(roi_output_size=(3, 3), n_classes=4, use_context=True, hidden_dim=96,
               bbox_hidden_dim=32, n_additional_feat=0, drop_prob=0.2)
    wcfg = {k: v for k, v in cfg.items() if k != "drop_prob"}
    sd = weights.seeded_state_dict(77, logit_gain=4.0, **wcfg)
    batch = synthetic.make_batch(2, img_h=96, boxes_per_page=[30, 17], context_size=12, seed=77)
    N, T = 47, 576 + 32 + 96
    rs = np.random.RandomState(3)
    masks = [torch.from_numpy((rs.uniform(size=(N, T)) > 0.2).astype(np.uint8)) for _ in range(2)]
    params = {k: v.to(DEV) for k, v in sd.items() if k in O.param_keys(sd)}
    buffers = {k: v.to(DEV) for k, v in sd.items() if k not in params}
    images, bboxes, addl, ctx = dev_batch(batch)
    logits, sv = engine.model_fwd(cfg, params, buffers, images, bboxes, addl, ctx, True,
                                  masks=[m.to(DEV) for m in masks])
    loss, dl, pred = engine.ce_sum(logits, batch["labels"].to(DEV))
    routing = routing_from_saved(sv)
    grads = engine.model_bwd(sv, dl, params)
    loss_ref, logits_ref, grads_ref, after, inter = O.loss_and_grads(
        sd, batch["images"], batch["bboxes"], batch["additional_feats"], batch["context_indices"],
        batch["labels"], cfg, [m.float() for m in masks], routing)
    assert_routing_near_ties(routing, inter, batch["bboxes"], (3, 3), 0.25)
    assert relerr(logits.cpu(), logits_ref) < LOGIT_TOL
    assert abs(loss.item() - float(loss_ref)) <= LOSS_TOL * abs(float(loss_ref))
    compare_grads(grads, grads_ref, rtol=GRAD_TOL, outlier_frac=0.0)
    for k in buffers:
        if not k.endswith("num_batches_tracked"):
            assert relerr(buffers[k].cpu(), after[k]) < 1e-4, k


def test_module_surface_and_no_cpu_fallback():
    m = CoVA((3, 3), 64, 4, True, 384, 32, 0, 0.2, ["BG", "Price", "Title", "Image"])
    assert [k for k in m.state_dict()] == [k for k, _ in weights.state_dict_spec()]
    assert m.n_classes == 4 and m.class_names[1] == "Price"
    assert (m.n_visual_feat, m.n_feat, m.n_total_feat) == (576, 608, 992)
    batch = synthetic.make_batch(1, img_h=64, boxes_per_page=11, seed=1)
    with pytest.raises(RuntimeError):
        m(batch["images"], batch["bboxes"], batch["additional_feats"], batch["context_indices"])
    m = m.to(DEV)
    m.train()
    args = dev_batch(batch)
    m.seed_dropout(5)
    a = m(*args)
    m.seed_dropout(5)
    for b in m.buffers():
        if b.dtype == torch.long:
            b.zero_()
    b2 = m(*args)
    assert a.shape == (11, 4) and torch.isfinite(a).all()
    assert torch.equal(a, b2)              # train mode: batch statistics + the same dropout stream -> bit-identical logits
    m2 = CoVA((3, 3), 64, 4, False, 384, 32, 0, 0.0, None).to(DEV)   # use_context=False branch
    out = m2(args[0], args[1], args[2], torch.empty((0, 0), dtype=torch.long, device=DEV))
    assert out.shape == (11, 4)


def test_reference_style_train_loop_matches_fused_trainer_and_oracle():
    """The loop of train.py:42-60 (zero_grad, forward, CrossEntropyLoss(sum), backward,
    torch.optim.Adam(lr 5e-4, weight_decay 1e-3) as in main.py:133-139) driving the drop-in module,
    against (a) HotPathTrainer (fused CE + flat-buffer Adam kernel) and (b) the CPU oracle."""
    from cova_web_object_detection_amd.trainer import HotPathTrainer
    cfg = dict(roi_output_size=(3, 3), n_classes=4, use_context=True, hidden_dim=64,
               bbox_hidden_dim=16, n_additional_feat=0, drop_prob=0.0)
    wcfg = {k: v for k, v in cfg.items() if k != "drop_prob"}
    sd = weights.seeded_state_dict(31, logit_gain=2.0, **wcfg)
    batch = synthetic.make_batch(2, img_h=64, boxes_per_page=[25, 14], context_size=12, seed=31)
    args = dev_batch(batch)
    labels = batch["labels"].to(DEV)
    # (1) reference-style loop on the nn.Module
    m = build(cfg, 64, sd)
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=5e-4, weight_decay=1e-3)
    crit = torch.nn.CrossEntropyLoss(reduction="sum")
    losses = []
    for _ in range(3):
        opt.zero_grad()
        out = m(*args)
        pred = out.argmax(dim=1)
        loss = crit(out, labels)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    # (2) fused trainer
    tr = HotPathTrainer(cfg, sd, DEV)
    dbatch = dict(images=args[0], bboxes=args[1], additional_feats=args[2], context_indices=args[3],
                  labels=labels)
    tlosses = [tr.train_step(dbatch)[0].item() for _ in range(3)]
    for a, b in zip(losses, tlosses):
        assert abs(a - b) <= 1e-3 * abs(a), (losses, tlosses)
    tsd = tr.state_dict()
    # Adam turns a ~0 gradient (round-off noise on shift-invariant parameters) into a +-lr step
    # of arbitrary sign, so parameters can only be compared to within steps * lr; the loss
    # trajectory above is the tight check.
    # (running statistics are not compared: they integrate the +-lr noise of the zero-gradient
    # parameters, e.g. gat.W_i, through the forward pass)
    bad = {k: float((v.detach() - tsd[k]).abs().max()) for k, v in m.named_parameters()
           if float((v.detach() - tsd[k]).abs().max()) > 2 * 3 * 5e-4 + 1e-3 * float(v.detach().abs().max())}
    assert not bad, bad
    assert all(torch.isfinite(v).all() for v in tsd.values())
    # (3) first step vs the oracle (loss, and Adam's first update has magnitude ~lr everywhere)
    loss_ref, _, grads_ref, _, _ = O.loss_and_grads(sd, batch["images"], batch["bboxes"],
                                                    batch["additional_feats"], batch["context_indices"],
                                                    batch["labels"], cfg, None)
    assert abs(losses[0] - float(loss_ref)) <= LOSS_TOL * abs(float(loss_ref))
    logits, pred = tr.predict(dbatch)
    assert logits.shape == (39, 4) and torch.equal(pred, logits.argmax(1))


def test_training_trajectory_follows_the_cpu_oracle(request):
    """Stand-in for the reference's accuracy claim (README.md:41-42; the CoVA dataset is not available offline):
    80 Adam steps (lr 5e-4, weight_decay 1e-3: /root/reference main.py:133-139) of the HIP trainer against the CPU oracle
    (oracle.loss_and_grads + torch.optim.Adam) on a memorisable synthetic task -- three batches of two 96-pixel pages
    cycling, the same dropout keep-masks injected on both sides.  The two runs are chaotic in the last bits (ReLU gates
    near zero), so the check is a band on the loss curve and equal decisions at the end: per-box arg-max on margin
    rows and the per-page / per-class top-1 box of train.py:144-153.  The curve goes to profiles/ when COVA_WRITE_CURVE
    names a file."""
    from cova_web_object_detection_amd.trainer import HotPathTrainer
    cfg = dict(roi_output_size=(3, 3), n_classes=4, use_context=True, hidden_dim=96,
               bbox_hidden_dim=16, n_additional_feat=0, drop_prob=0.2)
    wcfg = {k: v for k, v in cfg.items() if k != "drop_prob"}
    sd = weights.seeded_state_dict(77, logit_gain=2.0, **wcfg)
    batches = [synthetic.make_batch(2, img_h=96, boxes_per_page=[30 + 3 * i, 21 + 2 * i], context_size=12, seed=700 + i)
               for i in range(3)]
    T = sd["decoder.1.weight"].shape[0]
    steps = 80
    gen = torch.Generator().manual_seed(5)
    keys = O.param_keys(sd)
    # one CPU thread: the oracle side is then run-to-run reproducible (multi-threaded reductions are not, and an earlier test
    # of the session may have changed the thread count) -- with a chaotic trajectory the band below would otherwise
    # depend on what ran before
    n_threads = torch.get_num_threads()
    torch.set_num_threads(1)
    request.addfinalizer(lambda: torch.set_num_threads(n_threads))
    tr = HotPathTrainer(cfg, sd, DEV)
    ref_sd, state = O.clone_state_dict(sd), None
    curve = []
    for it in range(steps):
        b = batches[it % 3]
        n = b["bboxes"].shape[0]
        masks = [(torch.rand(n, T, generator=gen) > cfg["drop_prob"]) for _ in range(2)]
        db = {k: v.to(DEV) for k, v in b.items() if torch.is_tensor(v)}
        loss, _ = tr.train_step(db, masks=[m.to(torch.uint8).to(DEV) for m in masks])
        loss_ref, _, grads, after, _ = O.loss_and_grads(ref_sd, b["images"], b["bboxes"], b["additional_feats"],
                                                        b["context_indices"], b["labels"], cfg,
                                                        [m.float() for m in masks])
        new_p, state = O.adam_reference([ref_sd[k] for k in keys], [grads[k] for k in keys], state)
        for k, p in zip(keys, new_p):
            after[k] = p
        ref_sd = after
        curve.append((it, float(loss.item()), float(loss_ref)))
        if it == steps // 2 - 1:          # half-way snapshot of both sides for the decision check
            mid_sd = O.clone_state_dict(ref_sd)
            mid_tr = HotPathTrainer(cfg, tr.state_dict(), DEV)
    path = os.environ.get("COVA_WRITE_CURVE")
    if path:
        with open(path, "w") as f:
            f.write("# 80 Adam steps, HIP trainer vs CPU oracle (tests/test_model_gpu.py::test_training_trajectory_follows_the_cpu_oracle)\n")
            f.write("# step  loss_hip  loss_oracle  rel_diff\n")
            for it, a, r in curve:
                f.write("%3d %12.5f %12.5f %9.2e\n" % (it, a, r, abs(a - r) / max(abs(r), 1e-9)))
    first, last = curve[0], curve[-1]
    assert abs(first[1] - first[2]) <= LOSS_TOL * abs(first[2])                    # same start
    assert last[2] < 0.5 * first[2], "the oracle run did not learn: not a meaningful trajectory"
    # band: 5 % of the current loss + 0.2 % of the initial one.  The two runs are chaotic once the weights have separated
    # by Adam's +-lr steps on noise-level gradients (and the multi-threaded CPU side is not bit-reproducible itself):
    # measured 1e-7 at step 0, 1e-4 around step 15 at 1/4 of the initial loss, 1-5 % of the CURRENT loss from step ~45 on,
    # where the loss is 20-100x below its start (profiles/r03_trajectory.txt)
    for it, a, r in curve:
        assert abs(a - r) <= 0.05 * abs(r) + 2e-3 * first[2], "loss curves drifted apart at step %d: %.5f vs %.5f" % (it, a, r)
    # the first steps: Adam's first update is +-lr whatever the size of a gradient, so entries whose gradient is rounding
    # noise step either way and the runs separate from step 1 on -- measured 1e-6 at steps 1-2 and 3e-6 / 9e-5 at step 3
    # (conv1 on the f32-MFMA / bf16-split kernels: which noise-level signs differ is chance), <= 9e-4 up to step 11
    assert max(abs(a - r) / abs(r) for _, a, r in curve[:3]) < 2e-5
    assert max(abs(a - r) / abs(r) for _, a, r in curve[:5]) < 3e-4
    assert max(abs(a - r) / abs(r) for _, a, r in curve[:12]) < 2e-3       # the early ones tightly
    # decisions, eval mode, on every batch: half-way (step 40, where the two parameter sets are still within ~1 % in
    # logit space) on rows whose margin exceeds 10x the observed difference, and at the end on the rows that are still
    # decided by a clear margin
    def decisions(hip_tr, oracle_sd, min_frac):
        for b in batches:
            db = {k: v.to(DEV) for k, v in b.items() if torch.is_tensor(v)}
            logits, pred = hip_tr.predict(db)
            ref = O.forward(O.clone_state_dict(oracle_sd), b["images"], b["bboxes"], b["additional_feats"],
                            b["context_indices"], cfg, False)
            err = float((logits.cpu() - ref).abs().max())
            top2 = ref.topk(2, dim=1).values
            ok = (top2[:, 0] - top2[:, 1]) > 10 * err
            assert int(ok.sum()) >= min_frac * ok.numel(), ("too few margin rows for a decision check", int(ok.sum()), ok.numel(), err)
            assert torch.equal(pred.cpu()[ok], ref.argmax(1)[ok])
            # per page and class: the top-1 box (train.py:144-153) where the winner's margin exceeds the difference
            page = b["bboxes"][:, 0].long()
            for pg in page.unique():
                sel = page == pg
                lr, lh = ref[sel], logits.cpu()[sel]
                for c in range(1, 4):
                    col = lr[:, c].sort(descending=True).values
                    if col[0] - col[1] > 10 * err:
                        assert int(lh[:, c].argmax()) == int(lr[:, c].argmax())
        return err

    decisions(mid_tr, mid_sd, 0.5)
    decisions(tr, ref_sd, 0.0)
    hip_sd = tr.state_dict()
    assert all(torch.isfinite(v).all() for v in hip_sd.values())


def test_eval_decision_kernel_matches_reference_rule():
    """cova_page_class_topk against the reference's evaluate_model output (golden/evaluate.npz) and
    against torch.argsort on random logits with ragged pages, incl. k larger than a page."""
    from cova_web_object_detection_amd import _lib
    fx = np.load(GOLDEN + "/evaluate.npz")
    logits = torch.from_numpy(fx["logits"]).to(DEV)
    pages = fx["bboxes"][:, 0].astype(np.int64)
    counts = np.bincount(pages)
    start = torch.from_numpy(np.concatenate([[0], np.cumsum(counts)])).to(DEV)
    labels = torch.from_numpy(fx["labels"])
    for k in (1, 3):
        out = torch.empty((len(counts), 4, k), dtype=torch.int64, device=DEV)
        _lib.call("cova_page_class_topk", logits, start, len(counts), 4, k, out)
        out = out.cpu()
        acc = np.zeros((len(counts), 3), dtype=np.int32)
        for p in range(len(counts)):
            lab = labels[pages == p]
            o = logits.cpu()[pages == p]
            ref = torch.argsort(o, dim=0, descending=True)[:k]          # [k, 4], best first
            assert torch.equal(out[p].t(), ref)                           # no ties in random logits
            for c in (1, 2, 3):
                acc[p, c - 1] = int(int(torch.nonzero(lab == c)[0]) in out[p, c].tolist())
        assert np.array_equal(acc, fx["img_acc_k%d" % k][:, 1:])          # reference's own decisions
    # k larger than the page: padded with -1, prefix still sorted
    small = torch.randn(5, 4, device=DEV)
    st = torch.tensor([0, 2, 5], device=DEV)
    out = torch.empty((2, 4, 4), dtype=torch.int64, device=DEV)
    _lib.call("cova_page_class_topk", small, st, 2, 4, 4, out)
    assert (out[0, :, 2:] == -1).all() and (out[1, :, 3:] == -1).all() and (out[0, :, :2] >= 0).all()


def test_trainer_evaluate_matches_oracle_decision_rule():
    from cova_web_object_detection_amd.trainer import HotPathTrainer
    cfg = dict(roi_output_size=(3, 3), n_classes=4, use_context=True, hidden_dim=64,
               bbox_hidden_dim=16, n_additional_feat=0, drop_prob=0.2)
    wcfg = {k: v for k, v in cfg.items() if k != "drop_prob"}
    sd = weights.seeded_state_dict(41, logit_gain=3.0, **wcfg)
    counts = [25, 11, 40]
    batch = synthetic.make_batch(3, img_h=64, boxes_per_page=counts, context_size=12, seed=41)
    dbatch = {k: v.to(DEV) for k, v in batch.items() if torch.is_tensor(v)}
    tr = HotPathTrainer(cfg, sd, DEV)
    start = torch.tensor(np.concatenate([[0], np.cumsum(counts)]), device=DEV)
    for k in (1, 3):
        topk, correct = tr.evaluate(dbatch, start, k)
        logits, _ = tr.predict(dbatch)
        ref = np.asarray(O.eval_accuracy(logits.cpu(), batch["bboxes"], batch["labels"], 4, k))
        assert np.array_equal(correct.cpu().numpy().astype(np.int64), ref)
        assert topk.shape == (3, 4, k)
    # pages that break the "exactly one labelled box per class" habit of the dataset (README.md:17): a class
    # missing on a page scores False (the reference would raise there), and with two boxes of one class the FIRST
    # is the ground truth (train.py:146 takes `[0, 0]` of the matching rows)
    lab = batch["labels"].clone()
    p1 = slice(counts[0], counts[0] + counts[1])
    lab[p1][lab[p1] == 2] = 0                                  # page 1 loses its class-2 box
    p0 = lab[:counts[0]]
    first1 = int((p0 == 1).nonzero()[0])
    extra = (first1 + 5) % counts[0]
    if p0[extra] == 0:
        p0[extra] = 1                                          # a second class-1 box on page 0, after the first
    d2 = dict(dbatch, labels=lab.to(DEV))
    topk, correct = tr.evaluate(d2, start, 1)
    assert not bool(correct[1, 1])                             # (page 1, class 2)
    assert bool(correct[0, 0]) == bool(int(topk[0, 1, 0]) == first1)


@pytest.mark.parametrize("K", [2, 64, 256, 1024])
def test_gat_extreme_neighbour_counts(K):
    rs = np.random.RandomState(K)
    N, Fd, D = 70, 48, 24
    sd = {"gat.W_i.weight": torch.from_numpy(rs.uniform(-0.3, 0.3, (D, Fd)).astype(np.float32)),
          "gat.W_j.weight": torch.from_numpy(rs.uniform(-0.3, 0.3, (D, Fd)).astype(np.float32)),
          "gat.attention_layer.weight": torch.from_numpy(rs.uniform(-0.5, 0.5, (1, 2 * D)).astype(np.float32)),
          "gat.attention_layer.bias": torch.from_numpy(rs.uniform(-0.5, 0.5, (1,)).astype(np.float32))}
    h = torch.from_numpy(rs.standard_normal((N, Fd)).astype(np.float32))
    ctx = torch.from_numpy(synthetic.context_window_indices(N, K // 2))
    layer = GraphAttentionLayer(Fd, D)
    layer.load_state_dict({k[4:]: v for k, v in sd.items()})
    layer = layer.to(DEV)
    hg = h.to(DEV).requires_grad_(True)
    hp, attn = layer(hg, ctx.to(DEV), return_attn_wts=True)
    hr = h.clone().requires_grad_(True)
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    hp_ref, attn_ref = O.gat(hr, ctx, sdr, return_attn_wts=True)
    assert relerr(hp.detach().cpu(), hp_ref.detach()) < 1e-5 and relerr(attn.cpu(), attn_ref.detach()) < 1e-5
    g = torch.from_numpy(rs.standard_normal((N, D)).astype(np.float32))
    (hp * g.to(DEV)).sum().backward()
    (hp_ref * g).sum().backward()
    assert relerr(hg.grad.cpu(), hr.grad) < 1e-4
    for k, p in layer.named_parameters():
        ref = sdr["gat." + k].grad
        assert float((p.grad.cpu() - ref).abs().max()) <= 1e-4 * max(float(ref.abs().max()), 1e-3), k
    assert layer(hg, torch.zeros((N, 65), dtype=torch.long, device=DEV)).shape == (N, D)     # beyond one wavefront: fine
    with pytest.raises(ValueError):
        layer(hg, torch.zeros((N, engine.GAT_MAX_K + 1), dtype=torch.long, device=DEV))


@pytest.mark.parametrize("use_context,boxes", [(False, [7, 30]), (True, [3, 4]), (True, [230])])
def test_small_and_ragged_batches_match_oracle_forward(use_context, boxes):
    cfg = dict(roi_output_size=(3, 3), n_classes=4, use_context=use_context, hidden_dim=48,
               bbox_hidden_dim=16, n_additional_feat=0, drop_prob=0.0)
    wcfg = {k: v for k, v in cfg.items() if k != "drop_prob"}
    sd = weights.seeded_state_dict(len(boxes) * 13 + boxes[0], logit_gain=2.0, **wcfg)
    batch = synthetic.make_batch(len(boxes), img_h=64, boxes_per_page=boxes, context_size=12, seed=boxes[0])
    if not use_context:
        batch["context_indices"] = torch.empty((0, 0), dtype=torch.long)
    m = build(cfg, 64, sd)
    args = dev_batch(batch)
    for training in (False, True):
        m.train(training)
        with torch.no_grad():
            got = m(*args)
        ref = O.forward(O.clone_state_dict(sd), batch["images"], batch["bboxes"], batch["additional_feats"],
                        batch["context_indices"], cfg, training, None)
        assert relerr(got.cpu(), ref) < LOGIT_TOL, (training, relerr(got.cpu(), ref))
    empty = m(args[0], args[1][:0], args[2][:0], args[3][:0] if use_context else args[3])
    assert empty.shape == (0, 4)


def test_device_collate_matches_reference_fixture_bit_exact():
    """cova_images_u8_to_f32 + cova_collate_boxes against the reference's ToTensor / __getitem__ /
    custom_collate_fn output on the same raw inputs (tests/golden/collate_raw.npz)."""
    from cova_web_object_detection_amd.pipeline import DeviceCollate
    fx = np.load(GOLDEN + "/collate_raw.npz")
    rows = np.split(fx["rows"], np.cumsum(fx["counts"])[:-1])
    got = DeviceCollate(int(fx["context_size"]), DEV)(fx["u8_pages"], rows)
    for k in ("images", "bboxes", "labels", "context_indices"):
        assert np.array_equal(got[k].cpu().numpy(), fx[k]), k
    assert got["additional_feats"].shape == (fx["rows"].shape[0], 0)
    assert got["page_start"].cpu().tolist() == [0] + np.cumsum(fx["counts"]).tolist()


@pytest.mark.parametrize("B,H,W,counts,cs", [(1, 5, 7, [1], 3), (3, 64, 48, [2, 90, 17], 12),
                                             (2, 33, 35, [6, 0], 2), (2, 16, 16, [4, 9], 0)])
def test_device_collate_matches_oracle_edge_cases(B, H, W, counts, cs):
    """odd image sizes (scalar path), single-box and empty pages, context_size 0."""
    from cova_web_object_detection_amd.pipeline import DeviceCollate
    rs = np.random.RandomState(B * 100 + H)
    u8 = rs.randint(0, 256, (B, H, W, 3)).astype(np.uint8)
    rows = []
    for n in counts:
        r = np.concatenate([rs.uniform(0, 500, (n, 4)), rs.randint(0, 4, (n, 1))], 1).astype(np.float32)
        rows.append(r)
    got = DeviceCollate(cs, DEV)(u8, rows)
    ref = O.collate_reference(u8, rows, cs)
    for k in ("images", "bboxes", "labels"):
        assert np.array_equal(got[k].cpu().numpy(), ref[k].numpy()), k
    if cs > 0:
        assert np.array_equal(got["context_indices"].cpu().numpy(), ref["context_indices"].numpy())
    else:
        assert tuple(got["context_indices"].shape) == (0, 0)          # datasets.py:130


def test_attention_export_rows_match_reference_dump():
    """cova_attn_export_rows on the HIP eval forward against the reference's dump
    (tests/golden/attn_export.npz): geometry / label columns exact, attention weights 1e-5."""
    from cova_web_object_detection_amd.pipeline import attention_rows
    from cova_web_object_detection_amd.trainer import HotPathTrainer
    fx = np.load(GOLDEN + "/attn_export.npz")
    _, cfg, sd, batch = load_case(str(fx["source"]))
    tr = HotPathTrainer(cfg, sd, DEV)
    dbatch = {k: v.to(DEV) for k, v in batch.items() if torch.is_tensor(v)}
    rows = attention_rows(tr, dbatch).cpu().numpy()
    ref = fx["rows"]
    K = batch["context_indices"].shape[1]
    assert rows.shape == ref.shape
    assert np.array_equal(rows[:, :5 + 4 * K], ref[:, :5 + 4 * K])
    np.testing.assert_allclose(rows[:, 5 + 4 * K:], ref[:, 5 + 4 * K:], atol=1e-5)
    np.testing.assert_allclose(rows[:, 5 + 4 * K:].sum(1), 1.0, atol=1e-5)
    # and against the oracle on a bigger ragged batch
    _, cfg, sd, batch = load_case("cova_h128_ragged")
    tr = HotPathTrainer(cfg, sd, DEV)
    dbatch = {k: v.to(DEV) for k, v in batch.items() if torch.is_tensor(v)}
    rows = attention_rows(tr, dbatch).cpu()
    _, inter = O.forward(sd, batch["images"], batch["bboxes"], batch["additional_feats"],
                         batch["context_indices"], cfg, training=False, return_intermediates=True)
    ref = O.attention_rows(batch["bboxes"], batch["context_indices"], batch["labels"], inter["attn"])
    K = batch["context_indices"].shape[1]
    assert torch.equal(rows[:, :5 + 4 * K], ref[:, :5 + 4 * K])
    np.testing.assert_allclose(rows[:, 5 + 4 * K:].numpy(), ref[:, 5 + 4 * K:].numpy(), atol=2e-5)


def test_prefetcher_overlapped_upload_matches_direct_collate():
    """pipeline.Prefetcher (pinned staging, side-stream upload + collate kernels) yields exactly the
    batches of the synchronous DeviceCollate, in order, and a train step consumes them."""
    from cova_web_object_detection_amd.pipeline import DeviceCollate, Prefetcher
    from cova_web_object_detection_amd.trainer import HotPathTrainer
    rs = np.random.RandomState(5)
    items = []
    for counts in ([12, 30], [25, 9], [40, 18]):
        u8 = rs.randint(0, 256, (2, 64, 64, 3)).astype(np.uint8)
        rows = []
        for n in counts:
            xy, wh = rs.uniform(0, 30, (n, 2)), rs.uniform(8, 30, (n, 2))
            lab = np.zeros((n, 1)); lab[:3, 0] = [1, 2, 3]
            rows.append(np.concatenate([xy, wh, lab], 1).astype(np.float32))
        items.append((u8, rows))
    direct = [DeviceCollate(4, DEV)(*it) for it in items]
    cfg = dict(roi_output_size=(3, 3), n_classes=4, use_context=True, hidden_dim=32, bbox_hidden_dim=8,
               n_additional_feat=0, drop_prob=0.0)
    tr = HotPathTrainer(cfg, weights.seeded_state_dict(3, **{k: v for k, v in cfg.items() if k != "drop_prob"}), DEV)
    n = 0
    for got, ref in zip(Prefetcher(DeviceCollate(4, DEV, pin=True), items), direct):
        for k in ("images", "bboxes", "labels", "context_indices", "page_start"):
            assert torch.equal(got[k], ref[k]), k
        loss, _ = tr.train_step(got)
        assert torch.isfinite(loss).all()
        n += 1
    assert n == len(items)


def test_overlapped_gradient_exchange_path_on_rccl():
    """The two-phase gradient exchange (head all-reduced asynchronously under the conv-stack backward,
    conv stack at the end) through a real RCCL process group.  One GPU only allows a 1-rank group, so
    the sums are identities: the steps must track the single-process trainer (within Adam's +-lr
    noise), i.e. the asynchronous collective, its wait and Adam run in a consistent order."""
    import os
    import socket
    import torch.distributed as dist
    from cova_web_object_detection_amd.trainer import HotPathTrainer
    cfg = dict(roi_output_size=(3, 3), n_classes=4, use_context=True, hidden_dim=64, bbox_hidden_dim=16,
               n_additional_feat=0, drop_prob=0.2)
    wcfg = {k: v for k, v in cfg.items() if k != "drop_prob"}
    sd = weights.seeded_state_dict(17, **wcfg)
    batch = synthetic.make_batch(3, img_h=128, boxes_per_page=[20, 33, 9], context_size=6, seed=17)
    dbatch = {k: v.to(DEV) for k, v in batch.items() if torch.is_tensor(v)}
    ref = HotPathTrainer(cfg, sd, DEV, dropout_seed=5)
    for _ in range(3):
        loss_ref, _ = ref.train_step(dbatch)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        tr = HotPathTrainer(cfg, sd, DEV, world_size=2, dropout_seed=5)     # world_size > 1: exchange path on
        for _ in range(3):
            loss, _ = tr.train_step(dbatch)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    # the step contains no float atomics (RoIPool / GAT backward gather in a fixed order) and a 1-rank
    # all-reduce is the identity: three steps through the exchange path reproduce the plain trainer bit for bit
    assert float(loss) == float(loss_ref)
    a, b = tr.state_dict(), ref.state_dict()
    for k in a:
        assert torch.equal(a[k], b[k]), k

@pytest.mark.parametrize("kw", [dict(), dict(backbone="resnet50", n_heads=2, n_gat_layers=2)])
def test_train_steps_are_bit_reproducible(kw):
    """No float atomics in the step: two trainers fed the same batches end with bit-identical gradients,
    parameters, Adam moments and running statistics (RoIPool / GAT backward gather in a fixed order)."""
    from cova_web_object_detection_amd.trainer import HotPathTrainer
    cfg = dict(roi_output_size=(3, 3), n_classes=4, use_context=True, hidden_dim=64, bbox_hidden_dim=16,
               n_additional_feat=0, drop_prob=0.2, **kw)
    sd = weights.seeded_state_dict(19, **{k: v for k, v in cfg.items() if k != "drop_prob"})
    batches = [synthetic.make_batch(3, img_h=96, img_w=160, boxes_per_page=[70, 33, 90], context_size=12, seed=s)
               for s in (1, 2)]
    runs = []
    for rep in range(2):
        tr = HotPathTrainer(cfg, sd, DEV, dropout_seed=5)
        losses = []
        for step in range(4):
            b = {k: v.to(DEV) for k, v in batches[step % 2].items() if torch.is_tensor(v)}
            loss, _ = tr.train_step(b)
            losses.append(float(loss))
        runs.append((losses, tr.gbucket.flat.clone(), tr.pbucket.flat.clone(), tr.exp_avg_sq.clone(),
                     {k: v.clone() for k, v in tr.buffers.items()}))
    assert runs[0][0] == runs[1][0]
    for a, b in zip(runs[0][1:4], runs[1][1:4]):
        assert torch.equal(a, b)
    for k in runs[0][4]:
        assert torch.equal(runs[0][4][k], runs[1][4][k]), k


@pytest.mark.gpu
def test_malformed_input_raises_like_the_reference_forward():
    """The reference's forward (models.py:94-122) fails inside torch for these (observed by running it on
    CPU: ValueError from BatchNorm1d's train-mode batch-size check, RuntimeError from conv2d / cat / view,
    IndexError for a float index tensor, RuntimeError for a second backward); the HIP path takes raw
    pointers, so the same conditions are caught on the host with the same exception types."""
    cfg = dict(roi_output_size=(3, 3), n_classes=4, use_context=True, hidden_dim=48, bbox_hidden_dim=16,
               n_additional_feat=0, drop_prob=0.0)
    sd = weights.seeded_state_dict(3, logit_gain=2.0, **{k: v for k, v in cfg.items() if k != "drop_prob"})
    batch = synthetic.make_batch(2, img_h=64, boxes_per_page=12, context_size=2, seed=4)
    m = build(cfg, 64, sd)
    img, bb, af, ctx = dev_batch(batch)
    m.train()
    one = (img, bb[:1], af[:1], torch.full((1, 4), -1, dtype=torch.long, device=DEV))
    with pytest.raises(ValueError, match="more than 1 value per channel"):
        m(*one)
    with pytest.raises(ValueError, match="more than 1 value per channel"):
        m._get_bbox_features(bb[:1])
    m.eval()
    ref = O.forward(O.clone_state_dict(sd), batch["images"], batch["bboxes"][:1], batch["additional_feats"][:1],
                    one[3].cpu(), cfg, False, None)
    assert relerr(m(*one).detach().cpu(), ref) < LOGIT_TOL          # a single box is fine with running statistics
    m.train()
    for bad in ((img[:, :1], bb, af, ctx), (img, bb[:, :4], af, ctx), (img, bb, af, ctx[:5]),
                (img, bb, torch.rand(bb.shape[0], 3, device=DEV), ctx), (img[0], bb, af, ctx)):
        with pytest.raises(RuntimeError):
            m(*bad)
    with pytest.raises(IndexError):
        m(img, bb, af, ctx.float())
    with pytest.raises(RuntimeError):
        m.gat(torch.rand(24, 7, device=DEV), ctx)
    out = m(img, bb, af, ctx)
    out.sum().backward()
    with pytest.raises(RuntimeError, match="second time"):
        out.sum().backward()


@pytest.mark.gpu
def test_input_layout_and_index_dtype_do_not_change_the_result():
    """Strided / channels_last pages and int32 neighbour ids (both accepted by the reference's torch ops)
    give the bits of the contiguous int64 call; ids outside [0, N) act as pads (memory-safe; the reference
    raises IndexError for them on CPU, a device-side assert on a GPU); a page without boxes is legal."""
    cfg = dict(roi_output_size=(3, 3), n_classes=4, use_context=True, hidden_dim=48, bbox_hidden_dim=16,
               n_additional_feat=0, drop_prob=0.0)
    sd = weights.seeded_state_dict(5, logit_gain=2.0, **{k: v for k, v in cfg.items() if k != "drop_prob"})
    batch = synthetic.make_batch(2, img_h=64, boxes_per_page=12, context_size=2, seed=6)
    m = build(cfg, 64, sd).eval()
    img, bb, af, ctx = dev_batch(batch)
    with torch.no_grad():
        base = m(img, bb, af, ctx)
        assert torch.equal(m(img.to(memory_format=torch.channels_last), bb, af, ctx), base)
        wide = torch.rand(2, 3, 64, 128, device=DEV)
        wide[:, :, :, ::2] = img
        assert torch.equal(m(wide[:, :, :, ::2], bb, af, ctx), base)
        assert torch.equal(m(img, bb, af, ctx.int()), base)
        assert torch.equal(m(img, bb.double(), af, ctx), base)
        pad = ctx.clone()
        pad[3, 1] = -1
        far = ctx.clone()
        far[3, 1] = 10 ** 6
        assert torch.equal(m(img, bb, af, far), m(img, bb, af, pad))
        lone = batch["bboxes"].clone()
        lone[:, 0] = 1                                        # every box on page 1: page 0 has none
        got = m(img, lone.to(DEV), af, ctx)
    ref = O.forward(O.clone_state_dict(sd), batch["images"], lone, batch["additional_feats"],
                    batch["context_indices"], cfg, False, None)
    assert relerr(got.cpu(), ref) < LOGIT_TOL


@pytest.mark.gpu
@pytest.mark.parametrize("training", [True, False])
def test_gradient_with_respect_to_the_images(training):
    """`images.requires_grad` (the reference gets d loss / d images from autograd through nn.Conv2d, models.py:94-122): the
    module returns it too -- cova_pool_bwd_dy1 + cova_conv1_dgrad behind the conv stack's backward -- and the parameter
    gradients of that backward are the ones of a call without it.  Against the CPU oracle forced to the HIP forward's
    discrete decisions; also through _get_visual_features."""
    cfg = dict(roi_output_size=(3, 3), n_classes=4, use_context=True, hidden_dim=48, bbox_hidden_dim=16,
               n_additional_feat=0, drop_prob=0.0)
    sd = weights.seeded_state_dict(5, logit_gain=2.0, **{k: v for k, v in cfg.items() if k != "drop_prob"})
    batch = synthetic.make_batch(2, img_h=96, boxes_per_page=[14, 9], context_size=3, seed=6)
    img, bb, af, ctx = dev_batch(batch)
    m = build(cfg, 96, sd)
    m.train(training)
    x = img.clone().requires_grad_(True)
    logits = m(x, bb, af, ctx)
    routing = routing_from_saved(logits.grad_fn.sv)
    loss = torch.nn.CrossEntropyLoss(reduction="sum")(logits, batch["labels"].to(DEV))
    loss.backward()
    assert x.grad is not None and x.grad.shape == img.shape and torch.isfinite(x.grad).all()
    xi = batch["images"].clone().requires_grad_(True)
    _, _, grads_ref, _, _ = O.loss_and_grads(sd, xi, batch["bboxes"], batch["additional_feats"], batch["context_indices"],
                                             batch["labels"], cfg, None, routing, training=training)
    scale = float(xi.grad.abs().max())
    err = float((x.grad.cpu() - xi.grad).abs().max()) / scale
    print("d loss / d images (%s): max error %.2e of the scale %.3e" % ("train" if training else "eval", err, scale))
    assert err < GRAD_TOL, err
    compare_grads({k: p.grad for k, p in m.named_parameters()}, grads_ref, rtol=GRAD_TOL, outlier_frac=0.0)
    # the same parameter gradients without the image gradient
    m2 = build(cfg, 96, sd)
    m2.train(training)
    torch.nn.CrossEntropyLoss(reduction="sum")(m2(img, bb, af, ctx), batch["labels"].to(DEV)).backward()
    for (k, p), (_, p2) in zip(m.named_parameters(), m2.named_parameters()):
        # two instances of the module built from one state_dict: bit-identical gradients, wherever their parameters landed
        # in memory (engine._adjacent: the one-GEMM form of [W_i; W_j] only for views of ONE storage)
        assert torch.equal(p.grad, p2.grad), k
    # the piecewise surface (extract_attn_wts_and_visualize.py:117 calls it under no_grad; autograd works through it too)
    x3 = img.clone().requires_grad_(True)
    vis = m2._get_visual_features(x3, bb)
    vis.square().sum().backward()
    assert x3.grad is not None and torch.isfinite(x3.grad).all() and float(x3.grad.abs().max()) > 0


@pytest.mark.gpu
def test_no_kernel_reads_memory_nobody_wrote_and_steps_do_not_depend_on_the_allocation_pattern(monkeypatch):
    """Every buffer the engine allocates is NaN-filled at allocation: a kernel that reads an element nobody wrote shows up as a
    NaN gradient.  Then the same step with its parameters in freshly allocated tensors, twice, with differently sized
    allocations in between: bit-identical gradients (tools/poison_check.py runs this over seven model configurations)."""
    real_empty, real_empty_like = torch.empty, torch.empty_like

    def poisoned(t):
        if t.is_floating_point():
            t.fill_(float("nan"))
        elif t.dtype != torch.bool:
            t.fill_(0x7F if t.dtype == torch.uint8 else 0x3FFFFFFF)
        return t

    proxy = type(os)("torch_proxy")
    proxy.__dict__.update(torch.__dict__)
    proxy.empty = lambda *a, **k: poisoned(real_empty(*a, **k))
    proxy.empty_like = lambda *a, **k: poisoned(real_empty_like(*a, **k))
    monkeypatch.setattr(engine, "torch", proxy)
    cfg = dict(roi_output_size=(3, 3), n_classes=4, use_context=True, hidden_dim=48, bbox_hidden_dim=16,
               n_additional_feat=0, drop_prob=0.2)
    sd = weights.seeded_state_dict(5, logit_gain=2.0, **{k: v for k, v in cfg.items() if k != "drop_prob"})
    batch = synthetic.make_batch(2, img_h=96, boxes_per_page=[14, 9], context_size=3, seed=6)
    keys = O.param_keys(sd)
    args = dev_batch(batch)
    labels = batch["labels"].to(DEV)
    runs = []
    for rep in range(3):
        junk = [torch.full((1 + 37 * rep, 129), float("nan"), device=DEV) for _ in range(3 * rep)]     # shift the allocation pattern
        params = {k: sd[k].to(DEV) for k in keys}
        buffers = {k: v.to(DEV) for k, v in sd.items() if k not in params}
        logits, sv = engine.model_fwd(cfg, params, buffers, *args, True, (11, 12))
        loss, dl, _ = engine.ce_sum(logits, labels)
        grads = engine.model_bwd(sv, dl, params)
        torch.cuda.synchronize()
        del junk
        assert torch.isfinite(loss).all() and all(torch.isfinite(g).all() for g in grads.values()), \
            [k for k, g in grads.items() if not torch.isfinite(g).all()]
        runs.append({k: g.clone() for k, g in grads.items()})
    for k in runs[0]:
        assert torch.equal(runs[0][k], runs[1][k]) and torch.equal(runs[0][k], runs[2][k]), k
