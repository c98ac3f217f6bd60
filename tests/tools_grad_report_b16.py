"""Parity report at the benchmark's FULL batch: configs[1] (16 pages of 1280 x 1280, 90 boxes, K = 24) -- the HIP train step
(forward + CE-sum + backward, dropout off) against the CPU oracle (oracle/cova_oracle.py, pinned to the reference), with
the bf16-split conv1 kernels on and off and the split / f32 F(4x4) main loops: logits, loss, per-parameter gradient error
against the oracle forced to the HIP forward's discrete decisions, and the decisions that differ from the UNFORCED oracle's
(count, place, distance of the pre-activation from zero).  Reference: train.py:47-60.
  python tests/tools_grad_report_b16.py [pages=16] > profiles/r05_grad_parity_1280_b16.txt      (about 4 minutes of host time)"""
import os as _os; _os.environ.setdefault("COVA_ALLOW_OPTION_CHANGES", "1")
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import cova_amd  # noqa
from cova_web_object_detection_amd import _lib, engine, synthetic, weights
from helpers import routing_from_saved
from oracle import cova_oracle as O

pages = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = "cuda:0"
cfg = dict(roi_output_size=(3, 3), n_classes=4, use_context=True, hidden_dim=384, bbox_hidden_dim=32, n_additional_feat=0,
           drop_prob=0.0)
wcfg = {k: v for k, v in cfg.items() if k != "drop_prob"}
sd = weights.seeded_state_dict(123, logit_gain=4.0, **wcfg)
b = synthetic.make_boxes_only(pages, 1280, 1280, 90, 12, 123)
images = torch.rand((pages, 3, 1280, 1280), generator=torch.Generator().manual_seed(123))
keys = O.param_keys(sd)
torch.set_num_threads(os.cpu_count() or 8)
print("# configs[1] at %d pages: 1280 x 1280, %d boxes, K = 24, seeded weights (logit gain 4), dropout off" % (pages, b["bboxes"].shape[0]))
t0 = time.time()
tap = {}
loss_u, logits_u, grads_u, _, _ = O.loss_and_grads(sd, images, b["bboxes"], b["additional_feats"], b["context_indices"], b["labels"],
                                                   cfg, None, {"_tap": tap})
print("# unforced oracle: loss %.6f (%.0f s on %d threads)" % (float(loss_u), time.time() - t0, torch.get_num_threads()))
args = [images.to(dev), b["bboxes"].to(dev), b["additional_feats"].to(dev), b["context_indices"].to(dev)]
for conv1_f32, w4_f32 in ((0, 0), (1, 0), (0, 1)):
    _lib.query("cova_set_option", 7, conv1_f32)
    _lib.query("cova_set_option", 9, w4_f32)
    params = {k: sd[k].to(dev) for k in keys}
    buffers = {k: v.to(dev) for k, v in sd.items() if k not in params}
    logits, sv = engine.model_fwd(cfg, params, buffers, *args, True)
    loss, dl, pred = engine.ce_sum(logits, b["labels"].to(dev))
    grads = engine.model_bwd(sv, dl, params)
    torch.cuda.synchronize()
    routing = routing_from_saved(sv)
    t0 = time.time()
    loss_f, logits_f, grads_f, _, _ = O.loss_and_grads(sd, images, b["bboxes"], b["additional_feats"], b["context_indices"],
                                                       b["labels"], cfg, None, routing)
    print("\n== conv1 %s, 3x3 forward / data gradient %s   (forced oracle: %.0f s)"
          % ("f32 MFMA" if conv1_f32 else "bf16 split", "f32 MFMA loop" if w4_f32 else "bf16 split loop", time.time() - t0))
    ls = float(logits_f.abs().max())
    print("logits: max err / scale  vs forced oracle %.2e   vs unforced oracle %.2e;   loss %.6f  forced %.6f (rel %.1e)  unforced %.6f (rel %.1e)"
          % (float((logits.cpu() - logits_f).abs().max()) / ls, float((logits.cpu() - logits_u).abs().max()) / ls, float(loss),
             float(loss_f), abs(float(loss) - float(loss_f)) / abs(float(loss_f)), float(loss_u), abs(float(loss) - float(loss_u)) / abs(float(loss_u))))
    print("argmax equal to the unforced oracle's on %d of %d boxes" % (int((logits.argmax(1).cpu() == logits_u.argmax(1)).sum()), logits.shape[0]))
    tot = nf = 0
    for key, pre in tap.items():
        if key == "pool_in" or key not in routing:
            continue
        gate = routing[key].to(torch.bool).reshape(pre.shape)
        diff = gate != (pre > 0)
        n = int(diff.sum())
        tot += diff.numel(); nf += n
        if n:
            print("  decisions that differ from the unforced oracle's: %-16s %3d of %10d, farthest pre-activation %.2e (scale %.2e)"
                  % (key, n, diff.numel(), float(pre[diff].abs().max()), float(pre.abs().max())))
    if "pool_in" in tap and "pool_idx" in routing:
        ref = torch.nn.functional.max_pool2d(tap["pool_in"], 3, 2, 1, return_indices=True)[1]
        n = int((routing["pool_idx"].reshape(ref.shape) != ref).sum())
        print("  max-pool arg-max positions that differ: %d of %d" % (n, ref.numel()))
    print("  gate decisions that differ in total: %d of %d" % (nf, tot))
    gscale = max(float(g.abs().max()) for g in grads_f.values())
    print("%-34s %10s %12s %12s" % ("parameter", "max|g|", "err/forced", "err/unforced"))
    worst = (0.0, "")
    for k in keys:
        g = grads[k].cpu().view_as(grads_f[k])
        sc = max(float(grads_f[k].abs().max()), 0.01 * gscale)
        ef, eu = float((g - grads_f[k]).abs().max()) / sc, float((g - grads_u[k]).abs().max()) / sc
        worst = max(worst, (ef, k))
        print("%-34s %10.3e %12.2e %12.2e" % (k, float(grads_f[k].abs().max()), ef, eu))
    print("worst gradient error against the forced oracle: %.2e (%s); gate of the tests: 1e-4" % worst)
_lib.query("cova_set_option", 7, 0)
_lib.query("cova_set_option", 9, 0)
