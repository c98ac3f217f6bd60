"""Parity report of the ResNet-50 extension at full page size: BASELINE.json configs[2]'s architecture (ResNet-50 stem + three
Bottlenecks, 2-head GAT) on `pages` pages of 1280 x 1280 with 90 boxes each, K = 24 -- the HIP train step (forward + CE-sum +
backward, dropout off; the 1x1 products on the bf16 matrix pipe, csrc/conv1x1.hip / conv1x1_lin.hip) against the build's CPU
self-oracle (oracle/cova_oracle.py) forced to the HIP forward's discrete decisions: logits, loss, per-parameter gradient error.
The small-size parity tests (tests/test_extension_gpu.py) cannot show a coherent rounding direction that only adds up over
millions of pixel rows; this can (the 16-page report of configs[1] did for the 3x3 kernels).
  python tests/tools_grad_report_r50.py [pages=4] > profiles/r05_grad_parity_1280_r50.txt      (a few minutes of host time)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import cova_amd  # noqa
from cova_web_object_detection_amd import engine, synthetic, weights
from helpers import routing_from_saved
from oracle import cova_oracle as O

pages = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = "cuda:0"
kw = dict(backbone="resnet50", n_heads=2)
cfg = dict(roi_output_size=(3, 3), n_classes=4, use_context=True, hidden_dim=384, bbox_hidden_dim=32, n_additional_feat=0,
           drop_prob=0.0, **kw)
wcfg = {k: v for k, v in cfg.items() if k != "drop_prob"}
sd = weights.seeded_state_dict(321, logit_gain=4.0, **wcfg)
b = synthetic.make_boxes_only(pages, 1280, 1280, 90, 12, 321)
images = torch.rand((pages, 3, 1280, 1280), generator=torch.Generator().manual_seed(321))
keys = O.param_keys(sd)
torch.set_num_threads(os.cpu_count() or 8)
print("# configs[2] architecture (ResNet-50 stem + 3 Bottlenecks, 2-head GAT) at %d pages: 1280 x 1280, %d boxes, K = 24, seeded "
      "weights (logit gain 4), dropout off" % (pages, b["bboxes"].shape[0]))
args = [images.to(dev), b["bboxes"].to(dev), b["additional_feats"].to(dev), b["context_indices"].to(dev)]
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
print("# forced self-oracle: %.0f s on %d threads" % (time.time() - t0, torch.get_num_threads()))
ls = float(logits_f.abs().max())
print("logits: max err / scale %.2e;   loss %.6f  oracle %.6f (rel %.1e);   argmax equal on %d of %d boxes"
      % (float((logits.cpu() - logits_f).abs().max()) / ls, float(loss), float(loss_f),
         abs(float(loss) - float(loss_f)) / abs(float(loss_f)),
         int((logits.argmax(1).cpu() == logits_f.argmax(1)).sum()), logits.shape[0]))
gscale = max(float(g.abs().max()) for g in grads_f.values())
print("%-40s %10s %12s" % ("parameter", "max|g|", "err/forced"))
worst = (0.0, "")
for k in keys:
    g = grads[k].cpu().view_as(grads_f[k])
    sc = max(float(grads_f[k].abs().max()), 0.01 * gscale)
    ef = float((g - grads_f[k]).abs().max()) / sc
    worst = max(worst, (ef, k))
    print("%-40s %10.3e %12.2e" % (k, float(grads_f[k].abs().max()), ef))
print("worst gradient error against the forced oracle: %.2e (%s); gate of the tests: 1e-4" % worst)
