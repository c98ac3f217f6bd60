"""Per-parameter gradient error report of the HIP path vs the CPU oracle (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cova_amd  # noqa
from cova_web_object_detection_amd import engine, synthetic, weights
from oracle import cova_oracle as O

img_h = int(sys.argv[1]) if len(sys.argv) > 1 else 128
boxes = [int(b) for b in os.environ.get("BOXES", "40,23").split(",")]
seed = int(os.environ.get("SEED", 123))
dev = "cuda:0"
cfg = dict(roi_output_size=(3, 3), n_classes=4, use_context=True, hidden_dim=384,
           bbox_hidden_dim=32, n_additional_feat=0, drop_prob=0.0)
if os.environ.get("ARCH") == "r50":               # the extension model of BASELINE.json configs[2]
    cfg.update(backbone="resnet50", n_heads=2)
wcfg = {k: v for k, v in cfg.items() if k != "drop_prob"}
sd = weights.seeded_state_dict(seed, logit_gain=4.0, **wcfg)
batch = synthetic.make_batch(len(boxes), img_h=img_h, boxes_per_page=boxes, context_size=12, seed=seed)
keys = O.param_keys(sd)
loss_ref, logits_ref, grads_ref, _, _ = O.loss_and_grads(
    sd, batch["images"], batch["bboxes"], batch["additional_feats"], batch["context_indices"],
    batch["labels"], cfg, None)
# fp64 oracle for an error bar on the fp32 oracle itself
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
_, _, grads64, _, _ = O.loss_and_grads(
    sd64, batch["images"].double(), batch["bboxes"].double(), batch["additional_feats"].double(),
    batch["context_indices"], batch["labels"], cfg, None) if os.environ.get("F64") else (0, 0, None, 0, 0)
for trial in range(2):
    params = {k: sd[k].to(dev) for k in keys}
    buffers = {k: v.to(dev) for k, v in sd.items() if k not in params}
    args = [batch[k].to(dev) for k in ("images", "bboxes", "additional_feats", "context_indices")]
    logits, sv = engine.model_fwd(cfg, params, buffers, *args, True)
    loss, dl, pred = engine.ce_sum(logits, batch["labels"].to(dev))
    if trial == 0 and os.environ.get("FORCED"):
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
        from helpers import routing_from_saved
        routing = routing_from_saved(sv)
        loss_ref, logits_ref, grads_ref, _, _ = O.loss_and_grads(
            sd, batch["images"], batch["bboxes"], batch["additional_feats"], batch["context_indices"],
            batch["labels"], cfg, None, routing)
    grads = engine.model_bwd(sv, dl, params)
    gscale = max(float(g.abs().max()) for g in grads_ref.values())
    print("trial", trial, "loss", loss.item(), float(loss_ref), "gscale", gscale)
    for k, g in grads_ref.items():
        d = (grads[k].cpu().view_as(g) - g).abs().max().item()
        line = "%-34s max|g| %.3e  abs err %.3e  rel %.2e" % (k, g.abs().max().item(), d, d / max(g.abs().max().item(), 1e-30))
        if grads64 is not None:
            line += "  oracle32-vs-64 %.3e" % (grads64[k].float() - g).abs().max().item()
        print(line)
