"""Per-tensor gradient error of one model-level case (HIP vs the forced-routing oracle): python tools/case_probe.py cs n0,n1 [hidden]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch, torch.nn.functional as F
import cova_amd  # noqa
from cova_web_object_detection_amd import synthetic, weights
from cova_web_object_detection_amd.models import CoVA
from helpers import routing_from_saved
from oracle import cova_oracle as O
cs = int(sys.argv[1]); boxes = [int(x) for x in sys.argv[2].split(",")]; hidden = int(sys.argv[3]) if len(sys.argv) > 3 else 96
seed = int(sys.argv[4]) if len(sys.argv) > 4 else 53
cfg = dict(roi_output_size=(3, 3), n_classes=4, use_context=True, hidden_dim=hidden, bbox_hidden_dim=32, n_additional_feat=0, drop_prob=0.0)
sd = weights.seeded_state_dict(seed, logit_gain=4.0, **{k: v for k, v in cfg.items() if k != "drop_prob"})
b = synthetic.make_batch(len(boxes), img_h=64, img_w=64, boxes_per_page=boxes, context_size=cs, seed=seed)
m = CoVA((3, 3), 64, 4, True, hidden, 32, 0, 0.0, None); m.load_state_dict(sd); m = m.cuda().train()
args = [b[k].cuda() for k in ("images", "bboxes", "additional_feats", "context_indices")]
logits = m(*args); routing = routing_from_saved(logits.grad_fn.sv)
F.cross_entropy(logits, b["labels"].cuda(), reduction="sum").backward()
_, lref, gref, _, _ = O.loss_and_grads(sd, b["images"], b["bboxes"], b["additional_feats"], b["context_indices"], b["labels"], cfg, None, routing)
gs = max(float(g.abs().max()) for g in gref.values())
print("logit err %.2e" % float((logits.detach().cpu() - lref).abs().max() / lref.abs().max()))
for k, p in m.named_parameters():
    g = gref[k]; sc = max(float(g.abs().max()), 0.01 * gs)
    d = (p.grad.cpu().view_as(g) - g).abs() / sc
    print("%-34s max|g| %.2e (%.1e of largest)  err/scale %.2e  >1e-4: %d / %d" % (k, float(g.abs().max()), float(g.abs().max()) / gs, float(d.max()), int((d > 1e-4).sum()), d.numel()))
