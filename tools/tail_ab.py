"""Gradient difference of one train step between the in-launch finalize paths (OPTIONS.bn_tail) on and off, per key."""
import os as _os; _os.environ.setdefault("COVA_ALLOW_OPTION_CHANGES", "1")
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "tests")))
import cova_amd  # noqa
import torch
from cova_web_object_detection_amd import engine, synthetic, weights
from cova_web_object_detection_amd.trainer import HotPathTrainer

CFG = dict(roi_output_size=(3, 3), n_classes=4, use_context=True, hidden_dim=48, bbox_hidden_dim=16,
           n_additional_feat=0, drop_prob=0.0)
WCFG = {k: v for k, v in CFG.items() if k != "drop_prob"}
batch = synthetic.make_batch(4, img_h=128, boxes_per_page=[21, 34, 9, 40], context_size=5, seed=23)
full = {k: v.to("cuda:0") for k, v in batch.items() if torch.is_tensor(v)}
res = {}
for on in (True, False):
    engine.OPTIONS.bn_tail = on
    tr = HotPathTrainer(CFG, weights.seeded_state_dict(23, **WCFG), "cuda:0")
    loss, _ = tr.forward_backward(full)
    res[on] = (float(loss), tr.gbucket.flat.clone(), tr.gbucket.offsets)
print("loss", res[True][0], res[False][0])
ga, gb = res[True][1], res[False][1]
scale = float(gb.abs().max())
for k, (o, m, shape) in res[True][2].items():
    d = (ga[o:o + m] - gb[o:o + m]).abs().max().item()
    print("%-40s max|d|/scale %.3e   max|g| %.3e" % (k, d / scale, gb[o:o + m].abs().max().item()))

# ---- forward intermediates of the decoder's BatchNorm1d under both paths
from cova_web_object_detection_amd import models
svs = {}
for on in (True, False):
    engine.OPTIONS.bn_tail = on
    tr = HotPathTrainer(CFG, weights.seeded_state_dict(23, **WCFG), "cuda:0")
    logits, sv = engine.model_fwd(tr.cfg if hasattr(tr, "cfg") else CFG, tr.params, tr.buffers, full["images"], full["bboxes"],
                                  full["additional_feats"], full["context_indices"], True)
    svs[on] = sv
a, b = svs[True]["dec"], svs[False]["dec"]
print("decoder z equal:", torch.equal(a["z"], b["z"]), float((a["z"] - b["z"]).abs().max()))
print("comb equal:", torch.equal(svs[True]["comb"], svs[False]["comb"]), float((svs[True]["comb"] - svs[False]["comb"]).abs().max()))
for nm in ("scale", "shift", "mean", "invstd"):
    x, y = getattr(a["st"], nm), getattr(b["st"], nm)
    print(nm, float(((x - y).abs() / y.abs().clamp_min(1e-6)).max()))
ga_, gb_ = a["y"] > 0, b["y"] > 0
diff = (ga_ != gb_)
print("gate flips:", int(diff.sum()), "of", diff.numel())
if int(diff.sum()):
    idx = diff.nonzero()
    for r, c in idx[:10].tolist():
        print("  row %d col %d  y_on %.3e y_off %.3e  z %.6e mean %.6e invstd %.3e" % (r, c, float(a["y"][r, c]), float(b["y"][r, c]), float(a["z"][r, c]), float(a["st"].mean[c]), float(a["st"].invstd[c])))
print("y maxdiff", float((a["y"] - b["y"]).abs().max()))
