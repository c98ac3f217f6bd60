"""Uninitialised-read hunt: the caching allocator's free blocks are filled with NaN (or a large finite value) before every
train step of a small model; a kernel that reads memory nobody wrote then shows up as NaN / a changed gradient.
  python tools/poison_check.py [nan|big] [img_h=96]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import cova_amd  # noqa: F401
from cova_web_object_detection_amd import engine, synthetic, weights
from oracle import cova_oracle as O
mode = sys.argv[1] if len(sys.argv) > 1 else "nan"
img_h = int(sys.argv[2]) if len(sys.argv) > 2 else 96
dev = "cuda:0"
BASE = dict(roi_output_size=(3, 3), n_classes=4, use_context=True, hidden_dim=48, bbox_hidden_dim=16, n_additional_feat=0, drop_prob=0.0)
CASES = [("resnet18, 1 head (the reference's model)", {}, True),
         ("resnet18, eval-mode backward (frozen BatchNorm)", {}, False),
         ("dropout 0.2", dict(drop_prob=0.2), True),
         ("CoVA++ (3 additional features), no positional encoder", dict(n_additional_feat=3, bbox_hidden_dim=0), True),
         ("no context", dict(use_context=False), True),
         ("resnet50 stem, 2 heads x 2 layers", dict(backbone="resnet50", n_heads=2, n_gat_layers=2), True),
         ("RoIAlign", dict(roi_op="align", sampling_ratio=2, roi_aligned=False), True)]
case = int(os.environ.get("CASE", "0"))
cfg = dict(BASE, **CASES[case][1])
TRAINING = CASES[case][2]
print("== case %d: %s" % (case, CASES[case][0]))
wkeys = [k for k in cfg if k not in ("drop_prob", "roi_op", "sampling_ratio", "roi_aligned")]
sd = weights.seeded_state_dict(5, logit_gain=2.0, **{k: cfg[k] for k in wkeys})
batch = synthetic.make_batch(2, img_h=img_h, boxes_per_page=[14, 9], context_size=3, seed=6, n_additional_feat=cfg["n_additional_feat"])
keys = O.param_keys(sd)
args = [batch[k].to(dev) for k in ("images", "bboxes", "additional_feats", "context_indices")]
labels = batch["labels"].to(dev)


def poison(value):
    # grab (nearly) everything the allocator would hand out for this step, fill it, give it back
    torch.cuda.synchronize()
    blocks = []
    for mb in (512, 256, 128, 64, 32, 16, 8, 4, 2, 1, 1, 1, 1):
        for _ in range(6):
            blocks.append(torch.full((mb * 262144,), value, device=dev))
    for kb in (512, 256, 128, 64, 32, 16, 8, 4, 2, 1):
        for _ in range(40):
            blocks.append(torch.full((kb * 256,), value, device=dev))
    torch.cuda.synchronize()
    del blocks


def step(want_dimg=False):
    params = {k: sd[k].to(dev) for k in keys}
    buffers = {k: v.to(dev) for k, v in sd.items() if k not in params}
    logits, sv = engine.model_fwd(cfg, params, buffers, *args, TRAINING, (11, 12))
    loss, dl, pred = engine.ce_sum(logits, labels)
    grads = engine.model_bwd(sv, dl, params, want_dimg=want_dimg)
    torch.cuda.synchronize()
    return float(loss), {k: v.clone() for k, v in grads.items()}


# every buffer the engine allocates starts as NaN (floats) / a large value (integers): engine._empty and torch.empty_like
_real_empty, _real_empty_like = torch.empty, torch.empty_like


def _poisoned(t):
    if t.is_floating_point():
        t.fill_(float("nan"))
    elif t.dtype != torch.bool:
        t.fill_(0x7F if t.dtype == torch.uint8 else 0x3FFFFFFF)
    return t


if os.environ.get("POISON_ALLOC", "1") == "1":
    engine.torch = type(sys)("torch_proxy")
    engine.torch.__dict__.update(torch.__dict__)
    engine.torch.empty = lambda *a, **k: _poisoned(_real_empty(*a, **k))
    engine.torch.empty_like = lambda *a, **k: _poisoned(_real_empty_like(*a, **k))
if os.environ.get("POISON_LDS") == "1":
    # ... and every CU's LDS is NaN-filled in front of EVERY product launch (LDS keeps its predecessor's data otherwise)
    import probe_lib
    from cova_web_object_detection_amd import _lib
    probe_lib.load()
    _real_launch = _lib._launch
    n_cu = torch.cuda.get_device_properties(0).multi_processor_count

    def _launch_poisoned(L, name, a, cargs, stream):
        if not name.startswith("cova_probe"):
            _real_launch(L, "cova_probe_lds_fill", (0x7FC00000, 2 * n_cu), [0x7FC00000, 2 * n_cu], stream)
        return _real_launch(L, name, a, cargs, stream)
    _lib._launch = _launch_poisoned
    print("(LDS of every CU NaN-filled in front of every launch)")
ref_loss, ref = step()
print("first step with every engine buffer NaN-filled at allocation: loss %.6f, NaN gradients: %s" % (
    ref_loss, [k for k, v in ref.items() if torch.isnan(v).any()] or "none"))
for rep, value in enumerate([float("nan") if mode == "nan" else 1e30, 0.0, float("nan") if mode == "nan" else -1e30]):
    poison(value)
    loss, g = step(want_dimg=rep == 2)
    bad = [(k, float((g[k] - ref[k]).abs().max()), bool(torch.isnan(g[k]).any())) for k in ref if not torch.equal(g[k], ref[k])]
    print("free memory filled with %r: loss %.6f (clean %.6f); gradients that differ from the clean step: %s" % (value, loss, ref_loss, [b[0] for b in bad] or "none"))
