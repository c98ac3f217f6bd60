"""Parity margins of the two 3x3 kernel families, F(2x2,3x3) and F(4x4,3x3), on the four reference fixtures
(tests/golden/*.npz, captured from the imported reference) and on a full-resolution page: forward logits and loss against
the reference's, every gradient against the CPU oracle whose backward is forced to the HIP forward's discrete decisions,
the number of discrete decisions (ReLU gates, max-pool arg-maxes) that differ from the UNFORCED oracle, and the
distance of each flipped gate from zero.  Output: the table committed as profiles/r03_wino4_margin.txt.
usage: python tools/wino4_margin.py [out.txt]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import cova_amd  # noqa: F401
from cova_web_object_detection_amd import _lib, engine, synthetic, weights
from cova_web_object_detection_amd.models import CoVA
from helpers import FULL_CASES, load_case, routing_from_saved
from oracle import cova_oracle as O

DEV = "cuda:0"
lines = []


def emit(s=""):
    print(s)
    lines.append(s)


def relerr(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-12))


def run(cfg, img_h, sd, batch, ref_logits=None, ref_loss=None, img_w=None):
    out = {}
    for name, w4, f32loop in (("F(2x2,3x3)", False, 0), ("F(4x4) f32 loop", True, 1), ("F(4x4) bf16x3", True, 0)):
        engine.OPTIONS.wino4 = w4
        _lib.query("cova_set_option", 9, f32loop)
        m = CoVA(cfg["roi_output_size"], img_h, cfg["n_classes"], cfg["use_context"], cfg["hidden_dim"],
                 cfg["bbox_hidden_dim"], cfg["n_additional_feat"], cfg["drop_prob"], None)
        m.load_state_dict(sd, strict=True)
        m = m.to(DEV).train()
        args = [batch[k].to(DEV) for k in ("images", "bboxes", "additional_feats", "context_indices")]
        logits = m(*args)
        routing = routing_from_saved(logits.grad_fn.sv)
        loss = torch.nn.functional.cross_entropy(logits, batch["labels"].to(DEV), reduction="sum")
        loss.backward()
        grads = {k: p.grad.detach().cpu() for k, p in m.named_parameters()}
        loss_ref, logits_ref, grads_ref, _, _ = O.loss_and_grads(sd, batch["images"], batch["bboxes"], batch["additional_feats"],
                                                                 batch["context_indices"], batch["labels"], cfg, None, routing)
        tap = {}
        O.loss_and_grads(sd, batch["images"], batch["bboxes"], batch["additional_feats"], batch["context_indices"],
                         batch["labels"], cfg, None, {"_tap": tap})
        flips, total, worst_gate = 0, 0, 0.0
        for key, pre in tap.items():
            if key == "pool_in" or key not in routing:
                continue
            gate = routing[key].to(torch.bool).reshape(pre.shape)
            diff = gate != (pre > 0)
            flips += int(diff.sum())
            total += diff.numel()
            if int(diff.sum()):
                worst_gate = max(worst_gate, float(pre[diff].abs().max()) / float(pre.abs().max()))
        pool_flips = 0
        if "pool_in" in tap and "pool_idx" in routing:
            ref_idx = torch.nn.functional.max_pool2d(tap["pool_in"], 3, 2, 1, return_indices=True)[1]
            pool_flips = int((routing["pool_idx"].reshape(ref_idx.shape) != ref_idx).sum())
        gscale = max(float(g.abs().max()) for g in grads_ref.values())
        gerr, gkey = 0.0, ""
        for k, g in grads_ref.items():
            e = float((grads[k].view_as(g) - g).abs().max()) / max(float(g.abs().max()), 0.01 * gscale)
            if e > gerr:
                gerr, gkey = e, k
        out[name] = dict(
            logits_vs_reference=relerr(logits.detach().cpu(), ref_logits) if ref_logits is not None else None,
            loss_vs_reference=abs(loss.item() - ref_loss) / abs(ref_loss) if ref_loss is not None else None,
            logits_vs_oracle=relerr(logits.detach().cpu(), logits_ref), loss_vs_oracle=abs(loss.item() - float(loss_ref)) / abs(float(loss_ref)),
            worst_grad=gerr, worst_grad_key=gkey, gate_flips=flips, gates=total, worst_flipped_gate=worst_gate,
            pool_flips=pool_flips)
    return out


def show(title, res):
    emit(title)
    emit("  %-16s %12s %12s %12s %12s %12s  %-28s %10s %12s %10s" % (
        "kernels", "logit/ref", "loss/ref", "logit/oracle", "loss/oracle", "worst grad", "(parameter)", "gate flips",
        "worst |pre|", "pool flips"))
    for name, r in res.items():
        f = lambda v: "%12.2e" % v if v is not None else "%12s" % "-"
        emit("  %-16s %s %s %s %s %s  %-28s %4d/%-7d %12.2e %10d" % (
            name, f(r["logits_vs_reference"]), f(r["loss_vs_reference"]), f(r["logits_vs_oracle"]), f(r["loss_vs_oracle"]),
            f(r["worst_grad"]), r["worst_grad_key"][:28], r["gate_flips"], r["gates"], r["worst_flipped_gate"],
            r["pool_flips"]))
    emit()


emit("# Parity margins of the 3x3 convolution kernels: F(2x2,3x3) (csrc/conv_wino.hip), F(4x4,3x3) with the transform-domain products")
emit("# on the f32 MFMA (csrc/conv_wino4.hip, cova_set_option(9, 1)) and on the bf16 matrix pipe with three-piece operands (the default,")
emit("# csrc/conv_wino4_split.h); tools/wino4_margin.py on one MI355X.  Test gates: logits 5e-5, loss 2e-5, gradients 1e-4 of")
emit("# each tensor's scale against the forced-routing oracle; flipped gates must sit within 2e-5 of zero and be fewer than")
emit("# 1e-4 of a layer's decisions (tests/helpers.py).  'ref' = the fixture captured from the imported reference.")
emit()
for name in FULL_CASES:
    fx, cfg, sd, batch = load_case(name)
    res = run(cfg, int(fx["meta/img_h"]), sd, batch, torch.from_numpy(fx["train/logits"]), float(fx["train/loss"]))
    show("fixture %s" % name, res)
# full resolution: one 1280 x 1280 page, 90 boxes (the benchmark geometry; oracle on the CPU)
cfg = dict(roi_output_size=(3, 3), n_classes=4, use_context=True, hidden_dim=384, bbox_hidden_dim=32, n_additional_feat=0,
           drop_prob=0.0)
sd = weights.seeded_state_dict(123, logit_gain=4.0, **{k: v for k, v in cfg.items() if k != "drop_prob"})
batch = synthetic.make_batch(1, img_h=1280, boxes_per_page=[90], context_size=12, seed=123)
show("one 1280 x 1280 page, 90 boxes, K = 24 (oracle only; no reference fixture at this size)", run(cfg, 1280, sd, batch))
engine.OPTIONS.wino4 = True
_lib.query("cova_set_option", 9, 0)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(lines) + "\n")
