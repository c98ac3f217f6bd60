"""Per-tensor distance of the HIP gradients from a reference fixture's (nothing forced), the gate flips against the
unforced oracle, and the forced-routing distance: python tools/fixture_diag.py cova_h64_n90   (COVA_CONV1_F32=1: A/B)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cova_amd  # noqa: F401 (registers the package alias)
import test_model_gpu as T
from helpers import *  # noqa
name = sys.argv[1] if len(sys.argv) > 1 else "cova_h64_n90"
fx, cfg, sd, batch = T.load_case(name)
img_h = int(fx["meta/img_h"])
args = T.dev_batch(batch)
m = T.build(cfg, img_h, sd); m.train()
logits = m(*args)
routing = routing_from_saved(logits.grad_fn.sv)
loss = torch.nn.CrossEntropyLoss(reduction="sum")(logits, batch["labels"].to(T.DEV)); loss.backward()
print("train logits err %.3e   loss rel err %.3e" % (T.relerr(logits.detach().cpu(), fx["train/logits"]),
      abs(loss.item() - float(fx["train/loss"])) / abs(float(fx["train/loss"]))))
grads = {k: p.grad for k, p in m.named_parameters()}
frac, total, worst, per = unforced_fraction_above(fx, grads, rtol=2e-4)
print("unforced: %.3f %% of %d entries beyond 2e-4" % (100 * frac, total))
for k in sorted(per, key=lambda k: -per[k][2])[:40]:
    print("   %-40s bad %6d / %6d   max %.2e" % (k, *per[k]))
tap = {}
T.O.loss_and_grads(sd, batch["images"], batch["bboxes"], batch["additional_feats"], batch["context_indices"], batch["labels"], cfg, None, {"_tap": tap})
print("flips vs unforced oracle:", assert_gate_flips_near_zero(routing, tap))
for key, pre in tap.items():
    if key in routing and key != "pool_in":
        gate = routing[key].to(torch.bool).reshape(pre.shape); diff = gate != (pre > 0)
        if int(diff.sum()):
            print("   ", key, "flipped at pre-activations", pre[diff].flatten()[:6].tolist(), "scale", float(pre.abs().max()))
