"""Index-arithmetic check at batch sizes whose activations exceed 2^31 / 2^32 elements: a batch made of
R copies of a 16-page batch has the same BatchNorm statistics, so every copy's logits must equal the
16-page logits and every gradient must be R times the 16-page gradient (up to summation order)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cova_amd  # noqa
from cova_web_object_detection_amd import engine, synthetic, weights
from cova_web_object_detection_amd.trainer import HotPathTrainer

dev = "cuda:0"
CFG = dict(roi_output_size=(3, 3), n_classes=4, use_context=True, hidden_dim=384, bbox_hidden_dim=32,
           n_additional_feat=0, drop_prob=0.0)
wcfg = {k: v for k, v in CFG.items() if k != "drop_prob"}
sd = weights.seeded_state_dict(5, **wcfg)
P = int(os.environ.get("PAGES", 16))
base = synthetic.make_batch(P, img_h=1280, boxes_per_page=[90] * P, context_size=12, seed=5)
base = {k: v.to(dev) for k, v in base.items() if torch.is_tensor(v)}
n = base["bboxes"].shape[0]


def run(batch):
    tr = HotPathTrainer(CFG, sd, dev)
    logits, sv = engine.model_fwd(tr.cfg, tr.params, tr.buffers, batch["images"], batch["bboxes"],
                                  batch["additional_feats"], batch["context_indices"], True, (1, 2), None)
    loss, dl, pred = engine.ce_sum(logits, batch["labels"])
    engine.model_bwd(sv, dl, tr.params, tr.grads)
    torch.cuda.synchronize()
    return logits.clone(), float(loss), tr.gbucket.flat.clone()


l16, loss16, g16 = run(base)
for R in [int(r) for r in os.environ.get("REPS", "6,11").split(",")]:
    bb = torch.cat([base["bboxes"] + torch.tensor([r * P, 0, 0, 0, 0.0], device=dev) for r in range(R)])
    ctx = torch.cat([torch.where(base["context_indices"] >= 0, base["context_indices"] + r * n,
                                 base["context_indices"]) for r in range(R)])
    big = dict(images=base["images"].repeat(R, 1, 1, 1), bboxes=bb, context_indices=ctx,
               additional_feats=base["additional_feats"].repeat(R, 1), labels=base["labels"].repeat(R))
    lg, loss, g = run(big)
    y1_elems = R * P * 640 * 640 * 64
    e_log = float((lg.view(R, n, -1) - l16.unsqueeze(0)).abs().max()) / float(l16.abs().max())
    e_g = float((g / R - g16).abs().max()) / float(g16.abs().max())
    print("pages %4d (conv1 output %.2f G elements, %.1f GB): logits err %.2e  loss ratio %.6f  grad err %.2e  peak mem %.1f GB"
          % (R * P, y1_elems / 1e9, y1_elems * 4 / 1e9, e_log, loss / (R * loss16), e_g,
             torch.cuda.max_memory_allocated() / 1e9))
    del big, lg, g
    torch.cuda.empty_cache()
