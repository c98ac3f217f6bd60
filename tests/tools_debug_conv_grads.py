import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import cova_amd  # noqa
from cova_web_object_detection_amd import engine, synthetic, weights
from oracle import cova_oracle as O
dev = "cuda:0"
cfg = dict(roi_output_size=(3, 3), n_classes=4, use_context=True, hidden_dim=384,
           bbox_hidden_dim=32, n_additional_feat=0, drop_prob=0.0)
wcfg = {k: v for k, v in cfg.items() if k != "drop_prob"}
sd = weights.seeded_state_dict(123, logit_gain=4.0, **wcfg)
batch = synthetic.make_batch(2, img_h=128, boxes_per_page=[40, 23], context_size=12, seed=123)
keys = O.param_keys(sd)
work = O.clone_state_dict(sd)
for k in keys:
    work[k] = work[k].clone().requires_grad_(True)
logits, inter = O.forward(work, batch["images"], batch["bboxes"], batch["additional_feats"], batch["context_indices"], cfg, True, None, True)
inter["feat"].retain_grad(); inter["visual"].retain_grad()
loss = F.cross_entropy(logits, batch["labels"], reduction="sum"); loss.backward()
params = {k: sd[k].to(dev) for k in keys}
buffers = {k: v.to(dev) for k, v in sd.items() if k not in params}
args = [batch[k].to(dev) for k in ("images", "bboxes", "additional_feats", "context_indices")]
lg, sv = engine.model_fwd(cfg, params, buffers, *args, True)
_, dl, _ = engine.ce_sum(lg, batch["labels"].to(dev))
# replicate model_bwd step by step
dcomb, grads = engine.decoder_bwd(sv["dec"], dl, params)
grads.update(engine.gat_stack_bwd(sv["gat"], dcomb, sv["T"], sv["N"], sv["F"], sv["D"], params))
def rel(a, b, name):
    print("%-20s rel err %.3e  (max ref %.3e)" % (name, (a - b).abs().max().item() / b.abs().max().item(), b.abs().max().item()))
feat = engine.block_out(sv["conv"]["blocks"][1]).cpu().permute(0, 3, 1, 2)
rel(feat, inter["feat"].detach(), "feat fwd")
rel(sv["comb"][:, :576].cpu(), inter["visual"].detach(), "visual fwd")
rel(dcomb[:, :576].cpu(), inter["visual"].grad, "d visual")
arg_ref = O.roi_pool_argmax(inter["feat"].detach(), batch["bboxes"], (3, 3), 0.25)[1].reshape(63, 576)
print("argmax mismatches:", (sv["roi"]["argmax"].cpu() != arg_ref).sum().item(), "of", arg_ref.numel())
dfeat = engine.roipool_bwd(sv["roi"], dcomb, sv["T"])
rel(dfeat.cpu().permute(0, 3, 1, 2), inter["feat"].grad, "d feat")
print("sum dfeat", dfeat.sum().item(), inter["feat"].grad.sum().item())
