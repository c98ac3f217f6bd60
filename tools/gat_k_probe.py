"""GAT layer (drop-in module) vs the CPU oracle for several neighbour-table widths: max error / scale of h', attention,
dh and the parameter gradients (the K > 64 paths of csrc/roi_gat.hip)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cova_amd  # noqa
from cova_web_object_detection_amd.models import GraphAttentionLayer
from oracle import cova_oracle as O

def rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-9))

for K, N in ((24, 300), (64, 300), (100, 300), (128, 300), (130, 300), (200, 300), (256, 300)):
    rs = np.random.RandomState(K)
    Fd, D = 96, 32
    sd = {"gat.W_i.weight": torch.from_numpy(rs.uniform(-0.3, 0.3, (D, Fd)).astype(np.float32)),
          "gat.W_j.weight": torch.from_numpy(rs.uniform(-0.3, 0.3, (D, Fd)).astype(np.float32)),
          "gat.attention_layer.weight": torch.from_numpy(rs.uniform(-0.5, 0.5, (1, 2 * D)).astype(np.float32)),
          "gat.attention_layer.bias": torch.from_numpy(rs.uniform(-0.5, 0.5, (1,)).astype(np.float32))}
    h = torch.from_numpy(rs.standard_normal((N, Fd)).astype(np.float32))
    ctx = rs.randint(-1, N, (N, K)).astype(np.int64)
    ctx[3] = -1
    ctx[5, K // 2:] = -1
    g = torch.from_numpy(rs.standard_normal((N, D)).astype(np.float32))
    ho = h.clone().requires_grad_(True)
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    hp_o, at_o = O.gat(ho, torch.from_numpy(ctx), sdo, return_attn_wts=True)
    (hp_o * g).sum().backward()
    layer = GraphAttentionLayer(Fd, D)
    layer.load_state_dict({k[4:]: v for k, v in sd.items()})
    layer = layer.cuda()
    hg = h.cuda().requires_grad_(True)
    hp, at = layer(hg, torch.from_numpy(ctx).cuda(), return_attn_wts=True)
    (hp * g.cuda()).sum().backward()
    errs = {"hp": rel(hp, hp_o), "attn": rel(at, at_o), "dh": rel(hg.grad, ho.grad)}
    for k, p in layer.named_parameters():
        errs[k] = rel(p.grad, sdo["gat." + k].grad)
    print("K=%3d" % K, " ".join("%s %.2e" % kv for kv in errs.items()))

# model-like neighbour tables (datasets.py:121-128 windows, -1 pads) and wide inputs
from cova_web_object_detection_amd import synthetic
for cs, boxes, scale in ((50, [130, 9], 1.0), (100, [230, 40], 1.0), (100, [230, 40], 4.0), (100, [100, 40], 1.0), (80, [230, 40], 1.0)):
    b = synthetic.make_batch(len(boxes), img_h=64, img_w=64, boxes_per_page=boxes, context_size=cs, seed=53)
    ctx = b["context_indices"]
    N, K = ctx.shape
    rs = np.random.RandomState(cs)
    Fd, D = 608, 96
    sd = {"gat.W_i.weight": torch.from_numpy(rs.uniform(-0.05, 0.05, (D, Fd)).astype(np.float32)),
          "gat.W_j.weight": torch.from_numpy(rs.uniform(-0.05, 0.05, (D, Fd)).astype(np.float32)),
          "gat.attention_layer.weight": torch.from_numpy(rs.uniform(-0.2, 0.2, (1, 2 * D)).astype(np.float32)),
          "gat.attention_layer.bias": torch.from_numpy(rs.uniform(-0.5, 0.5, (1,)).astype(np.float32))}
    h = torch.from_numpy((scale * np.abs(rs.standard_normal((N, Fd)))).astype(np.float32))
    g = torch.from_numpy(rs.standard_normal((N, D)).astype(np.float32))
    ho = h.clone().requires_grad_(True)
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    hp_o, at_o = O.gat(ho, ctx, sdo, return_attn_wts=True)
    (hp_o * g).sum().backward()
    layer = GraphAttentionLayer(Fd, D)
    layer.load_state_dict({k[4:]: v for k, v in sd.items()})
    layer = layer.cuda()
    hg = h.cuda().requires_grad_(True)
    hp, at = layer(hg, ctx.cuda(), return_attn_wts=True)
    (hp * g.cuda()).sum().backward()
    errs = {"hp": rel(hp, hp_o), "attn": rel(at, at_o), "dh": rel(hg.grad, ho.grad)}
    for k, p in layer.named_parameters():
        errs[k] = rel(p.grad, sdo["gat." + k].grad)
    print("windows cs=%d boxes=%s x%g N=%d K=%d" % (cs, boxes, scale, N, K), " ".join("%s %.2e" % kv for kv in errs.items()))
