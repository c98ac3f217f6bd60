import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cova_amd  # noqa
from cova_web_object_detection_amd._lib import call, query
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(3)
w = torch.randn(64, 3, 7, 7, device=dev, generator=g) * 0.05
B, H, W = 2, 150, 330
img = torch.rand(B, 3, H, W, device=dev, generator=g)
ref = torch.nn.functional.conv2d(img.double().cpu(), w.double().cpu(), stride=2, padding=3).permute(0, 2, 3, 1)
H1, W1 = ref.shape[1], ref.shape[2]
for cap in (0, 7, 3):
    call("cova_set_option", 2, cap)
    out = torch.zeros(B, H1, W1, 64, device=dev)
    part = torch.zeros(query("cova_conv1_num_partials", B, H, W), 2, 64, device=dev)
    call("cova_conv1_fwd_tail", img, w, out, part, B, H, W, None)
    torch.cuda.synchronize()
    err = (out.double().cpu() - ref).abs().amax(dim=3)       # [B, H1, W1]
    bad = err > 1e-4
    print("cap", cap, "max err", err.max().item(), "bad pixels", int(bad.sum()))
    if bad.any():
        tiles = {}
        for b, y, x in bad.nonzero().tolist():
            tiles.setdefault((b, y // 8, x // 32), set()).add((y % 8, x % 32))
        for k in sorted(tiles)[:12]:
            rows = sorted({r for r, _ in tiles[k]}); cols = sorted({c for _, c in tiles[k]})
            print("  tile", k, "n", len(tiles[k]), "rows", rows, "cols", cols[:8], "..")
call("cova_set_option", 2, 0)
print("--- prepared weights + wgrad in between (the test's sequence)")
wk = torch.empty(154, 64, device=dev)
call("cova_conv1_prep_weights", w, wk)
dy = torch.randn(B, H1, W1, 64, device=dev, generator=g)
ws = torch.empty(query("cova_conv1_wgrad_workspace_floats", B, H, W), device=dev)
for cap in (0, 7):
    call("cova_set_option", 2, cap)
    out = torch.zeros(B, H1, W1, 64, device=dev)
    part = torch.zeros(query("cova_conv1_num_partials", B, H, W), 2, 64, device=dev)
    call("cova_conv1_fwd", img, wk, out, part, B, H, W)
    torch.cuda.synchronize()
    err = (out.double().cpu() - ref).abs().amax(dim=3)
    print("cap", cap, "fwd max err", err.max().item(), "bad", int((err > 1e-4).sum()))
    dw = torch.zeros(64, 3, 7, 7, device=dev)
    call("cova_conv1_wgrad", img, dy, dw, ws, B, H, W)
    torch.cuda.synchronize()
    print("   wk changed:", not torch.equal(wk, wk.clone()), " ws floats", ws.numel(), "grid", query("cova_conv1_num_partials", B, H, W))
call("cova_set_option", 2, 0)
