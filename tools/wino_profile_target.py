"""Runs one Winograd conv configuration a few times (target for rocprofv3 --pmc)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cova_amd  # noqa
from cova_web_object_detection_amd import _lib
call, query = _lib.call, _lib.query
dev = "cuda:0"
B, H, W = 16, 320, 320
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
x, x2, z, add = (torch.randn(B, H, W, 64, device=dev) for _ in range(4))
w = torch.randn(64, 64, 3, 3, device=dev) * 0.05
uf, ud = torch.empty(16, 16, 4, 64, device=dev), torch.empty(16, 16, 4, 64, device=dev)
call("cova_conv3x3_prep_weights_wino", w, uf, ud)
out = torch.empty_like(x)
part = torch.empty(2 * query("cova_conv3x3_num_tiles", B, H, W), 2, 64, device=dev)   # enough for both Winograd geometries
abc = torch.randn(3, 64, device=dev)
mean, invstd = torch.randn(64, device=dev) * 0.1, torch.rand(64, device=dev) + 0.5
N = None
fns = {
    "plain": lambda: call("cova_conv3x3_wino", x, uf, N, N, N, N, N, out, N, B, H, W),
    "stats": lambda: call("cova_conv3x3_wino", x, uf, N, N, N, N, N, out, part, B, H, W),
    "full": lambda: call("cova_conv3x3_wino", x, ud, add, x2, z, mean, invstd, out, part, B, H, W),
    "pro": lambda: call("cova_conv3x3_wino_pro", x, x2, abc, 0, ud, N, N, N, N, N, N, N, out, N, B, H, W),
}
ws = torch.empty(query("cova_conv3x3_wgrad_workspace_floats", B, H, W), device=dev)
dw = torch.empty(64, 64, 3, 3, device=dev)
fns["wgrad"] = lambda: call("cova_conv3x3_wgrad_wino", x, x2, dw, ws, B, H, W)
fns["wgrad_pro"] = lambda: call("cova_conv3x3_wgrad_wino_pro", x, abc, 1, x2, z, abc, dw, ws, B, H, W)
img = torch.rand(B, 3, 1280, 1280, device=dev)
w1 = torch.randn(64, 3, 7, 7, device=dev) * 0.1
wk = torch.empty(154, 64, device=dev)
call("cova_conv1_prep_weights", w1, wk)
y1 = torch.empty(B, 640, 640, 64, device=dev)
part1 = torch.empty(query("cova_conv1_num_tiles", B, 1280, 1280), 2, 64, device=dev)
ws1 = torch.empty(query("cova_conv1_wgrad_workspace_floats", B, 1280, 1280), device=dev)
dw1 = torch.empty(64, 3, 7, 7, device=dev)
fns["conv1"] = lambda: call("cova_conv1_fwd", img, wk, y1, part1, B, 1280, 1280)
fns["conv1_wgrad"] = lambda: call("cova_conv1_wgrad", img, y1, dw1, ws1, B, 1280, 1280)
for _ in range(int(os.environ.get("N", 6))):
    fns[mode]()
torch.cuda.synchronize()
