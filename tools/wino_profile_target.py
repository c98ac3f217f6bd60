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
part = torch.empty(query("cova_conv3x3_num_tiles", B, H, W), 2, 64, device=dev)
abc = torch.randn(3, 64, device=dev)
mean, invstd = torch.randn(64, device=dev) * 0.1, torch.rand(64, device=dev) + 0.5
N = None
fns = {
    "plain": lambda: call("cova_conv3x3_wino", x, uf, N, N, N, N, N, out, N, B, H, W),
    "stats": lambda: call("cova_conv3x3_wino", x, uf, N, N, N, N, N, out, part, B, H, W),
    "full": lambda: call("cova_conv3x3_wino", x, ud, add, x2, z, mean, invstd, out, part, B, H, W),
    "pro": lambda: call("cova_conv3x3_wino_pro", x, x2, abc, 0, ud, N, N, N, N, N, N, N, out, N, B, H, W),
}
for _ in range(int(os.environ.get("N", 6))):
    fns[mode]()
torch.cuda.synchronize()
