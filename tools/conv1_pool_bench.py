"""Isolated timing: conv1 weight gradient, plain vs with the pooling backward folded in."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cova_amd  # noqa
from cova_web_object_detection_amd import _lib
call, query = _lib.call, _lib.query
dev = "cuda:0"
B, H, W = 16, 1280, 1280
H1 = W1 = 640
H2 = W2 = 320


def timeit(fn, n=10):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


img = torch.rand(B, 3, H, W, device=dev)
y1 = torch.randn(B, H1, W1, 64, device=dev)
scale, shift = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.1
p1 = torch.empty(B, H2, W2, 64, device=dev)
idx = torch.empty(B, H2, W2, 64, device=dev, dtype=torch.uint8)
call("cova_bn_relu_maxpool_fwd", y1, scale, shift, p1, idx, None, B, H1, W1)
dp = torch.randn(B, H2, W2, 64, device=dev) * (p1 > 0)
abc = torch.randn(3, 64, device=dev)
ws = torch.empty(query("cova_conv1_wgrad_workspace_floats", B, H, W), device=dev)
dw = torch.empty(64, 3, 7, 7, device=dev)
print("conv1 wgrad plain          %.3f ms" % timeit(lambda: call("cova_conv1_wgrad", img, y1, dw, ws, B, H, W)))
print("conv1 wgrad + pool backward %.3f ms" % timeit(lambda: call("cova_conv1_wgrad_poolbwd", img, y1, dp, idx, abc, dw, ws, B, H, W)))
