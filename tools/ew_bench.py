"""Streaming element-wise kernels at the bench size: GB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cova_amd  # noqa
from cova_web_object_detection_amd import _lib
call, query = _lib.call, _lib.query
dev = "cuda:0"
B, H, W = 16, 320, 320
R = B * H * W


def timeit(fn, n=20):
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


z, x, out = (torch.randn(B, H, W, 64, device=dev) for _ in range(3))
sc, sh = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.1
T = z.numel() * 4 / 1e9
t = timeit(lambda: call("cova_bn_act_fwd", z, 64, sc, sh, x, 64, out, 64, R, 64, 1))
print("bn_act_fwd +res  (3 maps): %.3f ms  %.2f TB/s" % (t, 3 * T / t))
t = timeit(lambda: call("cova_bn_act_fwd", z, 64, sc, sh, None, 0, out, 64, R, 64, 1))
print("bn_act_fwd       (2 maps): %.3f ms  %.2f TB/s" % (t, 2 * T / t))
y1 = torch.randn(B, 640, 640, 64, device=dev)
p1 = torch.empty(B, H, W, 64, device=dev)
idx = torch.empty(B, H, W, 64, device=dev, dtype=torch.uint8)
ymax = torch.empty_like(p1)
t = timeit(lambda: call("cova_bn_relu_maxpool_fwd", y1, sc, sh, p1, idx, ymax, B, 640, 640))
print("bn_relu_maxpool_fwd (+ymax): %.3f ms  %.2f TB/s" % (t, (4 * T + 2 * T + T / 4) / t))
t = timeit(lambda: out.copy_(z))
print("torch copy (2 maps): %.3f ms  %.2f TB/s" % (t, 2 * T / t))
