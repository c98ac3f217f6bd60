"""Isolated timing of conv1 forward, plain (cova_conv1_fwd_tail, + the pool pass) against the max-pool in its epilogue
(cova_conv1_fwd_pool); COVA_HIP_LIB selects an ablation build (tools/c1p_abl_build.sh)."""
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


def timeit(fn, n=20):
    for _ in range(40):
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
w = torch.randn(64, 3, 7, 7, device=dev) * 0.1
gamma = torch.rand(64, device=dev) + 0.5
y1 = torch.empty(B, H1, W1, 64, device=dev)
ymax = torch.empty(B, H2, W2, 64, device=dev)
p1 = torch.empty(B, H2, W2, 64, device=dev)
idx = torch.empty(B, H2, W2, 64, device=dev, dtype=torch.uint8)
scale, shift = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.1
part = torch.empty(query("cova_conv1_num_partials", B, H, W), 2, 64, device=dev)
partp = torch.empty(query("cova_conv1_pool_num_partials", B, H, W), 2, 64, device=dev)
t0 = timeit(lambda: call("cova_conv1_fwd_tail", img, w, y1, part, B, H, W, None))
t1 = timeit(lambda: call("cova_bn_relu_maxpool_fwd", y1, scale, shift, p1, idx, ymax, B, H1, W1))
t2 = timeit(lambda: call("cova_conv1_fwd_pool", img, w, gamma, y1, ymax, idx, partp, B, H, W, None))
print("conv1 fwd %.3f ms + pool pass %.3f ms = %.3f ms | conv1 fwd with the pool in its epilogue %.3f ms" % (t0, t1, t0 + t1, t2))
