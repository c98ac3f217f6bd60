import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cova_amd  # noqa
from cova_web_object_detection_amd import _lib
call, query = _lib.call, _lib.query
dev = torch.device("cuda", 0)


def b2b(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def setup():
    img = torch.rand(16, 3, 1280, 1280, device=dev)
    wk = torch.empty(154, 64, device=dev)
    call("cova_conv1_prep_weights", torch.randn(64, 3, 7, 7, device=dev) * 0.1, wk)
    y1 = torch.empty(16, 640, 640, 64, device=dev)
    part = torch.empty(query("cova_conv1_num_tiles", 16, 1280, 1280), 2, 64, device=dev)
    x = torch.randn(16, 320, 320, 64, device=dev)
    out = torch.empty_like(x)
    wf, wd = torch.empty(9, 64, 64, device=dev), torch.empty(9, 64, 64, device=dev)
    call("cova_conv3x3_prep_weights", torch.randn(64, 64, 3, 3, device=dev) * 0.05, wf, wd)
    f1 = lambda: call("cova_conv1_fwd", img, wk, y1, part, 16, 1280, 1280)
    f3 = lambda: call("cova_conv3x3_fwd", x, wf, None, out, None, 16, 320, 320)
    return f1, f3, (img, wk, y1, part, x, out, wf, wd)


f1, f3, keep = setup()
print("fresh process:           conv1 %.3f  conv3x3 %.3f" % (b2b(f1), b2b(f3)))
print("again:                   conv1 %.3f  conv3x3 %.3f" % (b2b(f1), b2b(f3)))
hog = [torch.zeros(1 << 28, device=dev) for _ in range(20)]      # 20 GiB touched
torch.cuda.synchronize()
print("same buffers, +20GiB:    conv1 %.3f  conv3x3 %.3f" % (b2b(f1), b2b(f3)))
g1, g3, keep2 = setup()
print("new buffers after hog:   conv1 %.3f  conv3x3 %.3f" % (b2b(g1), b2b(g3)))
del hog
torch.cuda.empty_cache()
print("old buffers, hog freed:  conv1 %.3f  conv3x3 %.3f" % (b2b(f1), b2b(f3)))
import time
t0 = time.time()
while time.time() - t0 < 3.0:
    f3()
torch.cuda.synchronize()
print("after 3 s of MFMA load:  conv1 %.3f  conv3x3 %.3f" % (b2b(f1), b2b(f3)))
