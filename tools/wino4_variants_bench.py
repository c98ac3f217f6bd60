"""Stand-alone timing of the five F(4x4,3x3) kernel variants the train step uses (16 x 320 x 320 x 64)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cova_amd  # noqa: F401
from cova_web_object_detection_amd._lib import call, query
dev = "cuda:0"
B, H, W = 16, 320, 320


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


x, x2, z, add, act = (torch.randn(B, H, W, 64, device=dev) for _ in range(5))
w = torch.randn(64, 64, 3, 3, device=dev) * 0.05
uf, ud = torch.empty(query("cova_conv3x3_wino4_u_floats"), device=dev), torch.empty(query("cova_conv3x3_wino4_u_floats"), device=dev)
call("cova_conv3x3_wino4_prep", w, uf, ud)
out = torch.empty_like(x)
part = torch.empty(query("cova_conv3x3_wino4_num_partials", B, H, W), 2, 64, device=dev)
abc = torch.randn(3, 64, device=dev)
mean, invstd = torch.randn(64, device=dev) * 0.1, torch.rand(64, device=dev) + 0.5
N = None
full = lambda *a: call("cova_conv3x3_wino4_full", *a, out, part, B, H, W)
cases = [
    ("<1,0,0,0> forward, statistics", lambda: full(x, N, N, 0, uf, N, N, N, N, N, N, N)),
    ("<1,1,0,0> forward, BN+ReLU on load", lambda: full(x, N, abc, 1, uf, N, N, N, N, N, N, N)),
    ("<1,0,0,1> dgrad (materialised operand), mask from z", lambda: full(x, N, N, 0, ud, N, N, abc[0], abc[2], z, mean, invstd)),
    ("<1,0,1,1> ... + addend", lambda: full(x, N, N, 0, ud, add, N, abc[0], abc[2], z, mean, invstd)),
    ("<1,0,1,2> ... + addend, mask from act", lambda: full(x, N, N, 0, ud, add, act, N, N, z, mean, invstd)),
    ("<1,2,0,1> dgrad, two tensors on load, mask from z", lambda: full(x, x2, abc, 0, ud, N, N, abc[0], abc[2], z, mean, invstd)),
    ("<1,2,1,2> ... + addend, mask from act", lambda: full(x, x2, abc, 0, ud, add, act, N, N, z, mean, invstd)),
]
print(" ".join("%.3f" % timeit(fn) for _, fn in cases), " ms :", " | ".join(n.split(">")[0] + ">" for n, _ in cases))
