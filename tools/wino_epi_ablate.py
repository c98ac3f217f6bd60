"""Ablation of the epilogue-heavy Winograd data-gradient launch (needs a COVA_ABLATE=1 build:
COVA_ABLATE=1 python cova-web-object-detection_amd/_build.py; see tools/conv_bench.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cova_amd  # noqa
from cova_web_object_detection_amd import _lib
call, query = _lib.call, _lib.query
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


x, x2, z, add = (torch.randn(B, H, W, 64, device=dev) for _ in range(4))
w = torch.randn(64, 64, 3, 3, device=dev) * 0.05
uf, ud = torch.empty(16, 16, 4, 64, device=dev), torch.empty(16, 16, 4, 64, device=dev)
call("cova_conv3x3_prep_weights_wino", w, uf, ud)
out = torch.empty_like(x)
part = torch.empty(2 * query("cova_conv3x3_num_tiles", B, H, W), 2, 64, device=dev)
abc = torch.randn(3, 64, device=dev)
mean, invstd = torch.randn(64, device=dev) * 0.1, torch.rand(64, device=dev) + 0.5
N = None
heavy = lambda: call("cova_conv3x3_wino_pro", x, x2, abc, 0, ud, add, x2, N, N, z, mean, invstd, out, part, B, H, W)
zmask = lambda: call("cova_conv3x3_wino_pro", x, x2, abc, 0, ud, N, N, abc[0], abc[2], z, mean, invstd, out, part, B, H, W)
plain = lambda: call("cova_conv3x3_wino", x, ud, N, N, N, N, N, out, N, B, H, W)
for name, fn in (("plain", plain), ("z-mask dgrad (PRO, BN1)", zmask), ("heavy dgrad (PRO2, ADD, BN2)", heavy)):
    for abl, what in ((0, "full"), (1, "no epilogue at all"), (64, "no statistics reduction"), (128, "no epilogue operand loads"),
                      (256, "no output stores"), (384, "no operand loads, no stores"), (448, "no loads/stores/stat reduce"),
                      (6, "no refill loads+stores"), (32, "no input transform"), (16, "no chunk barrier")):
        query("cova_set_option", 5, abl)
        print("%-30s abl %3d %-32s %.3f ms" % (name, abl, what, timeit(fn)))
    query("cova_set_option", 5, 0)
