"""Isolated timing of the affine-on-load Winograd kernels vs their plain forms (B=16, 320x320x64)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cova_amd  # noqa
from cova_web_object_detection_amd import _lib
call, query = _lib.call, _lib.query
dev = "cuda:0"
B, H, W = int(os.environ.get("B", 16)), 320, 320


def timeit(fn, n=20):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


x, x2, z = (torch.randn(B, H, W, 64, device=dev) for _ in range(3))
add = torch.randn(B, H, W, 64, device=dev)
w = torch.randn(64, 64, 3, 3, device=dev) * 0.05
uf, ud = torch.empty(16, 16, 4, 64, device=dev), torch.empty(16, 16, 4, 64, device=dev)
call("cova_conv3x3_prep_weights_wino", w, uf, ud)
out = torch.empty_like(x)
part = torch.empty(2 * query("cova_conv3x3_num_tiles", B, H, W), 2, 64, device=dev)   # enough for both Winograd geometries
abc = torch.randn(3, 64, device=dev)
mean, invstd = torch.randn(64, device=dev) * 0.1, torch.rand(64, device=dev) + 0.5
N = None
cases = [
    ("fwd plain +stats", lambda: call("cova_conv3x3_wino", x, uf, N, N, N, N, N, out, part, B, H, W)),
    ("fwd relu(A x + C) on load +stats", lambda: call("cova_conv3x3_wino_pro", x, N, abc, 1, uf, N, N, N, N, N, N, N, out, part, B, H, W)),
    ("dgrad plain +addend +act mask", lambda: call("cova_conv3x3_wino", x, ud, add, x2, z, mean, invstd, out, part, B, H, W)),
    ("dgrad A x+B x2+C on load +addend +act mask", lambda: call("cova_conv3x3_wino_pro", x, x2, abc, 0, ud, add, x2, N, N, z, mean, invstd, out, part, B, H, W)),
    ("dgrad A x+B x2+C on load, z mask", lambda: call("cova_conv3x3_wino_pro", x, x2, abc, 0, ud, N, N, abc[0], abc[2], z, mean, invstd, out, part, B, H, W)),
    ("dgrad A x+B x2+C on load +addend, no stats", lambda: call("cova_conv3x3_wino_pro", x, x2, abc, 0, ud, add, N, N, N, N, N, N, out, N, B, H, W)),
]
for name, fn in cases:
    print("%-48s %.3f ms" % (name, timeit(fn)))
ws = torch.empty(query("cova_conv3x3_wgrad_workspace_floats", B, H, W), device=dev)
dw = torch.empty(64, 64, 3, 3, device=dev)
cases = [
    ("wgrad plain", lambda: call("cova_conv3x3_wgrad_wino", x, x2, dw, ws, B, H, W)),
    ("wgrad act on load", lambda: call("cova_conv3x3_wgrad_wino_pro", x, abc, 1, x2, N, N, dw, ws, B, H, W)),
    ("wgrad dz on load", lambda: call("cova_conv3x3_wgrad_wino_pro", x, N, 0, x2, z, abc, dw, ws, B, H, W)),
    ("wgrad act + dz on load", lambda: call("cova_conv3x3_wgrad_wino_pro", x, abc, 1, x2, z, abc, dw, ws, B, H, W)),
]
for name, fn in cases:
    print("%-48s %.3f ms" % (name, timeit(fn)))
R = B * H * W
a1 = torch.empty_like(x)
print("%-48s %.3f ms" % ("bn_act_fwd (pass being removed)", timeit(lambda: call("cova_bn_act_fwd", x, 64, abc[0], abc[2], N, 0, a1, 64, R, 64, 1))))
coef = torch.randn(2, 64, device=dev)
print("%-48s %.3f ms" % ("bn_bwd_apply (pass being removed)", timeit(lambda: call("cova_bn_bwd_apply", x, 64, N, 0, z, 64, mean, invstd, abc[0], coef, a1, 64, N, 0, R, 64))))
if os.environ.get("COVA_ABLATE"):
    print("--- epilogue ablations (COVA_ABLATE build)")
    plain = lambda: call("cova_conv3x3_wino", x, uf, N, N, N, N, N, out, N, B, H, W)
    stats = lambda: call("cova_conv3x3_wino", x, uf, N, N, N, N, N, out, part, B, H, W)
    full = lambda: call("cova_conv3x3_wino", x, ud, add, x2, z, mean, invstd, out, part, B, H, W)
    for nm, fn in (("plain", plain), ("stats", stats), ("addend+act+z+stats", full)):
        for abl, what in ((0, "full"), (1, "no epilogue at all"), (64, "no stats reduction"), (128, "no operand loads"),
                          (256, "no stores"), (512, "no L2 prefetch"), (64 + 128 + 256, "transform only")):
            query("cova_set_option", 5, abl)
            print("%-20s abl %3d %-22s %.3f ms" % (nm, abl, what, timeit(fn)))
    query("cova_set_option", 5, 0)
print("--- geometry A (8x32 tiles, 1 block/CU) vs B (8x16 tiles, 2 blocks/CU)")
for geo in (1, 2):
    query("cova_set_option", 6, geo)
    cases = [
        ("fwd plain", lambda: call("cova_conv3x3_wino", x, uf, N, N, N, N, N, out, N, B, H, W)),
        ("fwd plain +stats", lambda: call("cova_conv3x3_wino", x, uf, N, N, N, N, N, out, part, B, H, W)),
        ("fwd relu(A x + C) on load +stats", lambda: call("cova_conv3x3_wino_pro", x, N, abc, 1, uf, N, N, N, N, N, N, N, out, part, B, H, W)),
        ("dgrad plain +addend +act mask", lambda: call("cova_conv3x3_wino", x, ud, add, x2, z, mean, invstd, out, part, B, H, W)),
        ("dgrad pro +addend +act mask", lambda: call("cova_conv3x3_wino_pro", x, x2, abc, 0, ud, add, x2, N, N, z, mean, invstd, out, part, B, H, W)),
        ("dgrad pro, z mask", lambda: call("cova_conv3x3_wino_pro", x, x2, abc, 0, ud, N, N, abc[0], abc[2], z, mean, invstd, out, part, B, H, W)),
    ]
    for name, fn in cases:
        print("geo %d  %-36s %.3f ms" % (geo, name, timeit(fn)))
query("cova_set_option", 6, 1)
