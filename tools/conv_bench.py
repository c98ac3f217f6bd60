"""Micro-benchmark of the conv kernels at the bench workload size (B=16, 1280x1280 pages)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cova_amd  # noqa
from cova_web_object_detection_amd import _lib
import probe_lib  # noqa: E402  (tools/probe_lib.py: builds + registers libcova_probe.so)
probe_lib.load()
import direct_lib  # noqa: E402  (tools/direct_lib.py: the direct-form 3x3 kernels, out of the product library since round 3)
direct_lib.load()
call, query = _lib.call, _lib.query
dev = "cuda:0"
B, H, W = int(os.environ.get("B", 16)), 320, 320
iters = 10


def timeit(fn, n=iters):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


scratch = torch.zeros(16, device=dev)
for blocks in (256, 512):
    iters = 4000
    t = timeit(lambda: call("cova_probe_mfma_f32", scratch, blocks, iters), 5)
    print("MFMA f32 32x32x2 probe, %d blocks x 8 waves: %.3f ms  %.1f TF/s" % (blocks, t, blocks * 8 * iters * 16 * 4096 / t / 1e9))
x = torch.randn(B, H, W, 64, device=dev)
dz = torch.randn(B, H, W, 64, device=dev)
w = torch.randn(64, 64, 3, 3, device=dev) * 0.05
wf, wd = torch.empty(9, 64, 64, device=dev), torch.empty(9, 64, 64, device=dev)
call("cova_conv3x3_prep_weights", w, wf, wd)
out = torch.empty_like(x)
part = torch.empty(2 * query("cova_conv3x3_num_tiles", B, H, W), 2, 64, device=dev)   # enough for both Winograd geometries
flop3 = 2 * 64 * 64 * 9 * B * H * W
for stats in (False, True):
    t = timeit(lambda: call("cova_conv3x3_fwd", x, wf, None, out, part if stats else None, B, H, W))
    print("conv3x3 fwd (direct) stats=%d: %.3f ms  %.1f TF/s" % (stats, t, flop3 / t / 1e9))
uf, ud = torch.empty(16, 16, 4, 64, device=dev), torch.empty(16, 16, 4, 64, device=dev)
call("cova_conv3x3_prep_weights_wino", w, uf, ud)
for stats in (False, True):
    t = timeit(lambda: call("cova_conv3x3_wino", x, uf, None, None, None, None, None, out, part if stats else None, B, H, W))
    print("conv3x3 WINOGRAD fwd stats=%d: %.3f ms  %.1f TF/s (direct-equivalent)" % (stats, t, flop3 / t / 1e9))
for abl, what in ((0, "full"), (1, "no epilogue"), (2, "no refill stores"), (6, "no refill loads+stores"), (8, "no weight restage"),
                  (16, "no chunk barrier"), (32, "no input transform"), (7, "no epi+refill"), (15, "MFMA+LDS+transform+barriers"),
                  (63, "MFMA + B reads only")):
    query("cova_set_option", 5, abl)
    t = timeit(lambda: call("cova_conv3x3_wino", x, uf, None, None, None, None, None, out, None, B, H, W))
    print("WINO ablation %2d %-30s: %.3f ms  %.1f TF/s" % (abl, what, t, flop3 / t / 1e9))
query("cova_set_option", 5, 0)
for abl, what in ((0, "full"), (1, "no epilogue"), (2, "no LDS refill"), (4, "no prefetch loads"), (8, "no weight restage"),
                  (16, "no tap barrier"), (6, "no refill+prefetch"), (7, "no epi+refill+prefetch"), (15, "MFMA + LDS reads + barriers"),
                  (31, "MFMA + LDS reads only")):
    query("cova_set_option", 5, abl)
    t = timeit(lambda: call("cova_conv3x3_fwd", x, wf, None, out, None, B, H, W))
    print("ablation %2d %-28s: %.3f ms  %.1f TF/s" % (abl, what, t, flop3 / t / 1e9))
query("cova_set_option", 5, 0)
sc, sh = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.1
mean, invstd = torch.randn(64, device=dev) * 0.1, torch.rand(64, device=dev) + 0.5
t = timeit(lambda: call("cova_conv3x3_dgrad_bnbwd", dz, wd, x, x, dz, mean, invstd, out, part, B, H, W))
print("conv3x3 dgrad_bnbwd (+addend, act): %.3f ms  %.1f TF/s" % (t, flop3 / t / 1e9))
ws = torch.empty(query("cova_conv3x3_wgrad_workspace_floats", B, H, W), device=dev)

dw = torch.empty(64, 64, 3, 3, device=dev)
for abl, what in ((0, "full"), (2, "no LDS refill"), (4, "no prefetch loads"), (6, "no refill+prefetch")):
    query("cova_set_option", 5, abl)
    t = timeit(lambda: call("cova_conv3x3_wgrad", x, dz, dw, ws, B, H, W))
    print("wgrad ablation %2d %-22s: %.3f ms  %.1f TF/s" % (abl, what, t, flop3 / t / 1e9))
query("cova_set_option", 5, 0)
t = timeit(lambda: call("cova_conv3x3_wgrad_wino", x, dz, dw, ws, B, H, W))
print("conv3x3 WINOGRAD wgrad (+reduce): %.3f ms  %.1f TF/s (direct-equivalent)" % (t, flop3 / t / 1e9))
for variant in (2,):
    t = timeit(lambda: call("cova_conv3x3_wgrad", x, dz, dw, ws, B, H, W))
    print("conv3x3 wgrad v%d (+reduce): %.3f ms  %.1f TF/s" % (variant, t, flop3 / t / 1e9))
img = torch.rand(B, 3, 1280, 1280, device=dev)
w1 = torch.randn(64, 3, 7, 7, device=dev) * 0.1
wk = torch.empty(154, 64, device=dev)
call("cova_conv1_prep_weights", w1, wk)
y1 = torch.empty(B, 640, 640, 64, device=dev)
part1 = torch.empty(query("cova_conv1_num_tiles", B, 1280, 1280), 2, 64, device=dev)
flop1 = 2 * 64 * 147 * B * 640 * 640
for abl, what in ((0, "full"), (1, "no epilogue"), (2, "no LDS refill"), (4, "no prefetch loads"), (7, "MFMA + LDS reads")):
    query("cova_set_option", 5, abl)
    t = timeit(lambda: call("cova_conv1_fwd", img, wk, y1, part1, B, 1280, 1280))
    print("conv1 fwd ablation %2d %-20s: %.3f ms  %.1f TF/s" % (abl, what, t, flop1 / t / 1e9))
query("cova_set_option", 5, 0)
for variant in (2,):
    t = timeit(lambda: call("cova_conv1_fwd", img, wk, y1, part1, B, 1280, 1280))
    print("conv1 fwd v%d: %.3f ms  %.1f TF/s" % (variant, t, flop1 / t / 1e9))
ws1 = torch.empty(query("cova_conv1_wgrad_workspace_floats", B, 1280, 1280), device=dev)
dw1 = torch.empty(64, 3, 7, 7, device=dev)
for variant in (2,):
    t = timeit(lambda: call("cova_conv1_wgrad", img, y1, dw1, ws1, B, 1280, 1280))
    print("conv1 wgrad v%d (+reduce): %.3f ms  %.1f TF/s" % (variant, t, flop1 / t / 1e9))
