"""Load bandwidth of the 1x1-convolution operand pattern on a [R,256] fp32 map: row per lane (csrc/conv1x1.hip) vs 8
adjacent lanes per 128-byte line, at 2 waves per SIMD (as the kernels run) and at full occupancy."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cova_amd  # noqa
from cova_web_object_detection_amd import _lib
import probe_lib  # noqa: E402
probe_lib.load()
dev = "cuda:0"
npix = 32 * 320 * 320 * 4            # in 64-float units: a [3,276,800 x 256] map
x = torch.randn(npix, 64, device=dev)
y = torch.empty(4096, device=dev)
yo = torch.empty_like(x)


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for lds, blocks, what in ((100 * 1024, 256, "1 block/CU (2 waves/SIMD)"), (0, 2048, "full occupancy")):
    for mode, name in ((3, "row per lane"), (4, "8 lanes per 128-byte line")):
        t = timeit(lambda: _lib.call("cova_probe_lane_pattern", x, y, npix, mode, 1, blocks, lds))
        print("%-28s %-28s %.3f ms  %.2f TB/s" % (what, name, t, x.numel() * 4 / 1e9 / t))
    for mode, name in ((5, "store: channel per lane, 4 B"), (6, "store: float4, contiguous")):
        t = timeit(lambda: _lib.call("cova_probe_lane_pattern", x, yo, npix, mode, 1, blocks, lds))
        print("%-28s %-28s %.3f ms  %.2f TB/s" % (what, name, t, x.numel() * 4 / 1e9 / t))
