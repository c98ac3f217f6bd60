"""How much does the lane -> address mapping of a float4 access cost on MI355X?  NHWC-64 copy /
load-only sweep with the Winograd epilogue's mapping vs contiguous lanes (cova_probe_lane_pattern)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cova_amd  # noqa
from cova_web_object_detection_amd import _lib
import probe_lib  # noqa: E402  (tools/probe_lib.py: builds + registers libcova_probe.so)
probe_lib.load()
dev = "cuda:0"
npix = 16 * 320 * 320
x = torch.randn(npix, 64, device=dev)
y = torch.empty_like(x)


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


names = {0: "epilogue mapping (16 B per lane, lanes 512 B apart)", 1: "contiguous (16 lanes = 256 B pixel)", 2: "4 lanes = 64 B segment"}
for _ in range(2):
    for lds, blocks, what in ((0, 2048, "full occupancy"), (100 * 1024, 256, "1 block/CU (2 waves/SIMD)")):
        for loads_only in (1, 0):
            for mode in (0, 1, 2):
                t = timeit(lambda: _lib.call("cova_probe_lane_pattern", x, y, npix, mode, loads_only, blocks, lds))
                gb = x.numel() * 4 * (1 if loads_only else 2) / 1e9
                print("%-26s %-10s %-52s %.3f ms  %.2f TB/s" % (what, "load" if loads_only else "copy", names[mode], t, gb / t))
