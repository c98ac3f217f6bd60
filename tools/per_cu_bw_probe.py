"""Is there a per-CU cap on HBM streaming?  cova_probe_lane_pattern with 1 block (8 waves, 8 float4 in
flight per lane) per CU on 8 ... 256 CUs: if the per-CU rate stays ~23 KB/us when few CUs stream, the
cap is per CU; if it rises, the chip-wide 5.9 TB/s / 256 is just the fair share."""
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


def timeit(fn, n=5):
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


for loads_only in (1, 0):
    for blocks in (8, 16, 32, 64, 128, 256, 512):
        t = timeit(lambda: _lib.call("cova_probe_lane_pattern", x, y, npix, 1, loads_only, blocks, 100 * 1024))
        gb = x.numel() * 4 * (1 if loads_only else 2) / 1e9
        print("%-5s %4d blocks (1 per CU): %.3f ms  %.2f TB/s  = %.1f KB/us per block"
              % ("load" if loads_only else "copy", blocks, t, gb / t, gb * 1e6 / t / 1e3 / min(blocks, 256) ))
