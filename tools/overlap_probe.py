"""Does f32 MFMA throughput survive concurrent HBM streaming on MI355X?  One launch, MFMA blocks and
streaming blocks co-resident on every CU (cova_probe_mix)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cova_amd  # noqa
from cova_web_object_detection_amd import _lib
import probe_lib  # noqa: E402  (tools/probe_lib.py: builds + registers libcova_probe.so)
probe_lib.load()
dev = "cuda:0"
scratch = torch.zeros(16, device=dev)
buf = torch.randn(1 << 28, device=dev)      # 1 GiB
n4 = buf.numel() // 4


def run(mb, sb, iters, passes):
    fn = lambda: _lib.call("cova_probe_mix", scratch, buf, n4, mb, sb, iters, passes)
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


ITERS, PASSES = 8000, 32
for sb in (256, 512, 1024):
    tm = run(256, 0, ITERS, PASSES)
    ts = run(0, sb, ITERS, PASSES)
    tb = run(256, sb, ITERS, PASSES)
    print("stream blocks %4d: mfma alone %.2f ms (%.1f TF/s) | stream alone %.2f ms (%.2f TB/s) | mixed %.2f ms | sum %.2f max %.2f"
          % (sb, tm, 256 * 8 * ITERS * 16 * 4096 / tm / 1e9, ts, PASSES * buf.numel() * 4 / ts / 1e9, tb, tm + ts, max(tm, ts)))


def run2(lpi, iters):
    fn = lambda: _lib.call("cova_probe_mfma_load", scratch, buf, n4, 256, iters, lpi)
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


for lpi in (0, 1, 2):
    t = run2(lpi, ITERS)
    gb = 256 * 512 * 16 * ITERS * lpi / 1e9
    print("same-wave: %d float4 loads per 16 MFMAs: %.2f ms  %.1f TF/s  %.2f TB/s" % (lpi, t, 256 * 8 * ITERS * 16 * 4096 / t / 1e9, gb / t))


def run3(vpm, mfma, iters=8000):
    fn = lambda: _lib.call("cova_probe_mfma_valu", scratch, 256, iters, vpm, mfma)
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


for vpm in (0, 2, 4, 6):
    tm = run3(vpm, 1)
    tv = run3(vpm, 0) if vpm else 0.0
    print("MFMA 16x16x4 + %d FMAs each: %.2f ms (%.1f TF/s MFMA) ; the FMAs alone %.2f ms" % (vpm, tm, 256 * 8 * 8000 * 16 * 2048 / tm / 1e9, tv))
