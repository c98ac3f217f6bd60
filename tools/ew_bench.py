"""Streaming element-wise kernels at the bench size: GB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cova_amd  # noqa
from cova_web_object_detection_amd import _lib
call, query = _lib.call, _lib.query
dev = "cuda:0"
B, H, W = 16, 320, 320
R = B * H * W


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


z, x, out = (torch.randn(B, H, W, 64, device=dev) for _ in range(3))
sc, sh = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev) * 0.1
T = z.numel() * 4 / 1e9
t = timeit(lambda: call("cova_bn_act_fwd", z, 64, sc, sh, x, 64, out, 64, R, 64, 1))
print("bn_act_fwd +res  (3 maps): %.3f ms  %.2f TB/s" % (t, 3 * T / t))
t = timeit(lambda: call("cova_bn_act_fwd", z, 64, sc, sh, None, 0, out, 64, R, 64, 1))
print("bn_act_fwd       (2 maps): %.3f ms  %.2f TB/s" % (t, 2 * T / t))
y1 = torch.randn(B, 640, 640, 64, device=dev)
p1 = torch.empty(B, H, W, 64, device=dev)
idx = torch.empty(B, H, W, 64, device=dev, dtype=torch.uint8)
ymax = torch.empty_like(p1)
for variant in (0, 7, 10, 12, 13, 15, 16, 17, 0, 12, 16):      # cova_set_option(13, .): strip height / grid cap / row prefetch
    query("cova_set_option", 13, variant)
    t = timeit(lambda: call("cova_bn_relu_maxpool_fwd", y1, sc, sh, p1, idx, ymax, B, 640, 640))
    print("bn_relu_maxpool_fwd (+ymax) variant %d: %.3f ms  %.2f TB/s" % (variant, t, (4 * T + 2 * T + T / 4) / t))
query("cova_set_option", 13, 12)
bits = torch.empty(R, 2, device=dev, dtype=torch.int32)
for un in (1, 2, 4, 1, 2, 4):                  # cova_set_option(17, .): elements of a thread in flight
    query("cova_set_option", 17, un)
    t = timeit(lambda: call("cova_bn_act_fwd_bits", z, sc, sh, x, out, bits, R))
    print("bn_act_fwd_bits +res (3 maps) unroll %d: %.3f ms  %.2f TB/s" % (un, t, 3 * T / t))
query("cova_set_option", 17, 1)
# the head's BatchNorm1d launches at configs[1]: 128-slice form (0) against the float4 form (1)
N, Tc = 1440, 976
for Cc, drop in ((Tc, 1), (32, 0)):
    xg, dout = torch.randn(N, Cc, device=dev), torch.randn(N, Cc, device=dev)
    gam, bet = torch.rand(Cc, device=dev) + 0.5, torch.randn(Cc, device=dev)
    o3, d3, dz3 = (torch.empty(N, Cc, device=dev) for _ in range(3))
    mask = (torch.rand(N, Cc, device=dev) > 0.2).to(torch.uint8)
    stv = [torch.empty(Cc, device=dev) for _ in range(4)]
    dg3, db3, cs3 = (torch.empty(Cc, device=dev) for _ in range(3))
    for variant in (0, 1, 0, 1):
        query("cova_set_option", 14, variant)
        tf = timeit(lambda: call("cova_bn1d_fwd", xg, Cc, N, Cc, gam, bet, None, None, None, 0.1, 1e-5, 1, o3, Cc,
                                 d3 if drop else None, Cc, mask if drop else None, 0.2, 0, 1, *stv), 50)
        tb = timeit(lambda: call("cova_bn1d_bwd", dout, Cc, mask if drop else None, 0.2, o3, Cc, xg, Cc, stv[2], stv[3], stv[0],
                                 N, Cc, dg3, db3, dz3, Cc, cs3), 50)
        print("bn1d [%d x %d] drop %d form %d: fwd %.1f us  bwd %.1f us" % (N, Cc, drop, variant, tf * 1e3, tb * 1e3))
query("cova_set_option", 14, 1)
t = timeit(lambda: out.copy_(z))
print("torch copy (2 maps): %.3f ms  %.2f TB/s" % (t, 2 * T / t))
