"""F(4x4,3x3) against F(2x2,3x3) on the benchmark map (16 x 320 x 320 x 64): plain forward launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cova_amd  # noqa: F401
from cova_web_object_detection_amd._lib import call, query
dev = "cuda:0"
B, H, W = 16, 320, 320
x = torch.randn(B, H, W, 64, device=dev).clamp_min(0)
w = torch.randn(64, 64, 3, 3, device=dev) * 0.05
uf4, ud4 = (torch.empty(query("cova_conv3x3_wino4_u_floats"), device=dev) for _ in range(2))
call("cova_conv3x3_wino4_prep", w, uf4, ud4)
uf2, ud2 = torch.empty(16, 4, 64, 16, device=dev), torch.empty(16, 4, 64, 16, device=dev)
call("cova_conv3x3_prep_weights_wino", w, uf2, ud2)
o4, o2 = torch.empty_like(x), torch.empty_like(x)
p4 = torch.empty(query("cova_conv3x3_wino4_num_partials", B, H, W), 2, 64, device=dev)
p2 = torch.empty(query("cova_conv3x3_wino_num_partials", B, H, W), 2, 64, device=dev)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
f4 = lambda: call("cova_conv3x3_wino4", x, uf4, o4, p4, B, H, W)
f2 = lambda: call("cova_conv3x3_wino", x, uf2, None, None, None, None, None, o2, p2, B, H, W)
print("F(2x2,3x3) %.3f ms   F(4x4,3x3) %.3f ms" % (t(f2), t(f4)))
print("max diff between the two:", float((o4 - o2).abs().max()), "of", float(o2.abs().max()))
