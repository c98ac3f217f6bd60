"""conv1 (7x7 stride 2, 3 -> 64) forward on the benchmark page batch (16 x 3 x 1280 x 1280): the bf16-split kernel
(csrc/conv.hip, conv1_7x7_bf3_kernel) against the f32-MFMA kernel (cova_set_option(7, 1)), time per launch and error of
both against a float64 convolution on a smaller batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cova_amd  # noqa: F401
from cova_web_object_detection_amd._lib import call, query
dev = "cuda:0"

def run(img, w, f32):
    call("cova_set_option", 7, 1 if f32 else 0)
    B, _, H, W = img.shape
    H1, W1 = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    out = torch.empty(B, H1, W1, 64, device=dev)
    part = torch.zeros(query("cova_conv1_num_partials", B, H, W), 2, 64, device=dev)
    call("cova_conv1_fwd_tail", img, w, out, part, B, H, W, None)
    call("cova_set_option", 7, 0)
    return out, part

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

g = torch.Generator(device=dev).manual_seed(3)
w = torch.randn(64, 3, 7, 7, device=dev, generator=g) * 0.05
for shape in (() if '--time-only' in sys.argv else ((2, 3, 250, 333), (1, 3, 1280, 1280))):
    img = torch.randn(*shape, device=dev, generator=g)
    ref = torch.nn.functional.conv2d(img.double().cpu(), w.double().cpu(), stride=2, padding=3).permute(0, 2, 3, 1)
    for f32 in (1, 0):
        out, part = run(img, w, f32)
        e = (out.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
        s = (part[:, 0].double().sum(0).cpu() - ref.sum((0, 1, 2))).abs().max().item() / ref.sum((0, 1, 2)).abs().max().item()
        print("%s %s  max err / max |y| = %.3e   channel-sum err %.3e" % (shape, "f32 mfma  " if f32 else "bf16 split", e, s), flush=True)
img = torch.randn(16, 3, 1280, 1280, device=dev, generator=g)
out = torch.empty(16, 640, 640, 64, device=dev)
for f32 in (1, 0, 1, 0):
    call("cova_set_option", 7, f32)
    part = torch.zeros(query("cova_conv1_num_partials", 16, 1280, 1280), 2, 64, device=dev)
    ms = t(lambda: call("cova_conv1_fwd_tail", img, w, out, part, 16, 1280, 1280, None))
    print("16 x 1280 x 1280: %s %.3f ms per launch" % ("f32 mfma  " if f32 else "bf16 split", ms), flush=True)
call("cova_set_option", 7, 0)
