"""Does a launch on the bf16 matrix pipe slow the matrix-bound launch behind it?  Times the plain F(4x4) forward launch
(HIP events around it) alone, behind the f32-MFMA conv1 forward and behind the bf16-split conv1 forward, back to back
on one stream, at the bench shapes."""
import os as _os; _os.environ.setdefault("COVA_ALLOW_OPTION_CHANGES", "1")
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cova_amd  # noqa
from cova_web_object_detection_amd._lib import call, query
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(1)
B = 16
img = torch.randn(B, 3, 1280, 1280, device=dev, generator=g)
w1 = torch.randn(64, 3, 7, 7, device=dev, generator=g) * 0.05
y1 = torch.empty(B, 640, 640, 64, device=dev)
x = torch.randn(B, 320, 320, 64, device=dev, generator=g)
w = torch.randn(64, 64, 3, 3, device=dev, generator=g) * 0.05
uf, ud = torch.empty(query("cova_conv3x3_wino4_u_floats"), device=dev), torch.empty(query("cova_conv3x3_wino4_u_floats"), device=dev)
call("cova_conv3x3_wino4_prep", w, uf, ud)
out = torch.empty_like(x)
part = torch.empty(query("cova_conv3x3_wino4_num_partials", B, 320, 320), 2, 64, device=dev)

def run(mode, n=40):
    if mode != "alone":
        call("cova_set_option", 7, 1 if mode == "f32" else 0)
        p1 = torch.zeros(query("cova_conv1_num_partials", B, 1280, 1280), 2, 64, device=dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for i in range(n + 5):
        if mode != "alone":
            call("cova_conv1_fwd_tail", img, w1, y1, p1, B, 1280, 1280, None)
        if i >= 5: ev[i - 5][0].record()
        call("cova_conv3x3_wino4", x, uf, out, part, B, 320, 320)
        if i >= 5: ev[i - 5][1].record()
    torch.cuda.synchronize()
    call("cova_set_option", 7, 0)
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2]

for rep in range(2):
    for mode in ("alone", "f32", "bf16"):
        print("F(4x4) forward launch %-6s %s: %.4f ms (median of 40)" % ("" if mode == "alone" else "behind", {"alone": "alone, back to back", "f32": "the f32-MFMA conv1", "bf16": "the bf16-split conv1"}[mode], run(mode)), flush=True)
