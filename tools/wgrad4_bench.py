"""Weight gradient of the 3x3 convolution on the benchmark map (16 x 320 x 320 x 64): F(4x4,3x3) (csrc/conv_wgrad4.hip),
per operand-prologue variant, partial-sum launches only + the finish launch (40 warm-up launches: the first case a process times runs slow).
COVA_HIP_LIB selects an ablation build (tools/wg4_abl_build.sh)."""
import os as _os; _os.environ.setdefault("COVA_ALLOW_OPTION_CHANGES", "1")
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cova_amd  # noqa: F401
from cova_web_object_detection_amd._lib import call, query
dev = "cuda:0"
B, H, W = 16, 320, 320
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(B, H, W, 64, device=dev, generator=g).clamp_min(0)
dy = torch.randn(B, H, W, 64, device=dev, generator=g) * (torch.rand(B, H, W, 64, device=dev, generator=g) > 0.5)
z = torch.randn(B, H, W, 64, device=dev, generator=g)
abc_a, abc_d = torch.randn(3, 64, device=dev, generator=g), torch.randn(3, 64, device=dev, generator=g)
ws4 = torch.empty(query("cova_conv3x3_wgrad4_workspace_floats", B, H, W), device=dev)
dw = torch.empty(64, 64, 3, 3, device=dev)

def t(fn, n=20):
    for _ in range(40): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

only = [a for a in sys.argv[1:] if not a.startswith("--")]          # e.g. "12" = PROA + two-tensor gradient prologue
for name, pa, pd in (("00", False, 0), ("10", True, 0), ("02", False, 2), ("12", True, 2)):
    if only and name not in only:
        continue
    args = (x, abc_a if pa else None, 1, dy, z if pd == 2 else None, abc_d if pd else None)
    dzo = torch.empty_like(dy) if (pd and "--emit" in sys.argv) else None
    t4 = t(lambda: call("cova_conv3x3_wgrad4_partial", *args, dzo, ws4, B, H, W))
    print("prologue act=%d grad=%d   F(4x4) %.3f ms" % (pa, pd, t4), flush=True)
f4 = t(lambda: call("cova_conv3x3_wgrad4_finish", ws4, dw, ws4, dw, ws4, dw, ws4, dw, B, H, W))
print("finish (4 convolutions)   F(4x4) %.3f ms" % f4)
