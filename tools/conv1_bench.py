"""conv1 (7x7 stride 2, 3 -> 64) forward on the benchmark page batch (16 x 3 x 1280 x 1280): the bf16-split kernel
(csrc/conv.hip, conv1_7x7_bf3_kernel) against the f32-MFMA kernel (cova_set_option(7, 1)), time per launch and error of
both against a float64 convolution on a smaller batch."""
import os as _os; _os.environ.setdefault("COVA_ALLOW_OPTION_CHANGES", "1")
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

# ---- weight gradient (the step's form: BatchNorm + ReLU + MaxPool backward folded into the operand)
def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()

def wgrad_case(B, H, W, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.rand(B, 3, H, W, device=dev, generator=g)
    H1, W1 = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    H2, W2 = (H1 - 1) // 2 + 1, (W1 - 1) // 2 + 1
    y = torch.randn(B, H1, W1, 64, device=dev, generator=g)
    scale = torch.rand(64, device=dev, generator=g) - 0.3
    shift = torch.randn(64, device=dev, generator=g) * 0.2
    p1 = torch.empty(B, H2, W2, 64, device=dev)
    idx = torch.empty(B, H2, W2, 64, device=dev, dtype=torch.uint8)
    call("cova_bn_relu_maxpool_fwd", y, scale, shift, p1, idx, None, B, H1, W1)
    dp = torch.randn(B, H2, W2, 64, device=dev, generator=g) * (p1 > 0)
    abc = torch.randn(3, 64, device=dev, generator=g) * 0.3
    return x, y, dp, idx, abc

def wgrad_run(case, f32, B, H, W):
    x, y, dp, idx, abc = case
    call("cova_set_option", 7, f32)
    ws = torch.empty(query("cova_conv1_wgrad_workspace_floats", B, H, W), device=dev)
    dw = torch.zeros(64, 3, 7, 7, device=dev)
    call("cova_conv1_wgrad_poolbwd", x, y, dp, idx, abc, dw, ws, B, H, W)
    call("cova_set_option", 7, 0)
    return dw

if "--time-only" not in sys.argv:
    B, H, W = 2, 300, 330
    case = wgrad_case(B, H, W, 5)
    x, y, dp, idx, abc = case
    # float64 reference: dy1 = A * route(dp) + B * y + C, then the convolution's weight gradient
    H1, W1 = y.shape[1], y.shape[2]
    yd = y.double().cpu().permute(0, 3, 1, 2)
    idxc, dpc = idx.cpu().long().permute(0, 3, 1, 2), dp.double().cpu().permute(0, 3, 1, 2)
    gsc = torch.zeros(B, 64, H1 + 2, W1 + 2, dtype=torch.float64)
    H2, W2 = dp.shape[1], dp.shape[2]
    ph, pw = torch.meshgrid(torch.arange(H2), torch.arange(W2), indexing="ij")
    ry = (2 * ph)[None, None] + idxc // 3          # padded row of the arg-max (pad 1)
    rx = (2 * pw)[None, None] + idxc % 3
    gsc.view(B, 64, -1).scatter_add_(2, (ry * (W1 + 2) + rx).view(B, 64, -1), dpc.reshape(B, 64, -1))
    route = gsc[:, :, 1:H1 + 1, 1:W1 + 1]
    A, Bc, C = (abc[i].double().cpu().view(1, 64, 1, 1) for i in range(3))
    dy1 = A * route + Bc * yd + C
    wr = torch.zeros(64, 3, 7, 7, dtype=torch.float64, requires_grad=True)
    (torch.nn.functional.conv2d(x.double().cpu(), wr, stride=2, padding=3) * dy1).sum().backward()
    for f32 in (1, 0):
        dw = wgrad_run(case, f32, B, H, W)
        e = (dw.double().cpu() - wr.grad).abs().max().item() / wr.grad.abs().max().item()
        print("wgrad (%d, %d, %d) %s  max err / max |dw| = %.3e" % (B, H, W, "f32 mfma  " if f32 else "bf16 split", e), flush=True)
B, H, W = 16, 1280, 1280
case = wgrad_case(B, H, W, 7)
x, y, dp, idx, abc = case
ws = torch.empty(query("cova_conv1_wgrad_workspace_floats", B, H, W), device=dev)
dw = torch.zeros(64, 3, 7, 7, device=dev)
for f32 in (1, 0, 1, 0):
    call("cova_set_option", 7, f32)
    ms = t(lambda: call("cova_conv1_wgrad_poolbwd", x, y, dp, idx, abc, dw, ws, B, H, W))
    print("wgrad 16 x 1280 x 1280: %s %.3f ms per call (partials + reduce)"
          % ("f32 mfma              " if f32 else "bf16 split, role split", ms), flush=True)
call("cova_set_option", 7, 0)
# the same without the folded pool backward (dy given): isolates the pooled-gradient window loads
dyt = torch.randn(B, 640, 640, 64, device=dev)
for f32 in (1, 0):
    call("cova_set_option", 7, f32)
    ms = t(lambda: call("cova_conv1_wgrad", x, dyt, dw, ws, B, H, W))
    print("wgrad (dy given) 16 x 1280 x 1280: %s %.3f ms per call" % ("f32 mfma  " if f32 else "bf16 split", ms), flush=True)
call("cova_set_option", 7, 0)
