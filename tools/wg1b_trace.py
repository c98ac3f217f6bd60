"""Phase timeline of the bf16-split conv1 weight-gradient kernel (trace build: C1B_EXTRA=-DC1B_TRACE tools/c1b_abl_build.sh 0):
the eight waves of block 0, tiles 8..27, shader cycles."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["COVA_HIP_LIB"] = os.path.join(ROOT, "tools", "lib", "libcova_c1babl_0.so")
sys.path.insert(0, ROOT)
import torch
import cova_amd  # noqa
from cova_web_object_detection_amd import _lib
from cova_web_object_detection_amd._lib import call, query
dev = "cuda:0"
sys.argv.append("--time-only")
g = torch.Generator(device=dev).manual_seed(7)
B, H, W = 16, 1280, 1280
x = torch.rand(B, 3, H, W, device=dev, generator=g)
H1, W1, H2, W2 = 640, 640, 320, 320
y = torch.randn(B, H1, W1, 64, device=dev, generator=g)
scale = torch.rand(64, device=dev, generator=g) - 0.3
shift = torch.randn(64, device=dev, generator=g) * 0.2
p1 = torch.empty(B, H2, W2, 64, device=dev)
idx = torch.empty(B, H2, W2, 64, device=dev, dtype=torch.uint8)
call("cova_bn_relu_maxpool_fwd", y, scale, shift, p1, idx, None, B, H1, W1)
dp = torch.randn(B, H2, W2, 64, device=dev, generator=g) * (p1 > 0)
abc = torch.randn(3, 64, device=dev, generator=g) * 0.3
ws = torch.empty(query("cova_conv1_wgrad_workspace_floats", B, H, W), device=dev)
dw = torch.zeros(64, 3, 7, 7, device=dev)
call("cova_set_option", 8, 1)          # the phase-structured form (the default is the role-split one: tools/wg1r_trace.py)
for _ in range(3):
    call("cova_conv1_wgrad_poolbwd", x, y, dp, idx, abc, dw, ws, B, H, W)
torch.cuda.synchronize()
n = 8 * 20 * 8
buf = (ctypes.c_ulonglong * n)()
assert _lib.lib().cdll.cova_wg1b_trace_read(buf) == 0
names = ["barrier", "windows", "finish_dy", "barrier", "patch+barrier"]
t0 = buf[5]
for wave in range(8):
    print("wave", wave)
    for it in (3, 4, 5):
        st = [buf[(wave * 20 + it) * 8 + k] for k in range(6)]
        prev5 = buf[(wave * 20 + it - 1) * 8 + 5]
        print("   tile %2d  loop start %7d  mfma loop %6d  " % (it, prev5 - t0, st[0] - prev5) + "  ".join("%s %5d" % (names[k], st[k + 1] - st[k]) for k in range(5)))
