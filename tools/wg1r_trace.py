"""Per-tile timeline of the role-split conv1 weight-gradient kernel (trace build: C1B_EXTRA=-DC1B_TRACE tools/c1b_abl_build.sh 0):
the eight waves of block 0 (0-3 multiply, 4-7 stage), tiles 40..59, shader cycles."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["COVA_HIP_LIB"] = os.path.join(ROOT, "tools", "lib", "libcova_c1babl_0.so")
sys.path.insert(0, ROOT)
import torch
import cova_amd  # noqa
from cova_web_object_detection_amd import _lib
from cova_web_object_detection_amd._lib import call, query
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(7)
B, H, W = 16, 1280, 1280
x = torch.rand(B, 3, H, W, device=dev, generator=g)
y = torch.randn(B, 640, 640, 64, device=dev, generator=g)
scale = torch.rand(64, device=dev, generator=g) - 0.3
shift = torch.randn(64, device=dev, generator=g) * 0.2
p1 = torch.empty(B, 320, 320, 64, device=dev)
idx = torch.empty(B, 320, 320, 64, device=dev, dtype=torch.uint8)
call("cova_bn_relu_maxpool_fwd", y, scale, shift, p1, idx, None, B, 640, 640)
dp = torch.randn(B, 320, 320, 64, device=dev, generator=g) * (p1 > 0)
abc = torch.randn(3, 64, device=dev, generator=g) * 0.3
ws = torch.empty(query("cova_conv1_wgrad_workspace_floats", B, H, W), device=dev)
dw = torch.zeros(64, 3, 7, 7, device=dev)
for _ in range(3):
    call("cova_conv1_wgrad_poolbwd", x, y, dp, idx, abc, dw, ws, B, H, W)
torch.cuda.synchronize()
n = 8 * 20 * 8
buf = (ctypes.c_ulonglong * n)()
assert _lib.lib().cdll.cova_wg1b_trace_read(buf) == 0
t0 = buf[(0 * 20 + 3) * 8 + 0]
for wave in range(8):
    print("wave %d (%s)" % (wave, "multiplies" if wave < 4 else "stages: column group %d, row pair %d" % ((wave - 4) >> 1, (wave - 4) & 1)))
    for it in (3, 4, 5, 6):
        st = [buf[(wave * 20 + it) * 8 + k] for k in range(4)]
        if wave < 4:
            print("   tile %2d  start %7d  MFMA loop %5d  barrier wait %5d" % (it, st[0] - t0, st[1] - st[0], st[2] - st[1]))
        else:
            print("   tile %2d  start %7d  dy1 %5d  loads issued %5d  barrier wait %5d" % (it, st[0] - t0, st[3] - st[0], st[1] - st[3], st[2] - st[1]))
