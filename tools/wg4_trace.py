"""Phase timeline of the F(4x4) weight-gradient kernel (trace build: WG4_EXTRA=-DWG4_TRACE tools/wg4_abl_build.sh 0): s_memtime
stamps of wave 0 (input role) and wave 4 (gradient role) of block 0 over iterations 100..103, in shader cycles relative
to the iteration's start.  slots: 0 iteration start | 1 copies landed (wait done) | 2 first stage done | 3 copies + touch
issued | 4 second stage + operand writes done | 5 MFMA phase done | 6 barrier passed"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["COVA_HIP_LIB"] = os.path.join(ROOT, "tools", "lib", "libcova_wg4abl_0.so")
sys.path.insert(0, ROOT)
import torch
import cova_amd  # noqa
from cova_web_object_detection_amd import _lib
from cova_web_object_detection_amd._lib import call, query
dev = "cuda:0"
B, H, W = 16, 320, 320
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(B, H, W, 64, device=dev, generator=g).clamp_min(0)
dy = torch.randn(B, H, W, 64, device=dev, generator=g)
z = torch.randn(B, H, W, 64, device=dev, generator=g)
abc_a, abc_d = torch.randn(3, 64, device=dev, generator=g), torch.randn(3, 64, device=dev, generator=g)
ws4 = torch.empty(query("cova_conv3x3_wgrad4_workspace_floats", B, H, W), device=dev)
for _ in range(3):
    call("cova_conv3x3_wgrad4_partial", x, abc_a, 1, dy, z, abc_d, None, ws4, B, H, W)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
rc = _lib.lib().cdll.cova_wg4_trace_read(buf)
assert rc == 0, rc
names = ["start", "landed", "stage1", "issued", "stage2", "mfma", "barrier"]
for role in (0, 1):
    print("wave %d (%s role)" % (4 * role, "gradient" if role else "input"))
    for it in range(4):
        s = [buf[(role * 4 + it) * 8 + k] for k in range(7)]
        order = sorted(range(7), key=lambda k: s[k])
        print("  iter %d: " % (100 + it) + "  ".join("%s +%d" % (names[k], s[k] - s[0]) for k in order),
              " | next start +%d" % ((buf[(role * 4 + it + 1) * 8] - s[0]) if it < 3 else 0))
