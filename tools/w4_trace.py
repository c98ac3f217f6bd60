"""Phase timeline of the F(4x4) forward / data-gradient kernel (trace build: W4_EXTRA=-DW4_TRACE tools/w4_abl_build.sh 0):
waves 0 / 4 of block 0, chunks 32..47 (tiles 4 and 5 of its stream) and their two tile epilogues, shader cycles.
argv[1]: variant plain | pro1 | bn (one-tensor data gradient with BatchNorm epilogue) | bnadd"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["COVA_HIP_LIB"] = os.path.join(ROOT, "tools", "lib", "libcova_w4abl_0.so")
sys.path.insert(0, ROOT)
import torch
import cova_amd  # noqa
from cova_web_object_detection_amd import _lib
from cova_web_object_detection_amd._lib import call, query
dev = "cuda:0"
var = sys.argv[1] if len(sys.argv) > 1 else "plain"
B, H, W = 16, 320, 320
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(B, H, W, 64, device=dev, generator=g)
z = torch.randn(B, H, W, 64, device=dev, generator=g)
add = torch.randn(B, H, W, 64, device=dev, generator=g)
w = torch.randn(64, 64, 3, 3, device=dev, generator=g) * 0.05
uf, ud = torch.empty(query("cova_conv3x3_wino4_u_floats"), device=dev), torch.empty(query("cova_conv3x3_wino4_u_floats"), device=dev)
call("cova_conv3x3_wino4_prep", w, uf, ud)
out = torch.empty_like(x)
part = torch.empty(query("cova_conv3x3_wino4_num_partials", B, H, W), 2, 64, device=dev)
abc = torch.randn(3, 64, device=dev, generator=g)
v4 = [torch.randn(64, device=dev, generator=g) for _ in range(4)]
def run():
    if var == "plain":
        call("cova_conv3x3_wino4", x, uf, out, part, B, H, W)
    elif var == "pro1":
        call("cova_conv3x3_wino4_pro", x, abc, 1, uf, out, part, B, H, W)
    else:
        call("cova_conv3x3_wino4_full", x, None, None, 0, ud, add if var == "bnadd" else None, None, v4[0], v4[1], z, v4[2], v4[3],
             out, part, B, H, W)
for _ in range(3):
    run()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 320)()
assert _lib.lib().cdll.cova_w4_trace_read(buf) == 0
names = ["start", "reads", "mfma16", "stage", "mfma36", "planes", "barrier"]
for role in (0, 1):
    print("wave %d (position half %d)" % (4 * role, role))
    base = buf[(role * 20) * 8]
    for r in range(16):
        s = [buf[(role * 20 + r) * 8 + k] for k in range(7)]
        nxt = buf[(role * 20 + r + 1) * 8] if r < 15 else 0
        print("  chunk %2d @%6d: " % (32 + r, s[0] - base) + " ".join("%s +%d" % (names[k], s[k] - s[0]) for k in range(1, 7)),
              "| next +%d" % (nxt - s[0]) if nxt else "")
    for e in range(2):
        s = [buf[(role * 20 + 16 + 2 * e) * 8 + k] for k in range(4)]
        print("  epilogue of tile %d @%6d: operands requested +0, transform done +%d, exchange done +%d, stores issued +%d"
              % (4 + e, s[0] - base, s[1] - s[0], s[2] - s[0], s[3] - s[0]))
