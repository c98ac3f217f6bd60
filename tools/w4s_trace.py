"""Phase timeline of the split F(4x4,3x3) main loop (trace build: tools/w4s_var_build.sh "trace:-DW4S_TRACE"): waves 0 / 4 of
block 0 over the six steps + epilogue of its fourth tile, in s_memrealtime ticks (100 MHz: 10 ns).
argv[1]: plain | pro1 | bn | bnadd"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["COVA_HIP_LIB"] = os.path.join(ROOT, "tools", "lib", "libcova_w4svar_trace.so")
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
NU = query("cova_conv3x3_wino4_u_floats")
uf, ud = torch.empty(NU, device=dev), torch.empty(NU, device=dev)
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
buf = (ctypes.c_ulonglong * (2 * 7 * 16))()
assert _lib.lib().cdll.cova_w4s_trace_read(buf) == 0
for role in (0, 1):
    print("wave %d (position half %d), ticks of 10 ns" % (4 * role, role))
    base = buf[role * 7 * 16]
    for t in range(6):
        s = [buf[(role * 7 + t) * 16 + k] for k in range(16)]
        row = "  step %d @%5d: copies +%d, first reads/row +%d |" % (t, s[0] - base, s[1] - s[0], s[2] - s[0])
        prev = s[2]
        for j in range(6):
            row += " M%d %d T %d" % (j, s[3 + 2 * j] - prev, s[4 + 2 * j] - s[3 + 2 * j])
            prev = s[4 + 2 * j]
        row += " | barrier +%d, step %d" % (s[15] - prev, s[15] - s[0])
        print(row)
    e = [buf[(role * 7 + 6) * 16 + k] for k in range(2)]
    print("  epilogue %d ticks; tile %d ticks" % (e[1] - e[0], e[1] - base))
