"""Phase timeline of the bf16-split conv1 forward kernel (trace build: C1B_EXTRA=-DC1B_TRACE tools/c1b_abl_build.sh 0):
the four waves of blocks 0 and grid/2, tiles 8..27 of their streams, shader cycles relative to wave 0's first stamp."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["COVA_HIP_LIB"] = os.path.join(ROOT, "tools", "lib", "libcova_c1babl_0.so")
sys.path.insert(0, ROOT)
import torch
import cova_amd  # noqa
from cova_web_object_detection_amd import _lib
from cova_web_object_detection_amd._lib import call, query
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(3)
w = torch.randn(64, 3, 7, 7, device=dev, generator=g) * 0.05
img = torch.randn(16, 3, 1280, 1280, device=dev, generator=g)
out = torch.empty(16, 640, 640, 64, device=dev)
part = torch.zeros(query("cova_conv1_num_partials", 16, 1280, 1280), 2, 64, device=dev)
for _ in range(3):
    call("cova_conv1_fwd_tail", img, w, out, part, 16, 1280, 1280, None)
torch.cuda.synchronize()
n = 2 * 4 * 20 * 8
buf = (ctypes.c_ulonglong * n)()
assert _lib.lib().cdll.cova_c1b_trace_read(buf) == 0
t0 = buf[0]
names = ["kloop", "refill", "barrier", "stores"]
for blk in (0, 1):
    for wave in range(4):
        base = (blk * 4 + wave) * 20 * 8
        hw = buf[base + 7]
        print("block %d wave %d  hw_id 0x%x (cu %d simd %d se %d)" % (blk, wave, hw, (hw >> 8) & 15, (hw >> 4) & 3, (hw >> 13) & 7))
        for it in range(0, 20, 1 if wave == 0 else 5):
            st = [buf[base + it * 8 + k] for k in range(5)]
            print("   tile %2d  start %8d  " % (it, st[0] - t0) + "  ".join("%s %5d" % (names[k], st[k + 1] - st[k]) for k in range(4)))
