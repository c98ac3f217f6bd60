"""Time of the F(4x4,3x3) variants the train step launches, split main loop (16 x 320 x 320 x 64); --f32 adds the f32 loop."""
import os as _os; _os.environ.setdefault("COVA_ALLOW_OPTION_CHANGES", "1")
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cova_amd  # noqa: F401
from cova_web_object_detection_amd._lib import call, query
dev = "cuda:0"
NU = query("cova_conv3x3_wino4_u_floats")


def timeit(fn, n=20):
    for _ in range(40):          # (the first case timed in a process otherwise runs on a chip that has not clocked up yet)
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


B, H, W = 16, 320, 320
x, z, add, act = (torch.randn(B, H, W, 64, device=dev) for _ in range(4))
w = torch.randn(64, 64, 3, 3, device=dev) * 0.05
if "--zeros" in sys.argv:          # data-dependent power: all-zero operands (guide: DVFS give-back)
    x.zero_(); w.zero_()
uf, ud = torch.empty(NU, device=dev), torch.empty(NU, device=dev)
call("cova_conv3x3_wino4_prep", w, uf, ud)
out = torch.empty_like(x)
part = torch.empty(query("cova_conv3x3_wino4_num_partials", B, H, W), 2, 64, device=dev)
abc = torch.randn(3, 64, device=dev)
mean, invstd = torch.randn(64, device=dev) * 0.1, torch.rand(64, device=dev) + 0.5
N = None
full = lambda *a: call("cova_conv3x3_wino4_full", *a, out, part, B, H, W)
cases = [
    ("<1,0,0,0>", lambda: full(x, N, N, 0, uf, N, N, N, N, N, N, N)),
    ("<1,1,0,0>", lambda: full(x, N, abc, 1, uf, N, N, N, N, N, N, N)),
    ("<1,0,0,1>", lambda: full(x, N, N, 0, ud, N, N, abc[0], abc[2], z, mean, invstd)),
    ("<1,0,1,1>", lambda: full(x, N, N, 0, ud, add, N, abc[0], abc[2], z, mean, invstd)),
    ("<1,0,1,2>", lambda: full(x, N, N, 0, ud, add, act, N, N, z, mean, invstd)),
]
for f32 in ((1, 0) if "--f32" in sys.argv else (0,)):
    query("cova_set_option", 9, f32)
    print(("f32 loop  " if f32 else "split loop") + " ms: " + "  ".join("%s %.3f" % (n, timeit(fn)) for n, fn in cases))
query("cova_set_option", 9, 0)
