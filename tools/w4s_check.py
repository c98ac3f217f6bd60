"""Split (bf16-piece) main loop of the F(4x4,3x3) kernels against the f32 main loop: values, statistics, error against
fp64, and time per variant on the benchmark map (16 x 320 x 320 x 64).  cova_set_option(9, 1) selects the f32 loop."""
import os as _os; _os.environ.setdefault("COVA_ALLOW_OPTION_CHANGES", "1")
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import cova_amd  # noqa: F401
from cova_web_object_detection_amd._lib import call, query
dev = "cuda:0"
nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(dev)
nchw = lambda t: t.permute(0, 3, 1, 2).cpu()
NU = query("cova_conv3x3_wino4_u_floats")


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def check(B, H, W, cap, relu_in=False):
    g = torch.Generator().manual_seed(7 * H + W + B)
    rnd = lambda *s: torch.randn(*s, generator=g)
    xc = rnd(B, 64, H, W)
    if relu_in:
        xc = xc.clamp_min(0)
    x, add, z = nhwc(xc), nhwc(rnd(B, 64, H, W)), nhwc(rnd(B, 64, H, W))
    w = rnd(64, 64, 3, 3) * 0.05
    abc = rnd(3, 64).to(dev)
    uf, ud = torch.empty(NU, device=dev), torch.empty(NU, device=dev)
    call("cova_conv3x3_wino4_prep", w.to(dev), uf, ud)
    msc, msh = rnd(64).to(dev), rnd(64).to(dev) * 0.3
    mean, invstd = rnd(64).to(dev) * 0.2, (torch.rand(64, generator=g) + 0.5).to(dev)
    query("cova_set_option", 2, cap)
    n4 = query("cova_conv3x3_wino4_num_partials", B, H, W)
    N = None
    cases = [("plain+stats", (x, N, N, 0, uf, N, N, N, N, N, N, N)),
             ("pro relu", (x, N, abc, 1, uf, N, N, N, N, N, N, N)),
             ("pro affine", (x, N, abc, 0, uf, N, N, N, N, N, N, N)),
             ("dgrad mask z", (x, N, N, 0, ud, N, N, msc, msh, z, mean, invstd)),
             ("dgrad add mask z", (x, N, N, 0, ud, add, N, msc, msh, z, mean, invstd))]
    worst = 0.0
    for name, args in cases:
        outs = []
        for f32 in (1, 0):
            query("cova_set_option", 9, f32)
            out, part = torch.full_like(x, 7.0), torch.zeros(n4, 2, 64, device=dev)
            call("cova_conv3x3_wino4_full", *args, out, part, B, H, W)
            torch.cuda.synchronize()
            outs.append((out, part))
        (o1, p1), (o0, p0) = outs
        scale = float(o1.abs().max())
        d = float((o1 - o0).abs().max()) / scale
        flips = int(((o1 == 0) != (o0 == 0)).sum()) if "mask" in name else 0
        ps = float((p1.double().sum(0) - p0.double().sum(0)).abs().max() / p1.double().sum(0).abs().max())
        worst = max(worst, d)
        print("  %-18s max|split - f32| / scale = %.2e   mask differences %d   partial sums rel %.2e" % (name, d, flips, ps))
    # against fp64
    ref = F.conv2d(xc.double(), w.double(), padding=1)
    errs = []
    for f32 in (1, 0):
        query("cova_set_option", 9, f32)
        out = torch.empty_like(x)
        call("cova_conv3x3_wino4", x, uf, out, None, B, H, W)
        errs.append(float((nchw(out).double() - ref).abs().max() / ref.abs().max()))
    print("  forward against fp64: f32 loop %.2e   split loop %.2e" % tuple(errs))
    query("cova_set_option", 9, 0)
    query("cova_set_option", 2, 0)
    return worst


for shape in [(1, 8, 32, 0), (2, 19, 45, 0), (3, 100, 200, 0), (3, 100, 200, 5), (1, 3, 5, 0), (2, 320, 320, 0)]:
    print("B,H,W,cap =", shape)
    check(*shape)
print("post-ReLU input, 4 x 160 x 160")
check(4, 160, 160, 0, relu_in=True)

# ---- timing
B, H, W = 16, 320, 320
x, z, add, act = (torch.randn(B, H, W, 64, device=dev) for _ in range(4))
w = torch.randn(64, 64, 3, 3, device=dev) * 0.05
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
for f32 in (1, 0):
    query("cova_set_option", 9, f32)
    print(("f32 loop  " if f32 else "split loop") + " ms: " + "  ".join("%s %.3f" % (n, timeit(fn)) for n, fn in cases))
query("cova_set_option", 9, 0)
xr = x.clamp_min(0)
print("split loop, post-ReLU input, plain: %.3f ms" % timeit(lambda: full(xr, N, N, 0, uf, N, N, N, N, N, N, N)))
