"""Signed (systematic) error of the F(4x4,3x3) launches against a float64 convolution: per-channel SUMS of the output -- what
BatchNorm-backward style reductions over millions of positions see -- on the f32 and on the split main loop."""
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
B, H, W = 4, 320, 320
for name, relu_in, scale in (("post-ReLU activations", True, 1.0), ("gradient-like (zero mean, 1e-3)", False, 1e-3)):
    g = torch.Generator().manual_seed(5)
    xc = torch.randn(B, 64, H, W, generator=g) * scale
    if relu_in:
        xc = xc.clamp_min(0)
    w = torch.randn(64, 64, 3, 3, generator=g) * 0.05
    x = nhwc(xc)
    uf, ud = torch.empty(NU, device=dev), torch.empty(NU, device=dev)
    call("cova_conv3x3_wino4_prep", w.to(dev), uf, ud)
    ref = F.conv2d(xc.double(), w.double(), padding=1)
    print(name)
    for f32 in (1, 0):
        query("cova_set_option", 9, f32)
        out = torch.empty_like(x)
        call("cova_conv3x3_wino4", x, uf, out, None, B, H, W)
        d = nchw(out).double() - ref
        per_ch = d.sum((0, 2, 3)) / ref.abs().sum((0, 2, 3))
        print("  %s: max |err| / max|ref| %.2e   mean signed err / mean|ref| %+.2e   per-channel signed sums: worst %+.2e, rms %.2e,  "
              "sign(err) = sign(ref) on %.4f of the elements" % ("f32 loop  " if f32 else "split loop", float(d.abs().max() / ref.abs().max()),
              float(d.sum() / ref.abs().sum()), float(per_ch[per_ch.abs().argmax()]), float(per_ch.pow(2).mean().sqrt()),
              float(((d * ref) > 0).double().mean())))
query("cova_set_option", 9, 0)

# ---- sign symmetry: every f32 operation of the path is sign-symmetric under round-to-nearest; is the bf16 MFMA's accumulation?
g = torch.Generator().manual_seed(9)
xc = torch.randn(2, 64, 160, 160, generator=g)
w = torch.randn(64, 64, 3, 3, generator=g) * 0.05
uf, ud = torch.empty(NU, device=dev), torch.empty(NU, device=dev)
ufn, udn = torch.empty(NU, device=dev), torch.empty(NU, device=dev)
call("cova_conv3x3_wino4_prep", w.to(dev), uf, ud)
call("cova_conv3x3_wino4_prep", (-w).to(dev), ufn, udn)
ref = F.conv2d(xc.double(), w.double(), padding=1)
for f32 in (1, 0):
    query("cova_set_option", 9, f32)
    o = [torch.empty(2, 160, 160, 64, device=dev) for _ in range(3)]
    call("cova_conv3x3_wino4", nhwc(xc), uf, o[0], None, 2, 160, 160)
    call("cova_conv3x3_wino4", nhwc(-xc), uf, o[1], None, 2, 160, 160)
    call("cova_conv3x3_wino4", nhwc(xc), ufn, o[2], None, 2, 160, 160)
    e = [nchw(t).double() for t in o]
    print("%s: conv(-x) == -conv(x) bitwise: %s;  conv(x; -w) == -conv(x; w): %s;  mean signed error / mean|ref|: x %+.2e, -x %+.2e, -w %+.2e"
          % ("f32 loop  " if f32 else "split loop", bool(torch.equal(o[1], -o[0])), bool(torch.equal(o[2], -o[0])),
             float((e[0] - ref).sum() / ref.abs().sum()), float((e[1] + ref).sum() / ref.abs().sum()), float((e[2] + ref).sum() / ref.abs().sum())))
query("cova_set_option", 9, 0)

# ---- the same question for conv1 (7x7 / stride 2, bf16-split since round 4): forward output and weight gradient
B, H, W = 2, 640, 640
g = torch.Generator().manual_seed(3)
img = torch.rand(B, 3, H, W, generator=g)
w7 = torch.randn(64, 3, 7, 7, generator=g) * 0.05
wr = w7.double().requires_grad_(True)
ref = F.conv2d(img.double(), wr, stride=2, padding=3)
H1, W1 = ref.shape[2], ref.shape[3]
dy = torch.randn(B, 64, H1, W1, generator=g) * 1e-3
(ref * dy.double()).sum().backward()
ws = torch.empty(query("cova_conv1_wgrad_workspace_floats", B, H, W), device=dev)
for f32 in (1, 0):
    query("cova_set_option", 7, f32)
    out = torch.zeros(B, H1, W1, 64, device=dev)
    call("cova_conv1_fwd_tail", img.to(dev), w7.to(dev), out, None, B, H, W, None)
    dw = torch.zeros(64, 3, 7, 7, device=dev)
    call("cova_conv1_wgrad", img.to(dev), nhwc(dy), dw, ws, B, H, W)
    d = nchw(out).double() - ref.detach()
    print("conv1 %s: forward max |err| / max|ref| %.2e, mean signed err / mean|ref| %+.2e;  weight gradient max err %.2e, mean signed %+.2e"
          % ("f32 MFMA  " if f32 else "bf16 split", float(d.abs().max() / ref.detach().abs().max()), float(d.sum() / ref.detach().abs().sum()),
             float((dw.double().cpu() - wr.grad).abs().max() / wr.grad.abs().max()), float((dw.double().cpu() - wr.grad).sum() / wr.grad.abs().sum())))
query("cova_set_option", 7, 0)
