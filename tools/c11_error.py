"""Error class of the 1x1 products on the bf16 matrix pipe (three bf16 pieces per f32 operand, six products; csrc/conv1x1.hip,
csrc/conv1x1_lin.hip) against fp64, beside a plain f32 GEMM of the same operands (torch.matmul on the GPU, fp32) and beside a
2-piece / 3-product emulation -- the same three numbers as tools/w4s_bias.py prints for the 3x3 kernels:
  max   = max |got - ref| / max |ref|        rms = rms(got - ref) / rms(ref)        bias = mean(got - ref) / mean |ref|
(the bias column is what a coherent rounding direction of the bf16 MFMA would show; odd row tiles / blocks are multiplied with
negated activations and negated back).  python tools/c11_error.py [R]"""
import os as _os; _os.environ.setdefault("COVA_ALLOW_OPTION_CHANGES", "1")
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cova_amd  # noqa: E402,F401  (registers cova_web_object_detection_amd)
from cova_web_object_detection_amd import engine  # noqa: E402
from cova_web_object_detection_amd._lib import call, query  # noqa: E402

DEV = "cuda:0"
R = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
rs = np.random.RandomState(3)


def rnd(*shape, scale=1.0, positive=False):
    x = rs.standard_normal(shape) * scale
    if positive:
        x = np.maximum(x + 0.5 * scale, 0.0)            # ReLU-like: one sign, as the activations the step multiplies
    return torch.from_numpy(x.astype(np.float32))


def row(name, got, ref):
    got, ref = got.detach().cpu().double(), ref.double()
    e = got - ref
    print("  %-46s max %.2e   rms %.2e   bias %+.2e" % (name, float(e.abs().max() / ref.abs().max()),
                                                         float(e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()),
                                                         float(e.mean() / ref.abs().mean())))


print("# tools/c11_error.py: R = %d pixel rows; operands N(0,1) or ReLU-like (one sign), weights N(0, 0.1)" % R)
for cin, cout in ((64, 256), (256, 64), (64, 64)):
    for positive in (False, True):
        x, w = rnd(R, cin, positive=positive), rnd(cout, cin, scale=0.1, positive=positive)
        ref = x.double() @ w.double().t()
        out = torch.empty((R, cout), device=DEV)
        engine.conv1x1(x.to(DEV), None, None, 0, w.to(DEV), 0, out, None, R, cin, cout)
        kind = "one-sign" if positive else "N(0,1)  "
        row("conv1x1 %3d->%3d %s bf16 x3 / 6 products" % (cin, cout, kind), out, ref)
        row("                 %s f32 GEMM (torch, GPU)" % kind, x.to(DEV) @ w.to(DEV).t(), ref)
# the linear-form reductions: P = v^T a, G = a^T a over the pixel rows (K = R: the longest accumulation chains of the family)
for positive in (False, True):
    act, v = rnd(R, 64, positive=positive), rnd(R, 256, positive=positive)
    nl = query("cova_conv1x1_lin_floats")
    lin = torch.empty(nl, device=DEV)
    ws = torch.empty(query("cova_conv1x1_vprod_workspace_floats", R), device=DEV)
    call("cova_conv1x1_vprod", v.to(DEV), act.to(DEV), None, 0, lin, ws, R)
    P, G = lin[:16384].view(256, 64), lin[16384:20480].view(64, 64)
    kind = "one-sign" if positive else "N(0,1)  "
    row("vprod P = v^T a   %s bf16 x3 / 6 products" % kind, P, v.double().t() @ act.double())
    row("                  %s f32 GEMM (torch, GPU)" % kind, v.to(DEV).t() @ act.to(DEV), v.double().t() @ act.double())
    row("vprod G = a^T a   %s bf16 x3 / 6 products" % kind, G, act.double().t() @ act.double())
    row("                  %s f32 GEMM (torch, GPU)" % kind, act.to(DEV).t() @ act.to(DEV), act.double().t() @ act.double())
