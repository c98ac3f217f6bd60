"""conv1 weight gradient (pool backward folded in) of a 4-page batch in one call against the sum over two 2-page calls
with bit-identical operands: isolates the kernel from the BatchNorm coefficients it is given."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cova_amd  # noqa
from cova_web_object_detection_amd._lib import call, query
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(11)
B, H, W = 4, 128, 128
x = torch.rand(B, 3, H, W, device=dev, generator=g)
H1 = W1 = 64; H2 = W2 = 32
y = torch.randn(B, H1, W1, 64, device=dev, generator=g) * 3
scale = torch.rand(64, device=dev, generator=g) - 0.3
shift = torch.randn(64, device=dev, generator=g) * 0.2
p1 = torch.empty(B, H2, W2, 64, device=dev); idx = torch.empty(B, H2, W2, 64, device=dev, dtype=torch.uint8)
call("cova_bn_relu_maxpool_fwd", y, scale, shift, p1, idx, None, B, H1, W1)
dp = torch.randn(B, H2, W2, 64, device=dev, generator=g) * (p1 > 0)
abc = torch.randn(3, 64, device=dev, generator=g) * 0.3
for f32 in (1, 0):
    call("cova_set_option", 7, f32)
    def wg(sl):
        n = sl.stop - sl.start
        ws = torch.empty(query("cova_conv1_wgrad_workspace_floats", n, H, W), device=dev)
        dw = torch.zeros(64, 3, 7, 7, device=dev)
        call("cova_conv1_wgrad_poolbwd", x[sl].contiguous(), y[sl].contiguous(), dp[sl].contiguous(), idx[sl].contiguous(), abc, dw, ws, n, H, W)
        return dw
    full = wg(slice(0, 4)); parts = wg(slice(0, 2)) + wg(slice(2, 4))
    print("f32" if f32 else "bf16", "max |full - sum of halves| / max|full| = %.3e" % (float((full - parts).abs().max()) / float(full.abs().max())))
    # forward too
    w = torch.randn(64, 3, 7, 7, device=dev, generator=torch.Generator(device=dev).manual_seed(1)) * 0.05
    def fw(sl):
        n = sl.stop - sl.start
        out = torch.zeros(n, H1, W1, 64, device=dev); part = torch.zeros(query("cova_conv1_num_partials", n, H, W), 2, 64, device=dev)
        call("cova_conv1_fwd_tail", x[sl].contiguous(), w, out, part, n, H, W, None)
        return out, part
    of, pf = fw(slice(0, 4)); o0, p0 = fw(slice(0, 2)); o1, p1_ = fw(slice(2, 4))
    print("    forward: outputs equal", torch.equal(of, torch.cat((o0, o1))), " partial rows equal", torch.equal(pf, torch.cat((p0, p1_))))
call("cova_set_option", 7, 0)
