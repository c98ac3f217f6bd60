"""Times cova_roipool_bwd(_bn) on the configs[1] geometry: full, and with zero boxes (= pure zero-fill + write-out)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cova_amd  # noqa: F401
from cova_web_object_detection_amd import synthetic
from cova_web_object_detection_amd._lib import call, query

dev = "cuda:0"
B, C, H, W = 16, int(os.environ.get("C", "64")), 320, 320
boxes = synthetic.make_boxes_only(B, 1280, 1280, 90, 12, 123)
rois = boxes["bboxes"].to(dev)
n = rois.shape[0]
feat = torch.randn(B, H, W, C, device=dev)
out = torch.empty(n, C * 9, device=dev)
arg = torch.empty(n, C * 9, device=dev, dtype=torch.int32)
call("cova_roipool_fwd", feat, rois, n, B, C, H, W, 3, 3, 0.25, out, C * 9, arg)
gout = torch.randn(n, C * 9, device=dev)
gfeat = torch.empty(B, H, W, C, device=dev)
pr = torch.empty(query("cova_roipool_bwd_workspace_words", n, B, C, 3, 3), dtype=torch.int32, device=dev)
z = torch.randn(B, H, W, C, device=dev)
mean, invstd = torch.zeros(C, device=dev), torch.ones(C, device=dev)
part = torch.empty(query("cova_roipool_bwd_bn_num_partials", n), 2, C, device=dev)
zmax = torch.randn(n, C * 9, device=dev)


def timeit(f, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print("plain full   %.1f us" % timeit(lambda: call("cova_roipool_bwd", gout, C * 9, rois, arg, n, B, C, H, W, 3, 3, 0.25, gfeat, pr)))
print("plain n=0    %.1f us" % timeit(lambda: call("cova_roipool_bwd", gout, C * 9, rois, arg, 0, B, C, H, W, 3, 3, 0.25, gfeat, pr)))
print("bn full      %.1f us" % timeit(lambda: call("cova_roipool_bwd_bn", gout, C * 9, out, C * 9, zmax, rois, arg, n, B, C, H, W, 3, 3, 0.25, mean, invstd, gfeat, part, pr)))
print("memset map   %.1f us" % timeit(lambda: gfeat.zero_()))
