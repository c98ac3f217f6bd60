"""Why is conv1_fwd slower inside a train step than in isolation?  (diagnostic)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cova_amd  # noqa
import bench
from cova_web_object_detection_amd import _lib, weights, engine
from cova_web_object_detection_amd.trainer import HotPathTrainer
call, query = _lib.call, _lib.query
dev = torch.device("cuda", 0)
wcfg = {k: v for k, v in bench.CFG.items() if k != "drop_prob"}
sd = weights.seeded_state_dict(123, **wcfg)
tr = HotPathTrainer(bench.CFG, sd, dev)
batch = bench.make_device_batch(123, dev, 16)
for _ in range(3):
    tr.train_step(batch)
torch.cuda.synchronize()
img = batch["images"]
wk = torch.empty(154, 64, device=dev)
call("cova_conv1_prep_weights", tr.params["convnet.0.weight"], wk)
y1 = torch.empty(16, 640, 640, 64, device=dev)
part = torch.empty(query("cova_conv1_num_tiles", 16, 1280, 1280), 2, 64, device=dev)


def t_conv1(n=5, pre=None):
    ts = []
    for _ in range(n):
        if pre:
            pre()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call("cova_conv1_fwd", img, wk, y1, part, 16, 1280, 1280)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return " ".join("%.3f" % t for t in ts)


print("isolated, trainer image:", t_conv1())
img2 = torch.rand(16, 3, 1280, 1280, device=dev)
print("trainer image again    :", t_conv1())
img_saved = img
img = img2
print("fresh torch.rand image :", t_conv1())
img = img_saved
print("after a full train step:", t_conv1(3, lambda: tr.train_step(batch)))
big = torch.empty(1 << 28, device=dev)
print("after 1 GiB fill       :", t_conv1(3, lambda: big.fill_(1.0)))
print("after adam only        :", t_conv1(3, lambda: tr.optimizer_step()))
print("image stats: min %.3f max %.3f mean %.3f" % (img.min().item(), img.max().item(), img.mean().item()))


def b2b(fn, n=10):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


f = lambda: call("cova_conv1_fwd", img, wk, y1, part, 16, 1280, 1280)
print("back-to-back x10, trainer weights: %.3f" % b2b(f))
w1 = torch.randn(64, 3, 7, 7, device=dev) * 0.1
call("cova_conv1_prep_weights", w1, wk)
print("back-to-back x10, randn*0.1 weights: %.3f" % b2b(f))
print("isolated, randn*0.1 weights:", t_conv1())
call("cova_conv1_prep_weights", tr.params["convnet.0.weight"], wk)
print("isolated, trainer weights again:", t_conv1())
print("trainer w stats: absmax %.4f std %.4f" % (tr.params["convnet.0.weight"].abs().max().item(), tr.params["convnet.0.weight"].std().item()))
x = torch.randn(16, 320, 320, 64, device=dev)
outc = torch.empty_like(x)
wf, wd = torch.empty(9, 64, 64, device=dev), torch.empty(9, 64, 64, device=dev)
call("cova_conv3x3_prep_weights", torch.randn(64, 64, 3, 3, device=dev) * 0.05, wf, wd)
g = lambda: call("cova_conv3x3_fwd", x, wf, None, outc, None, 16, 320, 320)
print("conv3x3 randn input, b2b: %.3f" % b2b(g))
x2 = tr.params["convnet.0.weight"].new_empty(16, 320, 320, 64).uniform_(0, 1)
x2 = torch.relu(torch.randn(16, 320, 320, 64, device=dev))
g2 = lambda: call("cova_conv3x3_fwd", x2, wf, None, outc, None, 16, 320, 320)
print("conv3x3 relu(randn) input (50%% zeros), b2b: %.3f" % b2b(g2))
