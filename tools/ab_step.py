"""Interleaved A/B of the train step at configs[1] in ONE process: alternates cova_set_option(<key>, 0 / 1) every 10 steps,
<rounds> times, and prints the step time of every leg plus the mean launch times of the big kernels in each mode.
  python tools/ab_step.py [key=9] [rounds=4]        key 9: F(4x4) split / f32 main loop, 7: conv1 split / f32"""
import os, sys, time
os.environ.setdefault("COVA_ALLOW_OPTION_CHANGES", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cova_amd  # noqa: F401
import bench
from cova_web_object_detection_amd import _lib, weights
from cova_web_object_detection_amd.trainer import HotPathTrainer
key = int(sys.argv[1]) if len(sys.argv) > 1 else 9
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
vals = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, 1)      # the two option values that alternate
dev = torch.device("cuda", 0)
wl = bench.WORKLOADS[2]
cfg = bench.model_cfg(wl)
sd = weights.seeded_state_dict(123, **bench.weight_cfg(cfg))
tr = HotPathTrainer(cfg, sd, dev, dropout_seed=123)
batch = bench.make_device_batch(123, dev, wl["pages"], 2)
names = ["cova_conv3x3_wino4_full", "cova_conv3x3_wino4_full_tail", "cova_conv3x3_wgrad4_partial", "cova_conv1_fwd_tail",
         "cova_conv1_wgrad_poolbwd", "cova_bn_relu_maxpool_fwd", "cova_bn_act_fwd_bits", "cova_bn1d_fwd", "cova_bn1d_bwd", "cova_gat_fwd", "cova_gat_bwd", "cova_sgemm", "cova_roipool_fwd_bn", "cova_roipool_bwd_bn_tail",
         "cova_conv3x3_wgrad4_finish"]
for _ in range(5):
    tr.train_step(batch)
torch.cuda.synchronize()
# option value 1 = the A/B alternative (f32 kernels for keys 7 and 9)
for r in range(rounds):
    for val in vals:
        _lib.query("cova_set_option", key, val)
        for _ in range(3):
            tr.train_step(batch)
        torch.cuda.synchronize()
        _lib.PROFILE = {n: [] for n in names}
        t0 = time.perf_counter()
        for _ in range(10):
            tr.train_step(batch)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        prof, _lib.PROFILE = _lib.PROFILE, None
        ms = {n.replace("cova_", ""): (sum(a.elapsed_time(b) for a, b, _ in v) / len(v) if v else 0.0) for n, v in prof.items()}
        print("round %d option(%d)=%d: %.3f ms/step | " % (r, key, val, dt * 1e3) + "  ".join("%s %.3f" % kv for kv in ms.items()), flush=True)
if key:
    _lib.query("cova_set_option", key, 0)
