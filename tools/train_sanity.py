import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cova_amd  # noqa
import bench
from cova_web_object_detection_amd import weights
from cova_web_object_detection_amd.trainer import HotPathTrainer
dev = torch.device("cuda:0")
wcfg = {k: v for k, v in bench.CFG.items() if k != "drop_prob"}
tr = HotPathTrainer(bench.CFG, weights.seeded_state_dict(123, **wcfg), dev)
batch = bench.make_device_batch(123, dev, 8)
for i in range(61):
    loss, pred = tr.train_step(batch)
    if i % 10 == 0:
        acc = float((pred == batch["labels"]).float().mean())
        print("step %3d  loss %10.3f  box acc %.4f  finite %s" % (i, float(loss), acc, bool(torch.isfinite(tr.pbucket.flat).all())))
