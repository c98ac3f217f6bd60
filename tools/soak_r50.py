import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cova_amd, bench
from cova_web_object_detection_amd import weights
from cova_web_object_detection_amd.trainer import HotPathTrainer
dev = torch.device("cuda", 0)
cfg = bench.model_cfg(bench.WORKLOADS[3])
tr = HotPathTrainer(cfg, weights.seeded_state_dict(123, **bench.weight_cfg(cfg)), dev)
batches = [bench.make_device_batch(100 + i, dev, 8, 3) for i in range(3)]
losses = []
for i in range(150):
    loss, _ = tr.train_step(batches[i % 3])
    if i % 30 == 0 or i == 149:
        torch.cuda.synchronize()
        losses.append(float(loss))
        print(i, float(loss), torch.cuda.memory_allocated() >> 20, "MiB", torch.cuda.max_memory_allocated() >> 20, flush=True)
assert all(l == l for l in losses), "NaN"
assert losses[-1] < losses[0], (losses[0], losses[-1])
print("ok: loss falls, memory flat")
