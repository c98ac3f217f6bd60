"""Soak: N training steps at the benchmark size; checks that memory stays flat, the step time is steady
and the loss stays finite (production sanity, not a benchmark)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cova_amd  # noqa
import bench
from cova_web_object_detection_amd import weights
from cova_web_object_detection_amd.trainer import HotPathTrainer

N = int(os.environ.get("STEPS", 1000))
dev = torch.device("cuda", 0)
wcfg = {k: v for k, v in bench.CFG.items() if k != "drop_prob"}
tr = HotPathTrainer(bench.CFG, weights.seeded_state_dict(123, **wcfg), dev)
batches = [bench.make_device_batch(100 + i, dev) for i in range(4)]
for i in range(5):
    tr.train_step(batches[i % 4])
torch.cuda.synchronize()
torch.cuda.reset_peak_memory_stats()
m0 = torch.cuda.memory_allocated()
t0 = time.perf_counter()
marks = []
for i in range(N):
    loss, _ = tr.train_step(batches[i % 4])
    if (i + 1) % (N // 10) == 0:
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        marks.append((i + 1, float(loss), (t1 - t0) / (N // 10) * 1e3, torch.cuda.memory_allocated() / 1e9,
                      torch.cuda.max_memory_allocated() / 1e9, torch.cuda.memory_reserved() / 1e9))
        t0 = t1
for m in marks:
    print("step %5d  loss %10.3f  %.3f ms/step  allocated %.2f GB  peak %.2f GB  reserved %.2f GB" % m)
assert all(torch.isfinite(torch.tensor(m[1])) for m in marks)
assert marks[-1][5] <= marks[0][5] * 1.01 + 0.1, "reserved memory grows"
print("soak ok")
