"""Train-step rate when every batch comes from HOST memory: uint8 pages + csv rows -> pipeline.Prefetcher
(pinned staging, side-stream upload, device-side ToTensor/collate) -> HotPathTrainer.train_step.
Compare with bench.py's device-resident rate."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cova_amd  # noqa
from cova_web_object_detection_amd import weights
from cova_web_object_detection_amd.pipeline import DeviceCollate, Prefetcher
from cova_web_object_detection_amd.trainer import HotPathTrainer

dev = "cuda:0"
CFG = dict(roi_output_size=(3, 3), n_classes=4, use_context=True, hidden_dim=384, bbox_hidden_dim=32,
           n_additional_feat=0, drop_prob=0.2)
PAGES, IMG, BOXES, CS, STEPS = 16, 1280, 90, 12, int(os.environ.get("STEPS", 20))
rs = np.random.RandomState(0)
host_batches = []
for _ in range(4):                      # four distinct host batches, cycled
    u8 = torch.from_numpy(rs.randint(0, 256, (PAGES, IMG, IMG, 3), dtype=np.uint8)).pin_memory()
    rows = []
    for _p in range(PAGES):
        wh = np.stack([rs.uniform(8, 400, BOXES), rs.uniform(8, 200, BOXES)], 1)
        xy = rs.uniform(0, 1, (BOXES, 2)) * (IMG - wh)
        lab = np.zeros((BOXES, 1)); lab[rs.permutation(BOXES)[:3], 0] = [1, 2, 3]
        rows.append(np.concatenate([xy, wh, lab], 1).astype(np.float32))
    host_batches.append((u8, rows))
wcfg = {k: v for k, v in CFG.items() if k != "drop_prob"}
tr = HotPathTrainer(CFG, weights.seeded_state_dict(123, **wcfg), dev)


def source(n):
    for i in range(n):
        yield host_batches[i % len(host_batches)]


for b in Prefetcher(DeviceCollate(CS, dev, pin=True), source(5)):
    tr.train_step(b)
torch.cuda.synchronize()
t0 = time.perf_counter()
for b in Prefetcher(DeviceCollate(CS, dev, pin=True), source(STEPS)):
    tr.train_step(b)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("host-fed (uint8 over PCIe, overlapped): %.1f pages/s, %.2f ms/step; upload %.1f MB/step"
      % (PAGES * STEPS / dt, 1e3 * dt / STEPS, PAGES * IMG * IMG * 3 / 1e6))
