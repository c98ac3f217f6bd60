"""How much of the step is launch gap?  Captures one HotPathTrainer.train_step in a HIP graph (dropout seeds and
the Adam step count are frozen by the capture -- a timing probe, not a training mode) and compares replay with
eager launches.  Usage: python tools/graph_probe.py [--config 2|3]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cova_amd  # noqa: F401  (registers the package alias)
import bench
from cova_web_object_detection_amd.trainer import HotPathTrainer
from cova_web_object_detection_amd import weights

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=2)
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
wl = bench.WORKLOADS[a.config]
cfg = bench.model_cfg(wl)
wcfg = bench.weight_cfg(cfg)
batch = bench.make_device_batch(123, dev, config=a.config)
tr = HotPathTrainer(cfg, weights.seeded_state_dict(123, **wcfg), dev)

def timed(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

for _ in range(3): tr.train_step(batch)
eager = timed(lambda: tr.train_step(batch), a.steps)
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): tr.train_step(batch)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    tr.train_step(batch)
graph = timed(g.replay, a.steps)
print("config %d: eager %.3f ms/step, graph replay %.3f ms/step (%.1f %%)" % (a.config, eager, graph, 100 * (eager - graph) / eager))
