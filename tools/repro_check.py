"""Bit-reproducibility of whole train steps at the benchmark sizes (configs[1] and configs[2] models): two trainers,
same batches, four steps -> identical parameters and gradients (no float atomics anywhere in the step)."""
import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, cova_amd, bench
from cova_web_object_detection_amd import weights
from cova_web_object_detection_amd.trainer import HotPathTrainer
dev = torch.device("cuda", 0)
for config in (2, 3):
    cfg = bench.model_cfg(bench.WORKLOADS[config])
    sd = weights.seeded_state_dict(123, **bench.weight_cfg(cfg))
    batches = [bench.make_device_batch(100 + i, dev, 8 if config == 3 else 16, config) for i in range(2)]
    outs = []
    for rep in range(2):
        tr = HotPathTrainer(cfg, sd, dev, dropout_seed=7)
        for i in range(4):
            loss, _ = tr.train_step(batches[i % 2])
        torch.cuda.synchronize()
        outs.append((float(loss), tr.pbucket.flat.clone(), tr.gbucket.flat.clone()))
    print("config", config, "loss", outs[0][0], outs[1][0], "params equal", torch.equal(outs[0][1], outs[1][1]),
          "grads equal", torch.equal(outs[0][2], outs[1][2]))
