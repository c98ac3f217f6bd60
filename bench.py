#!/usr/bin/env python3
"""Benchmark of the CoVA hot path on MI355X: webpages/s of one full training step
(forward + CrossEntropy(sum) + backward + gradient all-reduce + Adam; train.py:42-60).

Workload (BASELINE.json configs[1], the configuration the metric is quoted on): synthetic
1280x1280 screenshots, 16 pages per GPU, 90 boxes per page, K = 24 DOM-window neighbours,
ResNet-18 stem+layer1 representation network + 1-head GAT, Dropout p = 0.2, fp32.
Inputs are device resident before the timed region.  N > 1: one process per GPU (launched by
torch.distributed.run), every rank trains on its own 16 pages (weak scaling) and the flat
6.5 MB gradient bucket is all-reduced over RCCL once per step.

Prints ONE JSON line on rank 0.  Extra objects:
  roofline     -- the dominant kernel (conv3x3 64->64 implicit GEMM on f32 MFMA): algorithmic
                  FLOPs per launch / mean launch time measured with HIP events on the launching
                  stream inside the timed steps, against the 157.3 TFLOP/s f32-MFMA peak.
  cpu_baseline -- the CPU oracle (oracle/cova_oracle.py, a restatement of the reference's
                  torch-CPU path) timed on this host on a bounded sample (2 pages per step).
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC for RCCL: before the HIP runtime starts

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import cova_amd  # noqa: E402,F401
from cova_web_object_detection_amd import _lib, synthetic, weights  # noqa: E402
from cova_web_object_detection_amd.trainer import HotPathTrainer  # noqa: E402

CFG = dict(roi_output_size=(3, 3), n_classes=4, use_context=True, hidden_dim=384,
           bbox_hidden_dim=32, n_additional_feat=0, drop_prob=0.2)
IMG, PAGES_PER_GPU, BOXES, CS = 1280, 16, 90, 12
PEAK_F32_MFMA_TFLOPS = 157.3                       # MI355X_MICROARCH.md, f32-input MFMA
CONV3_FLOP_PER_PIXEL = 2 * 64 * 64 * 9


def make_device_batch(seed, device, pages=PAGES_PER_GPU, img=IMG):
    """Boxes / neighbour tables from the numpy generator; pixels drawn on the device (uniform
    [0,1) like datasets.py:41-45's ToTensor output) to keep start-up short."""
    g = torch.Generator(device=device).manual_seed(seed)
    images = torch.rand((pages, 3, img, img), generator=g, device=device, dtype=torch.float32)
    boxes = synthetic.make_boxes_only(pages, img, img, BOXES, CS, seed)
    out = {k: v.to(device) for k, v in boxes.items() if torch.is_tensor(v)}
    out["images"] = images
    return out


def cpu_baseline(steps=2):
    """Oracle train step (forward, CE-sum, backward, Adam) on the host cores, 2 pages per step."""
    from oracle import cova_oracle as O
    pages = 2
    cfg = dict(CFG, drop_prob=0.0)      # oracle takes explicit masks; p=0 costs the same FLOPs
    wcfg = {k: v for k, v in CFG.items() if k != "drop_prob"}
    sd = weights.seeded_state_dict(123, **wcfg)
    boxes = synthetic.make_boxes_only(pages, IMG, IMG, BOXES, CS, 123)
    images = torch.rand((pages, 3, IMG, IMG), generator=torch.Generator().manual_seed(123))
    keys = O.param_keys(sd)
    state = None

    def one():
        nonlocal sd, state
        _, _, grads, after, _ = O.loss_and_grads(sd, images, boxes["bboxes"], boxes["additional_feats"],
                                                 boxes["context_indices"], boxes["labels"], cfg, None)
        new_p, state = O.adam_reference([sd[k] for k in keys], [grads[k] for k in keys], state)
        for k, p in zip(keys, new_p):
            after[k] = p
        sd = after

    one()                                # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    dt = (time.perf_counter() - t0) / steps
    return {"value": round(pages / dt, 4), "unit": "webpages/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": "%d steps x %d pages of the same workload (1280x1280, 90 boxes, K=24), "
                      "torch-CPU fp32 oracle incl. Adam, %.2f s/step" % (steps, pages, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pages", type=int, default=PAGES_PER_GPU)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sync-bn", action="store_true",
                    help="N > 1: BatchNorm statistics over the whole data-parallel batch (default: per rank, as DDP)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d "
                         "--master-addr 127.0.0.1 --master-port P bench.py --gpus %d ..." % (args.gpus, args.gpus))
    # Diagnostic only (code-path check of the N > 1 launch on a 1-GPU box): COVA_BENCH_BACKEND=gloo lets
    # all ranks share cuda:0; the throughput of such a run means nothing and is labelled in `config`.
    backend = os.environ.get("COVA_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    group = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    wcfg = {k: v for k, v in CFG.items() if k != "drop_prob"}
    sd = weights.seeded_state_dict(123, **wcfg)
    trainer = HotPathTrainer(CFG, sd, device, world_size=world, process_group=group, dropout_seed=123 + rank,
                             sync_bn=args.sync_bn)
    batch = make_device_batch(123 + rank, device, args.pages)
    n_boxes = batch["bboxes"].shape[0]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.train_step(batch)
    barrier()
    _lib.PROFILE = {"cova_conv3x3_fwd": [], "cova_conv3x3_dgrad_bnbwd": [], "cova_conv3x3_wino": [],
                    "cova_conv3x3_wino_pro": [], "cova_conv1_fwd": [], "cova_conv1_wgrad_poolbwd": [],
                    "cova_conv1_wgrad": [], "cova_conv3x3_wgrad_wino_pro": [], "cova_conv3x3_wgrad_wino": [],
                    "cova_bn_relu_maxpool_fwd": []}
    if os.environ.get("COVA_PROFILE_ALL"):          # per-entry-point HIP-event timing (diagnostic)
        _lib.PROFILE = {name: [] for name in _lib.lib().protos}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = trainer.train_step(batch)
    barrier()
    dt = time.perf_counter() - t0
    prof = (_lib.PROFILE["cova_conv3x3_fwd"] + _lib.PROFILE["cova_conv3x3_dgrad_bnbwd"] +
            _lib.PROFILE["cova_conv3x3_wino"] + _lib.PROFILE["cova_conv3x3_wino_pro"])
    def mean_ms(*names):
        ev = [p for n in names for p in _lib.PROFILE.get(n, [])]
        return (sum(a.elapsed_time(b) for a, b in ev) / len(ev), len(ev)) if ev else (0.0, 0)

    others = {}
    hw = (IMG // 4) * (IMG // 4)
    f_conv1 = 2 * 64 * 147 * args.pages * (IMG // 2) * (IMG // 2)
    for key, names, flop, nbytes in (
            ("conv1_7x7_fwd", ("cova_conv1_fwd",), f_conv1, 0),
            ("conv1_7x7_wgrad_with_pool_backward", ("cova_conv1_wgrad_poolbwd", "cova_conv1_wgrad"), f_conv1, 0),
            ("conv3x3_wgrad_winograd", ("cova_conv3x3_wgrad_wino_pro", "cova_conv3x3_wgrad_wino"),
             CONV3_FLOP_PER_PIXEL * args.pages * hw, 0),
            # reads conv1's output, writes the pooled map, its arg-max pre-activation and uint8 indices
            ("bn_relu_maxpool_fwd", ("cova_bn_relu_maxpool_fwd",), 0,
             args.pages * 64 * (4 * (IMG // 2) * (IMG // 2) + (4 + 4 + 1) * hw))):
        ms, n = mean_ms(*names)
        if n:
            o = {"avg_launch_ms": round(ms, 4), "launches_timed": n}
            if flop:
                o.update(algorithmic_tflops=round(flop / ms / 1e9, 1),
                         frac_of_f32_mfma_peak=round(flop / ms / 1e9 / PEAK_F32_MFMA_TFLOPS, 3))
            if nbytes:
                o.update(algorithmic_tb_per_s=round(nbytes / ms / 1e9, 2), frac_of_hbm_peak=round(nbytes / ms / 1e9 / 8.0, 3))
            others[key] = o
    if os.environ.get("COVA_PROFILE_ALL") and rank == 0:
        rows = [(sum(a.elapsed_time(b) for a, b in v) / args.steps, len(v) // args.steps, k)
                for k, v in _lib.PROFILE.items() if v]
        for ms, n, k in sorted(rows, reverse=True):
            print("%-36s %3d calls/step %8.3f ms/step" % (k, n, ms), file=sys.stderr)
        print("sum %.3f ms/step" % sum(r[0] for r in rows), file=sys.stderr)
    _lib.PROFILE = None
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss_val = float(loss.item())

    # forward only (eval mode, running statistics): the second number SURVEY.md section 8d asks for.
    # Not the metric; reported as an extra object.
    for _ in range(2):
        trainer.predict(batch)
    barrier()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        trainer.predict(batch)
    barrier()
    dt_fwd = time.perf_counter() - t1
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt_fwd], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_fwd = float(t.item())

    if rank == 0:
        ms_per_step = 1e3 * dt / args.steps
        value = world * args.pages * args.steps / dt
        conv_ms = sum(a.elapsed_time(b) for a, b in prof) / max(len(prof), 1)
        flops = CONV3_FLOP_PER_PIXEL * args.pages * (IMG // 4) * (IMG // 4)
        achieved = flops / (conv_ms * 1e-3) / 1e12
        out = {
            "metric": "webpages/sec fwd+bwd (90 bboxes, K=24)", "value": round(value, 3),
            "unit": "webpages/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: synthetic 1280x1280 screenshots, %d pages/GPU, "
                                   "%d boxes/page, K=%d, ResNet-18 stem+layer1 RN + 1-head GAT, "
                                   "train step = fwd+CE+bwd+allreduce+Adam, dropout 0.2"
                                   % (args.pages, BOXES, 2 * CS),
                       "pages_per_gpu": args.pages, "global_pages": world * args.pages,
                       "boxes_per_gpu": n_boxes, "parallelism": "dp%d" % world + ("+syncbn" if args.sync_bn and world > 1 else "") +
                       ("" if backend == "nccl" else " (DIAGNOSTIC: %s backend, shared GPU)" % backend),
                       "loss": round(loss_val, 3)},
            # `achieved` counts the ALGORITHMIC (direct-convolution) FLOPs of SURVEY.md section 8d; the
            # kernel is Winograd F(2x2,3x3) and executes 2.25x fewer MFMA FLOPs, so the algorithmic rate
            # can exceed the MFMA peak; `executed_*` is the matrix-pipe view of the same launches.
            "roofline": {"bound": "mfma", "kernel": "conv3x3_c64_wino_kernel (4 forward + 4 data-gradient launches per step)",
                         "algorithm": "winograd F(2x2,3x3), exact f32 MFMA (v_mfma_f32_16x16x4_f32)",
                         "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                         "executed_flop_per_launch": int(flops / 2.25),
                         "executed_achieved": round(achieved / 2.25, 2),
                         "executed_frac": round(achieved / 2.25 / PEAK_F32_MFMA_TFLOPS, 4),
                         # HBM bytes per forward launch from PMC (separate --pmc FETCH_SIZE /
                         # WRITE_SIZE passes, FETCH doubled per the gfx950 correction):
                         # plain forward launch: FETCH_SIZE 2*236.6 MiB (gfx950 half-count correction) +
                         # WRITE_SIZE 408.6 MiB, profiles/r01_pmc_hbm_traffic_final.txt; algorithmic =
                         # 2 * 419.4 MB (one read + one write of [16,320,320,64] f32).  The fused
                         # variants read 1-3 more maps (BatchNorm operands) -- see DESIGN.md 4.6.
                         "traffic": 924.6e6 * args.pages / 16, "traffic_unit": "B/launch",
                         "algorithmic_bytes": 2 * 4 * 64 * args.pages * (IMG // 4) * (IMG // 4),
                         "launches_timed": len(prof), "avg_launch_ms": round(conv_ms, 4),
                         "flop_per_launch": flops},
            # the other large kernels of the step, same live HIP-event timing (algorithmic FLOPs / bytes)
            "other_kernels": others,
        }
        out["forward_only"] = {"value": round(world * args.pages * args.steps / dt_fwd, 2), "unit": "webpages/s",
                               "ms_per_step": round(1e3 * dt_fwd / args.steps, 3),
                               "mode": "eval forward (running statistics) + per-box argmax, same batch"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
