#!/usr/bin/env python3
"""Benchmark of the CoVA hot path on MI355X: webpages/s of one full training step
(forward + CrossEntropy(sum) + backward + gradient all-reduce + Adam; train.py:42-60).

Workloads = BASELINE.json's configs (`--config`, numbered as the list reads, 1-based):
  2  configs[1]  1280x1280, 16 pages/GPU, 90 boxes, K=24, ResNet-18 RN + 1-head GAT        (default; the
                 configuration the metric is quoted on)
  3  configs[2]  same pages, 32 pages/GPU, ResNet-50 RN + 2-head GAT                        (extension)
  4  configs[3]  configs[1]'s model, global batch 256 = 32 pages/GPU x 8 GPUs, RCCL all-reduce
  5  configs[4]  1280x4096 pages, 300 boxes, K=48, ResNet-50 RN + 2-head x 2-layer GAT      (extension)
Inputs are synthetic and device resident before the timed region.  N > 1: one process per GPU; `python
bench.py --gpus N` launches itself through torch.distributed.run when it is not already inside one.
`--scaling weak` (default) keeps the pages per GPU fixed, `--scaling strong` fixes the GLOBAL batch
(256 pages unless --global-pages) and splits it over the ranks.

Prints ONE JSON line on rank 0.  Extra objects:
  roofline     -- the dominant kernel family (Winograd F(4x4,3x3) conv3x3 64->64, forward + data gradient), timed live with
                  HIP events on the launching stream inside the timed steps (every 10th).  Its transform-domain products run on
                  the bf16 matrix pipe (f32 operands as three bf16 pieces, six products): the launch's matrix floor is below its
                  HBM floor, so `bound` = "hbm" and `achieved` = ALGORITHMIC bytes / time.  Since round 6 the bytes follow from
                  the launches the step really issues: every timed call's variant and operand count are derived from which of its
                  arguments were non-NULL (`variants`: per variant launches, maps moved, bytes, mean time, frac, PMC traffic);
                  the family `frac` = sum of bytes / sum of time (`hbm_algorithmic_frac` is the same number under a name that
                  says what it is; `f32_equivalent` keeps the round-1..4 definition, executed f32 multiply-adds over the f32-MFMA
                  peak, for comparison across rounds; `hbm_frac` prices the PMC traffic instead).  COVA_W4_F32=1 selects the
                  f32-MFMA main loop (`bound` = "mfma").
  other_kernels-- conv1 forward / weight gradient and the 3x3 weight gradients timed INSIDE the timed steps (`in_step`: the
                  kernels whose time differs between boxes), the rest in five separate steps behind them.
  ab           -- same process, same trainer, no events: 10 steps each on the default kernels and with conv1 / the 3x3 main
                  loop on their f32-MFMA forms (the process keeps the library's options mutable: COVA_ALLOW_OPTION_CHANGES).
  clock_leg    -- effective shader clock per kernel on this box (a 5-step child run under rocprofv3 --pmc GRBM_GUI_ACTIVE):
                  the MFMA-dense kernels run power-limited, and by how much differs from box to box.
  sustained    -- >= 5 s of back-to-back train steps after the headline measurement (same batch), reported
                  separately: long enough for an external utilisation sampler to see the GPU work.
  step         -- whole-step FLOP accounting: algorithmic TFLOP/s, fraction of the direct-convolution MFMA
                  ceiling, fraction of the executed-MFMA floor.
  cpu_baseline -- the CPU oracle (oracle/cova_oracle.py, a restatement of the reference's torch-CPU path)
                  timed on this host: 2-page and 16-page batches, forward and forward+backward+Adam, medians.
"""
import argparse
import datetime
import json
import os
import signal
import socket
import statistics
import subprocess
import sys
import threading
import time
import traceback

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC for RCCL: before the HIP runtime starts
os.environ.setdefault("COVA_ALLOW_OPTION_CHANGES", "1")     # the `ab` legs switch kernel forms inside this process (include/cova_hip.h)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3                       # MI355X_MICROARCH.md, f32-input MFMA
PEAK_BF16_MFMA_TFLOPS = 2500.0                     # ... dense bf16 MFMA (the three-piece split kernels run six products per f32 product)
PEAK_HBM_TBS = 8.0
SPLIT_PRODUCTS = 6                                 # bf16 MFMA products per f32 multiply-add of the split kernels
WINO4_RATIO = 4.0                                  # ... / Winograd F(4x4,3x3): forward and data-gradient launches
TRAFFIC_FILES = [os.path.join("profiles", "r06_hbm_traffic.json"), os.path.join("profiles", "r05_hbm_traffic.json"),
                 os.path.join("profiles", "r04_hbm_traffic.json"), os.path.join("profiles", "r03_hbm_traffic.json")]

WORKLOADS = {
    2: dict(name="configs[1]", H=1280, W=1280, pages=16, boxes=90, cs=12, backbone="resnet18", n_heads=1,
            n_gat_layers=1, desc="ResNet-18 stem+layer1 RN + 1-head GAT"),
    3: dict(name="configs[2]", H=1280, W=1280, pages=32, boxes=90, cs=12, backbone="resnet50", n_heads=2,
            n_gat_layers=1, desc="ResNet-50 stem+layer1 RN + 2-head GAT (extension: not in the reference)"),
    4: dict(name="configs[3]", H=1280, W=1280, pages=32, boxes=90, cs=12, backbone="resnet18", n_heads=1,
            n_gat_layers=1, desc="ResNet-18 RN + 1-head GAT, global batch 256 = 32 pages/GPU at 8 GPUs"),
    5: dict(name="configs[4]", H=4096, W=1280, pages=8, boxes=300, cs=24, backbone="resnet50", n_heads=2,
            n_gat_layers=2, desc="long pages, ResNet-50 RN + 2-head x 2-layer GAT (extension)"),
}


def model_cfg(wl):
    return dict(roi_output_size=(3, 3), n_classes=4, use_context=True, hidden_dim=384, bbox_hidden_dim=32,
                n_additional_feat=0, drop_prob=0.2, backbone=wl["backbone"], n_heads=wl["n_heads"],
                n_gat_layers=wl["n_gat_layers"])


def weight_cfg(cfg):
    return {k: v for k, v in cfg.items() if k not in ("drop_prob", "roi_op", "sampling_ratio", "roi_aligned")}


CFG = model_cfg(WORKLOADS[2])                      # the quoted configuration's model (tools/ import it)


def make_device_batch(seed, device, pages=None, config=2):
    """One synthetic batch of a workload on the device: boxes / neighbour tables from the seeded numpy generator,
    pixels drawn on the device (uniform [0,1) like datasets.py:41-45's ToTensor output) to keep start-up short."""
    import torch
    from cova_web_object_detection_amd import synthetic
    wl = WORKLOADS[config]
    pages = pages or wl["pages"]
    g = torch.Generator(device=device).manual_seed(seed)
    batch = {k: v.to(device) for k, v in synthetic.make_boxes_only(pages, wl["H"], wl["W"], wl["boxes"], wl["cs"],
                                                                    seed).items() if torch.is_tensor(v)}
    batch["images"] = torch.rand((pages, 3, wl["H"], wl["W"]), generator=g, device=device, dtype=torch.float32)
    return batch


# ------------------------------------------------------------------------------------ FLOP accounting
def flop_model(wl):
    """Algorithmic FLOPs per page (SURVEY.md 8d conventions: 2*MACs, W_j once per node, backward = data +
    weight gradient except conv1 (weight only), head x3) and the part of them executed as Winograd."""
    H1, W1 = wl["H"] // 2, wl["W"] // 2
    px = (H1 // 2) * (W1 // 2)
    conv1 = 2 * 64 * 147 * H1 * W1
    c3 = 2 * 64 * 64 * 9 * px
    if wl["backbone"] == "resnet18":
        n3, c1x1, cfeat = 4, 0, 64
    else:
        n3 = 3
        c1x1 = 2 * px * (64 * 64 + 2 * 64 * 256 + 2 * (256 * 64 + 64 * 256))
        cfeat = 256
    n, K, D = wl["boxes"], 2 * wl["cs"], 384
    F = cfeat * 9 + 32
    T = F + D
    gat = 0
    for l in range(wl["n_gat_layers"]):
        fin = F if l == 0 else D
        gat += n * (2 * fin * 2 * D + 2 * K * D)
    head = gat + n * (2 * T * T + 2 * T * 4)
    fwd = conv1 + n3 * c3 + c1x1 + head
    bwd = conv1 + 2 * n3 * c3 + 2 * c1x1 + 2 * head
    wino = 3 * n3 * c3                                  # fwd + dgrad + wgrad all run as Winograd
    return dict(total=fwd + bwd, fwd=fwd, wino=wino, conv3_launch_per_page=c3)


def read_traffic(kernel_key, pages):
    """HBM bytes per launch of the dominant kernel from the committed PMC summary (separate FETCH_SIZE /
    WRITE_SIZE passes, FETCH doubled per the gfx950 correction; tools/hbm_traffic.py writes the file).  Since round 5
    the file holds the launches of ONE train step keyed by kernel variant (`step_kernels`) and their per-family means
    (`step_families`): the number is the mean over the variant mix the step launches, not over every launch of the
    profiling run (which also carried the eval and drop-in legs' variants).
    -> (family bytes per launch, source, step bytes, {kernel variant name: bytes per launch})"""
    for rel in TRAFFIC_FILES:          # the newest PMC pass that has the kernel (an older round's file is a stale number:
        path = os.path.join(ROOT, rel)  # the source is named in the line)
        if not os.path.exists(path):
            continue
        try:
            d = json.load(open(path))
            e = d.get("step_families", {}).get(kernel_key) or d["kernels"][kernel_key]
            step = d.get("step_traffic_bytes")
            scale = pages / d["pages"]
            per_variant = {k: v["traffic_bytes_per_launch"] * scale for k, v in d.get("step_kernels", {}).items()}
            return (e["traffic_bytes_per_launch"] * scale, rel + (":step_families" if "step_families" in d else ""),
                    step * scale if step else None, per_variant)
        except Exception:
            continue
    return None, None, None, {}


# Operands of the F(4x4,3x3) entry points by position (include/cova_hip.h); a profile record carries which were non-NULL
W4_FULL_ARGS = ("in", "in2", "abc", "relu", "u", "addend", "act", "msc", "msh", "z", "mean", "invstd", "out", "part", "B", "H", "W")
W4_TAIL_ARGS = W4_FULL_ARGS[:7] + ("act_bits",) + W4_FULL_ARGS[7:] + ("tail",)


def w4_variant(name, present):
    """The kernel variant a cova_conv3x3_wino4_full(_tail) call launches (the dispatch of csrc/conv_wino4.hip: launch_w4_pro) and
    the 64-channel maps it must move: -> (template arguments "<STATS, PRO, ADD, BN>", PRO, maps read + written, role).
    One map = B*H*W*64 floats.  Read: the input (two tensors under the two-tensor prologue), the residual-branch gradient
    (ADD), z for the ReLU mask / xhat of the BatchNorm-backward sums (BN >= 1), the mask source (BN == 2: a full map, or 1/32
    of one as bits); written: the output."""
    d = dict(zip(W4_TAIL_ARGS if name.endswith("_tail") else W4_FULL_ARGS, present))
    pro = 0 if not d["abc"] else (2 if d["in2"] else 1)
    add = bool(d["addend"])
    bn = 0 if not d["z"] else (1 if not (d["act"] or d.get("act_bits")) else 2)
    stats = bool(d["part"]) or bn > 0
    maps = 1 + (1 if pro == 2 else 0) + (1 if add else 0) + (1 if bn >= 1 else 0) + 1
    if bn == 2:
        maps += 1 if d["act"] else 1.0 / 32
    role = ("data gradient + residual gradient" if add else "data gradient") if bn else \
           ("forward, BatchNorm+ReLU on load" if pro else "forward")
    targs = "<%s, %d, %s, %d>" % ("true" if stats else "false", pro, "true" if add else "false", bn)
    return targs, pro, maps, role


# ------------------------------------------------------------------------------------ CPU baseline
def cpu_info():
    model, cores = "unknown", set()
    try:
        phys = core = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model == "unknown":
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                phys = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core = line.split(":", 1)[1].strip()
                cores.add((phys, core))
    except OSError:
        pass
    return model, (len(cores) or os.cpu_count() or 1)


def cpu_baseline(wl, full=False):
    """The oracle's train step (reference formulation: torch-CPU conv / batch_norm / linear, own RoIPool,
    the [N,K,F] gather GAT) on the host cores, same synthetic generator: batches of 2 and 16
    pages, forward-only and forward+backward+Adam, median of the timed steps."""
    import torch
    from oracle import cova_oracle as O
    from cova_web_object_detection_amd import synthetic, weights
    cfg = model_cfg(wl)
    model, phys = cpu_info()
    torch.set_num_threads(phys)
    out = {"unit": "webpages/s", "cores": torch.get_num_threads(), "physical_cores": phys, "cpu_model": model,
           "kind": "port", "statistic": "median"}
    legs = []
    for pages, warm, timed in ((2, 3, 5), (16, 3 if full else 1, 5 if full else 3)):      # default: full
        sd = weights.seeded_state_dict(123, **weight_cfg(cfg))
        b = synthetic.make_boxes_only(pages, wl["H"], wl["W"], wl["boxes"], wl["cs"], 123)
        images = torch.rand((pages, 3, wl["H"], wl["W"]), generator=torch.Generator().manual_seed(123))
        n, T = b["bboxes"].shape[0], None
        keys = O.param_keys(sd)
        state = [None]
        box = [sd]

        def masks():
            nonlocal T
            if T is None:
                T = box[0]["decoder.1.weight"].shape[0]
            return [(torch.rand(n, T) > cfg["drop_prob"]).float() for _ in range(2)]

        def train():
            sd_ = box[0]
            _, _, grads, after, _ = O.loss_and_grads(sd_, images, b["bboxes"], b["additional_feats"],
                                                     b["context_indices"], b["labels"], cfg, masks())
            new_p, state[0] = O.adam_reference([sd_[k] for k in keys], [grads[k] for k in keys], state[0])
            for k, p in zip(keys, new_p):
                after[k] = p
            box[0] = after

        def fwd():
            with torch.no_grad():
                O.forward(O.clone_state_dict(box[0]), images, b["bboxes"], b["additional_feats"],
                          b["context_indices"], cfg, False)

        res = {}
        for name, fn in (("fwd_bwd_adam", train), ("fwd", fwd)):
            for _ in range(warm):
                fn()
            ts = []
            for _ in range(timed):
                t0 = time.perf_counter()
                fn()
                ts.append(time.perf_counter() - t0)
            res[name] = {"pages_per_s": round(pages / statistics.median(ts), 4),
                         "s_per_step": round(statistics.median(ts), 3), "warmup": warm, "timed": timed}
        legs.append((pages, res))
        out["batch_%d" % pages] = res
    out["value"] = legs[-1][1]["fwd_bwd_adam"]["pages_per_s"]
    out["sample"] = ("%s workload (%dx%d, %d boxes, K=%d, dropout masks drawn per step), torch-CPU fp32 oracle; value = "
                     "16-page batches, forward+backward+Adam, median of %d steps after %d warm-up; 2-page batches: "
                     "%d after %d" % (wl["name"], wl["H"], wl["W"], wl["boxes"], 2 * wl["cs"],
                                      legs[-1][1]["fwd"]["timed"], legs[-1][1]["fwd"]["warmup"], 5, 3))
    return out


# ------------------------------------------------------------------------------------ per-kernel clock
def clock_leg(args):
    """Effective shader clock of the step's big kernels on THIS box: a short run of this script (3 timed steps) under
    `rocprofv3 --pmc GRBM_GUI_ACTIVE` (counters only: no tracing domain beside them), clock = counter / 8 XCDs / launch
    duration -- the recipe of MI355X_MICROARCH.md.  The MFMA-dense kernels run power-limited (conv1 forward 1.96 GHz on the
    round-5 boxes against 2.35 GHz for the memory-bound passes): a box on which conv1 forward takes 1.0 instead of 0.77 ms
    shows here whether it is the clock.  Failures are reported, never raised."""
    import shutil
    import sqlite3
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    tmp = tempfile.mkdtemp(prefix="cova_clk_", dir="/tmp")
    cmd = [exe, "--pmc", "GRBM_GUI_ACTIVE", "-d", tmp, "--", sys.executable, os.path.abspath(__file__), "--steps", "3", "--warmup", "2",
           "--no-cpu-baseline", "--sustained-seconds", "0", "--no-ab", "--no-clock-leg", "--config", str(args.config)]
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        dbs = [os.path.join(d, f) for d, _, fs in os.walk(tmp) for f in fs if f.endswith(".db")]
        if not dbs:
            return {"error": "no rocprofv3 database (rc %d): %s" % (r.returncode, r.stderr.decode(errors="replace")[-300:])}
        db = sqlite3.connect(dbs[0])
        rows = db.execute("select kernel_name, count(*), avg(value), avg(duration), sum(duration) from counters_collection "
                          "where counter_name = 'GRBM_GUI_ACTIVE' group by kernel_name order by sum(duration) desc").fetchall()
        out = {"how": "GRBM_GUI_ACTIVE / 8 XCDs / launch duration over every launch of a 2 + 3 step run under rocprofv3 --pmc "
                      "(launch times under the counter pass run ~1 % long)", "kernels": []}
        for name, n, val, dur, _ in rows[:14]:
            name = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            out["kernels"].append({"kernel": name[:72], "launches": n, "mean_us": round(dur / 1e3, 1),
                                   "effective_ghz": round(val / 8.0 / dur, 3)})
        return out
    except Exception as e:          # noqa: BLE001
        return {"error": "%s: %s" % (type(e).__name__, e)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


# ------------------------------------------------------------------------------------ launch
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks ourselves."""
    env = dict(os.environ)      # (fewer GPUs than ranks: the preflight of every rank reports it; COVA_BENCH_BACKEND=gloo
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))     #  opts into the shared-device code-path check)


# ------------------------------------------------------------------------------------ N > 1 first-contact hardening
class Guard:
    """Makes sure a multi-rank run ends with ONE JSON line on rank 0 instead of a hang: every phase has a deadline
    (`--collective-timeout`, also the process group's timeout); when it passes, when another rank dies (the launcher
    then SIGTERMs this one) or when this rank raises, rank 0 prints {"error": ...} in the bench line's shape and the
    process exits non-zero."""

    def __init__(self, args, rank, world):
        self.args, self.rank, self.world = args, rank, world
        self.done = False
        self.phase = "start"
        self.deadline = None
        self.lock = threading.Lock()
        if rank == 0:
            signal.signal(signal.SIGTERM, lambda *_: self.fail("terminated by the launcher (another rank failed or was killed) "
                                                               "during phase '%s'" % self.phase, 4))
        t = threading.Thread(target=self._watch, daemon=True)
        t.start()

    def enter(self, phase, seconds=None):
        """Name the phase; multi-rank runs also arm its deadline (a single rank has no collective to hang in, and its
        CPU-baseline leg legitimately takes minutes)."""
        self.phase = phase
        self.deadline = time.monotonic() + (seconds or self.args.collective_timeout) if self.world > 1 else None

    def _watch(self):
        while not self.done:
            time.sleep(0.5)
            d = self.deadline
            if d is not None and time.monotonic() > d and not self.done:
                self.fail("phase '%s' exceeded --collective-timeout %.0f s on rank %d (a collective that never completed?)"
                          % (self.phase, self.args.collective_timeout, self.rank), 3)

    def fail(self, msg, code):
        with self.lock:
            if not self.done:
                self.done = True
                if self.rank == 0:
                    print(json.dumps({"metric": "webpages/sec fwd+bwd (90 bboxes, K=24)", "value": None, "unit": "webpages/s",
                                      "n_gpus": self.world, "steps": self.args.steps, "warmup": self.args.warmup,
                                      "higher_is_better": True, "error": msg}), flush=True)
                else:
                    print("bench.py rank %d: %s" % (self.rank, msg), file=sys.stderr, flush=True)
        os._exit(code)

    def finish(self):
        self.done = True


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, choices=sorted(WORKLOADS),
                    help="BASELINE.json config, 1-based (default 2 = configs[1], the quoted one)")
    ap.add_argument("--pages", type=int, default=0, help="pages per GPU (default: the config's)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--global-pages", type=int, default=256, help="--scaling strong: global batch")
    ap.add_argument("--roi-op", choices=("pool", "align"), default="pool",
                    help="pool = the reference's RoIPool (the metric's configuration); align = the RoIAlign variant")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ab", action="store_true", help="skip the A/B legs (f32-MFMA forms of conv1 / the 3x3 main loop)")
    ap.add_argument("--no-clock-leg", action="store_true",
                    help="skip the per-kernel effective-clock leg (a 5-step child run under rocprofv3 --pmc GRBM_GUI_ACTIVE)")
    ap.add_argument("--cpu-baseline-quick", action="store_true",
                    help="16-page CPU leg with 1 warm-up + 3 timed steps (default: 3 + 5 as SURVEY.md 8d asks)")
    ap.add_argument("--sustained-seconds", type=float, default=5.0,
                    help="length of the `sustained` leg after the headline measurement (0 = skip)")
    ap.add_argument("--sync-bn", action="store_true",
                    help="N > 1: BatchNorm statistics over the whole data-parallel batch (default: per rank, as DDP)")
    ap.add_argument("--collective-timeout", type=float, default=300.0,
                    help="deadline (s) of every phase of a multi-rank run and of the process group's collectives: past it "
                         "rank 0 prints a JSON line with \"error\" instead of hanging")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ:
        self_launch(args)
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    guard = Guard(args, rank, world)
    try:
        run(args, guard, rank, local_rank, world)
    except SystemExit:
        raise
    except BaseException as e:          # noqa: BLE001 -- whatever it is, the line goes out and the other ranks get torn down
        traceback.print_exc()
        guard.fail("rank %d, phase '%s': %s: %s" % (rank, guard.phase, type(e).__name__, e), 2)


def run(args, guard, rank, local_rank, world):
    import torch
    import cova_amd  # noqa: F401
    from cova_web_object_detection_amd import _lib, synthetic, weights
    from cova_web_object_detection_amd.trainer import HotPathTrainer, shard_pages

    # ---- preflight: one GPU per rank (COVA_BENCH_BACKEND=gloo opts into the shared-device DIAGNOSTIC run of the code path)
    guard.enter("preflight")
    n_dev = torch.cuda.device_count()
    if n_dev < 1:
        raise RuntimeError("no GPU visible (torch.cuda.device_count() == 0)")
    backend = os.environ.get("COVA_BENCH_BACKEND", "nccl")
    if backend == "nccl" and world > n_dev:
        raise RuntimeError("--gpus %d needs %d visible GPUs, torch.cuda.device_count() = %d (set COVA_BENCH_BACKEND=gloo "
                           "for a shared-device code-path check; its throughput is meaningless)" % (world, world, n_dev))
    if backend != "nccl":
        local_rank %= n_dev
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    group = None
    ranks_seen = 1
    if world > 1:
        import torch.distributed as dist
        guard.enter("init_process_group")
        tmo = datetime.timedelta(seconds=args.collective_timeout)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device, timeout=tmo)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)
        # first contact: one all-reduce of ones over the data-path backend; every rank must see the world size
        guard.enter("checksum all-reduce")
        ones = torch.ones((1,), device=device, dtype=torch.float32)
        dist.all_reduce(ones)
        torch.cuda.synchronize()
        ranks_seen = int(round(float(ones.item())))
        if ranks_seen != world:
            raise RuntimeError("checksum all-reduce of ones returned %d, expected the world size %d" % (ranks_seen, world))

    guard.enter("model + batch + warm-up")
    wl = dict(WORKLOADS[args.config])
    if args.scaling == "strong":
        lo, hi = shard_pages(args.global_pages, rank, world)
        pages, global_pages = hi - lo, args.global_pages
    else:
        pages = args.pages or wl["pages"]
        global_pages = pages * world
    cfg = model_cfg(wl)
    if args.roi_op == "align":
        cfg.update(roi_op="align", sampling_ratio=2, roi_aligned=False)
        wl["desc"] += ", RoIAlign (sampling_ratio 2) instead of RoIPool"
    sd = weights.seeded_state_dict(123, **weight_cfg(cfg))
    trainer = HotPathTrainer(cfg, sd, device, world_size=world, process_group=group, dropout_seed=123 + rank,
                             sync_bn=args.sync_bn)
    batch = make_device_batch(123 + rank, device, pages, args.config)
    n_boxes = batch["bboxes"].shape[0]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([x], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return x

    for _ in range(args.warmup):
        trainer.train_step(batch)
    barrier()
    guard.enter("timed steps")
    timed = ["cova_conv3x3_wino4_full", "cova_conv3x3_wino4_full_tail", "cova_conv1_fwd_tail", "cova_conv3x3_wgrad4_partial",
             "cova_conv3x3_wgrad4_finish", "cova_conv1_fwd", "cova_conv1_wgrad_poolbwd", "cova_conv1_wgrad",
             "cova_bn_relu_maxpool_fwd", "cova_conv1x1", "cova_conv1x1_wgrad", "cova_bn_act_fwd", "cova_bn_act_fwd_bits",
             "cova_bn_act2_fwd", "cova_roipool_fwd_bn", "cova_roipool_bwd_bn", "cova_roipool_bwd_bn_tail", "cova_bn1d_fwd",
             "cova_bn1d_bwd", "cova_sgemm", "cova_gat_fwd", "cova_gat_bwd"]
    timed = [n for n in timed if n in _lib.lib().protos]
    # Inside the timed steps only the DOMINANT family's launches are bracketed by HIP events (8 per such step: the roofline's
    # live measurement); the other kernels of `other_kernels` are timed in a separate short leg behind it -- bracketing all
    # ~45 launches of a step cost the headline 2 % (round 4's line against its own `sustained` leg)
    dominant = [n for n in ("cova_conv3x3_wino4_full", "cova_conv3x3_wino4_full_tail") if n in timed]
    # ... and only in every EVENT_EVERY-th timed step (steps 0, 10, ...: 14 pairs per such step, ~0.3 % of the 20 steps' time): an event pair around a launch is two marker packets in
    # the stream, ~20 us of lost back-to-back dispatch each pair -- with all 8 x 20 launches bracketed the headline of round 5's
    # last pass read 9.25 ms beside 8.93 ms of the same kernels in the event-free `ab.default` leg.  The same steps also bracket
    # conv1 forward, conv1 weight gradient and the four 3x3 weight gradients (6 more pairs): the kernels whose time differs
    # from box to box (round 5: conv1 forward 0.75 ms on the builder's boxes, 1.01 ms on the driver's), IN the step.
    EVENT_EVERY = 1 if os.environ.get("COVA_PROFILE_ALL") else 10
    in_step = dominant + [n for n in ("cova_conv1_fwd_tail", "cova_conv1_fwd", "cova_conv1_wgrad_poolbwd",
                                      "cova_conv3x3_wgrad4_partial") if n in timed]
    prof = {name: [] for name in (_lib.lib().protos if os.environ.get("COVA_PROFILE_ALL") else in_step)}
    t0 = time.perf_counter()
    for i in range(args.steps):
        _lib.PROFILE = prof if i % EVENT_EVERY == 0 else None
        loss, _ = trainer.train_step(batch)
    barrier()
    dt = time.perf_counter() - t0
    _lib.PROFILE = None
    if not os.environ.get("COVA_PROFILE_ALL"):
        _lib.PROFILE = {name: [] for name in timed if name not in in_step}
        for _ in range(min(args.steps, 5)):
            trainer.train_step(batch)
        barrier()
        prof.update(_lib.PROFILE)
        _lib.PROFILE = None
    dt_rank = dt
    dt = max_over_ranks(dt)
    loss_val = float(loss.item())
    per_rank_ms, exposed_ms = [round(1e3 * dt_rank / args.steps, 3)], [round(trainer.exposed_allreduce_ms(), 4)]
    if world > 1:
        import torch.distributed as dist
        gathered = [None] * world
        dist.all_gather_object(gathered, (per_rank_ms[0], exposed_ms[0]))
        per_rank_ms, exposed_ms = [g[0] for g in gathered], [g[1] for g in gathered]

    def mean_ms(names, pred=None):
        ev = [p for n in names for p in prof.get(n, []) if pred is None or pred(p[2])]
        return (sum(p[0].elapsed_time(p[1]) for p in ev) / len(ev), len(ev)) if ev else (0.0, 0)

    if os.environ.get("COVA_PROFILE_ALL") and rank == 0:
        rows = [(sum(p[0].elapsed_time(p[1]) for p in v) / args.steps, len(v) / args.steps, k) for k, v in prof.items() if v]
        for ms, n, k in sorted(rows, reverse=True):
            print("%-36s %5.1f calls/step %8.3f ms/step" % (k, n, ms), file=sys.stderr)
        print("sum %.3f ms/step" % sum(r[0] for r in rows), file=sys.stderr)

    # sustained leg: >= N seconds of back-to-back train steps (reported separately from the headline)
    sustained = None
    if args.sustained_seconds > 0:
        guard.enter("sustained leg", args.collective_timeout + 2 * args.sustained_seconds)
        # step count fixed up front from the headline rate (+5 %: every rank runs the same number of steps and the leg
        # must last AT LEAST the stated time)
        n_s = max(args.steps, int(1.05 * args.sustained_seconds / max(dt / args.steps, 1e-4)) + 2)
        barrier()
        t_s = time.perf_counter()
        for _ in range(n_s):
            trainer.train_step(batch)
        barrier()
        dt_s = max_over_ranks(time.perf_counter() - t_s)
        sustained = {"value": round(global_pages * n_s / dt_s, 2), "unit": "webpages/s", "steps": n_s,
                     "seconds": round(dt_s, 2), "ms_per_step": round(1e3 * dt_s / n_s, 3)}

    # forward only (eval mode, running statistics): the second number SURVEY.md section 8d asks for
    guard.enter("forward-only leg")
    for _ in range(2):
        trainer.predict(batch)
    barrier()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        trainer.predict(batch)
    barrier()
    dt_fwd = max_over_ranks(time.perf_counter() - t1)

    # A/B legs in the same process, same trainer (the engine queries workspace sizes per step): the f32-MFMA forms of the
    # kernels that run on the bf16 matrix pipe by default -- conv1 forward + weight gradient (cova_set_option 7) and the
    # F(4x4,3x3) forward / data-gradient main loop (option 9) -- each 3 warm-up + 10 timed steps
    ab = {}
    if world == 1 and not args.no_ab:
        guard.enter("A/B legs")
        pre7, pre9 = os.environ.get("COVA_CONV1_F32") == "1", os.environ.get("COVA_W4_F32") == "1"
        for key, opt, pre, what in (("default", 0, False, "the headline's kernels, timed like the other two legs (no events)"),
                                    ("conv1_f32", 7, pre7, "conv1 forward + weight gradient on v_mfma_f32_32x32x2_f32"),
                                    ("wino4_f32", 9, pre9, "3x3 forward + data gradient on v_mfma_f32_16x16x4_f32")):
            if pre:
                continue
            if opt:
                _lib.query("cova_set_option", opt, 1)
            try:
                for _ in range(3):
                    trainer.train_step(batch)
                barrier()
                ta = time.perf_counter()
                for _ in range(10):
                    trainer.train_step(batch)
                barrier()
                dta = time.perf_counter() - ta
            finally:
                if opt:
                    _lib.query("cova_set_option", opt, 0)
            ab[key] = {"value": round(pages * 10 / dta, 2), "unit": "webpages/s", "ms_per_step": round(1e3 * dta / 10, 3),
                       "steps": 10, "what": what + (", everything else as the headline" if opt else "")}
        for _ in range(2):                      # back on the default kernels before the next leg
            trainer.train_step(batch)
        barrier()

    # the drop-in nn.Module route with the reference's loop cadence (train.py:45-60: zero_grad, forward,
    # argmax + .item(), CE-sum + .item(), backward, torch.optim.Adam.step) -- two host reads per step
    dropin = None
    guard.enter("report")
    if world == 1 and args.config in (2, 3) and args.roi_op == "pool":
        import contextlib
        import warnings
        from cova_web_object_detection_amd.models import CoVA
        # (the constructor prints its parameter count like the reference's does, models.py:92: keep stdout = one JSON line)
        with warnings.catch_warnings(), contextlib.redirect_stdout(sys.stderr):
            warnings.simplefilter("ignore")
            m = CoVA((3, 3), wl["H"], 4, True, 384, 32, 0, 0.2, None, backbone=wl["backbone"],
                     n_heads=wl["n_heads"], n_gat_layers=wl["n_gat_layers"])
        m.load_state_dict(sd)
        m = m.to(device).train()
        opt = torch.optim.Adam(m.parameters(), lr=5e-4, weight_decay=1e-3)
        crit = torch.nn.CrossEntropyLoss(reduction="sum")

        def ref_step():
            opt.zero_grad()
            out = m(batch["images"], batch["bboxes"], batch["additional_feats"], batch["context_indices"])
            n_ok = (out.argmax(dim=1) == batch["labels"]).sum().item()
            ls = crit(out, batch["labels"])
            lv = ls.item()
            ls.backward()
            opt.step()
            return n_ok, lv

        nd = max(3, min(args.steps, 10))
        for _ in range(2):
            ref_step()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(nd):
            ref_step()
        torch.cuda.synchronize()
        dtd = time.perf_counter() - t2
        dropin = {"value": round(pages * nd / dtd, 2), "unit": "webpages/s", "ms_per_step": round(1e3 * dtd / nd, 3),
                  "steps": nd, "mode": "models.CoVA drop-in module + torch.optim.Adam, train.py:45-60 cadence "
                                       "(two .item() host reads per step)"}
        del m, opt

    if rank == 0:
        ms_per_step = 1e3 * dt / args.steps
        value = global_pages * args.steps / dt
        fm = flop_model(wl)
        px_pages = pages
        w4_ms, w4_n = mean_ms(["cova_conv3x3_wino4_full", "cova_conv3x3_wino4_full_tail"])
        use4 = True                                                   # (F(4x4,3x3) is the only 3x3 family of the product since round 6)
        split = os.environ.get("COVA_W4_F32") != "1"                 # the bf16 three-piece main loop (default since round 5)
        conv_ms, conv_n, ratio = w4_ms, w4_n, WINO4_RATIO
        kname = "conv3x3_c64_wino4s_kernel" if split else "conv3x3_c64_wino4_kernel"
        alg = fm["conv3_launch_per_page"] * px_pages                     # algorithmic FLOPs per launch
        executed = alg / ratio                                            # f32 multiply-adds the Winograd form executes
        map_bytes = 4 * 64 * px_pages * (wl["H"] // 4) * (wl["W"] // 4)
        # Compulsory bytes of the launches AS THEY RAN: every timed launch's variant and operand count follow from which
        # arguments of its call were non-NULL (w4_variant: besides one input and one output map the fused variants read the
        # residual-branch gradient and the mask / xhat operands of their epilogues).  ResNet-18 step: forward
        # 4 x 2 maps, conv2 data gradient 2 x 3, conv1 data gradient 4 + 4 1/32 = 22 maps over 8 launches.
        vrows, alg_bytes_total, conv_ms_total = {}, 0.0, 0.0
        if use4:
            for n in ("cova_conv3x3_wino4_full", "cova_conv3x3_wino4_full_tail"):
                for p in prof.get(n, []):
                    targs, pro, maps, role = w4_variant(n, p[3])
                    # (the two-tensor prologue runs the f32 main loop whatever the default is)
                    kn = ("conv3x3_c64_wino4s_kernel" if split and pro != 2 else "conv3x3_c64_wino4_kernel") + targs
                    e = vrows.setdefault(kn, {"variant": kn, "role": role, "launches": 0, "maps": round(maps, 4), "ms": 0.0})
                    e["launches"] += 1
                    e["ms"] += p[0].elapsed_time(p[1])
        roof = {"kernel": kname + " (forward + data-gradient launches of the step)", "launches_timed": conv_n}
        if conv_n:
            traffic, src, step_traffic, tv = read_traffic(kname, px_pages)
            variants = []
            for kn, e in sorted(vrows.items()):
                vb = e["maps"] * map_bytes
                ms = e["ms"] / e["launches"]
                row = {"variant": kn, "role": e["role"], "launches": e["launches"], "maps": e["maps"],
                       "algorithmic_bytes": int(vb), "avg_ms": round(ms, 4), "achieved_gb_per_s": round(vb / ms / 1e6, 1),
                       "frac": round(vb / ms / 1e6 / (PEAK_HBM_TBS * 1e3), 4), "traffic": tv.get(kn)}
                if tv.get(kn):
                    row["traffic_over_algorithmic"] = round(tv[kn] / vb, 3)
                variants.append(row)
                alg_bytes_total += vb * e["launches"]
                conv_ms_total += e["ms"]
            if variants:
                alg_bytes = int(alg_bytes_total / conv_n)                 # mean over the launches timed
                if all(r["traffic"] for r in variants):                  # family traffic = the same launch mix as `achieved`
                    traffic = sum(r["traffic"] * r["launches"] for r in variants) / conv_n
                    src = src.replace(":step_families", ":step_kernels")
            else:
                alg_bytes = 2 * map_bytes
            f32_equiv = executed / conv_ms / 1e9                          # TFLOP/s of f32 multiply-adds executed
            if split:
                # Six bf16-MFMA products per f32 multiply-add at 16x the f32-MFMA rate: the matrix floor of a launch is
                # 6 * executed / 2.5 PFLOP/s = 0.07 ms against algorithmic_bytes / 8 TB/s = 0.14 ms -- in this
                # formulation the launch is bounded by HBM, and that is the roof it is priced against.
                ach = alg_bytes / conv_ms / 1e6                           # GB/s  (= sum of bytes / sum of time over the launches)
                roof.update(bound="hbm", algorithm="winograd F(4x4,3x3); transform-domain products on the bf16 matrix pipe, f32 "
                            "operands as three bf16 pieces, six products accumulated in f32 (v_mfma_f32_16x16x32_bf16)",
                            achieved=round(ach, 1), peak=PEAK_HBM_TBS * 1e3, unit="GB/s", frac=round(ach / (PEAK_HBM_TBS * 1e3), 4),
                            hbm_algorithmic_frac=round(ach / (PEAK_HBM_TBS * 1e3), 4),
                            achieved_is="ALGORITHMIC bytes of the launches timed (per-variant operand table `variants`, derived from "
                                        "each call's arguments) / their total time",
                            floors_ms={"hbm_algorithmic_bytes_at_8_TBps": round(alg_bytes / PEAK_HBM_TBS / 1e9, 4),
                                       "bf16_mfma_six_products_at_2.5_PFLOPs": round(SPLIT_PRODUCTS * executed / PEAK_BF16_MFMA_TFLOPS / 1e9, 4),
                                       "weight_stream_L2_to_registers_at_64_B_per_clk_CU": round(
                                           884736.0 * (px_pages * (wl["H"] // 4) * (wl["W"] // 4) / 256.0) / (256 * 64 * 2.4e9) * 1e3, 4)},
                            mfma={"pipe": "bf16", "executed_tflops": round(SPLIT_PRODUCTS * f32_equiv, 1), "peak": PEAK_BF16_MFMA_TFLOPS,
                                  "frac": round(SPLIT_PRODUCTS * f32_equiv / PEAK_BF16_MFMA_TFLOPS, 4),
                                  "note": "six bf16 products per f32 multiply-add of the F(4x4) form"},
                            f32_equivalent={"executed_tflops": round(f32_equiv, 2), "frac_of_f32_mfma_peak": round(f32_equiv / PEAK_F32_MFMA_TFLOPS, 4),
                                            "note": "the round-4 line's `frac` (f32 multiply-adds executed over the 157.3 TFLOP/s the f32-MFMA "
                                                    "main loop is bounded by); the wino4_f32 A/B leg runs that loop"})
            else:
                roof.update(bound="mfma", algorithm="winograd F(4x4,3x3), exact f32 MFMA (v_mfma_f32_16x16x4_f32)",
                            achieved=round(f32_equiv, 2), peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s",
                            frac=round(f32_equiv / PEAK_F32_MFMA_TFLOPS, 4),
                            achieved_is="EXECUTED MFMA FLOP/s (algorithmic / %g)" % ratio)
            if traffic:
                roof.update(hbm_achieved_tb_per_s=round(traffic / conv_ms / 1e9, 3), hbm_frac=round(traffic / conv_ms / 1e9 / PEAK_HBM_TBS, 4),
                            traffic_over_algorithmic=round(traffic / alg_bytes, 3))
            if step_traffic:
                roof.update(step_traffic_bytes=int(step_traffic))
            roof.update(avg_launch_ms=round(conv_ms, 4), executed_flop_per_launch=int(executed), algorithmic_flop_per_launch=int(alg),
                        algorithmic_achieved=round(alg / conv_ms / 1e9, 2),
                        algorithmic_frac_of_direct_conv_peak=round(alg / conv_ms / 1e9 / PEAK_F32_MFMA_TFLOPS, 4),
                        traffic=traffic, traffic_unit="B/launch", traffic_source=src, algorithmic_bytes=alg_bytes,
                        algorithmic_bytes_plain_launch=2 * map_bytes, variants=variants,
                        profiling="HIP events on the launching stream INSIDE the timed steps, in every %d-th of them (steps 0, %d, ...), "
                                  "around this family's launches and around conv1 forward, conv1 weight gradient and the 3x3 weight "
                                  "gradients (`other_kernels.*.in_step`): an event pair costs ~20 us of back-to-back dispatch; the "
                                  "remaining kernels of `other_kernels` are timed in %d separate steps behind them; `sustained` and "
                                  "the `ab` legs run without events" % (EVENT_EVERY, EVENT_EVERY, min(args.steps, 5)))
        step_alg = fm["total"] * pages                                   # per rank
        # executed multiply-adds: every 3x3 convolution at the Winograd share of the kernel that ran
        wg_ratio = WINO4_RATIO
        step_exec = (fm["total"] - fm["wino"] + fm["wino"] / 3 / wg_ratio + 2 * fm["wino"] / 3 / ratio) * pages
        step = {"algorithmic_gflop_per_page": round(fm["total"] / 1e9, 2),
                "algorithmic_tflops": round(step_alg / (ms_per_step * 1e-3) / 1e12, 2),
                "frac_of_direct_ceiling": round(step_alg / (ms_per_step * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                "executed_gflop_per_page": round(step_exec / pages / 1e9, 2),
                "frac_of_executed_floor": round(step_exec / (ms_per_step * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                "note": "per GPU; executed = 3x3 convolutions counted at Winograd's share of the direct multiplies (weight "
                        "gradient 1/%g, forward and data gradient 1/%g), priced against the f32-MFMA peak although conv1 and the "
                        "3x3 forward / data-gradient launches run six bf16 products per multiply-add on the 16x faster pipe"
                        % (wg_ratio, ratio)}
        others = {}
        hw = (wl["H"] // 4) * (wl["W"] // 4)
        f_conv1 = 2 * 64 * 147 * pages * (wl["H"] // 2) * (wl["W"] // 2)
        c1_split = os.environ.get("COVA_CONV1_F32") != "1"
        specs = [("conv1_7x7_fwd", ["cova_conv1_fwd", "cova_conv1_fwd_tail"], None, f_conv1, 0, c1_split),
                 ("conv1_7x7_wgrad_with_pool_backward", ["cova_conv1_wgrad_poolbwd", "cova_conv1_wgrad"], None, f_conv1, 0, c1_split),
                 ("conv3x3_wgrad_winograd_f4x4", ["cova_conv3x3_wgrad4_partial"], None,
                  fm["conv3_launch_per_page"] * pages / WINO4_RATIO, 0),
                 ("conv3x3_wgrad_finish_all_convs", ["cova_conv3x3_wgrad4_finish"], None, 0, 0),
                 ("bn_act_fwd_bits", ["cova_bn_act_fwd_bits"], None, 0, pages * 64 * 4 * 3 * hw),
                 ("roipool_fwd_lazy_feature", ["cova_roipool_fwd_bn"], None, 0, 0),
                 ("roipool_bwd_with_bn_tail", ["cova_roipool_bwd_bn", "cova_roipool_bwd_bn_tail"], None, 0, 0),
                 ("bn_relu_maxpool_fwd", ["cova_bn_relu_maxpool_fwd"], None, 0,
                  pages * 64 * (4 * 4 * hw + (4 + 4 + 1) * hw))]
        if wl["backbone"] == "resnet50":
            R = pages * hw
            for cin, cout in ((64, 64), (64, 256), (256, 64)):
                specs.append(("conv1x1_%d_to_%d_fwd_dgrad" % (cin, cout), ["cova_conv1x1"],
                              (lambda a, ci=cin, co=cout: a[-2:] == (ci, co)), 2 * cin * cout * R, 4 * (cin + cout) * R))
                specs.append(("conv1x1_wgrad_%dx%d" % (cout, cin), ["cova_conv1x1_wgrad"],
                              (lambda a, ci=cin, co=cout: a[-2:] == (co, ci)), 2 * cin * cout * R, 4 * (cin + cout) * R))
        for spec in specs:
            key, names, pred, flop, nbytes = spec[:5]
            on_bf16 = len(spec) > 5 and spec[5]
            ms, n = mean_ms(names, pred)
            if n:
                o = {"avg_launch_ms": round(ms, 4), "launches_timed": n, "in_step": any(x in in_step for x in names)}
                if flop and on_bf16:       # six bf16 products per f32 multiply-add, against the pipe the kernel uses
                    o.update(f32_equivalent_tflops=round(flop / ms / 1e9, 1), executed_tflops_bf16_pipe=round(SPLIT_PRODUCTS * flop / ms / 1e9, 1),
                             frac_of_bf16_mfma_peak=round(SPLIT_PRODUCTS * flop / ms / 1e9 / PEAK_BF16_MFMA_TFLOPS, 3))
                elif flop:
                    o.update(executed_tflops=round(flop / ms / 1e9, 1), frac_of_f32_mfma_peak=round(flop / ms / 1e9 / PEAK_F32_MFMA_TFLOPS, 3))
                if nbytes:
                    o.update(compulsory_tb_per_s=round(nbytes / ms / 1e9, 2), frac_of_hbm_peak=round(nbytes / ms / 1e9 / PEAK_HBM_TBS, 3))
                others[key] = o
        diag = "" if backend == "nccl" else " (DIAGNOSTIC: %s backend, ranks share GPUs; throughput meaningless)" % backend
        out = {
            "metric": "webpages/sec fwd+bwd (90 bboxes, K=24)", "value": round(value, 3),
            "unit": "webpages/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: synthetic %dx%d (HxW) screenshots, %d pages/GPU, %d boxes/page, K=%d, %s, "
                                   "train step = fwd+CE+bwd+allreduce+Adam, dropout 0.2"
                                   % (wl["name"], wl["H"], wl["W"], pages, wl["boxes"], 2 * wl["cs"], wl["desc"]),
                       "baseline_config": args.config, "pages_per_gpu": pages, "global_pages": global_pages,
                       "boxes_per_gpu": n_boxes, "world_size": world, "collective_backend": "rccl" if backend == "nccl" else backend,
                       "rccl_ranks_seen": ranks_seen, "gpus_visible": n_dev,
                       "parallelism": "dp%d" % world + ("+syncbn" if args.sync_bn and world > 1 else "") + diag,
                       "arithmetic": "f32 throughout (weights, activations, gradients, transforms, accumulation).  3x3 weight gradients, "
                                     "1x1 convolutions, GEMMs: f32 MFMA.  conv1 (7x7) forward and weight gradient%s and the 3x3 forward / "
                                     "data-gradient launches' transform-domain products%s: each f32 operand as three round-to-nearest bf16 "
                                     "pieces (<= 2^-26 left), six bf16-MFMA products accumulated in f32 -- error against fp64 at or below "
                                     "the f32-MFMA kernels' (tests/test_kernels_gpu.py::test_conv1_bf16_split_error_class, "
                                     "::test_conv3x3_winograd_f4x4_split_error_class), same parity gates; the `ab` legs time the f32-MFMA forms"
                                     % (" [here: f32-MFMA kernels, COVA_CONV1_F32=1]" if os.environ.get("COVA_CONV1_F32") == "1" else "",
                                        " [here: f32-MFMA main loop, COVA_W4_F32=1]" if os.environ.get("COVA_W4_F32") == "1" else ""),
                       "loss": round(loss_val, 3)},
            "roofline": roof, "step": step, "other_kernels": others,
            "forward_only": {"value": round(global_pages * args.steps / dt_fwd, 2), "unit": "webpages/s",
                             "ms_per_step": round(1e3 * dt_fwd / args.steps, 3),
                             "mode": "eval forward (running statistics) + per-box argmax, same batch"},
        }
        if ab:
            out["ab"] = ab
        if sustained:
            out["sustained"] = sustained
        out["per_rank"] = {"ms_per_step": per_rank_ms, "allreduce_exposed_ms": exposed_ms,
                           "note": "allreduce_exposed_ms = HIP-event time of optimizer_step's collective waits per step "
                                   "(0 on one rank)"}
        if dropin:
            out["dropin_module_loop"] = dropin
        if world == 1 and not args.no_clock_leg and args.roi_op == "pool":
            torch.cuda.synchronize()
            out["clock_leg"] = clock_leg(args)
        if world == 1 and not args.no_cpu_baseline and args.roi_op == "pool":
            out["cpu_baseline"] = cpu_baseline(wl, not args.cpu_baseline_quick)
        print(json.dumps(out), flush=True)
    guard.finish()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
