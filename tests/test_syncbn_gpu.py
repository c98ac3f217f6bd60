"""Data-parallel step with real HIP kernels in TWO processes: both on cuda:0 with the collectives through gloo
(which stages device tensors through the host -- a 1-GPU box cannot host a 2-rank RCCL group), and, wherever two GPUs
are visible, one rank per GPU over RCCL.

With sync_bn=True a 2-rank step over page shards must equal the single-process step over the whole
batch ("exact large-batch mode"): loss, every parameter after Adam, every
running statistic.  Without it, the ranks' BatchNorm statistics are local (reference DDP semantics):
each rank's gradient then equals the CPU oracle's on that rank's shard, and the exchanged sum is
checked against the sum of the two oracle gradients.  Both runs go through the trainer's overlapped
two-phase gradient exchange at world_size 2."""
import os
import socket
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.abspath(os.path.dirname(__file__)))
import cova_amd  # noqa: E402,F401  (spawned workers re-import this module without conftest.py)

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cova_web_object_detection_amd import synthetic, weights
from cova_web_object_detection_amd.trainer import HotPathTrainer, shard_batch

pytestmark = pytest.mark.gpu

CFG = dict(roi_output_size=(3, 3), n_classes=4, use_context=True, hidden_dim=48, bbox_hidden_dim=16,
           n_additional_feat=0, drop_prob=0.0)
WCFG = {k: v for k, v in CFG.items() if k != "drop_prob"}
STEPS = 2
EXT = dict(backbone="resnet50", n_heads=2)     # the extension model (linear-form Bottleneck backward)


def _cfgs(arch):
    cfg = dict(CFG, **EXT) if arch == "resnet50" else dict(CFG)
    return cfg, {k: v for k, v in cfg.items() if k != "drop_prob"}


def _batch():
    return synthetic.make_batch(4, img_h=128, boxes_per_page=[21, 34, 9, 40], context_size=5, seed=23)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir, sync_bn, arch="resnet18", backend="gloo"):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dev = "cuda:%d" % rank if backend == "nccl" else "cuda:0"       # RCCL: one GPU per rank; gloo: both on cuda:0
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg, wcfg = _cfgs(arch)
        shard = {k: v.to(dev) for k, v in shard_batch(_batch(), rank, world).items()}
        # rank 1 starts from DIFFERENT weights: the trainer's start-up broadcast must make the replicas equal
        tr = HotPathTrainer(cfg, weights.seeded_state_dict(23 + 100 * rank, **wcfg), dev, world_size=world,
                            sync_bn=sync_bn)
        # this rank's own gradient (no exchange): a single-process trainer on the shard
        solo = HotPathTrainer(cfg, weights.seeded_state_dict(23, **wcfg), dev)
        solo.forward_backward(shard)
        local = solo.gbucket.flat.clone()
        ref_states = os.path.join(out_dir, "ref_state_%d.pt")
        losses, grads = [], []
        for i in range(STEPS):
            if os.path.exists(ref_states % i):
                # every step starts from the single-process run's state at that step: the comparison of step i is then
                # not blurred by Adam's +-lr moves on noise-level gradients in the steps before it
                st = torch.load(ref_states % i)
                tr.load_state_dict(st["sd"])
                tr.load_optimizer_state_dict(st["opt"])
            loss, _ = tr.forward_backward(shard)
            tr.optimizer_step()
            grads.append(tr.gbucket.flat.clone().cpu())          # after the exchange: sum over ranks
            losses.append(float(loss))
        torch.cuda.synchronize()
        torch.save(dict(losses=losses, local=local.cpu(), grads=grads,
                        sd={k: v.cpu() for k, v in tr.state_dict().items()},
                        exposed_ms=tr.exposed_allreduce_ms()),
                   os.path.join(out_dir, "r%d.pt" % rank))
    finally:
        dist.destroy_process_group()


def _run(tmp_path, sync_bn, arch="resnet18", backend="gloo"):
    port = _free_port()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    mp.spawn(_worker, args=(2, port, str(tmp_path), sync_bn, arch, backend), nprocs=2, join=True)
    return [torch.load(os.path.join(str(tmp_path), "r%d.pt" % r)) for r in range(2)]


def _reference_run(tmp_path, arch):
    """The single-process run on the concatenated batch; its state BEFORE every step goes to files the ranks load."""
    dev = "cuda:0"
    cfg, wcfg = _cfgs(arch)
    full = {k: v.to(dev) for k, v in _batch().items() if torch.is_tensor(v)}
    ref = HotPathTrainer(cfg, weights.seeded_state_dict(23, **wcfg), dev)
    out = dict(losses=[], grads=[])
    for i in range(STEPS):
        torch.save(dict(sd={k: v.cpu() for k, v in ref.state_dict().items()},
                        opt={k: (v.cpu() if torch.is_tensor(v) else v) for k, v in ref.optimizer_state_dict().items()}),
                   os.path.join(str(tmp_path), "ref_state_%d.pt" % i))
        loss, _ = ref.forward_backward(full)
        out["grads"].append(ref.gbucket.flat.clone().cpu())
        ref.optimizer_step()
        out["losses"].append(float(loss))
    out["sd"] = {k: v.cpu() for k, v in ref.state_dict().items()}
    out["offsets"] = ref.gbucket.offsets
    return out


def _check_syncbn(r0, r1, ref, arch):
    # the ranks agree with each other exactly (same all-reduced gradient, same Adam) ...
    for g0, g1 in zip(r0["grads"], r1["grads"]):
        assert torch.equal(g0, g1)
    for k in r0["sd"]:
        assert torch.equal(r0["sd"][k], r1["sd"][k]), k
    # ... and EVERY step (started from the single-process run's state) with the single-process step on the concatenated
    # batch to fp32 re-association accuracy: a wrong count ratio in any step is an O(1) error
    for i in range(STEPS):
        tot = r0["losses"][i] + r1["losses"][i]
        assert abs(tot - ref["losses"][i]) <= 2e-4 * abs(ref["losses"][i]), (i, tot, ref["losses"][i])
        g_ref = ref["grads"][i]
        scale = float(g_ref.abs().max())
        d = (r0["grads"][i] - g_ref).double().abs()
        # The two runs finalize their BatchNorm statistics in different kernels (the single process in the producing
        # launches' tails and the one-launch BatchNorm1d, the ranks in stand-alone kernels around the all-reduce of the
        # fp32 rows): their scale / shift agree to ~1e-5 (E[x^2] - mean^2 in fp64 from fp32 partial sums), so a
        # pre-activation within ~1e-7 of zero may gate differently and moves a handful of gradient entries by ~1e-3 of
        # the scale (tools/tail_ab.py shows the case this batch hits: ONE of the decoder's 66,560
        # ReLU gates, at y = 3e-8, which moves one row of decoder.1.weight's gradient).  The bulk is held to the tight
        # bound, the peak to a loose one (which statistics rows a launch geometry produces decides which near-ties flip); the
        # deeper stack takes ~4x more ReLU decisions.
        # Which near-tie falls the other way depends on the last bits of every kernel in front of it: with conv1 on the
        # f32-MFMA kernels (COVA_CONV1_F32=1) this batch shows none (0.999-quantile 1e-6, max 6e-6 of the scale), with
        # the bf16-split ones the decoder gate above flips and, through the pooled gradient, moves 30 % of conv1's weight
        # gradient by 2e-4 .. 1e-3 (0.9-quantile 1.4e-5, 0.99 9e-5, 0.999 3.9e-4, max 1.0e-3).  The kernels themselves do not
        # depend on the partition: tests/test_kernels_gpu.py::test_conv1_kernels_do_not_depend_on_the_batch_partition.
        q = tuple(float(torch.quantile(d[::7].float(), p_)) / scale for p_ in (0.5, 0.9, 0.99, 0.999))
        print("syncbn %s step %d: |d|/scale quantiles 0.5 %.2e  0.9 %.2e  0.99 %.2e  0.999 %.2e  max %.2e"
              % ((arch, i) + q + (float(d.max()) / scale,)))
        if os.environ.get("COVA_SYNCBN_DIAG"):
            rows = sorted(((float(d[o:o + m].max()) / scale, int((d[o:o + m] > 2e-4 * scale).sum()), m, k)
                           for k, (o, m, _) in ref["offsets"].items()), reverse=True)
            for r in rows[:12]:
                print("    %-40s max %.2e  beyond 2e-4: %6d of %6d" % (r[3], r[0], r[1], r[2]))
        assert q[0] <= 1e-6 and q[1] <= 2e-4, (i, arch, q)
        assert float(torch.quantile(d[::7].float(), 0.999)) <= 1e-3 * scale, i
        assert float(d.max()) <= 1e-2 * scale, (i, float(d.max()) / scale)     # (seen: 1e-3 ResNet-18 model, 5.3e-3 ResNet-50 stack)
    # state after the last step: each step started from identical parameters, so one Adam step separates the runs
    for k, v in ref["sd"].items():
        if not v.is_floating_point():
            assert torch.equal(r0["sd"][k], v), k
        elif k.endswith(("running_mean", "running_var")):
            assert torch.allclose(r0["sd"][k], v, rtol=2e-4, atol=2e-5), k
        else:
            # Adam moves a weight by <= 3.2 lr per step; entries whose gradient is rounding noise around zero may step
            # the other way, the ones with a solid gradient must agree far better
            dd = (r0["sd"][k] - v).abs()
            assert float(dd.max()) <= 2 * 3.2 * 5e-4 + 1e-5, (k, float(dd.max()))
            off = ref["offsets"][k]
            gk = ref["grads"][-1][off[0]:off[0] + off[1]].view(dd.shape)
            solid = gk.abs() > 1e-3 * float(ref["grads"][-1].abs().max())
            if bool(solid.any()):
                assert float(dd[solid].mean()) <= 0.02 * 5e-4 + 1e-6, k


@pytest.mark.parametrize("arch", ["resnet18", "resnet50"])
def test_two_rank_step_with_syncbn_equals_single_process_large_batch(tmp_path, arch):
    ref = _reference_run(tmp_path, arch)
    r0, r1 = _run(tmp_path, True, arch)
    _check_syncbn(r0, r1, ref, arch)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: a 2-rank RCCL group, one rank per device")
@pytest.mark.parametrize("sync_bn", [True, False])
def test_two_rank_step_over_rccl(tmp_path, sync_bn):
    """The same two assertions with the collectives on RCCL (backend "nccl"), one rank per GPU over xGMI -- runs
    wherever `pytest -m gpu` finds two devices; the 1-GPU boxes skip it and keep the gloo variants."""
    if sync_bn:
        ref = _reference_run(tmp_path, "resnet18")
        r0, r1 = _run(tmp_path, True, "resnet18", backend="nccl")
        _check_syncbn(r0, r1, ref, "resnet18")
    else:
        r0, r1 = _run(tmp_path, False, backend="nccl")
        assert torch.equal(r0["grads"][0], r1["grads"][0])
        assert torch.equal(r0["local"] + r1["local"], r0["grads"][0])
    assert r0["exposed_ms"] >= 0.0


def test_two_rank_step_with_local_batchnorm_sums_the_shard_gradients(tmp_path):
    from oracle import cova_oracle as O
    r0, r1 = _run(tmp_path, False)
    assert torch.equal(r0["grads"][0], r1["grads"][0])
    summed = r0["local"] + r1["local"]
    # no float atomics anywhere in the step (RoIPool / GAT backward gather in a fixed order): exact
    assert torch.equal(summed, r0["grads"][0])
    # each rank's local gradient against the oracle on its shard
    sd = weights.seeded_state_dict(23, **WCFG)
    tr = HotPathTrainer(CFG, sd, "cpu")            # (flat layout only; no device work)
    for rank, res in ((0, r0), (1, r1)):
        sh = shard_batch(_batch(), rank, 2)
        _, _, grads, _, _ = O.loss_and_grads(sd, sh["images"], sh["bboxes"], sh["additional_feats"],
                                             sh["context_indices"], sh["labels"], CFG, None)
        got = {k: res["local"][o:o + m].view(shape) for k, (o, m, shape) in tr.gbucket.offsets.items()}
        # heads do not depend on the max-pool / RoIPool routing: tight; the conv stack gets the loose
        # bound of tests/test_model_gpu.py (one near-tie flip moves its gradients by ~1e-3); the
        # routing-forced tight comparison of the conv stack lives there
        gscale = max(float(g.abs().max()) for g in grads.values())
        for k, g in got.items():
            scale = max(float(grads[k].abs().max()), 0.01 * gscale)
            err = float((g - grads[k]).abs().max()) / scale
            assert err < (5e-2 if k.startswith("convnet.") else 2e-3), (rank, k, err)
