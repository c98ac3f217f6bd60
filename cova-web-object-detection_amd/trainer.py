"""Training step of the hot path (train.py:42-60) as one object: forward, CE-sum, backward,
data-parallel gradient exchange, Adam -- all device work through libcova_hip.so, the gradient
exchange through torch.distributed (backend "nccl" == RCCL over xGMI on ROCm).

Data parallelism: pages are independent units, so each rank takes whole
pages; the model (1.6 M parameters) is replicated.  Every parameter lives in ONE flat fp32 buffer
and every gradient in ONE flat bucket, so a step needs exactly one all-reduce (6.47 MB, latency
bound on xGMI) and one Adam launch.  The loss is a SUM over boxes (main.py:139), so SUM-reduced
gradients equal the single-device large-batch gradient except for BatchNorm, whose batch
statistics stay per-rank (standard DDP behaviour; the reference has no SyncBN).  ``sync_bn=True``
switches on the exact large-batch mode: every BatchNorm statistic row is summed over the ranks
(engine.StatSync, 2*C floats per message), after which a data-parallel step equals the
single-device step on the concatenated batch.
"""
from collections import OrderedDict

import os

import torch

from . import engine
from .weights import state_dict_spec

_BUF_SUFFIX = ("running_mean", "running_var", "num_batches_tracked")


def is_param_key(k):
    return not k.endswith(_BUF_SUFFIX)


class FlatBucket:
    """One contiguous fp32 buffer with a named view per tensor (device agnostic)."""

    def __init__(self, shapes, device):
        self.offsets, n = OrderedDict(), 0
        for k, shape in shapes.items():
            numel = int(torch.Size(shape).numel())
            self.offsets[k] = (n, numel, tuple(shape))
            n += (numel + 3) // 4 * 4           # keep every view 16-byte aligned
        self.flat = torch.zeros(n, dtype=torch.float32, device=device)
        self.views = OrderedDict((k, self.flat[o:o + m].view(shape))
                                 for k, (o, m, shape) in self.offsets.items())

    def all_reduce_sum(self, group=None):
        """One collective for the whole bucket."""
        import torch.distributed as dist
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)

    def split_at(self, first_key):
        """Flat offset at which tensor ``first_key`` starts (the bucket keeps state_dict order)."""
        return self.offsets[first_key][0]

    def all_reduce_range(self, lo, hi, group=None, async_op=False):
        """SUM all-reduce of flat[lo:hi]; with async_op the collective runs on the backend's own
        stream (RCCL) and the returned work handle must be waited on before the range is read."""
        import torch.distributed as dist
        return dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def shard_pages(n_pages, rank, world_size):
    """Contiguous page range [lo, hi) of this rank (whole pages only)."""
    base, rem = divmod(n_pages, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(batch, rank, world_size):
    """Cut a collated batch (datasets.py:183-190 layout) into this rank's pages: boxes keep their
    order, page indices and context indices are re-based to the shard."""
    n_pages = batch["images"].shape[0]
    lo, hi = shard_pages(n_pages, rank, world_size)
    page = batch["bboxes"][:, 0]
    sel = (page >= lo) & (page < hi)
    first = int(torch.nonzero(sel)[0]) if bool(sel.any()) else 0
    bb = batch["bboxes"][sel].clone()
    bb[:, 0] -= lo
    ctx = batch["context_indices"][sel].clone()
    ctx[ctx >= 0] -= first
    return dict(images=batch["images"][lo:hi].contiguous(), bboxes=bb,
                additional_feats=batch["additional_feats"][sel].contiguous(),
                context_indices=ctx, labels=batch["labels"][sel].contiguous())


class HotPathTrainer:
    """Owns flat parameters / gradients / Adam moments on one GPU and runs training steps."""

    def __init__(self, cfg, state_dict, device, lr=5e-4, weight_decay=1e-3, betas=(0.9, 0.999),
                 eps=1e-8, world_size=1, process_group=None, dropout_seed=123, sync_bn=False):
        self.cfg = dict(cfg)
        self.device = torch.device(device)
        spec = state_dict_spec(**{k: cfg[k] for k in ("roi_output_size", "n_classes", "use_context",
                                                     "hidden_dim", "bbox_hidden_dim", "n_additional_feat",
                                                     "backbone", "n_heads", "n_gat_layers") if k in cfg})
        pshapes = OrderedDict((k, s) for k, s in spec if is_param_key(k))
        self.pbucket = FlatBucket(pshapes, self.device)
        self.gbucket = FlatBucket(pshapes, self.device)
        self.params, self.grads = self.pbucket.views, self.gbucket.views
        self.exp_avg = torch.zeros_like(self.pbucket.flat)
        self.exp_avg_sq = torch.zeros_like(self.pbucket.flat)
        self.buffers = {}
        for k, _ in spec:
            v = state_dict[k]
            if is_param_key(k):
                self.params[k].copy_(v)
            else:
                self.buffers[k] = v.clone().to(self.device)
        self.hp = dict(lr=lr, weight_decay=weight_decay, betas=betas, eps=eps)
        self.world_size, self.group = world_size, process_group
        self.step_count, self.dropout_seed = 0, int(dropout_seed)
        self.sync_bn = bool(sync_bn) and world_size > 1
        self._ar_events = []              # (start, end) HIP events around the collective waits of optimizer_step
        self.measure_allreduce = True     # record them (up to 4096 steps; exposed_allreduce_ms() drains the list)
        if world_size > 1:
            self.broadcast_state(src=0)

    def broadcast_state(self, src=0):
        """Rank `src`'s parameters, BatchNorm buffers, Adam moments and step count to every rank: replicas start (and
        resume) from ONE state even when the ranks were built from rank-local checkpoints."""
        import torch.distributed as dist
        for t in [self.pbucket.flat, self.exp_avg, self.exp_avg_sq] + [self.buffers[k] for k in sorted(self.buffers)]:
            dist.broadcast(t, src=src, group=self.group)
        step = torch.tensor([self.step_count], dtype=torch.int64, device=self.device)
        dist.broadcast(step, src=src, group=self.group)
        self.step_count = int(step.item())

    def exposed_allreduce_ms(self):
        """Mean time per step the stream spent in optimizer_step's gradient collectives (HIP events on the compute
        stream around the all-reduce calls / waits: what the overlap did not hide); 0 on a single rank."""
        if not self._ar_events:
            return 0.0
        torch.cuda.synchronize(self.device)
        ms = sum(a.elapsed_time(b) for a, b in self._ar_events) / len(self._ar_events)
        self._ar_events = []
        return ms

    def state_dict(self):
        sd = OrderedDict()
        for k in list(self.params) + list(self.buffers):
            sd[k] = (self.params[k] if k in self.params else self.buffers[k]).detach().clone()
        return sd

    def _all_ranks_ok(self, ok, what):
        """Collective: every rank learns whether ALL ranks validated `what`; raises the same error everywhere instead of
        leaving the ranks that did validate blocked in the broadcast that follows."""
        import torch.distributed as dist
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        if int(flag.item()) == 0:
            raise KeyError("%s: at least one rank could not validate its argument (this rank: %s)" % (what, "ok" if ok else "failed"))

    def load_state_dict(self, state_dict, broadcast=False):
        """Restore parameters and BatchNorm buffers from a reference-layout state_dict -- the save-best /
        reload-best cycle of train.py:84,94 (`torch.save(model.state_dict())`, `load_state_dict(torch.load(...))`).
        The flat buffers keep their addresses (views, moments and bucket stay valid).

        Local by default: a rank-0-only reload ("reload best, evaluate on rank 0") involves no collective.  With
        `broadcast=True` the call IS a collective that EVERY rank must make: the keys are validated on all ranks first
        (one all-reduce of an ok flag, so a KeyError on one rank raises on all of them), then rank 0's parameters and
        buffers -- what was loaded, not the Adam moments or the step count -- replace every rank's."""
        missing = [k for k in list(self.params) + list(self.buffers) if k not in state_dict]
        # sizes are part of the validation every rank agrees on BEFORE any copy / broadcast: a tensor of the wrong size on
        # one rank raising inside copy_ would leave the other ranks blocked in dist.broadcast
        bad = [k for k in list(self.params) + list(self.buffers)
               if k in state_dict and (not torch.is_tensor(state_dict[k]) or
                                       state_dict[k].numel() != (self.params[k] if k in self.params else self.buffers[k]).numel())]
        if broadcast and self.world_size > 1:
            self._all_ranks_ok(not missing and not bad, "load_state_dict")
        if missing:
            raise KeyError("state_dict lacks %s" % missing[:4])
        if bad:
            raise ValueError("state_dict entries of the wrong size: %s" % bad[:4])
        for k, p in self.params.items():
            p.copy_(state_dict[k].to(p.device).view_as(p))
        for k in self.buffers:
            self.buffers[k].copy_(state_dict[k].to(self.device).view_as(self.buffers[k]))
        if broadcast and self.world_size > 1:
            import torch.distributed as dist
            for t in [self.pbucket.flat] + [self.buffers[k] for k in sorted(self.buffers)]:
                dist.broadcast(t, src=0, group=self.group)

    def optimizer_state_dict(self):
        """Adam moments + step count (the reference keeps no optimizer state in its checkpoints, train.py:84;
        with this a run can be resumed exactly)."""
        return dict(step=self.step_count, exp_avg=self.exp_avg.detach().clone(),
                    exp_avg_sq=self.exp_avg_sq.detach().clone(), hp=dict(self.hp))

    def load_optimizer_state_dict(self, state, broadcast=False):
        """Adam moments, step count and hyper-parameters.  Local by default; `broadcast=True` makes it a collective every
        rank must call (validated on all ranks first) after which every rank holds rank 0's moments, step count and
        hyper-parameters."""
        ok = all(k in state for k in ("step", "exp_avg", "exp_avg_sq"))
        sized = ok and all(torch.is_tensor(state[k]) and state[k].numel() == self.exp_avg.numel() for k in ("exp_avg", "exp_avg_sq"))
        if broadcast and self.world_size > 1:
            self._all_ranks_ok(ok and sized, "load_optimizer_state_dict")       # (sizes too: see load_state_dict)
        if not ok:
            raise KeyError("optimizer state lacks one of step / exp_avg / exp_avg_sq")
        if not sized:
            raise ValueError("optimizer state moments of the wrong size (expected %d elements)" % self.exp_avg.numel())
        self.step_count = int(state["step"])
        self.exp_avg.copy_(state["exp_avg"].to(self.device).view_as(self.exp_avg))
        self.exp_avg_sq.copy_(state["exp_avg_sq"].to(self.device).view_as(self.exp_avg_sq))
        self.hp.update(state.get("hp", {}))
        if broadcast and self.world_size > 1:
            import torch.distributed as dist
            for t in (self.exp_avg, self.exp_avg_sq):
                dist.broadcast(t, src=0, group=self.group)
            b1, b2 = self.hp["betas"]
            meta = torch.tensor([float(self.step_count), self.hp["lr"], self.hp["weight_decay"], b1, b2, self.hp["eps"]],
                                dtype=torch.float64, device=self.device)
            dist.broadcast(meta, src=0, group=self.group)
            m = meta.tolist()
            self.step_count = int(m[0])
            self.hp.update(lr=m[1], weight_decay=m[2], betas=(m[3], m[4]), eps=m[5])

    def forward_backward(self, batch, masks=None):
        """Forward + CE(sum) + backward into the flat gradient bucket.  Returns (loss, pred)."""
        # With SyncBN a one-box shard is legal (the statistics are over the whole batch, as torch.nn.SyncBatchNorm
        # accepts it): the train-mode "more than 1 value per channel" check then applies to the GLOBAL box count, which
        # _stat_sync has from its all-reduce -- every rank raises together instead of one rank leaving the others
        # blocked in a collective.
        engine.check_batch(self.cfg, batch["images"], batch["bboxes"], batch["additional_feats"],
                           batch["context_indices"], not self.sync_bn)
        self.step_count += 1
        base = (self.dropout_seed * 0x9E3779B1 + 2 * self.step_count) & 0xFFFFFFFFFFFF
        if self.sync_bn:
            engine.STAT_SYNC = self._stat_sync(batch)
        try:
            logits, sv = engine.model_fwd(self.cfg, self.params, self.buffers, batch["images"],
                                          batch["bboxes"], batch["additional_feats"],
                                          batch["context_indices"], True, (base, base + 1), masks)
            loss, dl, pred = engine.ce_sum(logits, batch["labels"])
            self._head_work = None
            overlap = self.world_size > 1 and engine.OPTIONS.overlap_allreduce
            engine.model_bwd(sv, dl, self.params, self.grads,
                             after_head=self._reduce_head if overlap else None)
        finally:
            engine.STAT_SYNC = None
        return loss, pred

    def _stat_sync(self, batch):
        """SyncBN bookkeeping of one step: whole-batch / local element-count ratios for the page-shaped
        (conv stack) and box-shaped (BatchNorm1d) statistics; one tiny all-reduce + host read."""
        import torch.distributed as dist
        n_pages, n_boxes = int(batch["images"].shape[0]), int(batch["bboxes"].shape[0])      # host ints: no device read
        total = torch.tensor([n_pages, n_boxes], dtype=torch.float64, device=self.device)
        dist.all_reduce(total, op=dist.ReduceOp.SUM, group=self.group)
        tot = total.tolist()                                                                   # the step's one host read
        r = [tot[0] / max(float(n_pages), 1.0), tot[1] / max(float(n_boxes), 1.0)]
        if tot[1] == 1.0:
            raise ValueError("Expected more than 1 value per channel when training (1 box in the whole batch)")
        return engine.StatSync(self.group, r[0], r[1])

    # Gradient exchange: the head (positional encoder, GAT, decoder = 96 % of the 6.5 MB bucket, the
    # tail of the flat buffer) is complete before the conv-stack backward starts, so its all-reduce is
    # issued there and runs over xGMI under ~6 ms of convolutions; only the 0.6 MB conv-stack part
    # is exchanged at the end of the step.
    def _head_offset(self):
        for k in self.gbucket.offsets:               # state_dict order: conv stack first
            if not k.startswith("convnet."):
                return self.gbucket.split_at(k)
        return self.gbucket.flat.numel()

    def _reduce_head(self):
        lo = self._head_offset()
        if lo >= self.gbucket.flat.numel():
            return
        self._head_work = self.gbucket.all_reduce_range(lo, self.gbucket.flat.numel(), self.group,
                                                        async_op=True)

    def optimizer_step(self):
        if self.world_size > 1:
            timing = self.measure_allreduce and len(self._ar_events) < 4096
            if timing:                      # events on THIS trainer's device / stream (not the process' current device)
                st = torch.cuda.current_stream(self.device)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(st)
            if getattr(self, "_head_work", None) is not None:
                self.gbucket.all_reduce_range(0, self._head_offset(), self.group)
                self._head_work.wait()
                self._head_work = None
            else:
                self.gbucket.all_reduce_sum(self.group)
            if timing:
                e1.record(st)
                self._ar_events.append((e0, e1))
        b1, b2 = self.hp["betas"]
        engine.call("cova_adam_step", self.pbucket.flat, self.gbucket.flat, self.exp_avg,
                    self.exp_avg_sq, self.pbucket.flat.numel(), self.step_count, self.hp["lr"], b1, b2,
                    self.hp["eps"], self.hp["weight_decay"])

    def train_step(self, batch, masks=None):
        """optimizer.zero_grad(); forward; loss; backward; optimizer.step()  (train.py:45-60).
        Gradients are fully overwritten each step, so zero_grad is implicit."""
        loss, pred = self.forward_backward(batch, masks)
        self.optimizer_step()
        return loss, pred

    @torch.no_grad()
    def evaluate(self, batch, page_start, k=1):
        """Eval-mode decisions of train.py:131-154 for one batch.  ``page_start`` int64 [n_pages+1]
        box offsets.  Returns (topk [n_pages, n_classes, k] page-local box indices, best first;
        correct [n_pages, n_classes-1] bool = the labelled box of class c is among the top k)."""
        logits, _ = self.predict(batch)
        n_pages, nc = page_start.numel() - 1, logits.shape[1]
        topk = torch.empty((n_pages, nc, k), dtype=torch.int64, device=logits.device)
        engine.call("cova_page_class_topk", logits, page_start.contiguous(), n_pages, nc, k, topk)
        # train.py:146: the labelled box of class c is the FIRST box of that class within the page;
        # a page without one scores False (the reference would raise an IndexError there)
        labels = batch["labels"]
        n_boxes = labels.numel()
        page_of = torch.searchsorted(page_start[1:].contiguous(), torch.arange(n_boxes, device=labels.device),
                                     right=True)
        local = torch.arange(n_boxes, device=labels.device) - page_start[page_of]
        correct = []
        for c in range(1, nc):
            first = torch.full((n_pages,), n_boxes, dtype=torch.int64, device=labels.device)
            sel = labels == c
            first.scatter_reduce_(0, page_of[sel], local[sel], reduce="amin")
            has = first < n_boxes
            correct.append(has & (topk[:, c, :] == first.view(-1, 1)).any(dim=1))
        return topk, torch.stack(correct, dim=1)

    @torch.no_grad()
    def predict(self, batch):
        """Eval-mode forward (running statistics) -> (logits, per-box argmax)."""
        engine.check_batch(self.cfg, batch["images"], batch["bboxes"], batch["additional_feats"],
                           batch["context_indices"], False)
        logits, _ = engine.model_fwd(self.cfg, self.params, self.buffers, batch["images"],
                                     batch["bboxes"], batch["additional_feats"],
                                     batch["context_indices"], False, save=False)
        _, _, pred = engine.ce_sum(logits, None, want_grad=False)
        return logits, pred
