"""Device-side input pipeline and attention export around the hot path.

``DeviceCollate`` replaces the tensor work of ``WebDataset.__getitem__`` + ``custom_collate_fn``
(reference datasets.py:94-132,159-190): the host hands over uint8 HWC pixels and the raw
``x,y,w,h,label`` rows; ToTensor (/255, CHW), the xywh->xyxy conversion, the page-index column,
the labels and the batch-global context-window table are produced on the GPU
(cova_images_u8_to_f32, cova_collate_boxes).  4x fewer bytes over PCIe than fp32 images.

``attention_rows`` is the dump of extract_attn_wts_and_visualize.py:104-135.
"""
import numpy as np
import torch

from . import engine
from ._lib import call


class DeviceCollate:
    def __init__(self, context_size, device, n_additional_feat=0, pin=False):
        assert context_size >= 0
        self.cs, self.device, self.A = int(context_size), torch.device(device), int(n_additional_feat)
        self.pin = bool(pin)            # stage host arrays in pinned memory: H2D copies become asynchronous

    def __call__(self, u8_pages, rows_per_page, additional_feats=None):
        """u8_pages: uint8 [B,H,W,3] (numpy or torch, host or device); rows_per_page: list of
        float32 [n,5] arrays.  Returns the batch dict the trainer / CoVA.forward consume."""
        u8 = torch.as_tensor(np.ascontiguousarray(u8_pages) if isinstance(u8_pages, np.ndarray)
                             else u8_pages)
        assert u8.dtype == torch.uint8 and u8.dim() == 4 and u8.shape[3] == 3
        B, H, W, _ = u8.shape
        assert len(rows_per_page) == B
        counts = [int(np.asarray(r).reshape(-1, 5).shape[0]) for r in rows_per_page]
        N = sum(counts)
        rows = np.concatenate([np.asarray(r, dtype=np.float32).reshape(-1, 5) for r in rows_per_page], 0) \
            if N else np.zeros((0, 5), np.float32)
        offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
        dev = self.device
        host = (lambda t: t.pin_memory()) if self.pin else (lambda t: t)
        u8 = (host(u8) if u8.device.type == "cpu" else u8).to(dev, non_blocking=True).contiguous()
        rows_d = host(torch.from_numpy(rows)).to(dev, non_blocking=True)
        offs_d = host(torch.from_numpy(offs)).to(dev, non_blocking=True)
        images = torch.empty((B, 3, H, W), dtype=torch.float32, device=dev)
        call("cova_images_u8_to_f32", u8, images, B, H, W)
        bboxes = torch.empty((N, 5), dtype=torch.float32, device=dev)
        labels = torch.empty((N,), dtype=torch.int64, device=dev)
        K = 2 * self.cs
        ctx = torch.empty((N, K) if K else (0, 0), dtype=torch.int64, device=dev)   # datasets.py:130
        call("cova_collate_boxes", rows_d, offs_d, B, N, self.cs, bboxes, labels, ctx if K else None)
        if additional_feats is None:
            addl = torch.empty((N, 0), dtype=torch.float32, device=dev)
        else:
            addl = torch.as_tensor(additional_feats, dtype=torch.float32).to(dev).contiguous()
        return dict(images=images, bboxes=bboxes, additional_feats=addl, context_indices=ctx,
                    labels=labels, page_start=offs_d.to(torch.int64))


class Prefetcher:
    """Iterates device batches while the NEXT one is uploaded and collated on a side stream.

    ``source`` yields ``(u8_pages, rows_per_page)`` (or with a third ``additional_feats`` item).  The
    uint8 upload (19.7 MB per 1280x1280 page fp32 -> 4.9 MB) and the two collate kernels of batch i+1
    run on their own HIP stream under the train step of batch i; ``__next__`` makes the consumer's
    stream wait on the upload's event -- the PCIe time never shows in the step time."""

    def __init__(self, collate, source):
        self.collate, self.it = collate, iter(source)
        self.stream = torch.cuda.Stream(device=collate.device)
        self._pending = None
        self._preload()

    def _preload(self):
        try:
            item = next(self.it)
        except StopIteration:
            self._pending = None
            return
        with torch.cuda.stream(self.stream):
            batch = self.collate(*item)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self._pending = (batch, ev)

    def __iter__(self):
        return self

    def __next__(self):
        if self._pending is None:
            raise StopIteration
        batch, ev = self._pending
        cur = torch.cuda.current_stream(self.collate.device)
        cur.wait_event(ev)
        for v in batch.values():
            if torch.is_tensor(v):
                v.record_stream(cur)         # allocated on the side stream, consumed on this one
        self._preload()
        return batch


@torch.no_grad()
def attention_rows(trainer, batch):
    """float32 [M, 5+5K] rows for the boxes with label > 0, eval mode (running statistics)."""
    _, sv = engine.model_fwd(trainer.cfg, trainer.params, trainer.buffers, batch["images"],
                             batch["bboxes"], batch["additional_feats"], batch["context_indices"],
                             False, save=True)
    attn, ctx = sv["gat"][-1]["heads"][0]["attn"], batch["context_indices"]
    N, K = ctx.shape
    out = torch.empty((N, 5 + 5 * K), dtype=torch.float32, device=attn.device)
    call("cova_attn_export_rows", batch["bboxes"], ctx, attn, batch["labels"], N, K, out)
    return out[batch["labels"] > 0]
