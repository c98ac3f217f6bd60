"""Synthetic webpage batches in the exact layout ``datasets.custom_collate_fn`` emits.

Input contract of the hot path (reference datasets.py:183-190): a batch is
``(img_ids, images[B,3,H,W] f32 in [0,1), bboxes[N,5] f32 = [page_idx,x1,y1,x2,y2] pixels,
additional_feats[N,A] f32, context_indices[N,2*cs] i64, labels[N] i64)``.

* context window: for box i of a page with n boxes the neighbours are
  ``max(0,i-cs)..i-1`` then ``i+1..min(n,i+cs+1)-1``, padded with -1 to 2*cs
  (datasets.py:117-128), then shifted by the number of boxes of the preceding pages,
  -1 pads untouched (datasets.py:170-178).
* labels: 0 = BG and exactly one box each of classes 1..3 per page (README.md:17).

Everything is drawn from numpy ``RandomState`` so that the same seed gives bit-identical
batches on the dev container and on the GPU box.
"""
import numpy as np
import torch


def context_window_indices(n, context_size):
    """[n, 2*context_size] int64 page-local neighbour table (datasets.py:117-128)."""
    k = 2 * context_size
    out = np.full((n, k), -1, dtype=np.int64)
    for i in range(n):
        ctx = list(range(max(0, i - context_size), i)) + \
            list(range(i + 1, min(n, i + context_size + 1)))
        out[i, :len(ctx)] = ctx
    return out


def collate_context(per_page_ctx):
    """Concatenate page-local tables, offsetting valid ids (datasets.py:170-178)."""
    outs, seen = [], 0
    for ctx in per_page_ctx:
        c = ctx.copy()
        c[c != -1] += seen
        outs.append(c)
        seen += ctx.shape[0]
    return np.concatenate(outs, axis=0)


def _draw_boxes(rs, counts, img_h, img_w, context_size, n_additional_feat, n_classes,
                border_fraction):
    boxes, ctxs, labels = [], [], []
    for p, n in enumerate(counts):
        bw = rs.uniform(8, min(400, img_w), n)
        bh = rs.uniform(8, min(200, img_h), n)
        x1 = rs.uniform(0, 1, n) * np.maximum(img_w - bw, 1.0)
        y1 = rs.uniform(0, 1, n) * np.maximum(img_h - bh, 1.0)
        cross = rs.uniform(0, 1, n) < border_fraction
        x1 = np.where(cross, img_w - 0.5 * bw, x1)
        y1 = np.where(cross, img_h - 0.5 * bh, y1)
        b = np.stack([np.full(n, p, dtype=np.float64), x1, y1, x1 + bw, y1 + bh], axis=1)
        boxes.append(b.astype(np.float32))
        ctxs.append(context_window_indices(n, context_size))
        lab = np.zeros(n, dtype=np.int64)
        pos = rs.permutation(n)[:n_classes - 1]
        lab[pos] = np.arange(1, n_classes)[:len(pos)]
        labels.append(lab)
    n_total = sum(counts)
    addl = rs.standard_normal((n_total, n_additional_feat)).astype(np.float32)
    return dict(
        bboxes=torch.from_numpy(np.concatenate(boxes, 0)),
        additional_feats=torch.from_numpy(addl),
        context_indices=torch.from_numpy(collate_context(ctxs)),
        labels=torch.from_numpy(np.concatenate(labels, 0)),
    )


def _counts(n_pages, boxes_per_page):
    counts = [boxes_per_page] * n_pages if np.isscalar(boxes_per_page) else list(boxes_per_page)
    assert len(counts) == n_pages
    return counts


def make_boxes_only(n_pages, img_h, img_w, boxes_per_page=90, context_size=12, seed=123,
                    n_additional_feat=0, n_classes=4, border_fraction=0.05):
    """Everything of a batch except the pixels (bench.py draws those on the device)."""
    rs = np.random.RandomState(seed)
    return _draw_boxes(rs, _counts(n_pages, boxes_per_page), img_h, img_w, context_size,
                       n_additional_feat, n_classes, border_fraction)


def make_batch(n_pages, img_h=1280, img_w=None, boxes_per_page=90, context_size=12,
               n_additional_feat=0, n_classes=4, seed=123, border_fraction=0.05,
               device=None):
    """Seeded synthetic batch.

    ``boxes_per_page`` may be an int or a per-page sequence (ragged batches, 11..230 per
    splits/bbox_stats.txt).  About ``border_fraction`` of the boxes cross the right/bottom
    image border so that RoIPool clamping is exercised.
    """
    img_w = img_h if img_w is None else img_w
    rs = np.random.RandomState(seed)
    counts = _counts(n_pages, boxes_per_page)
    images = rs.random_sample((n_pages, 3, img_h, img_w)).astype(np.float32)
    batch = _draw_boxes(rs, counts, img_h, img_w, context_size, n_additional_feat, n_classes,
                        border_fraction)
    batch["img_ids"] = np.arange(n_pages).astype(str)
    batch["images"] = torch.from_numpy(images)
    if device is not None:
        for k, v in batch.items():
            if torch.is_tensor(v):
                batch[k] = v.to(device)
    return batch
