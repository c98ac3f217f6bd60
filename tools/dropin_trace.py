"""The drop-in nn.Module route with the reference's loop cadence (train.py:45-60), a few steps -- for `rocprofv3 --kernel-trace`:
    rocprofv3 --kernel-trace -d /tmp/dt -- python tools/dropin_trace.py ; python tools/dropin_trace.py --report <results.db>
--report: one step (prep_wino4 to prep_wino4) in launch order with the gap in front of every launch, GPU busy vs wall, and the
time of the torch kernels (optimizer, loss, accuracy) beside the library's own."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if "--report" in sys.argv:
    import re, sqlite3
    db = sqlite3.connect(sys.argv[sys.argv.index("--report") + 1])
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kt = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    names = dict(cur.execute("select id, display_name from %s" % ks))
    rows = list(cur.execute("select start, end, kernel_id from %s order by start" % kt))
    marks = [i for i, r in enumerate(rows) if "prep_wino4_kernel" in names[r[2]]]
    a, b = marks[-3], marks[-2]
    seg = rows[a:b]
    wall = (rows[b][0] - rows[a][0]) / 1e3
    busy = sum(e - s for s, e, _ in seg) / 1e3
    torch_t = sum(e - s for s, e, k in seg if "at::" in names[k] or "rocclr" in names[k]) / 1e3
    ntorch = sum(1 for s, e, k in seg if "at::" in names[k] or "rocclr" in names[k])
    print("# one drop-in step: %d launches, wall %.1f us, GPU busy %.1f us (torch kernels: %d launches, %.1f us), idle %.1f us"
          % (len(seg), wall, busy, ntorch, torch_t, wall - busy))
    prev = rows[a][0]
    for s, e, k in seg:
        n = names[k].replace("(anonymous namespace)::", "").replace("void ", "")
        gap = (s - prev) / 1e3
        if gap > 4.0 or "at::" in n:
            print("%9.1f %8.1f %7.1f  %s" % ((s - rows[a][0]) / 1e3, (e - s) / 1e3, gap, n[:90]))
        prev = e
    sys.exit(0)

import contextlib, warnings
import torch
import cova_amd  # noqa: F401
from cova_web_object_detection_amd import synthetic, weights
from cova_web_object_detection_amd.models import CoVA

dev = "cuda:0"
cfg = dict(roi_output_size=(3, 3), n_classes=4, use_context=True, hidden_dim=384, bbox_hidden_dim=32, n_additional_feat=0)
sd = weights.seeded_state_dict(1, **cfg)
batch = synthetic.make_batch(16, img_h=1280, boxes_per_page=[90] * 16, context_size=12, seed=1)
batch = {k: v.to(dev) for k, v in batch.items() if torch.is_tensor(v)}
with warnings.catch_warnings(), contextlib.redirect_stdout(sys.stderr):
    warnings.simplefilter("ignore")
    m = CoVA((3, 3), 1280, 4, True, 384, 32, 0, 0.2, None)
m.load_state_dict(sd)
m = m.to(dev).train()
opt = torch.optim.Adam(m.parameters(), lr=5e-4, weight_decay=1e-3)
crit = torch.nn.CrossEntropyLoss(reduction="sum")
import time
for i in range(int(os.environ.get("STEPS", 8))):
    if i == 3:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    opt.zero_grad()
    out = m(batch["images"], batch["bboxes"], batch["additional_feats"], batch["context_indices"])
    n_ok = (out.argmax(dim=1) == batch["labels"]).sum().item()
    ls = crit(out, batch["labels"])
    lv = ls.item()
    ls.backward()
    opt.step()
torch.cuda.synchronize()
print("drop-in loop: %.3f ms per step" % ((time.perf_counter() - t0) / (int(os.environ.get("STEPS", 8)) - 3) * 1e3))
