"""Gaps of build/gap_probe's launch sequences out of a rocprofv3 kernel trace (rocpd database):
usage: python tools/gap_probe.py <results.db>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kt = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
names = dict(cur.execute("select id, display_name from %s" % ks))
rows = [(s, e, names[k]) for s, e, k in cur.execute("select start, end, kernel_id from %s order by start" % kt)]
KINDS = ["wr plain 400 MB", "wr nontemporal 400 MB", "wr sc0 sc1 400 MB", "wr plain 4 MB", "rd 400 MB", "lds 160 KB x 256 blocks",
         "wr plain 400 MB x 2", "wr nontemporal 400 MB x 2"]
marks = [i for i, r in enumerate(rows) if "mark_kernel" in r[2]]
res = {}
for n, i in enumerate(marks):
    kind = n % 8
    if n // 8 == 0:
        continue                      # first repetition: warm-up
    j = i + 1
    seq = []
    while j < len(rows) and "mark_kernel" not in rows[j][2]:
        seq.append(rows[j])
        j += 1
    bigs = [r for r in seq if "tiny" not in r[2]]
    tiny = [r for r in seq if "tiny" in r[2]]
    e = res.setdefault(kind, {"dur": [], "gap_mark_to_big": [], "gap_big_to_big": [], "gap_big_to_tiny": [], "gap_tiny_to_tiny": []})
    e["dur"].append((bigs[-1][1] - bigs[-1][0]) / 1e3)
    e["gap_mark_to_big"].append((bigs[0][0] - rows[i][1]) / 1e3)
    if len(bigs) > 1:
        e["gap_big_to_big"].append((bigs[1][0] - bigs[0][1]) / 1e3)
    e["gap_big_to_tiny"].append((tiny[0][0] - bigs[-1][1]) / 1e3)
    e["gap_tiny_to_tiny"].append((tiny[1][0] - tiny[0][1]) / 1e3)
mean = lambda v: sum(v) / len(v) if v else float("nan")
print("%-30s %9s %12s %11s %12s %13s" % ("kind of A", "dur_us", "mark->A_us", "A->A_us", "A->tiny_us", "tiny->tiny_us"))
for k in sorted(res):
    e = res[k]
    print("%-30s %9.1f %12.2f %11.2f %12.2f %13.2f" % (KINDS[k], mean(e["dur"]), mean(e["gap_mark_to_big"]), mean(e["gap_big_to_big"]),
                                                    mean(e["gap_big_to_tiny"]), mean(e["gap_tiny_to_tiny"])))
