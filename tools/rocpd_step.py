"""One training step out of a rocprofv3 kernel trace (rocpd sqlite database): per-kernel time inside the
step, GPU busy time vs wall time (idle gaps between kernels).  Steps are delimited by the Adam launch.
usage: python tools/rocpd_step.py <results.db> [step index, default 1]"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
which = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kt = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
names = dict(cur.execute("select id, display_name from %s" % ks))
rows = list(cur.execute("select start, end, kernel_id from %s order by start" % kt))
adam = [i for i, r in enumerate(rows) if "adam_kernel" in names[r[2]]]
seg = rows[adam[which - 1] + 1:adam[which] + 1]


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([\w:]+(<[^(]*>)?)", n)
    return (m.group(1) if m else n)[:64]


agg = collections.OrderedDict()
for s, e, k in seg:
    a = agg.setdefault(short(names[k]), [0, 0.0])
    a[0] += 1
    a[1] += (e - s) / 1e3
busy = sum(v[1] for v in agg.values())
wall = (seg[-1][1] - rows[adam[which - 1]][1]) / 1e3
print("# training step %d of the trace: %d kernels, wall %.1f us, GPU busy %.1f us, idle between kernels %.1f us (%.1f %%)"
      % (which, len(seg), wall, busy, wall - busy, 100 * (wall - busy) / wall))
print("%-66s %5s %10s %6s" % ("kernel", "calls", "total_us", "%"))
small = [0, 0.0]
for n, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1]):
    if t >= 40:
        print("%-66s %5d %10.1f %6.1f" % (n, c, t, 100 * t / busy))
    else:
        small[0] += c
        small[1] += t
print("%-66s %5d %10.1f %6.1f" % ("(all kernels below 40 us in total)", small[0], small[1], 100 * small[1] / busy))

if "--order" in sys.argv:       # launch order of the step: offset from the previous Adam's end, duration, gap in front
    print("\n# launch order: t_start_us  dur_us  gap_before_us  kernel")
    t0, prev = rows[adam[which - 1]][1], rows[adam[which - 1]][1]
    for s, e, k in seg:
        print("%9.1f %8.1f %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, short(names[k])))
        prev = e
