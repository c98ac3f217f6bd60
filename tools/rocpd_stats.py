"""Kernel-time summary from a rocprofv3 rocpd sqlite database (what --stats would print as CSV)."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                   "from kernels group by %s order by sum(end-start) desc" % (name_col, name_col)).fetchall()
tot = sum(r[2] for r in rows)
print("%-72s %7s %12s %12s %12s %12s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
for n, c, s, a, mn, mx in rows:
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    n = n.split("(")[0][:72]
    print("%-72s %7d %12.1f %12.1f %12.1f %12.1f %6.2f" % (n, c, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
print("TOTAL_us %.1f" % (tot / 1e3))
