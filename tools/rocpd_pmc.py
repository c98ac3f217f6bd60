"""Per-kernel average of one PMC counter from a rocprofv3 rocpd database (values in KiB for
FETCH_SIZE / WRITE_SIZE).  Usage: rocpd_pmc.py <db> [<db> ...]"""
import sqlite3
import sys

for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from "
                      "counters_collection group by kernel_name, counter_name order by sum(value) desc").fetchall()
    print("# %s" % path)
    print("%-60s %-12s %6s %14s %12s" % ("kernel", "counter", "calls", "avg_MB", "avg_us"))
    for n, c, k, v, d in rows[:24]:
        n = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]
        print("%-60s %-12s %6d %14.1f %12.1f" % (n, c, k, v / 1024.0, d / 1e3))
