"""Per-kernel average of PMC counters from a rocprofv3 rocpd database.
Usage: rocpd_pmc.py [--raw] <db> [<db> ...]   (default: values shown /1024, i.e. MB for *_SIZE)"""
import sqlite3
import sys

args = sys.argv[1:]
raw = "--raw" in args
args = [a for a in args if a != "--raw"]
for path in args:
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, counter_name, count(*), avg(value), avg(duration) from "
                      "counters_collection group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
    print("# %s" % path)
    print("%-50s %-34s %6s %16s %12s" % ("kernel", "counter", "calls", "avg" if raw else "avg/1024", "avg_us"))
    for n, c, k, v, d in rows:
        n = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:50]
        if raw and ("wino" not in n and "conv" not in n and "roipool" not in n):
            continue
        print("%-50s %-34s %6d %16.1f %12.1f" % (n, c, k, v if raw else v / 1024.0, d / 1e3))
