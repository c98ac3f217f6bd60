"""HBM traffic per launch of the step's kernels from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE
cannot share a pass: MI355X_MICROARCH.md "rocprofv3 PMC slots") -> profiles/rNN_hbm_traffic.json, the file
bench.py's roofline.traffic is read from.

  python tools/hbm_traffic.py <fetch.db> <write.db> <pages> <out.json> [--txt summary.txt]

Correction (MI355X_MICROARCH.md, HBM section): on gfx950 FETCH_SIZE counts 64 B per 128-B request of wide
coalesced 16-B/lane reads, so it is DOUBLED; WRITE_SIZE is taken as reported.  Both counters are in KiB.
Every kernel here streams with 16-B/lane loads; narrower accesses are uncalibrated (lower bounds)."""
import json
import sqlite3
import sys


def per_kernel(path, counter):
    db = sqlite3.connect(path)
    rows = db.execute("select kernel_name, count(*), avg(value), avg(duration) from counters_collection "
                      "where counter_name = ? group by kernel_name", (counter,)).fetchall()
    out = {}
    for name, calls, v, dur in rows:
        out[clean(name)] = (calls, v, dur / 1e3)
    return out


def family(name):
    return name.split("<")[0]


def step_rows(path, counter, which=2):
    """(kernel name, value) of every launch of ONE training step (delimited by the Adam launches, as tools/rocpd_step.py),
    or None when the database has no usable launch order."""
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    order = next((c for c in ("dispatch_id", "start", "id") if c in cols), None)
    if order is None:
        return None
    rows = db.execute("select kernel_name, value from counters_collection where counter_name = ? order by %s" % order,
                      (counter,)).fetchall()
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
    if len(adam) <= which:
        return None
    return [(clean(n), v) for n, v in rows[adam[which - 1] + 1:adam[which] + 1]]


def clean(name):
    return name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]


def main():
    fetch_db, write_db, pages, out_path = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    txt = sys.argv[sys.argv.index("--txt") + 1] if "--txt" in sys.argv else None
    f, w = per_kernel(fetch_db, "FETCH_SIZE"), per_kernel(write_db, "WRITE_SIZE")
    lines = ["%-72s %6s %12s %12s %14s %10s" % ("kernel", "calls", "FETCH_KiB", "WRITE_KiB", "traffic_MB", "avg_us")]
    fam = {}
    for name in sorted(set(f) & set(w), key=lambda n: -f[n][0] * (2 * f[n][1] + w[n][1])):
        calls, fk, us = f[name]
        wk = w[name][1]
        traffic = (2.0 * fk + wk) * 1024.0
        lines.append("%-72s %6d %12.1f %12.1f %14.1f %10.1f" % (name[:72], calls, fk, wk, traffic / 1e6, us))
        e = fam.setdefault(family(name), [0, 0.0, 0.0, 0.0])
        e[0] += calls
        e[1] += calls * fk
        e[2] += calls * wk
        e[3] += calls * us
    kernels = {}
    for k, (calls, fk, wk, us) in fam.items():
        kernels[k] = {"launches": calls, "fetch_size_kib_raw": round(fk / calls, 1), "write_size_kib": round(wk / calls, 1),
                      "traffic_bytes_per_launch": round((2.0 * fk + wk) / calls * 1024.0), "avg_us": round(us / calls, 1)}
    doc = {"pages": pages, "correction": "traffic = 2*FETCH_SIZE + WRITE_SIZE (gfx950 half-count of 16-B/lane reads)",
           "kernels": kernels}
    rf, rw = step_rows(fetch_db, "FETCH_SIZE"), step_rows(write_db, "WRITE_SIZE")
    if rf is not None and rw is not None and [n for n, _ in rf] == [n for n, _ in rw]:
        # every launch of one training step (Adam to Adam), the same correction: per kernel VARIANT (template instance) and
        # per family over the variant mix the step launches -- what bench.py's roofline.traffic reads
        doc["step_traffic_bytes"] = round((2.0 * sum(v for _, v in rf) + sum(v for _, v in rw)) * 1024.0)
        doc["step_traffic_note"] = "sum over the launches of one training step of the trace (second one), KiB counters * 1024"
        sk, sfam = {}, {}
        for (n, fv), (_, wv) in zip(rf, rw):
            for key, table in ((n, sk), (family(n), sfam)):
                e = table.setdefault(key, [0, 0.0])
                e[0] += 1
                e[1] += (2.0 * fv + wv) * 1024.0
        doc["step_kernels"] = {k: {"launches_in_step": c, "traffic_bytes_per_launch": round(t / c)} for k, (c, t) in sk.items()}
        doc["step_families"] = {k: {"launches_in_step": c, "traffic_bytes_per_launch": round(t / c)} for k, (c, t) in sfam.items()}
        lines.append("# one training step (Adam to Adam): 2*FETCH + WRITE = %.2f GB" % (doc["step_traffic_bytes"] / 1e9))
        lines.append("# launches of that step by kernel variant: count, traffic MB per launch")
        for k, (c, t) in sorted(sk.items(), key=lambda kv: -kv[1][1]):
            lines.append("#   %-100s %3d %10.1f" % (k[:100], c, t / c / 1e6))
    json.dump(doc, open(out_path, "w"), indent=1, sort_keys=True)
    if txt:
        open(txt, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
