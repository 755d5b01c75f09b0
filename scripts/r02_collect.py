"""Summaries of a session's rocprofv3 outputs (rocpd SQLite databases, the
format this rocprofv3 writes by default): per-kernel stats as CSV and
per-launch PMC counter sums.  Usage: r02_collect.py <dir> [outdir]"""
import csv
import glob
import os
import sqlite3
import sys

root = sys.argv[1]
outdir = sys.argv[2] if len(sys.argv) > 2 else None
for path in sorted(glob.glob(os.path.join(root, "**", "*_results.db"),
                             recursive=True)):
    rel = os.path.relpath(path, root)
    db = sqlite3.connect(path)
    stats = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), "
        "max(duration) from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in stats) or 1
    print("==", rel)
    rows = [("Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs",
             "MaxNs", "Percentage")]
    for name, calls, tot, avg, lo, hi in stats:
        rows.append((name, calls, tot, "%.3f" % avg, lo, hi,
                     "%.4f" % (100.*tot/total)))
        print("   %-70s calls %5d  avg %12.1f ns  %6.2f %%" % (
            name[:70], calls, avg, 100.*tot/total))
    if outdir:
        os.makedirs(outdir, exist_ok=True)
        tag = rel.replace(os.sep, "_").replace("_results.db", "")
        with open(os.path.join(outdir, tag + "_kernel_stats.csv"), "w",
                  newline="") as f:
            csv.writer(f).writerows(rows)
    try:
        counters = db.execute(
            "select kernel_name, counter_name, count(distinct dispatch_id), "
            "sum(value) from counters_collection group by 1, 2").fetchall()
    except sqlite3.Error:
        counters = []
    crow = [("Kernel_Name", "Counter_Name", "Launches", "Sum_Per_Launch")]
    for kernel, counter, launches, value in counters:
        if "rt_trace" not in kernel:
            continue
        crow.append((kernel, counter, launches, "%.6g" % (value/launches)))
        print("   PMC %-26s per launch %.6g  (%d launches)" % (
            counter, value/launches, launches))
    if outdir and len(crow) > 1:
        with open(os.path.join(outdir, tag + "_pmc.csv"), "w",
                  newline="") as f:
            csv.writer(f).writerows(crow)
