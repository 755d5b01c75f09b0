"""Pick the numbers out of a session's rocprofv3 output directories."""
import csv
import glob
import os
import sys

root = sys.argv[1]
for path in sorted(glob.glob(os.path.join(root, "**", "*kernel_stats.csv"),
                             recursive=True)):
    print("==", os.path.relpath(path, root))
    with open(path) as f:
        for row in list(csv.DictReader(f))[:6]:
            print("   %-60s calls %6s  avg %12s ns  total %14s ns  %5s %%" % (
                row.get("Name", "")[:60], row.get("Calls"),
                row.get("AverageNs"), row.get("TotalDurationNs"),
                row.get("Percentage")))
for path in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"),
                             recursive=True)):
    sums, calls = {}, {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if "rt_trace" not in row.get("Kernel_Name", ""):
                continue
            k = row["Counter_Name"]
            sums[k] = sums.get(k, 0.) + float(row["Counter_Value"])
            calls[k] = calls.get(k, 0) + 1
    print("==", os.path.relpath(path, root))
    for k in sorted(sums):
        print("   %-28s per launch %.6g  (%d launches)" % (
            k, sums[k]/calls[k], calls[k]))
