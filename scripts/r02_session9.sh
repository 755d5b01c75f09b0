#!/bin/bash
# round 2, GPU session 9: L2 <-> fabric counters of the store pattern with /
# without the input read, the single-stream fill and the trace kernel
O=gpurun_out/r02_s9
mkdir -p $O
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum --output-format csv -d $O/p1 -- python scripts/r02_tcc_probe.py > $O/p1.log 2>&1; echo "p1 rc $?"
timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum --output-format csv -d $O/p2 -- python scripts/r02_tcc_probe.py > $O/p2.log 2>&1; echo "p2 rc $?"
timeout 300 rocprofv3 --kernel-trace --pmc TCC_CYCLE_sum GRBM_GUI_ACTIVE --output-format csv -d $O/p3 -- python scripts/r02_tcc_probe.py > $O/p3.log 2>&1; echo "p3 rc $?"
grep -h median $O/p1.log
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: [0., 0])
for path in glob.glob("gpurun_out/r02_s9/p*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        k = row["Kernel_Name"]
        if "probe" not in k and "rt_trace" not in k:
            continue
        key = (k.split("(")[0][-60:], row["Counter_Name"])
        acc[key][0] += float(row["Counter_Value"]); acc[key][1] += 1
out = open("gpurun_out/r02_s9/tcc_per_launch.csv", "w")
out.write("kernel,counter,launches,per_launch\n")
for (k, c), (v, n) in sorted(acc.items()):
    out.write('"%s",%s,%d,%.6g\n' % (k, c, n, v/n)); print(k, c, n, "%.6g" % (v/n))
PY
