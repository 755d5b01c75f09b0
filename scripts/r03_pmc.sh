#!/bin/bash
# SQ / TCC counters of the headline kernel at two (default) and seven
# workgroups per CU -- separate passes, --kernel-trace only (no other trace
# domains with --pmc)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r03_pmc
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY" \
           "GRBM_GUI_ACTIVE TCC_BUSY_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" \
           "TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_CYCLE_sum" \
           "SQ_WAIT_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv \
      -d "$OUT/p$i" -- python "$REPO/scripts/r03_pmc_probe.py" > "$OUT/p$i.log" 2>&1
  echo "pass $i rc=$?"
done
cd "$REPO"
python - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for p in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(p)) if "rt_trace_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    by = collections.defaultdict(list)
    for r in rows:
        by[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in by.items():
        half = len(v)//2
        acc[k] = (sum(v[1:half])/max(1, half - 1), sum(v[half + 1:])/max(1, len(v) - half - 1))
print("%-40s %18s %18s" % ("counter (per launch, 10^7 rays)", "2 WG/CU (default)", "7 WG/CU"))
for k in sorted(acc):
    print("%-40s %18.4g %18.4g" % (k, acc[k][0], acc[k][1]))
PY
find "$OUT" -name "*.db" -delete; find "$OUT" -type d -name "p*" -exec rm -rf {} + 2>/dev/null
ls $OUT
