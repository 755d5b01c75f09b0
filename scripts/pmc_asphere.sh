#!/bin/bash
# SQ / TCC counters of the trace kernel on the asphere config (C4) next to the
# headline config (C3): which side of the roofline each one sits on.
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/${1:-pmc_c4}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for cfg in asphere dgauss; do
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SMEM" \
             "GRBM_GUI_ACTIVE TCC_BUSY_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum"; do
    i=$((i+1))
    RT_AB_CONFIG=$cfg timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv \
        -d "$OUT/$cfg$i" -- python "$REPO/scripts/ab.py" "[{}]" > "$OUT/$cfg$i.log" 2>&1
    echo "$cfg pass $i rc=$?"
  done
done
cd "$REPO"
python - "$OUT" <<'PY' | tee "$OUT/summary.txt"
import csv, glob, sys, collections
out = sys.argv[1]
for cfg in ("asphere", "dgauss"):
    acc = collections.defaultdict(list)
    for p in glob.glob(out + "/%s*/**/*counter_collection.csv" % cfg, recursive=True):
        for r in csv.DictReader(open(p)):
            if "rt_trace_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("== %s (per launch, 10^7 rays)" % cfg)
    m = {k: sum(v)/len(v) for k, v in acc.items()}
    for k in sorted(m):
        print("%-40s %.6g  (n=%d)" % (k, m[k], len(acc[k])))
    if "SQ_INSTS_VALU" in m and "GRBM_GUI_ACTIVE" in m:
        # a 64-lane FP64 VALU instruction occupies its SIMD for 4 cycles
        # (transcendentals longer); 1024 SIMDs; GRBM_GUI_ACTIVE sums 8 XCDs
        print("VALU issue share >= %.2f" % (
            m["SQ_INSTS_VALU"]*4/1024/(m["GRBM_GUI_ACTIVE"]/8)))
    if "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" in m and "TCC_BUSY_sum" in m:
        print("write requests stalled on DRAM credits / TCC busy = %.3f" % (
            m["TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum"]/m["TCC_BUSY_sum"]))
PY
