#!/bin/bash
# round 4, session 12: per kind of trace: launch time + clocks (plain), SQ /
# TCC counters (separate --pmc passes) -> VALU issue roofline; the N sweep
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s12
mkdir -p "$OUT"
cd "$REPO"
timeout 400 python scripts/lab.py kinds --seconds 1.0 --telemetry 1 > "$OUT/kinds_plain.jsonl" 2> "$OUT/kinds_plain.err"
echo "kinds plain rc=$?"; tail -2 "$OUT/kinds_plain.err"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR" \
           "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv \
      -d "$OUT/pass$i" -- python "$REPO/scripts/lab.py" kinds > "$OUT/pass$i.jsonl" 2> "$OUT/pass$i.err"
  echo "pass $i rc=$? ($set)"
done
cd "$REPO"
python scripts/lab.py kinds-summary "$OUT" "$OUT/kinds_plain.jsonl" > "$OUT/kinds_summary.jsonl" 2> "$OUT/kinds_summary.err"
cat "$OUT/kinds_summary.jsonl" | cut -c1-700
find "$OUT" -name "*.db" -delete; find "$OUT" -name "*kernel_trace.csv" -size +2M -delete
timeout 900 python scripts/lab.py nsweep > "$OUT/nsweep.jsonl" 2> "$OUT/nsweep.err"
echo "nsweep rc=$?"; tail -3 "$OUT/nsweep.err"
