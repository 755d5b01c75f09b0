#!/bin/bash
# round 4, session 13: what grows with N?  C3 at 1e7 / 2e7 / 5e7 rays, four
# and two workgroups per CU: plain, then counters; then 4 GiB pieces
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s13
mkdir -p "$OUT"
cd "$REPO"
timeout 400 python scripts/lab.py sizes > "$OUT/sizes_plain.jsonl" 2> "$OUT/sizes_plain.err"
echo "sizes plain rc=$?"; tail -2 "$OUT/sizes_plain.err"; cut -c1-200 "$OUT/sizes_plain.jsonl"
timeout 400 python scripts/lab.py sizes --piece-mib 4096 > "$OUT/sizes_plain_4GiB.jsonl" 2> "$OUT/sizes_plain_4GiB.err"
echo "4GiB rc=$?"; cut -c1-200 "$OUT/sizes_plain_4GiB.jsonl"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_LEVEL_sum" \
           "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv \
      -d "$OUT/pass$i" -- python "$REPO/scripts/lab.py" sizes > "$OUT/pass$i.jsonl" 2> "$OUT/pass$i.err"
  echo "pass $i rc=$? ($set)"
done
cd "$REPO"
python scripts/lab.py kinds-summary "$OUT" "$OUT/sizes_plain.jsonl" > "$OUT/sizes_summary.jsonl" 2> "$OUT/sizes_summary.err"
cut -c1-600 "$OUT/sizes_summary.jsonl"
find "$OUT" -name "*.db" -delete; find "$OUT" -name "*kernel_trace.csv" -size +2M -delete
