#!/bin/bash
# round 4, session 1: what separates a good allocation from a bad one?
#  1 chunk_lab: synthetic store pattern on hipMalloc allocations and on 1 GiB
#    hipMemCreate chunks (per chunk / per allocation / fastest vs slowest)
#  2 lab.py placement without counters (timings + store-pattern correlation)
#  3 the same command under rocprofv3 --pmc, separate passes per counter set
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s1
mkdir -p "$OUT"
cd "$REPO"
( time timeout 300 scripts/labsrc/chunk_lab 6 44 1024 ) > "$OUT/chunk_lab.jsonl" 2> "$OUT/chunk_lab.err"
echo "chunk_lab rc=$?"; tail -3 "$OUT/chunk_lab.err"
( time timeout 300 python scripts/lab.py placement --contexts 6 --vmm 3 ) > "$OUT/placement_plain.jsonl" 2> "$OUT/placement_plain.err"
echo "placement rc=$?"; tail -3 "$OUT/placement_plain.err"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_sum" \
           "TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum" \
           "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_SERIALIZATION_STALL_sum" \
           "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-trace --pmc $set --output-format csv \
      -d "$OUT/pass$i" -- python "$REPO/scripts/lab.py" placement --contexts 5 --vmm 2 --probe 0 --launches 16 --warm 8 \
      > "$OUT/pass$i.jsonl" 2> "$OUT/pass$i.err"
  echo "pass $i rc=$? ($set)"
done
cd "$REPO"
python scripts/lab.py pmc-summary "$OUT" > "$OUT/pmc_summary.jsonl" 2> "$OUT/pmc_summary.err"
find "$OUT" -name "*.db" -delete
# keep the counter CSVs but not the per-dispatch kernel traces if they are huge
du -sh "$OUT"
head -c 3000 "$OUT/pmc_summary.jsonl"
