#!/bin/bash
# round 4, session 15: where above 10^7 rays does the trace slow down, and is
# it the row spacing or the batch?
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s15
mkdir -p "$OUT"
cd "$REPO"
timeout 600 python scripts/lab.py nsweep --sizes 10000000 11000000 12500000 15000000 17500000 20000000 > "$OUT/nsweep_knee.jsonl" 2> "$OUT/nsweep_knee.err"
echo "knee rc=$?"; tail -2 "$OUT/nsweep_knee.err"
timeout 600 python scripts/lab.py spacing > "$OUT/spacing.jsonl" 2> "$OUT/spacing.err"
echo "spacing rc=$?"; tail -2 "$OUT/spacing.err"; cut -c1-260 "$OUT/spacing.jsonl"
