#!/bin/bash
# round 4, session 17: non-temporal row stores in the shipped kernel, A/B of
# two builds in one process
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s17
mkdir -p "$OUT"
cd "$REPO"
timeout 600 python scripts/lab.py libab rayopt_amd/librt_mi355_nt.so > "$OUT/libab_nt.jsonl" 2> "$OUT/libab_nt.err"
echo "libab rc=$?"; tail -2 "$OUT/libab_nt.err"; cut -c1-400 "$OUT/libab_nt.jsonl"
