#!/bin/bash
# round 4, session 24: the store floor under the trace (pattern without
# arithmetic, ordinary and non-temporal stores) in placed arrays
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s24
mkdir -p "$OUT"
cd "$REPO"
timeout 300 python scripts/lab.py floor > "$OUT/floor.jsonl" 2> "$OUT/floor.err"
echo "floor rc=$?"; tail -2 "$OUT/floor.err"; cat "$OUT/floor.jsonl"
