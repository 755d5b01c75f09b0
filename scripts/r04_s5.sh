#!/bin/bash
# round 4, session 5: all pairs of 56 chunks (42 rows in each): which chunks
# behave like "the same memory" when they share the 84 row streams?
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s5
mkdir -p "$OUT"
cd "$REPO"
timeout 500 scripts/labsrc/stream_lab 200 pairs 56 > "$OUT/pairs.jsonl" 2> "$OUT/pairs.err"
echo "pairs rc=$?"; tail -3 "$OUT/pairs.err"; wc -l "$OUT/pairs.jsonl"
