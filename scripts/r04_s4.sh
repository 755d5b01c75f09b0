#!/bin/bash
# round 4, session 4: the 84 row streams dealt onto 1 GiB chunks through a
# pointer table -- what about a SET of chunks decides the store pattern's speed?
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s4
mkdir -p "$OUT"
cd "$REPO"
timeout 500 scripts/labsrc/stream_lab 200 > "$OUT/stream_lab.jsonl" 2> "$OUT/stream_lab.err"
echo "stream_lab rc=$?"; tail -3 "$OUT/stream_lab.err"; wc -l "$OUT/stream_lab.jsonl"
