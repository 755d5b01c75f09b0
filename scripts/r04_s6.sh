#!/bin/bash
# round 4, session 6: the pair matrix with 256 MiB and 64 MiB chunks: is
# "the same memory" (profiles/r04_probes/pairs_1GiB) decided below 1 GiB too?
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s6
mkdir -p "$OUT"
cd "$REPO"
timeout 300 scripts/labsrc/stream_lab 400 pairs 64 256 > "$OUT/pairs_256MiB.jsonl" 2> "$OUT/pairs_256.err"
echo "pairs256 rc=$?"; tail -3 "$OUT/pairs_256.err"
timeout 300 scripts/labsrc/stream_lab 800 pairs 64 64 > "$OUT/pairs_64MiB.jsonl" 2> "$OUT/pairs_64.err"
echo "pairs64 rc=$?"; tail -3 "$OUT/pairs_64.err"
