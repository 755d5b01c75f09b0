#!/bin/bash
# round 4, session 14 (run on several boxes): the pair matrix of 40 pieces
# and the shipped library's placement on the same box -- classes found, the
# measured store pattern, trace times at four and two workgroups per CU
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s14_$1
mkdir -p "$OUT"
cd "$REPO"
timeout 300 scripts/labsrc/stream_lab 120 pairs 40 > "$OUT/pairs.jsonl" 2> "$OUT/pairs.err"
echo "pairs rc=$?"
timeout 500 python scripts/lab.py placed --contexts 6 --all-placed 1 > "$OUT/placed.jsonl" 2> "$OUT/placed.err"
echo "placed rc=$?"; tail -2 "$OUT/placed.err"
timeout 300 python scripts/lab.py sizes --sizes 10000000 20000000 > "$OUT/sizes.jsonl" 2> "$OUT/sizes.err"
echo "sizes rc=$?"
rocm-smi --showpower --showclocks 2>/dev/null | head -30 > "$OUT/smi.txt"
