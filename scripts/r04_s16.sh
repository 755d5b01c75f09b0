#!/bin/bash
# round 4, session 16: the laboratory kernel's variants in mixed memory
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s16
mkdir -p "$OUT"
cd "$REPO"
timeout 600 python scripts/lab.py variants > "$OUT/variants.jsonl" 2> "$OUT/variants.err"
echo "variants rc=$?"; tail -2 "$OUT/variants.err"; cat "$OUT/variants.jsonl"
