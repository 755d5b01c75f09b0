#!/bin/bash
# round 4, session 19: the laboratory kernel's variants with non-temporal
# stores as the baseline
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s19
mkdir -p "$OUT"
cd "$REPO"
timeout 600 python scripts/lab.py variants --nt 1 > "$OUT/variants_nt.jsonl" 2> "$OUT/variants_nt.err"
echo "variants rc=$?"; tail -2 "$OUT/variants_nt.err"; cat "$OUT/variants_nt.jsonl"
