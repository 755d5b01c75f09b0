#!/bin/bash
# round 4, session 22: what would hiding the launch-row reads buy (upper
# bound), and the blocks-per-workgroup prefetch builds against the shipped one
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s22
mkdir -p "$OUT"
cd "$REPO"
timeout 300 python scripts/lab.py noinput > "$OUT/noinput.jsonl" 2> "$OUT/noinput.err"
echo "noinput rc=$?"; tail -2 "$OUT/noinput.err"; cat "$OUT/noinput.jsonl"
for K in 2 4; do
  timeout 300 python scripts/lab.py libab rayopt_amd/build/librt_tpw$K.so --reps 2 > "$OUT/libab_tpw$K.jsonl" 2> "$OUT/libab_tpw$K.err"
  echo "tpw $K rc=$?"; cut -c1-330 "$OUT/libab_tpw$K.jsonl" | head -2
done
