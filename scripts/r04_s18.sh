#!/bin/bash
# round 4, session 18: the row stores' cache policy (A = the shipped library,
# non-temporal; B = builds with -DRT_ROWS_NT=0 / 2 / 3 / 4), then the
# resident-workgroups sweep again with non-temporal stores
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s18
mkdir -p "$OUT"
cd "$REPO"
for fl in 0 2 3 4; do
  timeout 300 python scripts/lab.py libab rayopt_amd/build/librt_store$fl.so --reps 2 > "$OUT/libab_store$fl.jsonl" 2> "$OUT/libab_store$fl.err"
  echo "store $fl rc=$?"; cut -c1-330 "$OUT/libab_store$fl.jsonl" | head -3
done
timeout 600 python scripts/lab.py resident > "$OUT/resident_nt.jsonl" 2> "$OUT/resident_nt.err"
echo "resident rc=$?"; tail -2 "$OUT/resident_nt.err"
