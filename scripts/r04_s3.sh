#!/bin/bash
# round 4, session 3: (a) SoA against tile-major layouts in the same
# allocations (laboratory kernel throughout), (b) chunk sizes 2 / 4 / 12 GiB
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s3
mkdir -p "$OUT"
cd "$REPO"
timeout 500 python scripts/lab.py layouts --contexts 6 --vmm 3 > "$OUT/layouts.jsonl" 2> "$OUT/layouts.err"
echo "layouts rc=$?"; tail -3 "$OUT/layouts.err"
for mb in 2048 4096 12288; do
  timeout 200 python scripts/lab.py placement --contexts 0 --vmm 4 --vmm-mb $mb > "$OUT/placement_vmm_$mb.jsonl" 2> "$OUT/placement_vmm_$mb.err"
  echo "vmm $mb rc=$?"; tail -2 "$OUT/placement_vmm_$mb.err"
done
