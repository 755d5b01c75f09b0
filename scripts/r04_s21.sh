#!/bin/bash
# round 4, session 21: non-temporal LOADS of the launch rows (two builds, one
# process; bundles with per-ray directions: 40 B per ray read); C2 with
# 512 MiB pieces; placement tests
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s21
mkdir -p "$OUT"
cd "$REPO"
timeout 400 python scripts/lab.py libab rayopt_amd/build/librt_input_nt.so --per-ray-directions 1 --reps 2 > "$OUT/libab_input_nt.jsonl" 2> "$OUT/libab_input_nt.err"
echo "input nt rc=$?"; cut -c1-330 "$OUT/libab_input_nt.jsonl" | head -3
timeout 400 python scripts/lab.py nsweep --sizes 300000 1000000 2000000 3000000 > "$OUT/nsweep_small.jsonl" 2> "$OUT/nsweep_small.err"
echo "nsweep rc=$?"; tail -2 "$OUT/nsweep_small.err"
timeout 600 python -m pytest tests/test_placement_gpu.py tests/test_bench_contract.py -m gpu -q -x 2>&1 | tail -5
