#!/bin/bash
# round 3, GPU session 5: power / clocks per kind of launch (laboratory build)
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=gpurun_out/r03_s5
mkdir -p $OUT
export TMPDIR=/tmp
RT_MI355_LIB=$PWD/rayopt_amd/librt_mi355_probes.so timeout 800 python scripts/r03_lowocc_variants.py > $OUT/lowocc.jsonl 2> $OUT/lowocc.err
tail -3 $OUT/lowocc.err
wc -l $OUT/lowocc.jsonl
