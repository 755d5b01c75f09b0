#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=gpurun_out/r03_s9
mkdir -p $OUT
export TMPDIR=/tmp
RT_MI355_LIB=$PWD/rayopt_amd/librt_mi355_probes.so timeout 300 python scripts/r03_block2.py 2> $OUT/block2.err | tee $OUT/block2_$(date +%H%M%S).jsonl; tail -3 $OUT/block2.err
