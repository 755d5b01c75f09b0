#!/bin/bash
# round 3, GPU session 2: the split library + fast-asphere default on the GPU
# suite, the row-stride sweep (laboratory build), shipped vs laboratory A/B
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=gpurun_out/r03_s2
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1200 python -m pytest tests -m gpu -q --maxfail=25 --deselect tests/test_bench_contract.py ) > $OUT/pytest_gpu.txt 2>&1
tail -15 $OUT/pytest_gpu.txt
timeout 300 python scripts/r03_ab_libs.py > $OUT/ab_libs.json 2> $OUT/ab_libs.err
cat $OUT/ab_libs.json
RT_MI355_LIB=$PWD/rayopt_amd/librt_mi355_probes.so timeout 900 python scripts/r03_ldpad.py > $OUT/ldpad.jsonl 2> $OUT/ldpad.err
tail -2 $OUT/ldpad.err
wc -l $OUT/ldpad.jsonl
