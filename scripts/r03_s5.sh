#!/bin/bash
# round 3, GPU session 5: power / clocks per kind of launch (laboratory build)
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=gpurun_out/r03_s5
mkdir -p $OUT
export TMPDIR=/tmp
timeout 800 python scripts/r03_resident.py > $OUT/resident.jsonl 2> $OUT/resident.err
tail -3 $OUT/resident.err
wc -l $OUT/resident.jsonl
