#!/bin/bash
# round 3, GPU session 3: whole GPU suite (8-rank stand-in, chunked gather,
# bench contract with telemetry / configs / reference legs), default bench
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=gpurun_out/r03_s3
mkdir -p $OUT
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 ) > $OUT/pytest_gpu.txt 2>&1
grep -E "FAILED|passed|failed|ERROR" $OUT/pytest_gpu.txt | tail -30
( time timeout 900 python bench.py ) > $OUT/bench.json 2> $OUT/bench.err
tail -5 $OUT/bench.err
head -c 600 $OUT/bench.json
