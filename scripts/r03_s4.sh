#!/bin/bash
# round 3, GPU session 4: clock / power trace under sustained load
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=gpurun_out/r03_s4
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python scripts/r03_clock_trace.py > $OUT/clock_trace.jsonl 2> $OUT/clock_trace.err
tail -3 $OUT/clock_trace.err
wc -c $OUT/clock_trace.jsonl

