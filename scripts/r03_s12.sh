#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=gpurun_out/r03_s12
mkdir -p $OUT
export TMPDIR=/tmp
( time RT_MI355_EXACT_ASPHERE=1 timeout 1200 python tests/tools/fuzz_state.py 1000 4000 30 ) > $OUT/fuzz_soak.txt 2>&1
tail -6 $OUT/fuzz_soak.txt
( time RT_MI355_EXACT_ASPHERE=1 timeout 900 python tests/tools/soak_random.py 0 1500 ) > $OUT/soak_random.txt 2>&1
tail -4 $OUT/soak_random.txt
