#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=gpurun_out/r03_s14
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=25 ) > $OUT/pytest_gpu_full.txt 2>&1
grep -E "FAILED|ERROR|passed|failed|^real" $OUT/pytest_gpu_full.txt | tail -12 | tee $OUT/pytest_gpu_tail.txt
