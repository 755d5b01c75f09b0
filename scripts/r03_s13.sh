#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=gpurun_out/r03_s13
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python scripts/r03_isolated.py 2> $OUT/isolated.err | tee $OUT/isolated.json
tail -2 $OUT/isolated.err
