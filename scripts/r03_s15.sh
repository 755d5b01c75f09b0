#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=gpurun_out/r03_s15
mkdir -p $OUT
export TMPDIR=/tmp
timeout 500 python scripts/r03_codeaddr.py 2> $OUT/codeaddr.err | tee $OUT/codeaddr_$(date +%H%M%S).jsonl
tail -3 $OUT/codeaddr.err
