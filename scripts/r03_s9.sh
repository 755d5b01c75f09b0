#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=gpurun_out/r03_s9
mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 python scripts/r03_resident2.py 2> $OUT/resident2.err | tee $OUT/resident2_$(date +%H%M%S).jsonl
