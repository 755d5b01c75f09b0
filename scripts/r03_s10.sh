#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=gpurun_out/r03_s10
mkdir -p $OUT
export TMPDIR=/tmp
timeout 400 python scripts/r03_prefetch_ab.py 2> $OUT/prefetch_ab.err | tee $OUT/prefetch_ab_$(date +%H%M%S).jsonl
tail -3 $OUT/prefetch_ab.err
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_reference_digests.py tests/test_uniform_input_gpu.py tests/test_state_fuzz_gpu.py tests/test_generate.py -m gpu -q -x 2>&1 | tail -5
