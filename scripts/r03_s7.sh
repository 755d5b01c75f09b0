#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=gpurun_out/r03_s7
mkdir -p $OUT
export TMPDIR=/tmp
RT_MI355_EXACT_ASPHERE=1 timeout 900 python tests/tools/fuzz_state.py 0 240 30 > $OUT/fuzz_soak.txt 2>&1
tail -5 $OUT/fuzz_soak.txt
timeout 1200 python -m pytest tests/test_state_fuzz_gpu.py tests/test_gather_ranks_gpu.py tests/test_bench_contract.py tests/test_cabi_gpu.py -m gpu -q 2>&1 | tail -6
