#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=gpurun_out/r03_s8
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_uniform_input_gpu.py tests/test_chunked_trace_gpu.py tests/test_state_fuzz_gpu.py tests/test_gpu_parity.py tests/test_reference_digests.py -m gpu -q -x 2>&1 | tail -12
timeout 300 python scripts/r03_uniform_ab.py > $OUT/uniform_ab.jsonl 2> $OUT/uniform_ab.err
tail -2 $OUT/uniform_ab.err; cat $OUT/uniform_ab.jsonl
RT_MI355_EXACT_ASPHERE=1 timeout 600 python tests/tools/fuzz_state.py 0 200 30 2>&1 | tail -3
