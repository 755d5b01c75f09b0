#!/bin/bash
# the occupancy tuner under the state fuzz (laboratory library: thresholds of
# 2 rays / 2 launches), the plain fuzz, the tuner's own tests
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=gpurun_out/r03_s17
mkdir -p $OUT
export RT_MI355_EXACT_ASPHERE=1
( time RT_MI355_LIB=$PWD/rayopt_amd/librt_mi355_probes.so RT_FUZZ_TUNE=1 timeout 150 python tests/tools/fuzz_state.py 0 120 ) 2>&1 | tail -6 | tee $OUT/fuzz_tuner.txt
( time timeout 100 python tests/tools/fuzz_state.py 1000 1080 ) 2>&1 | tail -5 | tee $OUT/fuzz_plain.txt
timeout 200 python -m pytest tests/test_tuning_gpu.py tests/test_state_fuzz_gpu.py tests/test_chunked_trace_gpu.py tests/test_cabi_gpu.py -q 2>&1 | tail -3 | tee $OUT/pytest.txt
