#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=gpurun_out/r03_s6
mkdir -p $OUT
export TMPDIR=/tmp
RT_MI355_LIB=$PWD/rayopt_amd/librt_mi355_probes.so timeout 600 python scripts/r03_tile_lowocc.py > $OUT/tile_lowocc.jsonl 2> $OUT/tile_lowocc.err
tail -2 $OUT/tile_lowocc.err; cat $OUT/tile_lowocc.jsonl
timeout 900 python -m pytest tests/test_chunked_trace_gpu.py tests/test_analysis_replay.py tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -8
