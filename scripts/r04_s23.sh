#!/bin/bash
# round 4, session 23: two blocks of rays per workgroup (launch rows of both
# requested up front, the second staged in LDS) against one, same context
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s23
mkdir -p "$OUT"
cd "$REPO"
timeout 500 python scripts/lab.py optab blocks_per_group 1 2 > "$OUT/optab_blocks_per_group.jsonl" 2> "$OUT/optab.err"
echo "optab rc=$?"; tail -2 "$OUT/optab.err"; cut -c1-420 "$OUT/optab_blocks_per_group.jsonl"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_chunked_trace_gpu.py tests/test_uniform_input_gpu.py tests/test_reference_digests.py -m gpu -q -x 2>&1 | tail -4
