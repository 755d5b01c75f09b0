#!/bin/bash
# round 4, session 7: the shipped library with measured placement and range
# shortcuts: self-test, timings, bit identity, then the parity suites
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s7
mkdir -p "$OUT"
cd "$REPO"
timeout 600 python scripts/lab.py placed --contexts 6 > "$OUT/placed.jsonl" 2> "$OUT/placed.err"
echo "placed rc=$?"; tail -5 "$OUT/placed.err"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_reference_digests.py tests/test_uniform_input_gpu.py -m gpu -x -q 2>&1 | tail -8 | tee "$OUT/pytest.txt"
