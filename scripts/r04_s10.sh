#!/bin/bash
# round 4, session 10: the two failing tests with their output, the placement
# search with hops (12 contexts in one process + churn), bench
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s10
mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests/test_placement_gpu.py tests/test_chunked_trace_gpu.py tests/test_gather_ranks_gpu.py -m gpu -q 2>&1 | tail -30 | tee "$OUT/pytest.txt"
RT_FUZZ_ARITH=default RT_MI355_EXACT_ASPHERE=0 timeout 600 python tests/tools/fuzz_state.py 9000 9075 2>&1 | tail -30 | tee "$OUT/fuzz_default.txt"
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench rc=$?"; tail -3 "$OUT/bench.err"
timeout 600 python scripts/lab.py placed --contexts 12 --all-placed 1 > "$OUT/placed12.jsonl" 2> "$OUT/placed12.err"
echo "placed rc=$?"; tail -3 "$OUT/placed12.err"
