#!/bin/bash
# round 4, session 11: C4 in bench vs lab, bench with consumers, contract test
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s11
mkdir -p "$OUT"
cd "$REPO"
timeout 600 python scripts/lab.py c4check > "$OUT/c4check.jsonl" 2> "$OUT/c4check.err"
echo "c4check rc=$?"; tail -3 "$OUT/c4check.err"; cat "$OUT/c4check.jsonl"
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench rc=$?"; tail -3 "$OUT/bench.err"
timeout 900 python -m pytest tests/test_bench_contract.py tests/test_state_fuzz_gpu.py tests/test_aiming_reference.py tests/test_analysis_replay.py -m gpu -q -x 2>&1 | tail -15 | tee "$OUT/pytest.txt"
