#!/bin/bash
# round 4, session 9: the whole GPU suite (both arithmetics), bench
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s9
mkdir -p "$OUT"
cd "$REPO"
( time timeout 1500 python -m pytest tests -m gpu -q ) 2>&1 | tail -25 | tee "$OUT/pytest.txt"
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench rc=$?"; tail -3 "$OUT/bench.err"
