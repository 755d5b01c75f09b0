#!/bin/bash
# round 4, session 8: the whole GPU suite, the driver's bench command, the
# resident-workgroups sweep per kind of trace with placed arrays
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s8
mkdir -p "$OUT"
cd "$REPO"
( time timeout 900 python -m pytest tests -m gpu -x -q ) 2>&1 | tail -15 | tee "$OUT/pytest.txt"
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench rc=$?"; tail -5 "$OUT/bench.err"; head -c 600 "$OUT/bench.json"
timeout 600 python scripts/lab.py resident > "$OUT/resident.jsonl" 2> "$OUT/resident.err"
echo "resident rc=$?"; tail -3 "$OUT/resident.err"
