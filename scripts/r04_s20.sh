#!/bin/bash
# round 4, session 20: with non-temporal stores: the N sweep above 10^7 rays,
# the compacting kernel on over-filled bundles, the bench line with the
# per-ray-direction config
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s20
mkdir -p "$OUT"
cd "$REPO"
timeout 600 python scripts/lab.py nsweep --sizes 1000000 3000000 10000000 12500000 20000000 100000000 > "$OUT/nsweep_nt.jsonl" 2> "$OUT/nsweep_nt.err"
echo "nsweep rc=$?"; tail -2 "$OUT/nsweep_nt.err"
timeout 600 python scripts/lab.py compact > "$OUT/compact.jsonl" 2> "$OUT/compact.err"
echo "compact rc=$?"; tail -2 "$OUT/compact.err"; cat "$OUT/compact.jsonl"
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench rc=$?"; tail -12 "$OUT/bench.err"
