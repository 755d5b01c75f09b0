#!/bin/bash
# round 6: one fresh box -> gpurun_out/<tag>/.  The driver's command runs
# FIRST (bench.py as the first GPU process of the lease),
# then whatever the mode asks for.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
TAG=${1:-r06}
MODE=${2:-light}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( time RT_BENCH_DETAIL=$OUT/bench_detail.json timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench.json 2> $OUT/bench.err
wc -c $OUT/bench.json
grep "summary\]\|^real" $OUT/bench.err | head -40
timeout 300 python scripts/class_map.py > $OUT/class_map.json 2> $OUT/class_map.txt
grep "rt_place" $OUT/class_map.txt | cut -c1-400
case "$MODE" in
ladder*)
    CYC=${MODE#ladder}; CYC=${CYC:-30}
    ( time timeout 1500 python scripts/reserve_ladder.py $CYC ) > $OUT/reserve_ladder.jsonl 2> $OUT/reserve_ladder.stderr
    echo "ladder rc $?"; tail -1 $OUT/reserve_ladder.jsonl; tail -5 $OUT/reserve_ladder.stderr | cut -c1-300
    ;;
esac
shift 2 || true
for cmd in "$@"; do
    echo "== $cmd"
    bash -c "$cmd"
done
