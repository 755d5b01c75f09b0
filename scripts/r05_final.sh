#!/bin/bash
# round 5 evidence run -> gpurun_out/<tag>/ (what matters is copied to
# profiles/r05_final/).  "light": the driver's bench command and the
# placement statistics of the box only.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
TAG=${1:-r05_final}
MODE=${2:-full}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# the default command, as the driver runs it
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench.json 2> $OUT/bench.err
grep "configs\]\|^real" $OUT/bench.err | tail -9
( time timeout 300 python scripts/boxstat.py 3 1 ) > $OUT/boxstat.jsonl 2> $OUT/boxstat.err
[ "$MODE" = light ] && exit 0
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
( time timeout 1800 python -m pytest tests -m gpu -q -n 4 --maxfail=25 ) > $OUT/pytest_gpu_full.txt 2>&1
grep -E "FAILED|ERROR|passed|failed|^real" $OUT/pytest_gpu_full.txt | tail -12 > $OUT/pytest_gpu_tail.txt
cat $OUT/pytest_gpu_tail.txt
# the same command under rocprofv3 (kernel trace + stats); the side legs are
# left out there so that every rt_trace_kernel launch is the headline workload
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv \
    -d $OLDPWD/$OUT/rocprof -- python $OLDPWD/bench.py --gpus 1 --steps 20 --warmup 5 --counters off ) \
    > $OUT/bench_under_rocprofv3.json 2> $OUT/rocprof.err
find $OUT/rocprof -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/rocprof -name "*kernel_trace.csv" -exec sh -c 'head -400 "$1" > '$OUT'/kernel_trace_head.csv' _ {} \;
rm -rf $OUT/rocprof
head -6 $OUT/kernel_stats.csv
# the multi-process path as far as one GPU allows
RT_BENCH_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 > $OUT/bench_forced_dist_one_rank.json 2> $OUT/dist1.err
RT_BENCH_SHARE_DEVICE=1 RT_TRANSPORT_LIBRARY=$PWD/tests/stubs/librt_shm_transport.so \
    timeout 900 python bench.py --gpus 8 --total-rays 8000000 --steps 5 --warmup 2 \
    > $OUT/bench_eight_ranks_stand_in_transport_TEST_MODE.json 2> $OUT/dist8.err
timeout 600 python bench.py --extras --cpu-sample 0 > $OUT/bench_extras.json 2> $OUT/extras.err
ls -la $OUT
