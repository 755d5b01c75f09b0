#!/bin/bash
# round 6 evidence run on ONE fresh box -> gpurun_out/<tag>/ (what matters is
# copied to profiles/r06_final/).  The driver's command runs first, as the
# first GPU process of the lease.
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
TAG=${1:-r06_final}
OUT=gpurun_out/$TAG
mkdir -p $OUT/legs
export TMPDIR=/tmp
( time RT_BENCH_DETAIL=$OUT/bench_detail.json timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench.json 2> $OUT/bench.err
wc -c $OUT/bench.json; grep "summary\]\|^real" $OUT/bench.err | head -30
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
# the GPU suite the driver's way: ONE process, stderr kept
( time timeout 1500 python -m pytest tests -x -q -m gpu ) > $OUT/pytest_gpu_single_process.txt 2> $OUT/pytest_gpu_single_process.stderr
grep -E "FAILED|ERROR|passed|failed|^real" $OUT/pytest_gpu_single_process.txt | tail -6
timeout 300 python scripts/class_map.py > $OUT/class_map.json 2> $OUT/class_map.txt
( time timeout 900 python scripts/reserve_ladder.py 30 ) > $OUT/reserve_ladder.jsonl 2> $OUT/reserve_ladder.stderr; tail -1 $OUT/reserve_ladder.jsonl
# the headline command under rocprofv3 (kernel trace + stats; side legs off so
# that every rt_trace_kernel launch is the headline workload) ...
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv \
    -d $OLDPWD/$OUT/rocprof -- python $OLDPWD/bench.py --gpus 1 --steps 20 --warmup 5 \
    --counters off --no-configs --cpu-sample 0 ) > $OUT/bench_under_rocprofv3.json 2> $OUT/rocprof.err
find $OUT/rocprof -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/rocprof; head -4 $OUT/kernel_stats.csv
# ... and every other leg of the line on its own (bench.py --only-config)
for KEY in C2 C3p C4 C4x C5 gen; do
    ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv \
        -d $OLDPWD/$OUT/rocprof_$KEY -- python $OLDPWD/bench.py --only-config $KEY ) \
        > $OUT/legs/$KEY.json 2> $OUT/legs/$KEY.err
    find $OUT/rocprof_$KEY -name "*kernel_stats.csv" -exec cp {} $OUT/legs/${KEY}_kernel_stats.csv \;
    rm -rf $OUT/rocprof_$KEY
    cat $OUT/legs/$KEY.json | cut -c1-250; grep "rt_trace" $OUT/legs/${KEY}_kernel_stats.csv | head -2
done
# the multi-process path as far as one GPU allows
RT_BENCH_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 > $OUT/bench_forced_dist_one_rank.json 2> $OUT/dist1.err
RT_BENCH_SHARE_DEVICE=1 RT_TRANSPORT_LIBRARY=$PWD/tests/stubs/librt_shm_transport.so \
    timeout 900 python bench.py --gpus 8 --total-rays 8000000 --steps 5 --warmup 2 \
    > $OUT/bench_eight_ranks_stand_in_transport_TEST_MODE.json 2> $OUT/dist8.err
RT_BENCH_DETAIL=$OUT/bench_extras_detail.json timeout 600 python bench.py --extras --cpu-sample 0 > $OUT/bench_extras.json 2> $OUT/extras.err
( time timeout 300 python scripts/boxstat.py 3 1 ) > $OUT/boxstat.jsonl 2> $OUT/boxstat.err
ls -la $OUT $OUT/legs
