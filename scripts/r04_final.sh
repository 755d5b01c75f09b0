#!/bin/bash
# round 4 evidence run -> gpurun_out/<tag>/ (what matters is copied to
# profiles/r04_final/)
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
TAG=${1:-r04_final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1
( time timeout 1800 python -m pytest tests -m gpu -q --maxfail=25 ) > $OUT/pytest_gpu_full.txt 2>&1
grep -E "FAILED|ERROR|passed|failed|^real" $OUT/pytest_gpu_full.txt | tail -12 > $OUT/pytest_gpu_tail.txt
cat $OUT/pytest_gpu_tail.txt
# the default command, as the driver runs it
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $OUT/bench.json 2> $OUT/bench.err
tail -4 $OUT/bench.err
# the same command under rocprofv3 (kernel trace + stats); the per-config
# records and the image-row leg are left out there so that every
# rt_trace_kernel launch is the headline workload
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv \
    -d $OLDPWD/$OUT/rocprof -- python $OLDPWD/bench.py --gpus 1 --steps 20 --warmup 5 --traffic off ) \
    > $OUT/bench_under_rocprofv3.json 2> $OUT/rocprof.err
find $OUT/rocprof -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/rocprof -name "*kernel_trace.csv" -exec sh -c 'head -400 "$1" > '$OUT'/kernel_trace_head.csv' _ {} \;
rm -rf $OUT/rocprof
head -5 $OUT/kernel_stats.csv
# the multi-process path as far as one GPU allows
RT_BENCH_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 > $OUT/bench_forced_dist_one_rank.json 2> $OUT/dist1.err
RT_BENCH_SHARE_DEVICE=1 RT_TRANSPORT_LIBRARY=$PWD/tests/stubs/librt_shm_transport.so \
    timeout 900 python bench.py --gpus 8 --total-rays 8000000 --steps 5 --warmup 2 \
    > $OUT/bench_eight_ranks_stand_in_transport_TEST_MODE.json 2> $OUT/dist8.err
timeout 600 python bench.py --extras --no-configs --cpu-sample 0 > $OUT/bench_extras.json 2> $OUT/extras.err
ls -la $OUT
