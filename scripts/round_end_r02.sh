#!/bin/bash
# Round 2 evidence run (one GPU): smoke, full GPU suite, bench (default and
# --extras), every BASELINE config, the multi-process path as far as one GPU
# allows, rocprofv3 --kernel-trace --stats of the DEFAULT bench command, and
# the FETCH_SIZE / WRITE_SIZE PMC passes (separate runs) -> gpurun_out/<tag>/
set -u
TAG=${1:-r02_final}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$REPO"
export TMPDIR=/tmp
say() { echo "$@" | tee -a "$OUT/summary.txt"; }
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
say "smoke rc=$?"; tail -1 "$OUT/smoke.log" | tee -a "$OUT/summary.txt"
timeout 1500 python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1
say "pytest rc=$?"; grep -E "passed|failed" "$OUT/pytest_gpu.log" | tail -2 | tee -a "$OUT/summary.txt"
timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
say "bench rc=$?"; cat "$OUT/bench.json" | tee -a "$OUT/summary.txt"
timeout 600 python bench.py --extras > "$OUT/bench_extras.json" 2> "$OUT/bench_extras.err"
say "bench --extras rc=$?"
timeout 900 python scripts/configs_bench.py > "$OUT/configs.jsonl" 2> "$OUT/configs.err"
say "configs rc=$?"; cat "$OUT/configs.jsonl" | tee -a "$OUT/summary.txt"
RT_BENCH_FORCE_DIST=1 timeout 600 python bench.py --gpus 1 --steps 10 --warmup 2 --cpu-sample 0 \
   > "$OUT/bench_forced_dist.json" 2> "$OUT/bench_forced_dist.err"
say "forced-dist bench (self-spawned rank, host group, RCCL communicator) rc=$?"; cat "$OUT/bench_forced_dist.json" | tee -a "$OUT/summary.txt"
RT_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
   --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 2 --cpu-sample 0 --gather-every-step \
   > "$OUT/bench_forced_dist_every.json" 2> "$OUT/bench_forced_dist_every.err"
say "forced-dist under torch.distributed.run, gather every step rc=$?"; cat "$OUT/bench_forced_dist_every.json" | tee -a "$OUT/summary.txt"
RT_BENCH_SHARE_DEVICE=1 timeout 900 python bench.py --gpus 2 --steps 5 --warmup 1 --settle 0 \
   > "$OUT/bench_two_ranks_one_device_TEST_MODE.json" 2> "$OUT/bench_two_ranks.err"
say "two ranks on one device (host side only, test mode) rc=$?"; cat "$OUT/bench_two_ranks_one_device_TEST_MODE.json" | tee -a "$OUT/summary.txt"
RT_BENCH_SHARE_DEVICE=1 RT_TRANSPORT_LIBRARY=$REPO/tests/stubs/librt_shm_transport.so timeout 900 python bench.py --gpus 2 --rays 2000000 --steps 5 --warmup 1 --settle 0 --no-configs4 \
   > "$OUT/bench_two_ranks_stand_in_transport_TEST_MODE.json" 2> "$OUT/bench_two_ranks_stand_in.err"
say "two ranks on one device, rt_gather_final over the shared-memory stand-in for RCCL (test mode) rc=$?"; cat "$OUT/bench_two_ranks_stand_in_transport_TEST_MODE.json" | tee -a "$OUT/summary.txt"
timeout 300 python scripts/regen_ab.py > "$OUT/regen_ab.txt" 2>&1
say "re-trace of a generated batch, rays rebuilt vs read rc=$?"; cat "$OUT/regen_ab.txt" | tee -a "$OUT/summary.txt"
for ex in end_to_end spot_report optimize_spot tolerance_monte_carlo; do
  timeout 300 python examples/$ex.py > "$OUT/example_$ex.log" 2>&1
  say "examples/$ex.py rc=$?"; tail -2 "$OUT/example_$ex.log" | tee -a "$OUT/summary.txt"
done
timeout 300 python scripts/r02_probe.py D > "$OUT/host_overhead.jsonl" 2>&1
say "host overhead rc=$?"; cat "$OUT/host_overhead.jsonl" | tee -a "$OUT/summary.txt"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_stats" -- \
    python bench.py --cpu-sample 0 > "$OUT/prof_stats.json" 2> "$OUT/prof_stats.err"
say "rocprof stats rc=$?"
find "$OUT/prof_stats" -name "*kernel_stats*.csv" | head -1 | xargs -r head -8 | tee -a "$OUT/summary.txt"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/prof_pmc_$c" -- \
      python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-api-leg > "$OUT/prof_pmc_$c.json" 2> "$OUT/prof_pmc_$c.err"
  say "rocprof pmc $c rc=$?"
done
python scripts/pmc_traffic.py "$OUT" "$TAG" 2>&1 | tee -a "$OUT/summary.txt"
du -sh "$OUT"
