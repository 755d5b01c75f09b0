#!/bin/bash
# tests + probes + bench + kernel-trace stats + FETCH/WRITE PMC
set -u
TAG=${1:-r01c}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$REPO"
timeout 1200 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
echo "smoke rc=$?" | tee -a "$OUT/summary.txt"; tail -1 "$OUT/smoke.log" | tee -a "$OUT/summary.txt"
timeout 1200 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1
echo "pytest rc=$?" | tee -a "$OUT/summary.txt"; tail -3 "$OUT/pytest_gpu.log" | tee -a "$OUT/summary.txt"
timeout 300 python scripts/probe.py > "$OUT/probe.json" 2> "$OUT/probe.err"
echo "probe rc=$?" | tee -a "$OUT/summary.txt"; cat "$OUT/probe.json" | tee -a "$OUT/summary.txt"
timeout 600 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "bench rc=$?" | tee -a "$OUT/summary.txt"; cat "$OUT/bench.json" | tee -a "$OUT/summary.txt"
timeout 600 python bench.py --extras > "$OUT/bench_extras.json" 2> "$OUT/bench_extras.err"
echo "bench --extras rc=$?" | tee -a "$OUT/summary.txt"; cat "$OUT/bench_extras.json" | tee -a "$OUT/summary.txt"
timeout 300 python scripts/configs_bench.py > "$OUT/configs.jsonl" 2> "$OUT/configs.err"
echo "configs rc=$?" | tee -a "$OUT/summary.txt"
RT_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
   --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 2 --cpu-sample 0 \
   > "$OUT/bench_forced_dist.json" 2> "$OUT/bench_forced_dist.err"
echo "forced-dist bench rc=$?" | tee -a "$OUT/summary.txt"; cat "$OUT/bench_forced_dist.json" | tee -a "$OUT/summary.txt"
RT_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
   --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 10 --warmup 2 --cpu-sample 0 --gather-every-step \
   > "$OUT/bench_forced_dist_every.json" 2> "$OUT/bench_forced_dist_every.err"
echo "forced-dist (gather every step) rc=$?" | tee -a "$OUT/summary.txt"; cat "$OUT/bench_forced_dist_every.json" | tee -a "$OUT/summary.txt"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_stats" -- \
    python "$REPO/bench.py" > "$OUT/prof_stats.log" 2>&1
echo "rocprof stats rc=$?" | tee -a "$OUT/summary.txt"
find "$OUT/prof_stats" -name "*kernel_stats*.csv" | head -1 | xargs -r head -8 | tee -a "$OUT/summary.txt"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/prof_pmc_$c" -- \
      python "$REPO/bench.py" --steps 3 --warmup 1 --cpu-sample 0 > "$OUT/prof_pmc_$c.log" 2>&1
  echo "rocprof pmc $c rc=$?" | tee -a "$OUT/summary.txt"
done
cd "$REPO"
python scripts/pmc_traffic.py "$OUT" 2>&1 | tee -a "$OUT/summary.txt"
du -sh "$OUT"
