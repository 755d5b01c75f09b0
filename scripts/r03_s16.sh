#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=gpurun_out/r03_s16
mkdir -p $OUT
timeout 300 python -m pytest tests/test_tuning_gpu.py -x -q 2>&1 | tail -3
for k in 1 2 3; do
  timeout 200 python bench.py --no-configs --cpu-sample 0 --traffic off 2>/dev/null | tee $OUT/bench_tuned_$k.json | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print(d['ms_per_step'], r['frac'], {k: v for k, v in r['resident_workgroups'].items() if k != 'note'}, 'generated', d['generated_batch']['kernel_ms'])"
done
timeout 200 python scripts/r03_tune_ab.py 2>> $OUT/tune.err | tee $OUT/tune_ab2.jsonl | cut -c1-260
