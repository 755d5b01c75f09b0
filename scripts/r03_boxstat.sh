#!/bin/bash
# one short driver-style headline run on whatever box gpurun hands out:
# launch time, frac and the box state (clocks, power, limiter residency)
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=gpurun_out/r03_boxstat
mkdir -p $OUT
export TMPDIR=/tmp
TAG=$(date +%H%M%S)
timeout 300 python bench.py --no-configs --cpu-sample 0 --traffic off --settle 1.0 > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
python - <<PY
import json
d = json.load(open("$OUT/bench_$TAG.json"))
t = d.get("telemetry") or {}
print(json.dumps({"kernel_ms": d["roofline"]["kernel_ms"], "frac": d["roofline"]["frac"],
                  "generated_kernel_ms": d["generated_batch"]["kernel_ms"],
                  "generated_frac": d["generated_batch"]["frac"],
                  "settle": t.get("settle"), "loop": t.get("loop")}))
PY
