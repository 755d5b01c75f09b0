#!/bin/bash
# One gpurun call: smoke, GPU parity tests, bench (+ variants), rocprofv3.
# Usage (from the repo root, on the GPU box):  bash scripts/gpu_session.sh [tag] [what...]
# what: smoke tests bench variants prof pmc   (default: all)
set -u
TAG=${1:-r01}
shift || true
WHAT=${*:-smoke tests bench variants prof pmc}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$REPO"
has() { [[ " $WHAT " == *" $1 "* ]]; }

if has smoke; then
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
  echo "smoke rc=$?" | tee -a "$OUT/summary.txt"
fi
if has tests; then
  timeout 1500 python -m pytest tests -m gpu -x -q > "$OUT/pytest_gpu.log" 2>&1
  echo "pytest rc=$?" | tee -a "$OUT/summary.txt"
  tail -5 "$OUT/pytest_gpu.log" | tee -a "$OUT/summary.txt"
fi
if has bench; then
  timeout 900 python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
  echo "bench rc=$?" | tee -a "$OUT/summary.txt"
  cat "$OUT/bench.json" | tee -a "$OUT/summary.txt"
fi
if has variants; then
  for v in "rays_per_thread=1" "rays_per_thread=4" "nontemporal=1" "xcd_remap=1" \
           "block=128" "block=512" "rays_per_thread=4 nontemporal=1" "nontemporal=1 xcd_remap=1" \
           "rays_per_thread=1 nontemporal=1"; do
    opts=""; for kv in $v; do opts="$opts --option $kv"; done
    echo "== $v" >> "$OUT/variants.log"
    timeout 600 python bench.py --steps 20 --warmup 3 --cpu-sample 0 $opts >> "$OUT/variants.log" 2>> "$OUT/variants.err"
  done
  timeout 600 python bench.py --steps 20 --warmup 3 --cpu-sample 0 --no-clip > "$OUT/bench_noclip.json" 2>> "$OUT/variants.err"
  python - "$OUT/variants.log" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
name = None
for line in open(sys.argv[1]):
    line = line.strip()
    if line.startswith("=="):
        name = line[3:]
    elif line.startswith("{"):
        d = json.loads(line)
        print("%-34s %.3e ops/s  kernel %.3f ms  %.0f GB/s" % (
            name, d["value"], d["roofline"]["kernel_ms"], d["roofline"]["achieved"]))
PY
fi
if has prof; then
  cd /tmp && export TMPDIR=/tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_stats" -- \
      python "$REPO/bench.py" --steps 10 --warmup 2 --cpu-sample 0 > "$OUT/prof_stats.log" 2>&1
  echo "rocprof stats rc=$?" | tee -a "$OUT/summary.txt"
  cd "$REPO"
  find "$OUT/prof_stats" -name "*kernel_stats*.csv" | head -1 | xargs -r head -20 | tee -a "$OUT/summary.txt"
fi
if has pmc; then
  cd /tmp && export TMPDIR=/tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/prof_pmc_$c" -- \
        python "$REPO/bench.py" --steps 3 --warmup 1 --cpu-sample 0 > "$OUT/prof_pmc_$c.log" 2>&1
    echo "rocprof pmc $c rc=$?" | tee -a "$OUT/summary.txt"
  done
  cd "$REPO"
  python scripts/pmc_traffic.py "$OUT" 2>&1 | tee -a "$OUT/summary.txt"
fi
# keep the merged-back output small: drop rocprof's bulky per-run databases
find "$OUT" -name "*.db" -size +8M -delete 2>/dev/null
du -sh "$OUT" | tee -a "$OUT/summary.txt"
