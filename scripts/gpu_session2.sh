#!/bin/bash
# Second-style GPU session: probes, SQ/TCC counters, forced-dist bench, tests.
set -u
TAG=${1:-r01b}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$REPO"
timeout 900 python -m pytest tests -m gpu -x -q -k "rccl or golden or errors" > "$OUT/pytest_gpu.log" 2>&1
echo "pytest rc=$?" | tee -a "$OUT/summary.txt"; tail -3 "$OUT/pytest_gpu.log" | tee -a "$OUT/summary.txt"
timeout 600 python scripts/probe.py > "$OUT/probe.json" 2> "$OUT/probe.err"
echo "probe rc=$?" | tee -a "$OUT/summary.txt"; cat "$OUT/probe.json" | tee -a "$OUT/summary.txt"
RT_BENCH_FORCE_DIST=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
   --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 2 --cpu-sample 0 \
   > "$OUT/bench_forced_dist.json" 2> "$OUT/bench_forced_dist.err"
echo "forced-dist bench rc=$?" | tee -a "$OUT/summary.txt"; cat "$OUT/bench_forced_dist.json" | tee -a "$OUT/summary.txt"
tail -5 "$OUT/bench_forced_dist.err" | tee -a "$OUT/summary.txt"
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_ANY" \
           "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum" \
           "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_WR" \
           "GRBM_GUI_ACTIVE TCC_BUSY_sum TCC_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  # (a TCP_* set hung the profiler for its full timeout on this pool: left out)
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pmc$i" -- \
     python "$REPO/bench.py" --steps 2 --warmup 1 --cpu-sample 0 > "$OUT/pmc$i.log" 2>&1
  echo "pmc$i rc=$?" | tee -a "$OUT/summary.txt"
done
cd "$REPO"
python - "$OUT" <<'PY' | tee -a "$OUT/summary.txt"
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for p in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "rt_trace_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print("%-40s %.6g  (n=%d)" % (k, sum(v)/len(v), len(v)))
PY
du -sh "$OUT"
