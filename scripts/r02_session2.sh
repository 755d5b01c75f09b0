#!/bin/bash
# round 2, GPU session 2: new parity tests, fast-vs-exact audit, store-flavour
# probes, rocprofv3 kernel stats + PMC for C4 exact/fast and the bench
O=gpurun_out/r02_s2
mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
tail -4 $O/pytest.txt
timeout 300 python scripts/r02_fastcheck.py > $O/fastcheck.jsonl 2> $O/fastcheck.err; echo "fastcheck rc $?"; cat $O/fastcheck.jsonl
timeout 300 python scripts/r02_store_flavours.py > $O/store_flavours.jsonl 2> $O/store_flavours.err; echo "flavours rc $?"
# rocprofv3: kernel trace + stats of the default bench command
cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o bench -- python bench.py --cpu-sample 0 > $O/bench_prof.json 2> $O/bench_prof.err; echo "rocprof bench rc $?"
# C4 exact and fast: kernel stats, then PMC passes (separate runs)
for fast in 0 1; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_c4_fast$fast -o c4 -- python scripts/r02_c4_run.py $fast > $O/c4_fast$fast.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES -d $O/pmc_c4_fast$fast -o c4 -- python scripts/r02_c4_run.py $fast 5 > $O/c4_pmc_fast$fast.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc2_c4_fast$fast -o c4 -- python scripts/r02_c4_run.py $fast 5 > $O/c4_pmc2_fast$fast.log 2>&1
done
find $O -name "*.csv" | head -40
python scripts/r02_collect.py $O > $O/summary.txt 2>&1; cat $O/summary.txt | head -60
