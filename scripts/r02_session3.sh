#!/bin/bash
# round 2, GPU session 3: full parity suite, compaction timing, bench +
# rocprofv3 kernel stats of the default bench command
O=gpurun_out/r02_s3
mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
tail -6 $O/pytest.txt
timeout 600 python scripts/r02_probe.py C > $O/compaction.jsonl 2> $O/compaction.err; echo "compaction rc $?"; cat $O/compaction.jsonl
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_bench -o bench -- python bench.py --cpu-sample 0 > $O/bench_prof.json 2> $O/bench_prof.err; echo "rocprof bench rc $?"
python scripts/r02_collect.py $O $O/collected > $O/summary.txt 2>&1; cat $O/summary.txt
head -c 1500 $O/bench.json
