#!/bin/bash
# round 2, GPU session 1: parity suite, probes, bench
O=gpurun_out/r02_s1
mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
tail -5 $O/pytest.txt
timeout 900 python scripts/r02_probe.py ABCD > $O/probe.jsonl 2> $O/probe.err; echo "probe rc $?"
tail -3 $O/probe.err
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
tail -2 $O/bench.err
timeout 600 python bench.py --extras --cpu-sample 0 > $O/bench_extras.json 2> $O/bench_extras.err; echo "extras rc $?"
cat $O/bench.json | head -c 3000
