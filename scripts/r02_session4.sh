#!/bin/bash
# round 2, GPU session 4: full parity suite (reference-procedure aiming on the
# device, N=2 host path), compaction with sparser survivor counts
O=gpurun_out/r02_s4
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
grep -E "passed|failed|rc" $O/pytest.txt | tail -3
timeout 900 python scripts/r02_probe.py C > $O/compaction.jsonl 2> $O/compaction.err; echo "compaction rc $?"; cat $O/compaction.jsonl
