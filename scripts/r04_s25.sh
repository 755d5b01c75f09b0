#!/bin/bash
# round 4, session 25: soaks of the final code on the device -- random systems
# against the oracle on both asphere arithmetics, the state fuzz on both
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s25
mkdir -p "$OUT"
cd "$REPO"
( time RT_MI355_EXACT_ASPHERE=1 timeout 400 python tests/tools/soak_random.py 0 1500 ) 2>&1 | tail -5 | tee "$OUT/soak_random_exact.txt"
( time RT_MI355_EXACT_ASPHERE=0 timeout 400 python tests/tools/soak_random.py 0 1500 ) 2>&1 | tail -5 | tee "$OUT/soak_random_default.txt"
( time RT_MI355_EXACT_ASPHERE=1 timeout 400 python tests/tools/fuzz_state.py 1000 3000 30 ) 2>&1 | tail -5 | tee "$OUT/fuzz_exact.txt"
( time RT_FUZZ_ARITH=default RT_MI355_EXACT_ASPHERE=0 timeout 400 python tests/tools/fuzz_state.py 1000 3000 30 ) 2>&1 | tail -5 | tee "$OUT/fuzz_default.txt"
