#!/bin/bash
# round 4, session 2: a map of the device memory in 1 GiB chunks (store
# pattern per chunk, two / four workgroups per CU, back to back and one launch
# at a time), then the engine on 8 hipMalloc + 4 chunked allocations with the
# same two ways of timing
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/r04_s2
mkdir -p "$OUT"
cd "$REPO"
timeout 400 scripts/labsrc/chunk_lab 8 230 1024 > "$OUT/chunk_lab.jsonl" 2> "$OUT/chunk_lab.err"
echo "chunk_lab rc=$?"; tail -3 "$OUT/chunk_lab.err"; wc -l "$OUT/chunk_lab.jsonl"
timeout 400 python scripts/lab.py placement --contexts 8 --vmm 4 --serial 8 > "$OUT/placement_plain.jsonl" 2> "$OUT/placement_plain.err"
echo "placement rc=$?"; tail -3 "$OUT/placement_plain.err"
