#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
OUT=gpurun_out/r03_s16
mkdir -p $OUT
for o in "alloc_vmm_mb=1024 alloc_vmm_align_mb=1024" "alloc_vmm_mb=1024 alloc_vmm_align_mb=2" "alloc_vmm_mb=2048 alloc_vmm_align_mb=2048"; do
  timeout 200 python scripts/r03_alloc_lab.py $o 2>> $OUT/vmm.err | tee -a $OUT/alloc_vmm_align.jsonl
done
