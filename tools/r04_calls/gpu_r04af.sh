#!/bin/bash
# call af: item splits for a batch of 1024 users with the scalar candidate path
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04af
SWEEP_USERS=1024 SWEEP="2:-,2:8,2:12,2:16,2:24,2:32,2:40,2:48" timeout 200 python tools/eval_variants.py shipped 2>&1 | grep '^{' | tee gpurun_out/r04af/eval_sweep_1024.jsonl
