#!/bin/bash
# call ag: up to 64 item splits (merge of 4096 candidates) for small user batches
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04ag
timeout 300 python -m pytest tests -q -m gpu -x -k "eval or topk or predict or c_abi or metric" 2>&1 | tail -3
SWEEP_USERS=1024 SWEEP="2:-,2:48,2:56,2:64" timeout 200 python tools/eval_variants.py shipped 2>&1 | grep '^{' | tee gpurun_out/r04ag/eval_sweep_1024.jsonl
