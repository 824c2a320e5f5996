#!/bin/bash
# call x: evaluation after the round's experiments: item splits for many users, full_predict with batched mask loads
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04x; mkdir -p $O
timeout 500 python -m pytest tests -q -m gpu -x -k "eval or topk or predict or c_abi or metric" 2>&1 | tail -4
SWEEP="2:-,2:1,2:3" timeout 300 python tools/eval_variants.py shipped 2>&1 | grep '^{' | tee -a $O/eval_sweep.jsonl
