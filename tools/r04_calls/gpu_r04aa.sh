#!/bin/bash
# call aa: evaluation top-k with the few-candidate tiles handled by scalar control (EV_FEW_LANES = 0 / 3 / 6 / 12 / 24)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04aa; mkdir -p $O
timeout 500 python -m pytest tests -q -m gpu -x -k "eval or topk or predict or c_abi or metric" 2>&1 | tail -4
SWEEP="2:-,2:1" timeout 200 python tools/eval_variants.py few6 2>&1 | grep '^{' | tee -a $O/eval_sweep.jsonl
for v in few0 few3 few12 few24; do
  SSLREC_HIP_LIBRARY="$PWD/tools/variants/ev_$v.so" SWEEP="2:-,2:1" timeout 200 python tools/eval_variants.py $v 2>&1 | grep '^{' | tee -a $O/eval_sweep.jsonl
done
