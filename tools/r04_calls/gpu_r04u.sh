#!/bin/bash
# call u: evaluation top-k, all users: item splits (with the shared thresholds of round 4) x register sets, with and without candidates
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04u
SWEEP="2:-,3:-,2:1,2:2,2:3,2:4,2:5,2:6,2:8,2:10,3:5" timeout 300 python tools/eval_variants.py full 2>&1 | grep '^{' | tee -a gpurun_out/r04u/eval_sweep.jsonl
SSLREC_HIP_LIBRARY="$PWD/tools/variants/eval_no_cand.so" SWEEP="2:1,3:1,2:2,2:5,3:5,2:10" timeout 300 python tools/eval_variants.py nocand 2>&1 | grep '^{' | tee -a gpurun_out/r04u/eval_sweep.jsonl
