#!/bin/bash
# call w: what bounds the evaluation loop -- the no-candidate build without its item loads / without the look-ahead pair
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04w; mkdir -p $O
for v in nc nc_nomask nc_noload; do
  SSLREC_HIP_LIBRARY="$PWD/tools/variants/ev_$v.so" SWEEP="2:1,2:5,2:10" timeout 200 python tools/eval_variants.py $v 2>&1 | grep '^{' | tee -a $O/eval_sweep.jsonl
done
