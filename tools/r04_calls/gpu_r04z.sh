#!/bin/bash
# call z: the shipped evaluation loop without candidates / without its item loads (the train-mask merge intact), by item splits
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04z; mkdir -p $O
for v in nc nc_noload noload; do
  SSLREC_HIP_LIBRARY="$PWD/tools/variants/ev_$v.so" SWEEP="2:1,2:3,2:10" timeout 200 python tools/eval_variants.py $v 2>&1 | grep '^{' | tee -a $O/eval_sweep.jsonl
done
