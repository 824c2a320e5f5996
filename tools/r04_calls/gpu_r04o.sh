#!/bin/bash
O=gpurun_out/r04o; mkdir -p $O
for b in 1 0 2; do
  if [ $b = 0 ]; then unset SSLREC_INFONCE_BSPLIT; else export SSLREC_INFONCE_BSPLIT=$b; fi
  timeout 200 python tools/traj_margin.py sgl 2>/dev/null | tee -a $O/traj_margin.jsonl
done
