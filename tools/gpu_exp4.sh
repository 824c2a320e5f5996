#!/bin/bash
OUT=gpurun_out/r01d; mkdir -p $OUT; export TMPDIR=/tmp; R=$(pwd)
timeout 900 python -m pytest tests -m gpu -q --tb=short -k "spmm or propagate or tiny or amazon" > $OUT/test.log 2>&1; echo "== pytest exit $?"; tail -2 $OUT/test.log
for u in 8 4 16; do
  SSLREC_SPMM_UNROLL=$u python tools/spmm_sweep.py --only amazon-book 2>&1 | grep graph
  SSLREC_SPMM_UNROLL=$u python tools/spmm_sweep.py --only amazon-book --fold 4096 --order degree 2>&1 | grep graph
done
python tools/spmm_sweep.py --only yelp --order degree 2>&1 | grep graph
