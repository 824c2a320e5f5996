#!/bin/bash
OUT=gpurun_out/r01f; mkdir -p $OUT; export TMPDIR=/tmp; R=$(pwd)
timeout 900 python -m pytest tests -m gpu -q --tb=short -k "spmm or propagate or tiny or amazon or yelp" > $OUT/test.log 2>&1; echo "== pytest exit $?"; tail -3 $OUT/test.log
python tools/spmm_sweep.py --only amazon-book 2>&1 | grep graph
python tools/spmm_sweep.py --only amazon-book --fold 4096 --order degree 2>&1 | grep graph
for blk in 524288 1048576 4194304 33554432; do
  SSLREC_SWEEP_BLOCK_BYTES=$blk python tools/spmm_sweep.py --only amazon-book --order degree 2>&1 | grep graph
done
SSLREC_SPMM_MODE=stream python tools/spmm_sweep.py --only amazon-book --order degree 2>&1 | grep graph
python tools/spmm_sweep.py --only yelp --order degree 2>&1 | grep graph
python tools/spmm_sweep.py --only gowalla --order degree 2>&1 | grep graph
cd /tmp
for pmc in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pmc | tr ' ' '_')
  timeout 200 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $R/$OUT/pmc_$tag -o p -- python $R/tools/spmm_sweep.py --only amazon-book --order degree --reps 3 > $R/$OUT/pmc_$tag.log 2>&1
  echo "== pmc $pmc exit $?"
done
