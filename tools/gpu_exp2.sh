#!/bin/bash
OUT=gpurun_out/r01b; mkdir -p $OUT; export TMPDIR=/tmp; R=$(pwd)
python tools/spmm_sweep.py > $OUT/sweep_u8.log 2>&1; cat $OUT/sweep_u8.log | grep graph
SSLREC_SPMM_UNROLL=4 python tools/spmm_sweep.py --only amazon-book > $OUT/sweep_u4.log 2>&1; grep graph $OUT/sweep_u4.log
SSLREC_SPMM_UNROLL=16 python tools/spmm_sweep.py --only amazon-book > $OUT/sweep_u16.log 2>&1; grep graph $OUT/sweep_u16.log
cd /tmp
for ord in degree xcd; do
  for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    tag=$(echo $pmc | tr ' ' '_')
    timeout 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $R/$OUT/pmc_${ord}_$tag -o p -- python $R/tools/spmm_sweep.py --only amazon-book --order $ord --reps 5 > $R/$OUT/pmc_${ord}_$tag.log 2>&1
    echo "== pmc $ord $pmc exit $?"
  done
done
cd $R
timeout 600 python -m pytest tests -m gpu -q --tb=short -k "yelp" 2>&1 | tail -3
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o bench -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $R/$OUT/prof_bench.log 2>&1; echo "== rocprof exit $?")
find $OUT -name "*.csv" | head -30
