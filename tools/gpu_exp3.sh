#!/bin/bash
OUT=gpurun_out/r01c; mkdir -p $OUT; export TMPDIR=/tmp; R=$(pwd)
for grp in "spmm or propagate" "tiny" "yelp" "amazon"; do
  timeout 900 python -m pytest tests -m gpu -q --tb=short -k "$grp" > $OUT/test_$(echo $grp | tr ' ' '_').log 2>&1
  echo "== pytest -k '$grp' exit $?"; tail -2 $OUT/test_$(echo $grp | tr ' ' '_').log
done
python tools/spmm_sweep.py 2>&1 | grep graph
python tools/spmm_sweep.py --only amazon-book --fold 4096 2>&1 | grep graph
SSLREC_SPMM_UNROLL=4 python tools/spmm_sweep.py --only amazon-book 2>&1 | grep graph
SSLREC_SPMM_UNROLL=16 python tools/spmm_sweep.py --only amazon-book 2>&1 | grep graph
timeout 900 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/bench.log 2> $OUT/bench.err; echo "== bench exit $?"; cat $OUT/bench.log | cut -c1-1500
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o bench -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras > $R/$OUT/prof_bench.log 2>&1; echo "== rocprof exit $?"; head -5 $R/$OUT/prof/bench_kernel_stats.csv)
