#!/bin/bash
# Runs on the GPU box (via gpurun): parity tests in isolated processes, smoke, bench, rocprof summary.
# usage: tools/gpu_check.sh [tag]
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== rocminfo"; rocminfo | grep -E "Marketing Name|gfx9" | head -4
nproc
for grp in "spmm or propagate" "bpr" "infonce" "tiny" "yelp" "amazon"; do
  name=$(echo $grp | tr ' ' '_')
  timeout 900 python -m pytest tests -m gpu -q --tb=short -k "$grp" > $OUT/test_$name.log 2>&1
  echo "== pytest -k '$grp' exit $?"; tail -3 $OUT/test_$name.log
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "== smoke exit $?"; tail -2 $OUT/smoke.log
timeout 900 python bench.py --steps 50 --warmup 5 > $OUT/bench.log 2> $OUT/bench.err; echo "== bench exit $?"; tail -c 3000 $OUT/bench.log; tail -5 $OUT/bench.err
ROOTDIR=$(pwd)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTDIR/$OUT/prof -o bench -- python $ROOTDIR/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras > $ROOTDIR/$OUT/prof_bench.log 2>&1; echo "== rocprof exit $?")
find $OUT/prof -name "*kernel_stats*" | head; f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f"
for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pmc | tr ' ' '_')
  (cd /tmp && timeout 300 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $ROOTDIR/$OUT/pmc_$tag -o p -- python $ROOTDIR/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $ROOTDIR/$OUT/pmc_$tag.log 2>&1; echo "== pmc $pmc exit $?")
done
