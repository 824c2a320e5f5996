#!/bin/bash
# rocprofv3 kernel stats of the other single-GPU config lines (python bench.py --config cfg3 / cfg4)   usage: bash tools/gpu_prof_configs.sh [tag]
T=${1:-r03cfgprof}; O=gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
R=$(pwd)
for c in cfg3 cfg4 cfg1; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$c -o $c -- python $R/bench.py --config $c --steps 30 > $R/$O/prof_$c.log 2>&1; echo "== rocprof $c exit $?")
  f=$(find $O/prof_$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${c}_kernel_stats.csv && head -12 $O/${c}_kernel_stats.csv | cut -c1-140
  tail -1 $O/prof_$c.log | cut -c1-300
done
