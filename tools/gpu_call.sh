OUT=gpurun_out/r02zo; mkdir -p $OUT
export TMPDIR=/tmp
ROOTDIR=$(pwd)
for m in lightgcl sgl; do
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTDIR/$OUT/prof_$m -o p -- python $ROOTDIR/tools/epoch_demo.py $m 1 graph fused > $ROOTDIR/$OUT/demo_$m.log 2>&1; echo "== $m exit $?")
tail -2 $OUT/demo_$m.log
f=$(find $OUT/prof_$m -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${m}_kernel_stats.csv && head -14 $OUT/${m}_kernel_stats.csv | cut -c 1-150
done
