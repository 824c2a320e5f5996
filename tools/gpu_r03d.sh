#!/bin/bash
O=gpurun_out/r03d; mkdir -p $O
ROOTDIR=$(pwd); export TMPDIR=/tmp
python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench.err
SSLREC_SWEPT_NT_STORES=2 python bench.py --no-extras --no-cpu-baseline > $O/bench_sc1.json 2>> $O/bench.err
SSLREC_SPMM_SWEPT=0 python bench.py --no-extras --no-cpu-baseline > $O/bench_streamed.json 2>> $O/bench.err
SSLREC_SPMM_SWEPT=0 SSLREC_STREAM_PRIO=1 python bench.py --no-extras --no-cpu-baseline > $O/bench_streamed_prio.json 2>> $O/bench.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTDIR/$O/prof -o bench -- python $ROOTDIR/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras > $ROOTDIR/$O/prof_bench.log 2>&1; echo "== rocprof stats exit $?")
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats.csv && head -6 $O/bench_kernel_stats.csv
rm -rf $O/prof
for f in $O/bench*.json; do python - "$f" <<'PY'
import json,sys
try:
    l=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=l['roofline']
    print(sys.argv[1], 'ms/step %.4f frac %.4f launch_us %.2f'%(l['ms_per_step'], r['frac'], r['avg_launch_us']), r['kernel'][:24])
    e=l.get('extras',{})
    for k in e:
        if 'device_clock' in k or 'hip_events_same' in k: print('   ',k,e[k])
except Exception as e: print(sys.argv[1], 'unreadable', e)
PY
done
timeout 300 python tools/cfg5_step.py --scale 0.02 > $O/cfg5_step_small.json 2> $O/cfg5_small.err; echo "cfg5 small rc $?"; tail -c 1500 $O/cfg5_step_small.json; tail -5 $O/cfg5_small.err
SSLREC_PLAN_TIMING=1 timeout 900 python tools/cfg5_step.py --scale 1.0 > $O/cfg5_step.json 2> $O/cfg5.err; echo "cfg5 full rc $?"; tail -c 2500 $O/cfg5_step.json; tail -30 $O/cfg5.err
