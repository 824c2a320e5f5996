#!/bin/bash
# round 4, call f: one-launch reductions with two-level tickets vs the two-launch form (A/B on one box: step time + kernel stats)
O=gpurun_out/r04f; mkdir -p $O
export TMPDIR=/tmp
R=$(pwd)
timeout 300 python -m pytest tests -x -q -m gpu -k "bpr or sum_squares or regularizer or training_step_matches_reference_tiny or hip_graph_training_equals" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log | cut -c1-200
for m in 1 0 1 0; do
  SSLREC_ONE_LAUNCH_REDUCE=$m timeout 200 python bench.py --steps 50 --warmup 10 --no-extras --no-cpu-baseline --no-configs > $O/bench_one$m.json 2>$O/bench.err
  python - <<PY
import json
l = json.load(open('$O/bench_one$m.json'))
r = l['roofline']
print('one_launch=$m ms/step %.4f  launch %.2f us  graph %s' % (l['ms_per_step'], r['avg_launch_us'], r.get('step_as_one_hip_graph')))
PY
done
export SSLREC_SPARSE_GRAD=0
for m in 1 0; do
  (cd /tmp && SSLREC_ONE_LAUNCH_REDUCE=$m timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof$m -o bench -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras --no-configs > $R/$O/prof$m.log 2>&1; echo "== rocprof $m exit $?")
  f=$(find $O/prof$m -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats_one$m.csv && python - <<PY
import csv
tot = 0
for r in csv.DictReader(open('$O/bench_kernel_stats_one$m.csv')):
    if int(r['Calls']) >= 100:
        print('  %-46s calls %4d avg %7.1f us' % (r['Name'].split('(')[0][:46], int(r['Calls']), float(r['AverageNs']) / 1e3))
        tot += float(r['TotalDurationNs']) / 113 / 1e3
print('  GPU time per step %.1f us' % tot)
PY
done
rm -rf $O/prof0 $O/prof1
