#!/bin/bash
# round 4, call g: kept scatter table (two launches per BPR backward) vs fresh workspace + clearing launch: tests, step time, kernel stats
O=gpurun_out/r04g; mkdir -p $O
export TMPDIR=/tmp
R=$(pwd)
timeout 400 python -m pytest tests -x -q -m gpu -k "bpr or gather_backward or training_step_matches_reference or trajectory_matches or hip_graph_training or sharded_model or feature_sliced_ranks" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log | cut -c1-200
for m in 1 0 1 0; do
  SSLREC_KEPT_SCATTER=$m timeout 200 python bench.py --steps 50 --warmup 10 --no-extras --no-cpu-baseline --no-configs > $O/bench_kept$m.json 2>$O/bench.err
  python - <<PY
import json
l = json.load(open('$O/bench_kept$m.json'))
r = l['roofline']
print('kept=$m ms/step %.4f  launch %.2f us  graph %s' % (l['ms_per_step'], r['avg_launch_us'], r.get('step_as_one_hip_graph')))
PY
done
export SSLREC_SPARSE_GRAD=0
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o bench -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras --no-configs > $R/$O/prof.log 2>&1; echo "== rocprof exit $?")
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats.csv && python - <<PY
import csv
tot = 0
for r in csv.DictReader(open('$O/bench_kernel_stats.csv')):
    if int(r['Calls']) >= 100:
        print('  %-46s calls %4d avg %7.1f us' % (r['Name'].split('(')[0][:46], int(r['Calls']), float(r['AverageNs']) / 1e3))
        tot += float(r['TotalDurationNs']) / 113 / 1e3
print('  GPU time per step %.1f us' % tot)
PY
rm -rf $O/prof
