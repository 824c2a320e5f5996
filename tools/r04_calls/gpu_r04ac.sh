#!/bin/bash
# round 4, call ac (validation after the evaluation / SimGCL / event-sampling changes): the GPU suite as the driver runs it, smoke, the default bench line as the driver runs it, kernel stats, InfoNCE modes
O=gpurun_out/r04ac; mkdir -p $O
export TMPDIR=/tmp
R=$(pwd)
T0=$(date +%s); timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $? in $(( $(date +%s) - T0 )) s"; tail -3 $O/pytest.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
T0=$(date +%s); timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc $? in $(( $(date +%s) - T0 )) s"
python - <<PY
import json
l = json.load(open('$O/bench_line.json'))
print('headline ms/step %.4f value %.4g frac %.4f launch %.2f us  graph %s' % (l['ms_per_step'], l['value'], l['roofline']['frac'], l['roofline']['avg_launch_us'], l['roofline'].get('step_as_one_hip_graph')))
print('cpu', l.get('cpu_baseline', {}).get('sample'))
for m, v in l.get('roofline_infonce', {}).get('modes', {}).items():
    print('infonce', m, {k: (round(x, 4) if isinstance(x, float) else x) for k, x in v.items() if k.endswith('_ms') or k in ('frac',)})
for t, c in l.get('configs', {}).items():
    if 'error' in c: print(t, c); continue
    r = c['roofline']
    print(t, 'ms/step %.4f' % c['ms_per_step'], r['bound'], 'frac %.4f' % r['frac'], 'spmm', round(c['extras'].get('spmm_roofline', r).get('avg_launch_us', 0), 2), 'us frac', round(c['extras'].get('spmm_roofline', r).get('frac', 0), 3), {k: (round(v, 4) if isinstance(v, float) else v) for k, v in c['extras'].items() if k.startswith('ms_per')}, 'cpu', c.get('cpu_baseline', {}).get('ms_per_step'), c.get('cpu_baseline', {}).get('cores'))
x = l.get('extras', {})
print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in x.items() if 'eval' in k or 'step_ms' in k or 'spmm_plain' in k or 'streamed' in k})
PY
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
unset SSLREC_SPARSE_GRAD
for c in cfg3 cfg4; do
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$c -o $c -- python $R/bench.py --config $c --steps 30 --no-cpu-baseline > $R/$O/prof_$c.log 2>&1; echo "== rocprof $c exit $?")
  f=$(find $O/prof_$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${c}_kernel_stats.csv && head -8 $O/${c}_kernel_stats.csv | cut -c1-110
done
rm -rf $O/prof $O/prof_cfg3 $O/prof_cfg4
