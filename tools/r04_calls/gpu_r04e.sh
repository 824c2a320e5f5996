#!/bin/bash
# round 4, call e: full GPU suite, the default bench line, rocprofv3 stats + PMC traffic of the bench command (one counter group per
# pass), FETCH_SIZE of the co-clustered yelp layout, config 5's row-sharded step and 2x4 hybrid products.  Every piece under timeout.
O=gpurun_out/r04e; mkdir -p $O
export TMPDIR=/tmp
R=$(pwd)
timeout 600 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log | cut -c1-300
T0=$(date +%s); timeout 400 python bench.py > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc $? in $(( $(date +%s) - T0 )) s"
python - <<PY
import json
l = json.load(open('$O/bench_line.json'))
print('headline ms/step %.4f frac %.4f launch %.2f us  graph %s  cpu %s' % (l['ms_per_step'], l['roofline']['frac'], l['roofline']['avg_launch_us'], l['roofline'].get('step_as_one_hip_graph'), l.get('cpu_baseline', {}).get('sample')))
for m, v in l.get('roofline_infonce', {}).get('modes', {}).items():
    print('infonce', m, {k: (round(x, 4) if isinstance(x, float) else x) for k, x in v.items() if k.endswith('_ms') or k in ('frac',)})
for t, c in l.get('configs', {}).items():
    if 'error' in c: print(t, c); continue
    r = c['roofline']
    print(t, 'ms/step %.4f' % c['ms_per_step'], r['bound'], 'frac %.4f' % r['frac'], 'spmm us', round(c['extras'].get('spmm_roofline', r).get('avg_launch_us', 0), 2), {k: (round(v, 4) if isinstance(v, float) else v) for k, v in c['extras'].items() if k.startswith('ms_per')}, 'cpu', c.get('cpu_baseline', {}).get('ms_per_step'), c.get('cpu_baseline', {}).get('cores'))
PY
export SSLREC_SPARSE_GRAD=0
CMD="python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras --no-configs"
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o bench -- $CMD > $R/$O/prof_bench.log 2>&1; echo "== rocprof bench exit $?")
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats.csv && head -12 $O/bench_kernel_stats.csv | cut -c1-120
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  (cd /tmp && timeout 150 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $R/$O/pmc_$i -o p -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-configs > $R/$O/pmc_$i.log 2>&1; echo "== pmc [$pmc] exit $?")
done
python tools/pmc_summary.py $O $O/spmm_pmc_summary.json r04 > $O/pmc_summary.log 2>&1; tail -3 $O/pmc_summary.log | cut -c1-300
cp profiles/spmm_traffic.json $O/spmm_traffic.json 2>/dev/null
unset SSLREC_SPARSE_GRAD
for P in 0 auto; do
  (cd /tmp && SSLREC_X=1 timeout 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/yelp_$P -o p -- python $R/tools/xcd_cluster_gpu.py --graph yelp --only $([ $P = auto ] && echo 4 || echo 0) --reps 20 > /dev/null 2>&1; echo "== pmc yelp $P exit $?")
done
python - <<PY
import csv, glob, json
out = {}
for tag in ('0', 'auto'):
    v = [float(r['Counter_Value']) for f in glob.glob('$O/yelp_%s/*counter_collection.csv' % tag) for r in csv.DictReader(open(f)) if 'spmm_swept' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE']
    if v:
        out['cluster_' + tag] = {'launches': len(v), 'FETCH_SIZE_KB_raw_mean': sum(v) / len(v), 'fabric_read_MB_per_launch_x2_gfx950': sum(v) / len(v) * 1024 * 2 / 1e6}
json.dump(out, open('$O/yelp_cluster_fetch.json', 'w'), indent=1)
print(out)
PY
timeout 480 python tools/cfg5_round4.py --out $O/cfg5_round4.json 2> $O/cfg5.err | cut -c1-1500; echo "cfg5 rc $?"; tail -3 $O/cfg5.err | cut -c1-300
rm -rf $O/prof $O/pmc_*/ $O/yelp_0 $O/yelp_auto
