#!/bin/bash
# round 4, call d: new tests, the full default bench line, the locality-aware plan on four graphs (time, bit identity, FETCH_SIZE)
O=gpurun_out/r04d; mkdir -p $O
export TMPDIR=/tmp
R=$(pwd)
python -m pytest tests -x -q -m gpu -k "co_clustered or infonce or bundled or config_lines" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log | cut -c1-300
T0=$(date +%s); python bench.py > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc $? in $(( $(date +%s) - T0 )) s"
python - <<PY
import json
l = json.load(open('$O/bench_line.json'))
print('headline ms/step %.4f frac %.4f launch %.2f us' % (l['ms_per_step'], l['roofline']['frac'], l['roofline']['avg_launch_us']))
ri = l.get('roofline_infonce', {})
print('roofline_infonce', {k: v for k, v in ri.items() if k in ('achieved', 'frac', 'peak', 'error')})
for m, v in ri.get('modes', {}).items():
    print('infonce', m, {k: (round(x, 4) if isinstance(x, float) else x) for k, x in v.items() if k.endswith('_ms') or k in ('frac', 'achieved', 'fp32_equivalent_TFLOPs')})
for t, c in l.get('configs', {}).items():
    if 'error' in c: print(t, c); continue
    r = c['roofline']
    print(t, 'ms/step %.4f' % c['ms_per_step'], r['bound'], 'frac %.4f' % r['frac'], {k: (round(v, 4) if isinstance(v, float) else v) for k, v in c['extras'].items() if k.startswith('ms_per')}, 'cpu ms', c.get('cpu_baseline', {}).get('ms_per_step'), c.get('cpu_baseline', {}).get('error'))
PY
for g in headline item_exp1 planted yelp; do
  python tools/xcd_cluster_gpu.py --graph $g --out $O/xcd_cluster.jsonl 2>/dev/null | cut -c1-600
done
for g in headline planted yelp; do for P in 0 4; do
  (cd /tmp && timeout 200 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_${g}_$P -o p -- python $R/tools/xcd_cluster_gpu.py --graph $g --only $P --reps 20 > /dev/null 2>&1; echo "== pmc $g $P exit $?")
done; done
python - <<PY
import csv, glob, collections, json
out = {}
for d in sorted(glob.glob('$O/pmc_*')):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + '/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if 'spmm_swept' in r['Kernel_Name']:
                acc[r['Counter_Name']].append(float(r['Counter_Value']))
    out[d.split('pmc_')[1]] = {c: {'launches': len(v), 'mean': sum(v) / len(v)} for c, v in acc.items()}
json.dump(out, open('$O/xcd_cluster_pmc.json', 'w'), indent=1)
for k, v in out.items():
    print(k, {c: round(x['mean'], 1) for c, x in v.items()})
PY
rm -rf $O/pmc_*/
