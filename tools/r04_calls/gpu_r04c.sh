#!/bin/bash
# round 4, call c: GPU suite, the full default bench line (roofline_infonce + configs), rocprofv3 kernel stats of the bench command
# and of the InfoNCE call, MFMA-busy PMC of the InfoNCE kernels
O=gpurun_out/r04c; mkdir -p $O
export TMPDIR=/tmp
R=$(pwd)
python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log | cut -c1-300
/usr/bin/time -v python bench.py > $O/bench_line.json 2> $O/bench_line.err; echo "bench rc $?"; grep "Elapsed (wall" $O/bench_line.err
python - <<PY
import json
l = json.load(open('$O/bench_line.json'))
print('headline ms/step %.4f frac %.4f launch %.2f us' % (l['ms_per_step'], l['roofline']['frac'], l['roofline']['avg_launch_us']))
ri = l.get('roofline_infonce', {})
for m, v in ri.get('modes', {}).items():
    print('infonce', m, {k: (round(x, 4) if isinstance(x, float) else x) for k, x in v.items() if k.endswith('_ms') or k in ('frac', 'achieved', 'fp32_equivalent_TFLOPs')})
for t, c in l.get('configs', {}).items():
    if 'error' in c: print(t, c); continue
    r = c['roofline']
    print(t, 'ms/step %.4f' % c['ms_per_step'], r['bound'], 'frac %.4f' % r['frac'], {k: (round(v, 4) if isinstance(v, float) else v) for k, v in c['extras'].items() if k.startswith('ms_per')}, 'cpu', c.get('cpu_baseline', {}).get('ms_per_step'))
PY
export SSLREC_SPARSE_GRAD=0
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o bench -- python $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras --no-configs > $R/$O/prof_bench.log 2>&1; echo "== rocprof bench exit $?")
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats.csv && head -14 $O/bench_kernel_stats.csv | cut -c1-150
unset SSLREC_SPARSE_GRAD
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_inf -o inf -- python $R/tools/infonce_profile.py 20 > $R/$O/prof_inf.log 2>&1; echo "== rocprof infonce exit $?")
f=$(find $O/prof_inf -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/infonce_kernel_stats.csv && head -12 $O/infonce_kernel_stats.csv | cut -c1-170
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/$O/pmc_inf -o p -- python $R/tools/infonce_profile.py 5 > $R/$O/pmc_inf.log 2>&1; echo "== pmc infonce exit $?")
python - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob('$O/pmc_inf/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name'].split('(')[0].replace('void ', '').strip()
        acc[name][r['Counter_Name']].append(float(r['Counter_Value']))
out = {k: {c: {'launches': len(v), 'mean': sum(v) / len(v)} for c, v in cs.items()} for k, cs in acc.items()}
json.dump(out, open('$O/infonce_pmc.json', 'w'), indent=1)
for k, cs in out.items():
    if 'infonce' in k and 'lds' in k or 'rowsum' in k:
        print(k, {c: round(v['mean'], 1) for c, v in cs.items()})
PY
rm -rf $O/prof $O/prof_inf $O/pmc_inf
