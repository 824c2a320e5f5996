#!/bin/bash
# call i: the driver's sequence on the round's code -- full GPU suite, smoke, default bench line (live traffic passes inside), rocprofv3
# kernel stats and PMC passes of the bench command, kernel stats of the cfg3 / cfg4 steps, the InfoNCE call's stats and matrix-pipe counters
cd "$GRAFT_REPO_ROOT"
R=$PWD
O=gpurun_out/r05i; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -12 > $O/pytest_gpu_tail.txt; tail -4 $O/pytest_gpu_tail.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err || echo "bench failed"
bash tools/gpu_profile.sh r05i > $O/gpu_profile.log 2>&1; tail -6 $O/gpu_profile.log | cut -c1-300
for c in cfg3 cfg4; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$c -o $c -- python $R/tools/step_profile.py $c 40 > $R/$O/${c}_prof.log 2>&1; echo "== rocprof $c exit $?")
  f=$(find $O/prof_$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${c}_kernel_stats.csv; rm -rf $O/prof_$c
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_inf -o inf -- python $R/tools/infonce_profile.py > $R/$O/infonce_prof.log 2>&1; echo "== rocprof infonce exit $?")
f=$(find $O/prof_inf -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/infonce_kernel_stats.csv; rm -rf $O/prof_inf
(cd /tmp && timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$O/pmc_inf -o p -- python $R/tools/infonce_profile.py > $R/$O/infonce_pmc.log 2>&1; echo "== pmc infonce exit $?")
python - <<'PY'
import csv, glob, collections, json
O='gpurun_out/r05i/'
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O+'pmc_inf/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'].split('(')[0].replace('void ', '').strip()][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, cs in acc.items():
    if 'infonce' in k and 'SQ_VALU_MFMA_BUSY_CYCLES' in cs and 'GRBM_GUI_ACTIVE' in cs:
        busy, act = sum(cs['SQ_VALU_MFMA_BUSY_CYCLES']) / len(cs['SQ_VALU_MFMA_BUSY_CYCLES']), sum(cs['GRBM_GUI_ACTIVE']) / len(cs['GRBM_GUI_ACTIVE'])
        out[k] = {'launches': len(cs['GRBM_GUI_ACTIVE']), 'SQ_VALU_MFMA_BUSY_CYCLES': busy, 'GRBM_GUI_ACTIVE': act, 'matrix_pipe_busy_frac_per_simd': busy / act / 1024.0}
json.dump(out, open(O+'infonce_pmc.json', 'w'), indent=1)
print(json.dumps(out)[:900])
try:
    j = json.loads(open(O+'bench_line.json').read().strip().splitlines()[-1])
    r = j['roofline']
    print('bench', j['ms_per_step'], r['avg_launch_us'], r['frac'], r['traffic'], r['traffic_source'][:160])
    print(r.get('launch_us_by_position_in_step'), r.get('step_as_one_hip_graph'))
    ri = j.get('roofline_infonce', {})
    print('infonce', {k: (round(v['fwdbwd_ms'], 4), round(v['frac'], 4)) for k, v in ri.get('modes', {}).items()})
    for k, v in j.get('configs', {}).items():
        print(k, round(v.get('ms_per_step', 0), 4), v.get('headline_form'), round(v.get('roofline', {}).get('frac', 0), 4))
    e = j['extras']
    print({k: round(e[k], 3) for k in e if ('eval' in k or 'spmm_plain' in k) and isinstance(e[k], float)})
except Exception as ex:
    print('bench unreadable', ex)
PY
rm -rf $O/pmc_inf
