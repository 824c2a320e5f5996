#!/bin/bash
# call e: h3 as the library default -- the full GPU suite, the default bench line, cfg3 / cfg4 kernel stats
cd "$GRAFT_REPO_ROOT"
R=$PWD
O=gpurun_out/r05e; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu --durations=12 2>&1 | tail -40 > $O/pytest_tail.txt; tail -22 $O/pytest_tail.txt
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench_line.err || echo "bench failed"
for c in cfg3 cfg4; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$c -o $c -- python $R/bench.py --config $c --steps 30 --no-cpu-baseline > $R/$O/${c}_line.json 2> $R/$O/${c}_prof.err; echo "== rocprof $c exit $?")
  f=$(find $O/prof_$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${c}_kernel_stats.csv; rm -rf $O/prof_$c
done
python - <<'PY'
import json
O='gpurun_out/r05e/'
try:
    j = json.loads(open(O+'bench_line.json').read().strip().splitlines()[-1])
    r = j['roofline']
    print('bench', j['ms_per_step'], r['avg_launch_us'], r['frac'], r.get('launch_us_by_position_in_step'))
    ri = j.get('roofline_infonce', {})
    print('infonce', ri.get('mode', '')[:40], ri.get('frac'), {k: (round(v['fwdbwd_ms'], 4), round(v['frac'], 4)) for k, v in ri.get('modes', {}).items()})
    for k, v in j.get('configs', {}).items():
        print(k, v.get('ms_per_step'), v.get('headline_form'), v.get('roofline', {}).get('frac'), {a: round(b, 4) for a, b in v.get('extras', {}).items() if a.startswith('ms_per_step') and isinstance(b, float)})
except Exception as e:
    print('bench unreadable', e)
PY
