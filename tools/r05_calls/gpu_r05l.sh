#!/bin/bash
# call l: the driver's sequence after the evaluation work -- full GPU suite, smoke, default bench line
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05l; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -q -m gpu --durations=12 2>&1 | tail -30 > $O/pytest_gpu_tail.txt; tail -22 $O/pytest_gpu_tail.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
S=$(date +%s); timeout 900 python bench.py > $O/bench_line.json 2> $O/bench_line.err || echo "bench failed"; E=$(date +%s); echo "bench wall $((E-S)) s"
python - <<'PY'
import json
j = json.loads(open('gpurun_out/r05l/bench_line.json').read().strip().splitlines()[-1])
r = j['roofline']
print('bench', j['ms_per_step'], r['avg_launch_us'], r['frac'], r['traffic'], j['steps'], j['warmup'])
ri = j.get('roofline_infonce', {})
print('infonce', {k: (round(v['fwdbwd_ms'], 4), round(v['frac'], 4)) for k, v in ri.get('modes', {}).items()})
for k, v in j.get('configs', {}).items():
    print(k, v.get('ms_per_step'), v.get('headline_form'))
print({k: v for k, v in j.get('extras', {}).items() if 'eval' in k})
PY
