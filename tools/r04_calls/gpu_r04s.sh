#!/bin/bash
# call s: event pairs around every 5th SpMM launch in the timed regions -- the bench tests, the new hook test, the default bench line
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04s
timeout 600 python -m pytest tests -q -m gpu -x -k "bench or sampled_launch or zero_row_hint" 2>&1 | tail -4
t0=$(date +%s)
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04s/bench_line.json 2> gpurun_out/r04s/bench_line.err
echo "bench rc $? in $(( $(date +%s) - t0 )) s"
python - <<'PY'
import json
l = json.loads(open('gpurun_out/r04s/bench_line.json').read().strip().splitlines()[-1])
r = l['roofline']
print('headline ms/step %.4f value %.4g frac %.4f launch %.2f us launches %s timed %s | %s' % (l['ms_per_step'], l['value'], r['frac'], r['avg_launch_us'], r['launches'], r.get('launches_timed'), r['launch_timing']))
print('graph', r.get('step_as_one_hip_graph'), 'hint', (r.get('with_zero_row_hint') or {}).get('ms_per_step'))
for c in l.get('configs', []):
    x = c.get('extras', {})
    print(c['config']['workload'][:4], 'ms/step %.4f' % c['ms_per_step'], c['roofline']['bound'], 'frac %.4f' % c['roofline']['frac'],
          'spmm', (x.get('spmm_roofline') or c['roofline']).get('avg_launch_us'), 'graph', x.get('ms_per_step_as_one_hip_graph'), 'eager', x.get('ms_per_step_eager_launches'))
PY
