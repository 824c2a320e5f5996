#!/bin/bash
# call ab: SimGCL's three views share the whole backward chain (one linear map on the summed upstream gradients)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04ab; mkdir -p $O
timeout 700 python -m pytest tests -q -m gpu -x -k "simgcl or views or config_lines" 2>&1 | tail -4
timeout 300 python bench.py --config cfg3 --steps 30 --warmup 5 > $O/cfg3_line.json 2> $O/cfg3.err; echo "cfg3 rc $?"
python - <<'PY'
import json
l = json.loads(open('gpurun_out/r04ab/cfg3_line.json').read().strip().splitlines()[-1])
x = l['extras']
print('cfg3 ms/step %.4f frac %.4f' % (l['ms_per_step'], l['roofline']['frac']), 'spmm launches/step', x['spmm_launches_per_step'], 'spmm ms/step %.4f' % x['spmm_ms_per_step'],
      'graph', x.get('ms_per_step_as_one_hip_graph'), 'parity', x.get('ms_per_step_parity_mode_generator_on_device'))
PY
