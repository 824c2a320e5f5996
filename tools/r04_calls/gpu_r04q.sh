#!/bin/bash
O=gpurun_out/r04q; mkdir -p $O
timeout 600 python -m pytest tests -x -q -m gpu -k "infonce or training_step or trajectory or whole_training_step or sharded" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log | cut -c1-300
timeout 200 python tools/traj_margin.py sgl lightgcn 2>/dev/null | tee $O/traj_margin.json
for c in cfg3 cfg4; do
  timeout 200 python bench.py --config $c --steps 30 --no-cpu-baseline > $O/$c.json 2> $O/$c.err
  python - <<PY
import json
c = json.load(open('$O/$c.json'))
r = c['roofline']
print('$c', 'ms/step %.4f' % c['ms_per_step'], r['bound'], 'frac %.4f' % r['frac'], 'infonce ms %.4f' % r.get('ms_per_step', 0), 'spmm ms %.4f' % c['extras']['spmm_ms_per_step'], {k: (round(v, 4) if isinstance(v, float) else v) for k, v in c['extras'].items() if k.startswith('ms_per')})
PY
done
