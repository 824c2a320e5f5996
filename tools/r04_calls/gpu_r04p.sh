#!/bin/bash
O=gpurun_out/r04p; mkdir -p $O
for b in 1 2 3 4; do
  SSLREC_INFONCE_BSPLIT=$b timeout 200 python bench.py --config cfg4 --steps 30 --no-cpu-baseline > $O/cfg4_b$b.json 2>/dev/null
  python - <<PY
import json
c = json.load(open('$O/cfg4_b$b.json'))
r = c['roofline']
print('bsplit $b', 'cfg4 ms/step %.4f' % c['ms_per_step'], 'infonce ms %.4f frac %.4f' % (r.get('ms_per_step', 0), r['frac']), 'graph', round(c['extras']['ms_per_step_as_one_hip_graph'], 4))
PY
done
SSLREC_INFONCE_BSPLIT=2 timeout 200 python bench.py --config cfg3 --steps 30 --no-cpu-baseline > $O/cfg3_b2.json 2>/dev/null
python - <<PY
import json
c = json.load(open('$O/cfg3_b2.json'))
r = c['roofline']
print('bsplit 2', 'cfg3 ms/step %.4f' % c['ms_per_step'], 'infonce ms %.4f frac %.4f' % (r.get('ms_per_step', 0), r['frac']), 'graph', round(c['extras']['ms_per_step_as_one_hip_graph'], 4))
PY
