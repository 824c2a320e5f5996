#!/bin/bash
# round 4, call b: GPU suite (deferred layer sum, bench over RCCL) + bench A/B of the deferred layer sum
O=gpurun_out/r04b; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log | cut -c1-300
for m in 1 0 1 0; do
  SSLREC_DEFERRED_SUM=$m python bench.py --steps 50 --warmup 10 --no-extras --no-cpu-baseline > $O/bench_deferred$m.json 2>$O/bench.err
  python - <<PY
import json
l = json.load(open('$O/bench_deferred$m.json'))
r = l['roofline']
print('deferred=$m ms/step %.4f  launch %.2f us  frac %.4f  bytes %.1f MB  graph %s' % (l['ms_per_step'], r['avg_launch_us'], r['frac'], r['algorithmic_bytes_per_launch'] / 1e6, r.get('step_as_one_hip_graph')))
PY
done
