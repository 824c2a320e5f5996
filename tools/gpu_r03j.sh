#!/bin/bash
O=gpurun_out/r03j; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -k "draw_ahead or trajectory or hip_graph or host_generator or trainer_runs or training_step_matches_reference" --maxfail=10 > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -15 $O/pytest.log | cut -c1-300
python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<'PY'
import json
l=json.loads(open('gpurun_out/r03j/bench.json').read().strip().splitlines()[-1]); e=l['extras']
print('bench ms/step %.4f frac %.4f'%(l['ms_per_step'], l['roofline']['frac']))
for k in ('lightgcn_step_ms_device_rng','simgcl_step_ms_device_rng','lightgcn_step_ms_parity_generator_on_device','simgcl_step_ms_parity_generator_on_device','lightgcn_step_ms_cpu_rng_parity'): print(' ',k, e.get(k))
PY
