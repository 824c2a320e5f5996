OUT=gpurun_out/r02zm; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "host_generator_replay or training_trajectory_matches or hip_graph_training_in_parity" 2>&1 | tail -4
timeout 900 python - <<PY
import sys, torch, time
sys.path.insert(0, '.')
from sslrec_amd import rng
from bench import time_events
rep = rng.enable_host_replay('cuda:0')
torch.manual_seed(1)
t0 = time.time(); rep.rand((4761460,)); torch.cuda.synchronize(); print('first large draw incl. jump matrix', round(time.time() - t0, 3), 's')
for n in (300000, 4761460, 9231488, 55388928):
    ms = time_events(lambda: rep.rand((n,)), 5, 1)
    print('uniform', n, 'ms', round(ms, 3))
ms = time_events(lambda: rep.keep_mask(4761460, 0.5), 5, 1); print('mask ms', round(ms, 3))
rep.ahead = False
rng.disable_host_replay()
PY
timeout 900 python bench.py --steps 30 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; python - <<PY
import json
j=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1])
print({k:j[k] for k in ('value','ms_per_step')}, j['roofline']['frac'])
print({k:round(v,3) for k,v in j['extras'].items() if 'step_ms' in k})
PY
