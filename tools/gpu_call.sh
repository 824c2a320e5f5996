OUT=gpurun_out/r02zn; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "two_ranks or host_generator_replay or hip_graph_training_in_parity" 2>&1 | tail -4
timeout 300 python - <<PY
import sys, torch
sys.path.insert(0, '.')
from sslrec_amd import rng
from bench import time_events
rep = rng.enable_host_replay('cuda:0')
torch.manual_seed(1)
rep.rand((4761460,)); torch.cuda.synchronize()
for n in (4761460, 9231488, 55388928):
    print('uniform', n, 'ms', round(time_events(lambda: rep.rand((n,)), 5, 1), 3))
rep.ahead = False
PY
