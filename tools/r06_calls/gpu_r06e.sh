#!/bin/bash
# round 6, call e: the fused / pipelined LightGCL graph view on the real kernels (2 and 8 processes on this GPU), the Trainer's
# one-synchronisation epoch + the device-side loader, config 5's rank step in its three forms at full size
O=gpurun_out/r06e; mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu -k "two_ranks or sharded or lightgcl or trainer or Trainer or trajectory or epoch or loader or graph" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log | cut -c1-300
timeout 600 python tools/cfg5_row_sharded.py --scale 0.02 --reps 2 > $O/cfg5_small.json 2> $O/cfg5_small.err; echo "cfg5 small rc $?"; python - <<PY
import json
try:
    d=json.loads(open('$O/cfg5_small.json').read().strip().split('\n')[-1]); print(json.dumps(d['modes'])[:900])
except Exception as e: print('no json', e); print(open('$O/cfg5_small.err').read()[-1500:])
PY
timeout 1500 python tools/cfg5_row_sharded.py --out $O/cfg5_row_sharded_step.json > $O/cfg5_full.log 2> $O/cfg5_full.err; echo "cfg5 full rc $?"; python - <<PY
import json
try:
    d=json.load(open('$O/cfg5_row_sharded_step.json')); print(json.dumps(d['modes'])); print(json.dumps(d['breakdown_ms'])[:1500])
except Exception as e: print('no json', e); print(open('$O/cfg5_full.err').read()[-1500:])
PY
