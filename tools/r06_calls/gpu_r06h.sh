#!/bin/bash
# round 6, call h: h3 on the un-normalized variant after the padding-row fix -- the InfoNCE / LightGCL / sharded suites, then config 5's rank step
O=gpurun_out/r06h; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu -k "unnormalized or lightgcl or LightGCL or sharded or two_ranks or infonce or bench_config" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -8 $O/pytest.log | cut -c1-300
timeout 1500 python tools/cfg5_row_sharded.py --modes all_gather --out $O/cfg5_row_sharded_step_h3.json > $O/cfg5_full.log 2> $O/cfg5_full.err; echo "cfg5 full rc $?"; python - <<PY
import json
try:
    d=json.load(open('$O/cfg5_row_sharded_step_h3.json')); print(d['step_ms_compute_only']); print({k[:50]: v.get('ms') for k, v in d['breakdown_ms'].items()})
except Exception as e: print('no json', e); print(open('$O/cfg5_full.err').read()[-1500:])
PY
