#!/bin/bash
# round 6, call i: h3 on the un-normalized variant after the three-pass fix (the backward's own tile-transposed planes take the device scale)
O=gpurun_out/r06i; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu -k "unnormalized or lightgcl or LightGCL or sharded or two_ranks or infonce or bench_config" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -8 $O/pytest.log | cut -c1-300
SSLREC_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 6 --warmup 2 --no-cpu-baseline --no-extras --no-configs > $O/bench_gpus2_one_device.json 2> $O/bench_gpus2.err; echo "bench gpus2 rc $?"; python - <<PY
import json
try:
    d=json.loads(open('$O/bench_gpus2_one_device.json').read().strip().splitlines()[-1]); print(json.dumps(d['multi_gpu']['transport_proof'])[:1200]); print({k: v['rccl_ranks'] for k, v in d['decompositions'].items()})
except Exception as e: print('no json', e); print(open('$O/bench_gpus2.err').read()[-1500:])
PY
