#!/bin/bash
# round 6, call k: the SVD view as one node per table (LightGCL sharded suites), the library after the dead-kernel removal (InfoNCE suites, smoke),
# config 5's rank step once more
O=gpurun_out/r06k; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu -k "lightgcl or LightGCL or sharded or two_ranks or infonce or round6" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log | cut -c1-300
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 1500 python tools/cfg5_row_sharded.py --modes all_gather --out $O/cfg5_row_sharded_step_svd_node.json > $O/cfg5_full.log 2> $O/cfg5_full.err; echo "cfg5 full rc $?"; python - <<PY
import json
try:
    d=json.load(open('$O/cfg5_row_sharded_step_svd_node.json')); print(d['step_ms_compute_only'], d['breakdown_check']); print({k[:60]: v.get('ms') for k, v in d['breakdown_ms'].items()})
except Exception as e: print('no json', e); print(open('$O/cfg5_full.err').read()[-1500:])
PY
