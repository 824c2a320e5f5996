#!/bin/bash
# round 6, call n: the evaluation kernel with the item tile staged once per 8-wave workgroup in LDS (SSLREC_EVAL_STAGE, default on):
# the evaluation tests, then all users / user counts with the stage on and off, and a split sweep of the staged form
O=gpurun_out/r06n; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu -k "eval or topk or metric or Metric" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest.log | cut -c1-300
timeout 900 python - > $O/eval_stage.log 2>&1 <<'PY'
import json, os, subprocess, sys
code = r'''
import json, os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
from sslrec_amd import ops
from sslrec_amd.data_utils.synth import make_dataset
from bench import time_events
dev = 'cuda:0'
trn = make_dataset('amazon-book').tocsr(); trn.sort_indices()
U, I = trn.shape
csr = (torch.from_numpy(trn.indptr.astype(np.int64)).to(dev), torch.from_numpy(trn.indices.astype(np.int64)).to(dev))
out = {}
for d in (64, 32):
    gen = torch.Generator().manual_seed(d)
    ue, ie = (torch.randn(U, d, generator=gen) * 0.1).to(dev), (torch.randn(I, d, generator=gen) * 0.1).to(dev)
    users = torch.arange(U, device=dev)
    for n in (4096, 16384, 32768, U):
        for k in (20, 40):
            out['d%d_k%d_%d_users_ms' % (d, k, n)] = round(time_events(lambda: ops.eval_topk(ue, ie, users[:n], k, csr), 5, 2), 3)
    idx = ops.eval_topk(ue, ie, users, 40, csr)
    out['d%d_checksum' % d] = int(idx.sum().item())
print(json.dumps(out))
'''
res = {}
for tag, env in (('stage_on', {}), ('stage_off', {'SSLREC_EVAL_STAGE': '0'}), ('stage_on_split3', {'SSLREC_EVAL_SPLIT': '3'}), ('stage_on_split4', {'SSLREC_EVAL_SPLIT': '4'}),
                 ('stage_on_split5', {'SSLREC_EVAL_SPLIT': '5'}), ('stage_on_split8', {'SSLREC_EVAL_SPLIT': '8'}), ('stage_on_again', {}), ('stage_off_again', {'SSLREC_EVAL_STAGE': '0'})):
    r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
    try:
        res[tag] = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception:
        res[tag] = {'error': (r.stderr or r.stdout)[-600:]}
    print(tag, json.dumps(res[tag]), flush=True)
json.dump(res, open('gpurun_out/r06n/eval_stage_ab.json', 'w'), indent=1)
PY
echo "ab rc $?"; cut -c1-700 $O/eval_stage.log
