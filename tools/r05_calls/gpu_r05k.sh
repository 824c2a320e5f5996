#!/bin/bash
# call k: evaluation with FEW item splits (all users) -- the splits' published m-th bests as a bound: evaluation tests incl. the new
# h3 / few-splits parity test, A/B at all users (SSLREC_EVAL_SHARE_TOP1=1/0), the small batches again
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05k; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -k "evaluation or eval or predict or topk" 2>&1 | tail -8 > $O/pytest_eval_tail.txt; tail -5 $O/pytest_eval_tail.txt
cat > /tmp/ev_all.py <<'PY'
import json, os, sys, numpy as np, torch
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
from sslrec_amd import ops
from sslrec_amd.data_utils.synth import make_dataset
from bench import time_events
trn = make_dataset('amazon-book').tocsr(); trn.sort_indices()
U, I = trn.shape
csr = (torch.from_numpy(trn.indptr.astype(np.int64)).cuda(), torch.from_numpy(trn.indices.astype(np.int64)).cuda())
gen = torch.Generator().manual_seed(3)
for d in (64, 128):
    ue, ie = (torch.randn(U, d, generator=gen) * 0.1).cuda(), (torch.randn(I, d, generator=gen) * 0.1).cuda()
    users = torch.randperm(U, generator=gen).cuda()
    for k in (20, 40):
        for nu in (16384, 32768, U):
            ms = time_events(lambda: ops.eval_topk(ue, ie, users[:nu], k, csr), 8, 2)
            print(json.dumps({'share': os.environ.get('SSLREC_EVAL_SHARE_TOP1', '1'), 'users': nu, 'k': k, 'd': d, 'ms': round(ms, 4)}), flush=True)
PY
for s in 1 0 1 0; do SSLREC_EVAL_SHARE_TOP1=$s timeout 300 python /tmp/ev_all.py | tee -a $O/eval_all_users.jsonl | tr '\n' ' '; echo; done
