#!/usr/bin/env python
"""times the fused evaluation kernel of whatever library SSLREC_HIP_LIBRARY points at (experiments: variants of csrc/eval.hip built
into tools/variants/*.so).  usage: SSLREC_HIP_LIBRARY=... python tools/eval_variants.py [tag]"""
import json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import time_events
from sslrec_amd import ops
from sslrec_amd.data_utils.synth import make_dataset
dev = 'cuda:0'
trn = make_dataset('amazon-book').tocsr(); trn.sort_indices()
n_user, n_item = trn.shape
csr = (torch.from_numpy(trn.indptr.astype(np.int64)).to(dev), torch.from_numpy(trn.indices.astype(np.int64)).to(dev))
gen = torch.Generator().manual_seed(0)
ue = (torch.randn(n_user, 64, generator=gen) * 0.1).to(dev)
ie = (torch.randn(n_item + 32, 64, generator=gen) * 0.1).to(dev)[:n_item]      # (a view: the row-major experiment build reads the last tile's rows past n_item)
users = torch.arange(n_user, device=dev)
tag = sys.argv[1] if len(sys.argv) > 1 else os.environ.get('SSLREC_HIP_LIBRARY', 'default')
# SWEEP="bufs:split,bufs:split,..." (all users only) -- the library reads SSLREC_EVAL_BUFS / SSLREC_EVAL_SPLIT at every call
sweep = os.environ.get('SWEEP')
if sweep:
    for spec in sweep.split(','):
        bufs, split = spec.split(':')
        os.environ['SSLREC_EVAL_BUFS'] = bufs
        if split == '-':
            os.environ.pop('SSLREC_EVAL_SPLIT', None)
        else:
            os.environ['SSLREC_EVAL_SPLIT'] = split
        rec = {'tag': tag, 'bufs': bufs, 'split': split}
        for n in ((1024, n_user) if split == '-' else ((1024,) if os.environ.get('SWEEP_USERS') == '1024' else (n_user,))):
            rec['topk40_%d_users_ms' % n] = round(time_events(lambda: ops.eval_topk(ue, ie, users[:n], 40, csr), 6, 2), 4)
        print(json.dumps(rec), flush=True)
else:
    out = {'tag': tag, 'bufs': os.environ.get('SSLREC_EVAL_BUFS', 'default')}
    for n in (1024, n_user):
        out['topk40_%d_users_ms' % n] = round(time_events(lambda: ops.eval_topk(ue, ie, users[:n], 40, csr), 10, 2), 4)
    print(json.dumps(out), flush=True)
