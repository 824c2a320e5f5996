#!/usr/bin/env python
"""experiment: cut threshold and item splits of the fused evaluation kernel (SSLREC_EVAL_CUT / SSLREC_EVAL_SPLIT)"""
import json, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sslrec_amd import ops
from sslrec_amd.data_utils.synth import make_dataset
from bench import time_events
dev = 'cuda:0'
trn = make_dataset('amazon-book').tocsr(); trn.sort_indices()
U, I = trn.shape
csr = (torch.from_numpy(trn.indptr.astype(np.int64)).to(dev), torch.from_numpy(trn.indices.astype(np.int64)).to(dev))
d = 64
ue, ie = torch.randn(U, d, device=dev) * 0.1, torch.randn(I, d, device=dev) * 0.1
users = torch.arange(U, device=dev)
for k in (20, 40):
    for cap in (64, 128):
        for cut in (cap - 24, cap - 8, cap - 2):
            for nu, splits in ((U, (1, 2)), (1024, (32, 48))):
                for sp in splits:
                    os.environ['SSLREC_EVAL_CAP'] = str(cap); os.environ['SSLREC_EVAL_CUT'] = str(cut); os.environ['SSLREC_EVAL_SPLIT'] = str(sp)
                    ms = time_events(lambda: ops.eval_topk(ue, ie, users[:nu], k, csr), 3, 1)
                    print(json.dumps({'k': k, 'cap': cap, 'cut': cut, 'users': nu, 'split': sp, 'ms': round(ms, 3)}), flush=True)
