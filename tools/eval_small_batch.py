#!/usr/bin/env python
"""fused evaluation top-k of a BATCH of users at amazon-book size (the reference's test batch: 1024 users, metrics.py:99-108), d = 64:
time per call for k = 20 / 40 and 256 / 1024 / 2048 users.  SSLREC_EVAL_SHARE_TOP1=0 in the environment switches the published-maxima
bound off (csrc/eval.hip `share`) for the A/B.   usage: python tools/eval_small_batch.py [out.jsonl]"""
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
gen = torch.Generator().manual_seed(3)
ue, ie = (torch.randn(U, d, generator=gen) * 0.1).to(dev), (torch.randn(I, d, generator=gen) * 0.1).to(dev)
users = torch.randperm(U, generator=gen).to(dev)
out = open(sys.argv[1], 'a') if len(sys.argv) > 1 else None
for k in (20, 40):
    for nu in (256, 1024, 2048):
        ms = time_events(lambda: ops.eval_topk(ue, ie, users[:nu], k, csr), 20, 3)
        rec = {'share_top1': os.environ.get('SSLREC_EVAL_SHARE_TOP1', '1'), 'users': nu, 'k': k, 'd': d, 'items': I, 'ms': round(ms, 4)}
        print(json.dumps(rec), flush=True)
        if out:
            out.write(json.dumps(rec) + '\n')
