#!/usr/bin/env python
"""fused evaluation top-k, amazon-book size, d = 64: item splits per user group (SSLREC_EVAL_SPLIT) for batches of 2048 - 16384 users -- does
raising the splits to >= k (which switches the published-maxima bound on, csrc/eval.hip `share`) pay beyond the 1024-user batch?
usage: python tools/eval_split_sweep.py [out.jsonl]"""
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
for k in (40, 20):
    for nu in (2048, 4096, 8192, 16384, U):
        for sp in (0, k, 48, 64):
            if sp:
                os.environ['SSLREC_EVAL_SPLIT'] = str(sp)
            else:
                os.environ.pop('SSLREC_EVAL_SPLIT', None)
            ms = time_events(lambda: ops.eval_topk(ue, ie, users[:nu], k, csr), 8, 2)
            rec = {'users': nu, 'k': k, 'split': sp or 'default', 'ms': round(ms, 4)}
            print(json.dumps(rec), flush=True)
            if out:
                out.write(json.dumps(rec) + '\n')
