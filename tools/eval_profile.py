#!/usr/bin/env python
"""fused all-rank evaluation on the amazon-book-shaped data: 10 x 1024 users and 3 x all users (for rocprofv3 --stats)"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sslrec_amd import ops
from sslrec_amd.data_utils.synth import make_dataset
dev = 'cuda:0'
trn = make_dataset('amazon-book').tocsr(); trn.sort_indices()
U, I = trn.shape
csr = (torch.from_numpy(trn.indptr.astype(np.int64)).to(dev), torch.from_numpy(trn.indices.astype(np.int64)).to(dev))
ue, ie = torch.randn(U, 64, device=dev) * 0.1, torch.randn(I, 64, device=dev) * 0.1
users = torch.arange(U, device=dev)
for _ in range(10):
    ops.eval_topk(ue, ie, users[:1024], 40, csr)
torch.cuda.synchronize()
for _ in range(3):
    ops.eval_topk(ue, ie, users, 40, csr)
torch.cuda.synchronize()
