#!/usr/bin/env python
"""the fused evaluation kernel alone (all users of the amazon-book-shaped data, d = 64, k = 40), for rocprofv3 --pmc passes"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sslrec_amd import ops
from sslrec_amd.data_utils.synth import make_dataset
dev = 'cuda:0'
trn = make_dataset('amazon-book').tocsr(); trn.sort_indices()
U, I = trn.shape
csr = (torch.from_numpy(trn.indptr.astype(np.int64)).to(dev), torch.from_numpy(trn.indices.astype(np.int64)).to(dev))
d = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ue, ie = torch.randn(U, d, device=dev) * 0.1, torch.randn(I, d, device=dev) * 0.1
for _ in range(3):
    ops.eval_topk(ue, ie, None, 40, csr)
torch.cuda.synchronize()
