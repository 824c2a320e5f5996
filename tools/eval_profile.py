#!/usr/bin/env python
"""fused all-rank evaluation on the amazon-book-shaped data: batches of 1024 users and all users, k = 20 / 40, with the
reference's expression (dense mask assumed resident) beside it; also usable under rocprofv3 --stats
usage: python tools/eval_profile.py [out.json]"""
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
out = {'n_user': U, 'n_item': I}
for d in (64, 128):
    ue, ie = torch.randn(U, d, device=dev) * 0.1, torch.randn(I, d, device=dev) * 0.1
    users = torch.arange(U, device=dev)
    for k in (20, 40):
        out['d%d_k%d_1024_users_ms' % (d, k)] = round(time_events(lambda: ops.eval_topk(ue, ie, users[:1024], k, csr), 10, 2), 4)
        ms = time_events(lambda: ops.eval_topk(ue, ie, users, k, csr), 5, 1)
        out['d%d_k%d_all_users_ms' % (d, k)] = round(ms, 3)
        out['d%d_k%d_all_users_mfma_frac' % (d, k)] = round(2.0 * U * I * d / (ms * 1e-3) / 157.3e12, 4)
    # trained-like tables: a few popular items dominate every user's list early (the easy case for thresholds) is NOT assumed;
    # the adversarial order: item scores increasing with the item id, every item beats the running k-th best
    if d == 64:
        out['d64_k40_all_users_no_train_mask_ms'] = round(time_events(lambda: ops.eval_topk(ue, ie, users, 40, None), 5, 1), 3)
        ie_sorted = ie[torch.argsort((ie * ue[:1]).sum(1))]
        out['d64_k40_1024_users_ascending_scores_of_user0_ms'] = round(time_events(lambda: ops.eval_topk(ue, ie_sorted, users[:1024], 40, csr), 5, 1), 4)
    mask = torch.from_numpy(trn[:1024].toarray().astype(np.float32)).to(dev)

    def stock():
        sc = ue[:1024] @ ie.T
        return torch.topk(sc * (1 - mask) - 1e8 * mask, 40)[1]
    out['d%d_stock_torch_1024_users_mask_resident_ms' % d] = round(time_events(stock, 5, 1), 4)
    got = ops.eval_topk(ue, ie, users[:1024], 40, csr)
    want = stock()
    out['d%d_same_as_stock_frac' % d] = round((got == want).float().mean().item(), 5)
    del mask
print(json.dumps(out))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], 'w'), indent=1)
