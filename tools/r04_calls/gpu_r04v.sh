#!/bin/bash
# call v: evaluation kernels after the wait-count fixes (unconditional prefetch, look-ahead pair behind the MFMA chain, batched mask loads)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04v; mkdir -p $O
timeout 500 python -m pytest tests -q -m gpu -x -k "eval or topk or predict or c_abi" 2>&1 | tail -4
SWEEP="2:-,3:-,2:1,2:2,2:3,2:4,2:5,2:6,2:8,2:10,3:1,3:5" timeout 300 python tools/eval_variants.py full 2>&1 | grep '^{' | tee -a $O/eval_sweep.jsonl
SSLREC_HIP_LIBRARY="$PWD/tools/variants/eval_no_cand.so" SWEEP="2:1,3:1,2:2,2:5,3:5,2:10" timeout 300 python tools/eval_variants.py nocand 2>&1 | grep '^{' | tee -a $O/eval_sweep.jsonl
timeout 100 python - <<'PY'
import sys, json, numpy as np, torch
sys.path.insert(0, '.')
from bench import time_events
from sslrec_amd import ops
from sslrec_amd.data_utils.synth import make_dataset
dev = 'cuda:0'
trn = make_dataset('amazon-book').tocsr()
n_user, n_item = trn.shape
ue, ie = torch.randn(n_user, 64, device=dev) * 0.1, torch.randn(n_item, 64, device=dev) * 0.1
users = torch.arange(n_user, device=dev)
mask = torch.from_numpy(trn[:1024].toarray().astype(np.int64)).to(dev)
out = {'full_predict_1024_users_int64_mask_ms': round(time_events(lambda: ops.full_predict(ue, ie, users[:1024], mask), 10, 2), 4),
       'full_predict_1024_users_bool_mask_ms': round(time_events(lambda: ops.full_predict(ue, ie, users[:1024], mask.bool()), 10, 2), 4),
       'full_predict_1024_users_no_mask_ms': round(time_events(lambda: ops.full_predict(ue, ie, users[:1024]), 10, 2), 4)}
def stock():
    sc = ue[:1024] @ ie.T
    return sc * (1 - mask) - 1e8 * mask
out['stock_torch_full_predict_1024_users_ms'] = round(time_events(stock, 5, 1), 4)
print(json.dumps(out))
open('gpurun_out/r04v/full_predict.json', 'w').write(json.dumps(out, indent=1))
PY
