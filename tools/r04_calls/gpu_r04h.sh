#!/bin/bash
# round 4, call h: shared evaluation thresholds across item splits; zero fill of the gradient table fused into the BPR staging launch
O=gpurun_out/r04h; mkdir -p $O
export TMPDIR=/tmp
R=$(pwd)
timeout 400 python -m pytest tests -x -q -m gpu -k "evaluation or topk or full_predict or bpr or training_step_matches_reference or hip_graph_training or infonce_forward_that_keeps or sharded_model" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log | cut -c1-200
timeout 200 python - <<'PY'
import sys, json, numpy as np, scipy.sparse as sp, torch
sys.path.insert(0, '.')
from bench import time_events
from sslrec_amd import ops
from sslrec_amd.data_utils.synth import make_dataset
dev = 'cuda:0'
trn = make_dataset('amazon-book').tocsr(); trn.sort_indices()
n_user, n_item = trn.shape
csr = (torch.from_numpy(trn.indptr.astype(np.int64)).to(dev), torch.from_numpy(trn.indices.astype(np.int64)).to(dev))
ue, ie = torch.randn(n_user, 64, device=dev) * 0.1, torch.randn(n_item, 64, device=dev) * 0.1
users = torch.arange(n_user, device=dev)
out = {}
for n in (128, 1024, 4096, n_user):
    for k in (20, 40):
        out['topk%d_%d_users_ms' % (k, n)] = round(time_events(lambda: ops.eval_topk(ue, ie, users[:n], k, csr), 10, 2), 4)
mask = torch.from_numpy(trn[:1024].toarray().astype(np.int64)).to(dev)
out['full_predict_1024_users_int64_mask_ms'] = round(time_events(lambda: ops.full_predict(ue, ie, users[:1024], mask), 10, 2), 4)
def stock():
    sc = ue[:1024] @ ie.T
    return sc * (1 - mask) - 1e8 * mask
out['stock_torch_full_predict_1024_users_ms'] = round(time_events(stock, 5, 1), 4)
print(json.dumps(out))
open('gpurun_out/r04h/eval.json', 'w').write(json.dumps(out, indent=1))
PY
for m in 1 0 1 0; do
  SSLREC_KEPT_SCATTER=$m timeout 200 python bench.py --steps 50 --warmup 10 --no-extras --no-cpu-baseline --no-configs > $O/bench_kept$m.json 2>$O/bench.err
  python - <<PY
import json
l = json.load(open('$O/bench_kept$m.json'))
r = l['roofline']
print('kept+fused-zero=$m ms/step %.4f  launch %.2f us  graph %s' % (l['ms_per_step'], r['avg_launch_us'], r.get('step_as_one_hip_graph')))
PY
done
