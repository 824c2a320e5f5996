mkdir -p gpurun_out/r02j
timeout 600 python -m pytest tests -m gpu -q -x --tb=short -k "fused_evaluation or device_side" > gpurun_out/r02j/test_eval.log 2>&1; echo "pytest exit $?"; tail -2 gpurun_out/r02j/test_eval.log
timeout 300 python - <<'PY'
import torch, sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from sslrec_amd import ops
from sslrec_amd.data_utils.synth import make_dataset
from bench import time_events
dev='cuda:0'
trn = make_dataset('amazon-book').tocsr(); trn.sort_indices()
U, I = trn.shape
csr = (torch.from_numpy(trn.indptr.astype(np.int64)).to(dev), torch.from_numpy(trn.indices.astype(np.int64)).to(dev))
ue, ie = torch.randn(U, 64, device=dev)*0.1, torch.randn(I, 64, device=dev)*0.1
users = torch.arange(U, device=dev)
for n in (1024, U):
    print(n, 'users: %.3f ms' % time_events(lambda: ops.eval_topk(ue, ie, users[:n], 40, csr), 5, 1), flush=True)
PY
