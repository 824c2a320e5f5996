#!/bin/bash
# call j: evaluation of a user batch -- the splits' published best scores as a threshold bound (csrc/eval.hip `share`): the evaluation tests,
# A/B timings at 256 / 1024 / 2048 users, rocprofv3 kernel stats of both forms
cd "$GRAFT_REPO_ROOT"
R=$PWD
O=gpurun_out/r05j; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests -q -m gpu -k "evaluation or eval or predict or topk" 2>&1 | tail -8 > $O/pytest_eval_tail.txt; tail -5 $O/pytest_eval_tail.txt
for s in 1 0 1 0; do SSLREC_EVAL_SHARE_TOP1=$s timeout 300 python tools/eval_small_batch.py $O/eval_small_batch.jsonl | tr '\n' ' '; echo; done
cat > /tmp/ev_prof.py <<'PY'
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
from sslrec_amd import ops
from sslrec_amd.data_utils.synth import make_dataset
trn = make_dataset('amazon-book').tocsr(); trn.sort_indices()
U, I = trn.shape
csr = (torch.from_numpy(trn.indptr.astype(np.int64)).cuda(), torch.from_numpy(trn.indices.astype(np.int64)).cuda())
gen = torch.Generator().manual_seed(3)
ue, ie = (torch.randn(U, 64, generator=gen) * 0.1).cuda(), (torch.randn(I, 64, generator=gen) * 0.1).cuda()
users = torch.randperm(U, generator=gen).cuda()
for _ in range(30):
    ops.eval_topk(ue, ie, users[:1024], 40, csr)
torch.cuda.synchronize()
PY
for s in 1 0; do
  (cd /tmp && SSLREC_EVAL_SHARE_TOP1=$s timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$s -o ev -- python /tmp/ev_prof.py > $R/$O/prof_$s.log 2>&1; echo "== rocprof share=$s exit $?")
  f=$(find $O/prof_$s -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/eval_1024_share${s}_kernel_stats.csv && head -5 $f | cut -c1-160; rm -rf $O/prof_$s
done
