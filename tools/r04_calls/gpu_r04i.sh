#!/bin/bash
# round 4, call i: where the evaluation kernel's time goes (no-candidate floor; PMC) + the coalesced full_predict kernel
O=gpurun_out/r04i; mkdir -p $O
export TMPDIR=/tmp
R=$(pwd)
timeout 100 python tools/eval_variants.py shipped | tee -a $O/eval_variants.jsonl
SSLREC_HIP_LIBRARY=$R/tools/variants/libsslrec_nocand.so timeout 100 python tools/eval_variants.py no-candidates | tee -a $O/eval_variants.jsonl
timeout 120 python -m pytest tests -x -q -m gpu -k "full_predict or device_side_evaluation" 2>&1 | tail -2
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
open('gpurun_out/r04i/full_predict.json', 'w').write(json.dumps(out, indent=1))
PY
i=0
for pmc in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU"; do
  i=$((i+1))
  (cd /tmp && timeout 100 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $R/$O/pmc_$i -o p -- python $R/tools/eval_variants.py pmc > /dev/null 2>&1; echo "== pmc [$pmc] exit $?")
done
python - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob('$O/pmc_*/*counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        if 'eval_topk_kernel' in r['Kernel_Name']:
            acc[r['Grid_Size'] if 'Grid_Size' in r else 'k'][r['Counter_Name']].append(float(r['Counter_Value']))
out = {k: {c: {'n': len(v), 'mean': sum(v) / len(v)} for c, v in cs.items()} for k, cs in acc.items()}
json.dump(out, open('$O/eval_pmc.json', 'w'), indent=1)
for k, cs in out.items():
    print(k, {c: round(v['mean']) for c, v in cs.items()})
PY
rm -rf $O/pmc_*/
