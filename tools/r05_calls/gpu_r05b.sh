#!/bin/bash
# call b: the factorized chain with straight-line flush-record / row-factor loads; the tests call a failed
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05b; mkdir -p $O
timeout 600 python -m pytest tests -q -m gpu -k "propagate_sum or deferred_layer_sum or single_rank_equals_unsharded or round5 or scale_flags or factorized or (whole_training_step_at_amazon and simgcl)" 2>&1 | tail -60 > $O/pytest_tail.txt
tail -30 $O/pytest_tail.txt
B="python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline --no-configs"
run() { tag=$1; shift; env "$@" timeout 200 $B > $O/$tag.json 2> $O/$tag.err || echo "$tag failed"; }
run valued_1 SSLREC_SPMM_FACTORIZED=0
run factorized_1 SSLREC_SPMM_FACTORIZED=1
run valued_2 SSLREC_SPMM_FACTORIZED=0
run factorized_2 SSLREC_SPMM_FACTORIZED=1
SSLREC_SPMM_FACTORIZED=0 timeout 200 python tools/spmm_boundary.py > $O/boundary_valued.json 2> $O/boundary_valued.err
SSLREC_SPMM_FACTORIZED=1 timeout 200 python tools/spmm_boundary.py > $O/boundary_factorized.json 2> $O/boundary_factorized.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r05b/*.json')):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), 'unreadable', e); continue
    if 'roofline' in j:
        r = j['roofline']
        print('%-22s ms/step %.4f  launch %.2f us  frac %.4f  graph %s' % (os.path.basename(f), j['ms_per_step'], r['avg_launch_us'], r['frac'],
              {k: round(v, 4) if isinstance(v, float) else v for k, v in (r.get('step_as_one_hip_graph') or {}).items()}))
    else:
        print(os.path.basename(f), {k: (round(v, 2) if isinstance(v, float) else v) for k, v in j.items() if k != 'workload'})
PY
