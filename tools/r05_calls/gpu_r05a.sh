#!/bin/bash
# call a: the full GPU suite with the factorized chain as the default; SpMM A/B of the round's three levers on one box
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05a; mkdir -p $O
timeout 600 python -m pytest tests -q -m gpu 2>&1 | tail -25 > $O/pytest_tail.txt
tail -5 $O/pytest_tail.txt
B="python bench.py --steps 30 --warmup 5 --no-extras --no-cpu-baseline --no-configs"
run() { tag=$1; shift; env "$@" timeout 200 $B > $O/$tag.json 2> $O/$tag.err || echo "$tag failed"; }
run valued_1 SSLREC_SPMM_FACTORIZED=0
run factorized_1 SSLREC_SPMM_FACTORIZED=1
run accinit SSLREC_SPMM_FACTORIZED=0 SSLREC_HIP_LIBRARY=$PWD/sslrec_amd/csrc/exp/libsslrec_hip_accinit.so SSLREC_SWEPT_ACC_INIT=1
run accinit_off SSLREC_SPMM_FACTORIZED=0 SSLREC_HIP_LIBRARY=$PWD/sslrec_amd/csrc/exp/libsslrec_hip_accinit.so SSLREC_SWEPT_ACC_INIT=0
run stagger20 SSLREC_SPMM_FACTORIZED=1 SSLREC_XCD_STAGGER=20
run stagger40 SSLREC_SPMM_FACTORIZED=1 SSLREC_XCD_STAGGER=40
run stagger80 SSLREC_SPMM_FACTORIZED=1 SSLREC_XCD_STAGGER=80
run stagger40_valued SSLREC_SPMM_FACTORIZED=0 SSLREC_XCD_STAGGER=40
run valued_2 SSLREC_SPMM_FACTORIZED=0
run factorized_2 SSLREC_SPMM_FACTORIZED=1
SSLREC_SPMM_FACTORIZED=0 timeout 200 python tools/spmm_boundary.py > $O/boundary_valued.json 2> $O/boundary_valued.err
SSLREC_SPMM_FACTORIZED=1 timeout 200 python tools/spmm_boundary.py > $O/boundary_factorized.json 2> $O/boundary_factorized.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob('gpurun_out/r05a/*.json')):
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
