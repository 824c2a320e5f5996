#!/bin/bash
O=gpurun_out/r03c; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -q -k "row_bundled" > $O/pytest_bundled.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_bundled.log
for m in 2 4; do for b in 500 485; do SSLREC_XCD_BALANCE=$b SSLREC_SWEPT_PRIO=$m python bench.py --no-extras --no-cpu-baseline > $O/bench_prio${m}_xcd$b.json 2>> $O/bench.err; done; done
SSLREC_SWEPT_PRIO=4 python tools/spmm_trace.py > $O/trace_prio4.json 2>> $O/trace.err
SSLREC_SWEPT_PRIO=2 SSLREC_SWEPT_LATE_FLUSH=1 python bench.py --no-extras --no-cpu-baseline > $O/bench_prio2_late.json 2>> $O/bench.err
for f in $O/bench*.json; do python - "$f" <<'PY'
import json,sys
try:
    l=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=l['roofline']
    print(sys.argv[1], 'ms/step %.4f frac %.4f launch_us %.2f'%(l['ms_per_step'], r['frac'], r['avg_launch_us']))
except Exception as e: print(sys.argv[1], 'unreadable', e)
PY
done
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03c/trace_prio4.json'))
print({k:d[k] for k in d if k.startswith(('sweep_end_by','corr','block_slowest','slowest'))})
for x in d['launches'][0]['xcd']: print(x)
PY
