#!/bin/bash
O=gpurun_out/r03b; mkdir -p $O
python tools/spmm_trace.py > $O/trace.json 2> $O/trace.err
for m in 1 2 3; do SSLREC_SWEPT_PRIO=$m python bench.py --no-extras --no-cpu-baseline > $O/bench_prio$m.json 2>> $O/bench.err; done
SSLREC_SWEPT_PRIO=2 python tools/spmm_trace.py > $O/trace_prio2.json 2>> $O/trace.err
python bench.py --no-extras --no-cpu-baseline > $O/bench_base.json 2>> $O/bench.err
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
for f in ('trace','trace_prio2'):
    d=json.load(open('gpurun_out/r03b/%s.json'%f))
    print(f, {k:d[k] for k in d if k.startswith(('sweep_end_by','corr','block_slowest','slowest'))})
PY
