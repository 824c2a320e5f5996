#!/bin/bash
O=gpurun_out/r03g; mkdir -p $O
run() { tag=$1; shift; env "$@" python bench.py --config cfg4 --steps 30 > $O/cfg4_$tag.json 2>> $O/err.log; env "$@" python bench.py --config cfg1 --steps 30 > $O/cfg1_$tag.json 2>> $O/err.log; }
run default X=1
run prio0 SSLREC_SWEPT_PRIO=0
run plainstores SSLREC_SWEPT_NT_STORES=0
run both SSLREC_SWEPT_PRIO=0 SSLREC_SWEPT_NT_STORES=0
run late SSLREC_SWEPT_PRIO=0 SSLREC_SWEPT_NT_STORES=0 SSLREC_SWEPT_LATE_FLUSH=1
for f in $O/cfg*.json; do python - "$f" <<'PY'
import json,sys
try:
    l=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], 'ms/step %.3f launch_us %.1f'%(l['ms_per_step'], l['roofline']['avg_launch_us']))
except Exception as e: print(sys.argv[1], 'unreadable', e)
PY
done
