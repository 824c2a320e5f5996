#!/bin/bash
# round-3 GPU call A: the whole GPU suite on the new swept kernel, bench line, A/B switches, time line, N > 1 code path
O=gpurun_out/r03a; mkdir -p $O
python -m pytest tests -m gpu -q --maxfail=25 > $O/pytest.log 2>&1; echo "pytest rc $?" | tee $O/pytest.rc
python bench.py --steps 50 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
SSLREC_SWEPT_LATE_FLUSH=1 python bench.py --no-extras --no-cpu-baseline > $O/bench_late_flush.json 2>> $O/bench.err
SSLREC_SWEPT_NT_STORES=1 python bench.py --no-extras --no-cpu-baseline > $O/bench_nt_stores.json 2>> $O/bench.err
for b in 485 515; do SSLREC_XCD_BALANCE=$b python bench.py --no-extras --no-cpu-baseline > $O/bench_xcd_$b.json 2>> $O/bench.err; done
python tools/spmm_trace.py > $O/trace.json 2> $O/trace.err
SSLREC_SWEPT_LATE_FLUSH=1 python tools/spmm_trace.py > $O/trace_late.json 2>> $O/trace.err
SSLREC_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench_g2_one_device.json 2> $O/bench_g2.err; echo "bench g2 rc $?"
tail -3 $O/pytest.log
for f in $O/bench*.json; do python - "$f" <<'PY'
import json,sys
try:
    l=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=l['roofline']
    print(sys.argv[1], 'ms/step %.4f frac %.4f launch_us %.2f'%(l['ms_per_step'], r['frac'], r['avg_launch_us']))
except Exception as e: print(sys.argv[1], 'unreadable', e)
PY
done
