#!/bin/bash
O=gpurun_out/r03e; mkdir -p $O
python -m pytest tests -m gpu -q --maxfail=25 > $O/pytest.log 2>&1; echo "pytest rc $?" | tee $O/pytest.rc; tail -15 $O/pytest.log
python bench.py --steps 50 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
for c in cfg1 cfg3 cfg4; do python bench.py --config $c --steps 30 >> $O/configs.jsonl 2>> $O/bench.err; done
SSLREC_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --steps 5 --warmup 2 > $O/bench_g2_one_device.json 2> $O/bench_g2.err; echo "bench g2 rc $?"
for f in $O/bench.json $O/bench_g2_one_device.json; do python - "$f" <<'PY'
import json,sys
try:
    l=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=l['roofline']
    print(sys.argv[1], 'ms/step %.4f frac %.4f launch_us %.2f value %.3e'%(l['ms_per_step'], r['frac'], r['avg_launch_us'], l['value']))
except Exception as e: print(sys.argv[1], 'unreadable', e)
PY
done
python - <<'PY'
import json
for l in open('gpurun_out/r03e/configs.jsonl'):
    l=json.loads(l); print(l['config']['workload'][:40], 'ms/step %.3f frac %.3f launch_us %.1f'%(l['ms_per_step'], l['roofline']['frac'], l['roofline']['avg_launch_us']))
PY
timeout 900 python tools/cfg5_step.py --scale 1.0 > $O/cfg5_step.json 2> $O/cfg5.err; echo "cfg5 full rc $?"; tail -c 1500 $O/cfg5_step.json
