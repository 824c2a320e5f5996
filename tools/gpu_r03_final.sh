#!/bin/bash
O=gpurun_out/r03v; mkdir -p $O
python -m pytest tests -m gpu -q --maxfail=25 --durations=25 > $O/pytest.log 2>&1; echo "pytest rc $?" | tee $O/pytest.rc; grep -E "passed|failed|error" $O/pytest.log | tail -3; grep -A 28 "slowest" $O/pytest.log | head -32
python bench.py --steps 50 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
for c in cfg1 cfg3 cfg4; do python bench.py --config $c --steps 30 >> $O/configs.jsonl 2>> $O/bench.err; done
python - <<'PY'
import json
l=json.loads(open('gpurun_out/r03v/bench.json').read().strip().splitlines()[-1]); r=l['roofline']; e=l['extras']
print('bench ms/step %.4f frac %.4f launch_us %.2f value %.3e'%(l['ms_per_step'], r['frac'], r['avg_launch_us'], l['value']))
for k in ('spmm_plain_us','infonce_fwd_ms','infonce_fwdbwd_ms','infonce_fp32_fwdbwd_ms','lightgcn_step_ms_device_rng','simgcl_step_ms_device_rng','lightgcn_step_ms_parity_generator_on_device','simgcl_step_ms_parity_generator_on_device','spmm_fused_launch_us_by_device_clock','spmm_fused_launch_us_by_hip_events_same_launches'): print(' ',k, e.get(k))
for l in open('gpurun_out/r03v/configs.jsonl'):
    l=json.loads(l); print(l['config']['workload'][:40], 'ms/step %.3f frac %.3f launch_us %.1f'%(l['ms_per_step'], l['roofline']['frac'], l['roofline']['avg_launch_us']))
PY
python tools/mt_replay_bench.py > $O/mt_replay.json 2>> $O/bench.err
bash tools/gpu_profile.sh r03v_prof > $O/profile.log 2>&1; tail -12 $O/profile.log
