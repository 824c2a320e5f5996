#!/bin/bash
# The round's final evidence on one MI355X box: the GPU suite, the driver's bench command, the other single-GPU configs, the multi-GPU
# bench started the way the driver starts it (two ranks sharing the one device of the box), the replay timing, rocprofv3 kernel stats of
# the bench command and of the fused InfoNCE call, PMC passes (tools/gpu_profile.sh).   usage: bash tools/gpu_r03_final.sh [tag]
T=${1:-r03final}
O=gpurun_out/$T; mkdir -p $O
python -m pytest tests -m gpu -q --maxfail=25 --durations=10 > $O/pytest.log 2>&1; echo "pytest rc $?" | tee $O/pytest.rc; grep -E "passed|failed|error" $O/pytest.log | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py --steps 50 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
for c in cfg1 cfg3 cfg4; do python bench.py --config $c --steps 30 >> $O/configs.jsonl 2>> $O/bench.err; done
SSLREC_BENCH_ONE_DEVICE=1 python bench.py --gpus 2 --steps 10 --warmup 2 > $O/bench_gpus2_one_device.json 2>> $O/bench.err; echo "bench --gpus 2 (one device) rc $?"
python tools/mt_replay_bench.py > $O/mt_replay.json 2>> $O/bench.err
export TMPDIR=/tmp
R=$(pwd)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_infonce -o infonce -- python $R/tools/infonce_profile.py 20 > $R/$O/prof_infonce.log 2>&1; echo "== rocprof infonce exit $?")
f=$(find $O/prof_infonce -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/infonce_kernel_stats.csv
python - "$O" <<'PY'
import json, sys
O = sys.argv[1]
l = json.loads(open(O + '/bench.json').read().strip().splitlines()[-1]); r = l['roofline']; e = l['extras']
print('bench ms/step %.4f frac %.4f launch_us %.2f value %.3e' % (l['ms_per_step'], r['frac'], r['avg_launch_us'], l['value']))
print('  launch kinds', {k: (round(v['avg_launch_us'], 1), round(v['algorithmic_bytes'] / 1e6, 1)) for k, v in (r.get('launch_kinds') or {}).items()})
for k in ('spmm_plain_us', 'infonce_fwd_ms', 'infonce_fwdbwd_ms', 'lightgcn_step_ms_device_rng', 'simgcl_step_ms_device_rng', 'lightgcn_step_ms_parity_generator_on_device',
          'simgcl_step_ms_parity_generator_on_device', 'eval_topk40_all_52643_users_ms'):
    print(' ', k, e.get(k))
for ln in open(O + '/configs.jsonl'):
    c = json.loads(ln); print(c['config']['workload'][:40], 'ms/step %.3f frac %.3f launch_us %.1f' % (c['ms_per_step'], c['roofline']['frac'], c['roofline']['avg_launch_us']))
g = json.loads(open(O + '/bench_gpus2_one_device.json').read().strip().splitlines()[-1])
print('gpus 2 (one device): ms/step %.3f' % g['ms_per_step'], list((g.get('multi_gpu') or {}).keys()))
PY
bash tools/gpu_profile.sh ${T}_prof > $O/profile.log 2>&1; tail -9 $O/profile.log | cut -c1-300
