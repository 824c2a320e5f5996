#!/bin/bash
# call c: eight processes on one GPU (tests + bench --gpus 4 / 8 code checks), config 5 row-sharded with the complete breakdown,
# the full default bench line (multi_gpu_predicted, launch times by position)
cd "$GRAFT_REPO_ROOT"
R=$PWD
O=gpurun_out/r05c; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -k "(two_ranks_on_one_gpu and 8) or (feature_sliced_ranks_on_one_gpu and 8) or (feature_sliced_lightgcl_ranks_on_one_gpu and 8-) or (whole_training_step_at_amazon and simgcl) or round5" 2>&1 | tail -30 > $O/pytest_tail.txt
tail -12 $O/pytest_tail.txt
for n in 4 8; do
  SSLREC_BENCH_ONE_DEVICE=1 timeout 400 python bench.py --gpus $n --steps 5 --warmup 2 > $O/bench_gpus${n}_one_device.json 2> $O/bench_gpus${n}_one_device.err || echo "bench --gpus $n failed"
done
timeout 300 python tools/cfg5_row_sharded.py --scale 0.01 --reps 2 > $O/cfg5_small.json 2> $O/cfg5_small.err || { echo "cfg5 small failed"; tail -5 $O/cfg5_small.err; }
timeout 600 python tools/cfg5_row_sharded.py --out $O/cfg5_row_sharded_step.json > $O/cfg5_full.log 2> $O/cfg5_full.err || { echo "cfg5 full failed"; tail -5 $O/cfg5_full.err; }
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/cfg5_prof -o cfg5 -- python $R/tools/cfg5_row_sharded.py --reps 1 > $R/$O/cfg5_prof.log 2>&1; echo "== rocprof cfg5 exit $?")
f=$(find $O/cfg5_prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/cfg5_kernel_stats.csv && head -25 $O/cfg5_kernel_stats.csv | cut -c1-150
rm -rf $O/cfg5_prof
timeout 600 python bench.py > $O/bench_line.json 2> $O/bench_line.err || echo "bench failed"
python - <<'PY'
import json
O='gpurun_out/r05c/'
for n in (4, 8):
    try:
        j = json.loads(open(O+'bench_gpus%d_one_device.json' % n).read().strip().splitlines()[-1])
        print('gpus', n, 'value', j['value'], 'ms', j['ms_per_step'], {k: v for k, v in j.items() if k.startswith('value_')}, j.get('headline_decomposition'))
    except Exception as e:
        print('gpus', n, 'unreadable', e)
try:
    j = json.load(open(O+'cfg5_row_sharded_step.json'))
    print('cfg5 step', j['step_ms_compute_only'], json.dumps(j['breakdown_check']))
    for k, v in j['breakdown_ms'].items():
        print('   %-90s %s' % (k[:90], v.get('ms')))
    print(json.dumps(j['compute_beside_exchange_traffic']))
    for k, v in j['critical_path']['schedules'].items():
        print(k, {a: b for a, b in v.items() if a != 'timeline_ms_overlapped'})
except Exception as e:
    print('cfg5 unreadable', e)
try:
    j = json.loads(open(O+'bench_line.json').read().strip().splitlines()[-1])
    r = j['roofline']
    print('bench', j['ms_per_step'], r['avg_launch_us'], r['frac'], r.get('launch_us_by_position_in_step'), r.get('factorized_chain', {}).get('frac_on_those_bytes'))
    print(json.dumps(j.get('multi_gpu_predicted'))[:1500])
    print({k: (v.get('ms_per_step'), v.get('roofline', {}).get('frac')) for k, v in j.get('configs', {}).items()})
    print(json.dumps(j.get('roofline_infonce', {}).get('modes', {}))[:600])
except Exception as e:
    print('bench unreadable', e)
PY
