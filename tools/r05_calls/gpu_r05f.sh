#!/bin/bash
# call f: SimGCL / SGL steps as one autograd node -- parity tests, cfg3 / cfg4 lines with and without it, kernel stats
cd "$GRAFT_REPO_ROOT"
R=$PWD
O=gpurun_out/r05f; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -q -m gpu -x -k "round5 or training_step or whole_training or traj or views_equals or hip_graph or captured or bench_config_lines or smoke" 2>&1 | tail -30 > $O/pytest_tail.txt; tail -12 $O/pytest_tail.txt
for c in cfg3 cfg4; do
  for one in 1 0; do
    SSLREC_ONE_NODE_STEP=$one timeout 300 python bench.py --config $c --steps 40 --no-cpu-baseline > $O/${c}_one_node_$one.json 2> $O/${c}_one_node_$one.err || echo "$c $one failed"
  done
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$c -o $c -- python $R/tools/step_profile.py $c 40 > $R/$O/${c}_prof.log 2>&1; echo "== rocprof $c exit $?")
  f=$(find $O/prof_$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${c}_kernel_stats.csv; rm -rf $O/prof_$c
done
python - <<'PY'
import json, csv
O='gpurun_out/r05f/'
for c in ('cfg3', 'cfg4'):
    for one in (1, 0):
        try:
            j = json.loads(open(O+'%s_one_node_%d.json' % (c, one)).read().strip().splitlines()[-1])
            print(c, 'one_node', one, 'eager', round(j['ms_per_step_eager'], 4), 'graph', j['ms_per_step_graph'] and round(j['ms_per_step_graph'], 4), 'roofline', round(j['roofline']['frac'], 4), j['roofline'].get('ms_per_step'))
        except Exception as e:
            print(c, one, 'unreadable', e)
    try:
        rows = list(csv.DictReader(open(O+c+'_kernel_stats.csv')))
        steps = [int(r['Calls']) for r in rows if r['Name'].startswith('bpr_fwd_kernel')][0]
        tot = sum(float(r['TotalDurationNs']) for r in rows) / steps / 1e3
        stock = [(r['Name'][:60], int(r['Calls']) / steps, float(r['TotalDurationNs']) / steps / 1e3) for r in rows if 'at::native' in r['Name'] or 'rocclr' in r['Name']]
        print(c, 'steps', steps, 'GPU us/step', round(tot, 1), 'stock:', [(n, round(a, 2), round(b, 1)) for n, a, b in stock if a >= 0.5])
    except Exception as e:
        print(c, 'stats unreadable', e)
PY
