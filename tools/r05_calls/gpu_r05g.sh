#!/bin/bash
# call g: the two parameters as row ranges of one buffer (no concatenation per forward), compacted edge-dropped views on the row-bundled
# layout: the full GPU suite, the config-5-scale view timings, cfg1 / cfg3 / cfg4 lines
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05g; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu --durations=6 2>&1 | tail -24 > $O/pytest_tail.txt; tail -14 $O/pytest_tail.txt
timeout 400 python tools/bundled_compact_bench.py $O/bundled_compact.json 2> $O/bundled_compact.err | cut -c1-1500
for c in cfg1 cfg3 cfg4; do
  timeout 300 python bench.py --config $c --steps 40 --no-cpu-baseline > $O/${c}_line.json 2> $O/${c}_line.err || echo "$c failed"
done
python - <<'PY'
import json
O='gpurun_out/r05g/'
for c in ('cfg1', 'cfg3', 'cfg4'):
    try:
        j = json.loads(open(O+c+'_line.json').read().strip().splitlines()[-1])
        print(c, 'eager', round(j['ms_per_step_eager'], 4), 'graph', j['ms_per_step_graph'] and round(j['ms_per_step_graph'], 4), 'headline', j['headline_form'], round(j['ms_per_step'], 4))
    except Exception as e:
        print(c, 'unreadable', e)
PY
