OUT=gpurun_out/r02n5; mkdir -p $OUT
timeout 120 python tools/spmm_trace.py > $OUT/trace.json 2> $OUT/trace.err; echo "trace exit $?"; tail -2 $OUT/trace.err
python - <<'PY'
import json
o=json.load(open('gpurun_out/r02n5/trace.json'))
print(o['propagate_fwd_bwd_L3_us'], o['launches_traced'], o['blocks_per_wave'])
for l in o['launches']:
    print('slot', l['ring_slot'])
    for x in l['xcd']:
        print('  ', x)
PY
