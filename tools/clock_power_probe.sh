#!/bin/bash
# Shader clock and socket power while a kernel loop runs (rocm-smi sampled every 0.2 s): the InfoNCE forward+backward of the cfg-3 item term
# in the default h3 mode, in x6 and in exact-fp32 mode, the swept SpMM of the bench step, and idle.   usage: bash tools/clock_power_probe.sh <outdir>
O=${1:-gpurun_out/clock_probe}; mkdir -p $O
sample() {   # $1 = label, $2.. = command
  label=$1; shift
  "$@" > $O/$label.run.log 2>&1 &
  pid=$!
  sleep 6          # imports, graph build, warm-up
  for i in $(seq 1 15); do
    rocm-smi --showclocks --showpower --json 2>/dev/null | tr -d '\n' >> $O/$label.samples.jsonl; echo >> $O/$label.samples.jsonl
    sleep 0.2
  done
  wait $pid
}
sample infonce_h3 python tools/infonce_profile.py 6000          # (round 5: h3 is the default arithmetic)
sample infonce_x6 env SSLREC_INFONCE_PRECISION=x6 python tools/infonce_profile.py 5000
SSLREC_INFONCE_PRECISION=fp32 sample infonce_fp32 env SSLREC_INFONCE_PRECISION=fp32 python tools/infonce_profile.py 4000
sample spmm_step python bench.py --steps 15000 --warmup 5 --no-extras --no-cpu-baseline
sleep 2
for i in $(seq 1 5); do rocm-smi --showclocks --showpower --json 2>/dev/null | tr -d '\n' >> $O/idle.samples.jsonl; echo >> $O/idle.samples.jsonl; sleep 0.2; done
python - "$O" <<'PY'
import json, sys, glob, os, re
O = sys.argv[1]
out = {'what': 'rocm-smi --showclocks --showpower sampled every 0.2 s while the loop runs; samples after the loop ended (power < 700 W) are dropped'}
for f in sorted(glob.glob(O + '/*.samples.jsonl')):
    label = os.path.basename(f).split('.')[0]
    rows = []
    for ln in open(f):
        ln = ln.strip()
        if not ln:
            continue
        card = json.loads(ln).get('card0', {})

        def g(key):
            return next((float(re.search(r'([0-9.]+)', str(v)).group(1)) for k, v in card.items() if key in k.lower() and re.search(r'[0-9]', str(v))), None)
        rows.append((g('sclk clock speed'), g('mclk clock speed'), g('power')))
    live = [r for r in rows if r[2] and r[2] > 700] if label != 'idle' else rows
    if live:
        out[label] = {'samples': len(live), 'sclk_MHz_mean': round(sum(r[0] for r in live) / len(live), 1), 'sclk_MHz_min': min(r[0] for r in live),
                      'sclk_MHz_max': max(r[0] for r in live), 'mclk_MHz': sorted(set(r[1] for r in live)),
                      'socket_power_W_mean': round(sum(r[2] for r in live) / len(live), 1), 'socket_power_W_max': max(r[2] for r in live)}
json.dump(out, open(O + '/clock_power.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
PY
