mkdir -p gpurun_out/r02j
timeout 900 python bench.py --steps 50 --warmup 5 > gpurun_out/r02j/bench.log 2> gpurun_out/r02j/bench.err; echo "bench exit $?"; python - <<'PY'
import json
l=[x for x in open('gpurun_out/r02j/bench.log') if x.startswith('{')][-1]
j=json.loads(l)
print('ms_per_step', j['ms_per_step'], 'frac', j['roofline']['frac'])
e=j['extras']
for k in sorted(e):
    if 'eval' in k or 'sample' in k or 'error' in k: print(k, e[k])
PY
for m in lightgcn simgcl sgl lightgcl; do timeout 600 python tools/epoch_demo.py $m 3 graph fused > gpurun_out/r02j/epoch_$m.log 2>&1; echo "epoch $m exit $?"; tail -2 gpurun_out/r02j/epoch_$m.log; done
timeout 600 python tools/spmm_kernels.py > gpurun_out/r02j/spmm_kernels.log 2>&1; tail -9 gpurun_out/r02j/spmm_kernels.log
