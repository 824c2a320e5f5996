mkdir -p gpurun_out/r02g
timeout 900 python -m pytest tests -m gpu -q -x --tb=short -k "bpr or gather_backward or hip_graph or tiny or infonce_gathered or sharded" > gpurun_out/r02g/test_det.log 2>&1; echo "pytest exit $?"; tail -5 gpurun_out/r02g/test_det.log
timeout 900 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r02g/bench2.log 2> gpurun_out/r02g/bench2.err; echo "bench exit $?"; python - <<'PY'
import json
l=[x for x in open('gpurun_out/r02g/bench2.log') if x.startswith('{')][-1]
j=json.loads(l)
print('ms_per_step', j['ms_per_step'], 'frac', j['roofline']['frac'], 'avg_us', j['roofline']['avg_launch_us'])
e=j['extras']
for k in ('spmm_plain_us','masked_keep0.5_spmm_us','infonce_fwdbwd_ms','lightgcn_step_ms_cpu_rng_parity','lightgcn_step_ms_device_rng','simgcl_step_ms_cpu_rng_parity','simgcl_step_ms_device_rng'): print(k, e.get(k))
PY
