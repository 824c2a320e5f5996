mkdir -p gpurun_out/r02g
timeout 1500 python -m pytest tests -m gpu -q -x --tb=short --deselect "tests/test_gpu_parity.py::test_whole_training_step_at_amazon_book_size_matches_the_chunked_oracle" > gpurun_out/r02g/test_all.log 2>&1; echo "pytest exit $?"; tail -4 gpurun_out/r02g/test_all.log
timeout 900 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r02g/bench3.log 2> gpurun_out/r02g/bench3.err; echo "bench exit $?"; python - <<'PY'
import json
l=[x for x in open('gpurun_out/r02g/bench3.log') if x.startswith('{')][-1]
j=json.loads(l)
print('ms_per_step', j['ms_per_step'], 'frac', j['roofline']['frac'], 'avg_us', j['roofline']['avg_launch_us'])
e=j['extras']
for k in ('spmm_plain_us','lightgcn_step_ms_device_rng','simgcl_step_ms_device_rng'): print(k, e.get(k))
PY
