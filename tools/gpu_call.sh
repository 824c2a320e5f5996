mkdir -p gpurun_out/r02i
timeout 1500 python -m pytest tests -m gpu -q -x --tb=short --deselect "tests/test_gpu_parity.py::test_whole_training_step_at_amazon_book_size_matches_the_chunked_oracle" > gpurun_out/r02i/test_all.log 2>&1; echo "pytest exit $?"; tail -3 gpurun_out/r02i/test_all.log
timeout 900 python bench.py --steps 50 --warmup 5 > gpurun_out/r02i/bench.log 2> gpurun_out/r02i/bench.err; echo "bench exit $?"; python - <<'PY'
import json
l=[x for x in open('gpurun_out/r02i/bench.log') if x.startswith('{')][-1]
j=json.loads(l)
print('ms_per_step', j['ms_per_step'], 'frac', j['roofline']['frac'], 'avg_us', j['roofline']['avg_launch_us'], 'cpu', j['cpu_baseline']['value'])
e=j['extras']
for k in sorted(e): print(k, e[k])
PY
