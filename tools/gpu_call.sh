mkdir -p gpurun_out/r02d
timeout 1500 python -m pytest tests -m gpu -q --tb=short > gpurun_out/r02d/test_all.log 2>&1; echo "pytest exit $?"; tail -8 gpurun_out/r02d/test_all.log
timeout 900 python bench.py --steps 50 --warmup 5 > gpurun_out/r02d/bench.log 2> gpurun_out/r02d/bench.err; echo "bench exit $?"; tail -c 2500 gpurun_out/r02d/bench.log; tail -3 gpurun_out/r02d/bench.err
