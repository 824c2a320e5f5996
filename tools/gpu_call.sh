mkdir -p gpurun_out/r02e
timeout 1500 python -m pytest tests -m gpu -q -x --tb=short -k "two_ranks or sharded" > gpurun_out/r02e/test_new3.log 2>&1; echo "pytest exit $?"; tail -30 gpurun_out/r02e/test_new3.log
