mkdir -p gpurun_out/r02i
timeout 900 python -m pytest tests -m gpu -q -x --tb=short -k "two_ranks or edge_drop or device_rng" > gpurun_out/r02i/test_sgl.log 2>&1; echo "pytest exit $?"; tail -12 gpurun_out/r02i/test_sgl.log
