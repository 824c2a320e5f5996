mkdir -p gpurun_out/r02e
timeout 1500 python -m pytest tests -m gpu -q -x --tb=short -k "device_rng or device_sampler or d128 or precision_is or unnormalized" > gpurun_out/r02e/test_new.log 2>&1; echo "pytest exit $?"; tail -30 gpurun_out/r02e/test_new.log
