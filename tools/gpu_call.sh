mkdir -p gpurun_out/r02e
timeout 900 python -m pytest tests -m gpu -q -x --tb=short --durations=5 -k "whole_training_step" > gpurun_out/r02e/test_new4.log 2>&1; echo "pytest exit $?"; tail -15 gpurun_out/r02e/test_new4.log
