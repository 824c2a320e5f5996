OUT=gpurun_out/r02i; mkdir -p $OUT
timeout 110 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "whole_training_step_at_amazon" > $OUT/tests.log 2>&1; echo "tests exit $?"; tail -3 $OUT/tests.log
