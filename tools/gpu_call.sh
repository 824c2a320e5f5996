OUT=gpurun_out/r02zd; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused_evaluation or device_side_evaluation or trainer_runs" 2>&1 | tail -3
timeout 600 python tools/eval_profile.py $OUT/eval.json 2>&1 | tail -1
