mkdir -p gpurun_out/r02e
timeout 1500 python -m pytest tests -m gpu -q -x --tb=short -k "fused_evaluation or hip_negative or sharded_lightgcl or c_only or infonce or device_side or end_to_end" > gpurun_out/r02e/test_new2.log 2>&1; echo "pytest exit $?"; tail -30 gpurun_out/r02e/test_new2.log
