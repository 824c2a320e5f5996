mkdir -p gpurun_out/r02e
timeout 900 python -m pytest tests -m gpu -q -x --tb=short -k "bpr or gather_backward or hip_graph or tiny or infonce_gathered" > gpurun_out/r02e/test_new5.log 2>&1; echo "pytest exit $?"; tail -15 gpurun_out/r02e/test_new5.log
