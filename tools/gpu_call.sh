mkdir -p gpurun_out/r02c
timeout 900 python -m pytest tests -m gpu -q -x --tb=short -k "spmm or propagate or drop or swept or mask" > gpurun_out/r02c/test_spmm.log 2>&1; echo "pytest exit $?"; tail -5 gpurun_out/r02c/test_spmm.log
for deep in 0 1; do
SSLREC_SWEPT_DEEP=$deep timeout 600 python tools/spmm_xcd.py --split 1 > gpurun_out/r02c/spmm_deep$deep.log 2>&1; echo "deep $deep exit $?"; tail -2 gpurun_out/r02c/spmm_deep$deep.log
done
