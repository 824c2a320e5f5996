OUT=gpurun_out/r02n1; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "narrow or feature_sliced" > $OUT/tests.log 2>&1; echo "tests exit $?"; tail -15 $OUT/tests.log
timeout 300 python tools/spmm_narrow.py > $OUT/narrow.jsonl 2> $OUT/narrow.err; echo "narrow exit $?"; cat $OUT/narrow.jsonl; tail -3 $OUT/narrow.err
