# scratch: the command of the next gpurun call (see tools/gpu_check.sh / tools/gpu_profile.sh for the full recipes)
OUT=gpurun_out/next; mkdir -p $OUT
timeout 900 python -m pytest tests -q -m gpu > $OUT/tests.log 2>&1; echo "tests exit $?"; tail -3 $OUT/tests.log
