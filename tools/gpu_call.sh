OUT=gpurun_out/r02t; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "lowrank or lightgcl" 2>&1 | tail -5
timeout 300 python tools/micro/run_gather_ceiling.py $OUT/gather_sliced.json --sliced > $OUT/gather_sliced.log 2>&1; echo "sliced exit $?"; grep -c TBps $OUT/gather_sliced.log
timeout 900 python tools/cfg5_shard.py > $OUT/cfg5_shard.json 2> $OUT/cfg5_shard.err; echo "cfg5 exit $?"; cat $OUT/cfg5_shard.json
