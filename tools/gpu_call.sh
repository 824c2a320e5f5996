OUT=gpurun_out/r02g; mkdir -p $OUT
timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "feature_sliced_ranks or graphed_feature" > $OUT/tests.log 2>&1; echo "tests exit $?"; tail -12 $OUT/tests.log
SSLREC_BENCH_ONE_DEVICE=1 timeout 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 2 --steps 5 --warmup 2 > $OUT/bench_feature_2.log 2>&1; echo "bench N=2 exit $?"; tail -2 $OUT/bench_feature_2.log | cut -c1-1800
